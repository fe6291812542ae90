"""N > 1 path on CPU: two gloo ranks shard the sample list with the reference's partition rule and all-gather their
generated ids; the gathered order must equal the reference's `cat` of per-chunk outputs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modelcompose_amd.dist import gather_ids, get_chunk
    samples = list(range(8))
    mine = get_chunk(samples, world, rank)
    ids = torch.tensor([[s * 10 + t for t in range(3)] for s in mine], dtype=torch.int64)     # "generated ids" of my chunk
    allids = gather_ids(ids, world)
    if rank == 0:
        ret.put(allids)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    exp = torch.tensor([[s * 10 + t for t in range(3)] for s in range(8)], dtype=torch.int64)
    assert torch.equal(got, exp)
