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


def _grad_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modelcompose_amd.train.buckets import allreduce_buckets, bucket_ranges
    # 6 "layers" of 10 gradients each laid out last layer first, then 7 projector / token gradients
    n_layers, per = 6, 10
    layer_end = {l: (n_layers - l) * per for l in range(n_layers)}
    n_params = n_layers * per + 7
    ranges = bucket_ranges(layer_end, n_layers, 4, n_params)
    g = torch.arange(n_params, dtype=torch.float32) * (rank + 1)
    for h in allreduce_buckets(g, ranges):
        h.wait()
    if rank == 0:
        ret.put((ranges, g))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bucketed_gradient_allreduce():
    """Config 5's exchange step: buckets tile the flat gradient buffer exactly once, complete in backward order, and the
    bucketed asynchronous all-reduce equals one all-reduce of the whole buffer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ranges, g = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[0] for r in ranges] == [4, 0, -1]                       # layers 5..4 ready after layer 4, layers 3..0 after layer 0, rest last
    cover = sorted((lo, hi) for _, lo, hi in ranges)
    assert cover[0][0] == 0 and cover[-1][1] == 67 and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert torch.equal(g, torch.arange(67, dtype=torch.float32) * 3)   # rank 0 (x1) + rank 1 (x2)
