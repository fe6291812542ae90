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


def _ragged_ids(sample, rank):
    """"generated ids" of one sample: its length depends on the sample (EOS at different steps)."""
    n = 2 + (sample * 3) % 5
    return [sample * 100 + t for t in range(n)]


def _ragged_worker(rank, world, port, ret, n_samples, batch):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modelcompose_amd.dist import gather_ids, split_list
    samples = list(range(n_samples))
    chunks = split_list(samples, world)
    mine = chunks[rank] if rank < len(chunks) else []               # ceil(n/N) chunks: the last rank's may be shorter or missing
    got = []
    n_batches = -(-max(len(c) for c in chunks) // batch)
    for b in range(n_batches):                                       # every rank makes the same number of collective calls
        part = mine[b * batch:(b + 1) * batch]
        rows = [_ragged_ids(s, rank) for s in part]
        T = max((len(r) for r in rows), default=0)
        ids = torch.full((len(rows), T), 0, dtype=torch.int64)       # HF pads finished rows with pad_token_id (0)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
        allids, nrows = gather_ids(ids, world, return_rows=True)
        got.append((allids, nrows))
    if rank == 0:
        ret.put(got)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_of_unequal_chunks_and_lengths():
    """7 samples on 2 ranks = chunks of 4 and 3 (model_multimodal_qa_loader.py:25-33), batches of 2, rows that stop
    at different lengths.  The gathered rows, batch by batch, must be the reference's `cat` of the chunk outputs (MCUB-4.sh:60-70):
    rank-major, every row's own ids followed only by pad."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 2, port, q, 7, 2)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    chunks = [[0, 1, 2, 3], [4, 5, 6]]
    assert len(got) == 2
    per_rank = {0: [], 1: []}
    for b, (allids, nrows) in enumerate(got):
        exp_rows = [len(c[b * 2:(b + 1) * 2]) for c in chunks]
        assert nrows == exp_rows and allids.shape[0] == sum(exp_rows)
        row = 0
        for r, c in enumerate(chunks):
            for s in c[b * 2:(b + 1) * 2]:
                want = _ragged_ids(s, r)
                assert allids[row, :len(want)].tolist() == want and int(allids[row, len(want):].abs().sum()) == 0, (b, r, s, allids[row])
                per_rank[r].append(s)
                row += 1
    assert per_rank[0] + per_rank[1] == list(range(7))              # concatenating rank 0's answers then rank 1's = the question order


def test_gather_with_an_empty_rank():
    """5 samples on 4 ranks: ceil(5/4) = 2 -> chunks 2, 2, 1 and NO chunk for rank 3 (the reference's split_list returns 3 chunks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_worker, args=(r, 4, port, q, 5, 2)) for r in range(4)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (allids, nrows), = got
    assert nrows == [2, 2, 1, 0] and allids.shape[0] == 5
    for row, s in enumerate(range(5)):
        want = _ragged_ids(s, 0)
        assert allids[row, :len(want)].tolist() == want


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


# ------------------------------------------------------------------------------------------------------------------- round 5
def _soak_worker(rank, world, port, ret, n_samples, rounds):
    """An eval job in miniature on `world` gloo ranks: the reference's ceil(n / N) chunks (uneven, the last ranks short or EMPTY), per batch a
    gather of the ids (ragged lengths) AND of the step logits (north_star's wording), many rounds back to back."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from modelcompose_amd.dist import gather_ids, gather_logits, split_list
    V, T = 16, 3
    ok = True
    for rd in range(rounds):
        samples = list(range(rd * 1000, rd * 1000 + n_samples))
        chunks = split_list(samples, world)                       # the reference's rule: ceil(n / N) per chunk, possibly fewer than N chunks
        mine = chunks[rank] if rank < len(chunks) else []         # (its get_chunk raises for a rank without one, model_multimodal_qa_loader.py:31-33)
        ids = torch.tensor([[s * 10 + t for t in range(T)] for s in mine], dtype=torch.int64).reshape(len(mine), T)
        lg = torch.tensor([[[float(s) + 0.001 * t + 0.01 * v for v in range(V)] for t in range(T)] for s in mine], dtype=torch.float32).reshape(len(mine), T, V)
        all_ids = gather_ids(ids, world)
        all_lg = gather_logits(lg, world)
        exp_ids = torch.tensor([[s * 10 + t for t in range(T)] for s in samples], dtype=torch.int64)
        exp_lg = torch.tensor([[[float(s) + 0.001 * t + 0.01 * v for v in range(V)] for t in range(T)] for s in samples], dtype=torch.float32)
        ok = ok and torch.equal(all_ids, exp_ids) and torch.equal(all_lg, exp_lg)
    ret.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_soak_with_uneven_chunks_and_an_empty_rank():
    """8 ranks on one host (what the driver's SCALE run starts), 19 samples -> ceil(19 / 8) = 3 per rank: ranks 0-5 hold 3, rank 6
    holds 1, rank 7 NONE; every rank must see the reference's `cat` order (eval/model_multimodal_qa_loader.py:25-33) for ids and logits,
    20 rounds in a row."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_soak_worker, args=(r, world, port, q, 19, 20)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert got == [(r, True) for r in range(world)]
