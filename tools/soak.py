"""Soak: repeated generate() / generate_pipelined() calls must return identical ids and keep memory flat.
    python tools/soak.py            vision-only model, B = 16 (130 generations)
    python tools/soak.py iav        round 4: image + audio + video into the 3-way composed model, B = 16, the towers on side streams (50 generations)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import synthetic
from modelcompose_amd.model.builder import build_from_state_dict
dev = torch.device("cuda", 0)
if len(sys.argv) > 1 and sys.argv[1] == "iav":
    import bench
    meta = bench.workload_meta("iav", 32)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    m = build_from_state_dict(meta, sd, device=dev); del sd; m._raw = {}
    torch.cuda.empty_cache()
    B = 16
    ids = synthetic.synthetic_prompt(B, [-200, -203, -204], seed=0).to(dev)
    mi = bench.synthetic_inputs(("vision", "audio", "video"), B, dev, 100)
    assert getattr(m, "encode_streams", True)
    ref = m.generate(ids, modal_inputs=mi, max_new_tokens=16, ignore_eos=True)
    m.encode_streams = False
    one = m.generate(ids, modal_inputs=mi, max_new_tokens=16, ignore_eos=True)
    del m.encode_streams
    assert torch.equal(one, ref), "side streams changed the tokens"
    mem0 = torch.cuda.memory_allocated()
    t0 = time.perf_counter()
    for i in range(20):
        assert torch.equal(m.generate(ids, modal_inputs=mi, max_new_tokens=16, ignore_eos=True), ref), i
    outs = list(m.generate_pipelined(((ids, mi) for _ in range(30)), max_new_tokens=16, ignore_eos=True))
    assert all(torch.equal(o, ref) for o in outs)
    torch.cuda.synchronize()
    print(f"iav: 50 generations in {time.perf_counter()-t0:.1f} s, all outputs identical (side streams == one stream); allocated {mem0/2**30:.2f} -> "
          f"{torch.cuda.memory_allocated()/2**30:.2f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
    sys.exit(0)
meta = synthetic.vicuna7b_meta(("vision",), None, layers=32)
sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
m = build_from_state_dict(meta, sd, device=dev); del sd; m._raw = {}
torch.cuda.empty_cache()
ids = synthetic.synthetic_prompt(16, [-200], seed=0).to(dev)
px = torch.randn(16, 3, 336, 336, device=dev).to(torch.bfloat16)
ref = m.generate(ids, modal_inputs={"vision": px}, max_new_tokens=32, ignore_eos=True)
mem0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
for i in range(60):
    out = m.generate(ids, modal_inputs={"vision": px}, max_new_tokens=32, ignore_eos=True)
    assert torch.equal(out, ref), i
outs = list(m.generate_pipelined(((ids, {"vision": px}) for _ in range(60)), max_new_tokens=32, ignore_eos=True))
assert all(torch.equal(o, ref) for o in outs)
for i in range(10):
    out = m.generate(ids, modal_inputs={"vision": px}, max_new_tokens=32, do_sample=True, temperature=0.7, top_p=0.9, seed=i, ignore_eos=True)
torch.cuda.synchronize()
print(f"130 generations in {time.perf_counter()-t0:.1f} s, all greedy outputs identical; allocated {mem0/2**30:.2f} -> {torch.cuda.memory_allocated()/2**30:.2f} GiB, reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
