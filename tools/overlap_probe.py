"""Probe: does running two generate() pipelines concurrently on two HIP streams (decode of one batch beside the prefill of the next) raise
samples/s on one MI355X?  Two full model instances (no shared state) driven from two Python threads, each on its own torch stream."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import synthetic
from modelcompose_amd.model.builder import build_from_state_dict

dev = torch.device("cuda", 0)
B, NEW, STEPS = 16, 32, 6
models = []
for i in range(2):
    meta = synthetic.vicuna7b_meta(("vision",), None, layers=32)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    m = build_from_state_dict(meta, sd, device=dev)
    del sd
    m._raw = {}
    models.append(m)
torch.cuda.empty_cache()
ids = synthetic.synthetic_prompt(B, [-200], seed=0).to(dev)
px = torch.randn(B, 3, 336, 336, device=dev).to(torch.bfloat16)

def run(m, stream, n, delay=0.0):
    time.sleep(delay)
    with torch.cuda.stream(stream):
        for _ in range(n):
            m.generate(ids, modal_inputs={"vision": px}, max_new_tokens=NEW, ignore_eos=True)
        stream.synchronize()

for m in models:
    run(m, torch.cuda.current_stream(), 2)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(models[0], torch.cuda.current_stream(), STEPS)
torch.cuda.synchronize()
t1 = time.perf_counter() - t0
print(f"sequential: {STEPS} steps {t1*1e3/STEPS:.1f} ms/step  {B*STEPS/t1:.1f} samples/s")
s = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
for delay in (0.0, 0.13):
    th = [threading.Thread(target=run, args=(models[i], s[i], STEPS, delay * i)) for i in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t0
    print(f"two pipelines (second starts {delay*1e3:.0f} ms later): {2*STEPS} steps in {t2*1e3:.0f} ms = {t2*1e3/(2*STEPS):.1f} ms/step  {2*B*STEPS/t2:.1f} samples/s")
