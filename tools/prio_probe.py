import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from modelcompose_amd import synthetic, _lib
from modelcompose_amd.model.builder import build_from_state_dict
from modelcompose_amd.train import MultimodalTrainStep
dev = torch.device("cuda", 0)
meta = synthetic.vicuna7b_meta(("vision",), None, layers=32)
meta["lora_dropout"] = 0.0
sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
model = build_from_state_dict(meta, sd, device=dev)
st = MultimodalTrainStep(model, lr=2e-4)
del sd; model._raw = {}
B = 4
ids = synthetic.synthetic_prompt(B, [-200], seed=0).to(dev)
labels = ids.clone(); labels[:, :40] = -100
px = torch.randn(B, 3, 336, 336, device=dev).to(torch.bfloat16)
L = _lib.lib()
def bench(tag, stream, tile192):
    orig = st._forward_backward
    def run():
        for _ in range(2): st.step(ids, labels, {"vision": px})
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): st.step(ids, labels, {"vision": px})
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5
    if tile192:
        # bypass the policy switch: call the inner function directly with tile192 left on
        st.forward_backward = lambda *a, **k: orig(*a, **k)
    if stream is None: t = run()
    else:
        with torch.cuda.stream(stream): t = run()
    if tile192: del st.forward_backward
    print(f"{tag}: {t*1e3:.1f} ms/step")
hp = torch.cuda.Stream(device=dev, priority=-1)
bench("default stream, 256 tiles (shipping)", None, False)
bench("high-priority main stream, 256 tiles", hp, False)
bench("default stream, 192 tiles + overlap", None, True)
bench("high-priority main stream, 192 tiles + overlap", hp, True)
bench("default stream, 256 tiles (shipping) again", None, False)
