"""How long the CPU draw of the 32-layer synthetic weights takes on this host, sequential and with MC_SYNTH_THREADS workers (same bits)."""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import fullwidth_cases as fc
name = sys.argv[1] if len(sys.argv) > 1 else "fulldepth_iav8"
dig = {}
for th in (int(x) for x in (sys.argv[2:] or ["0", "32"])):
    os.environ["MC_SYNTH_THREADS"] = str(th)
    t = time.time()
    meta, sd = fc.build_weights(name)
    dt = time.time() - t
    h = hashlib.sha256()
    for k in sorted(sd)[::37]:
        h.update(sd[k].contiguous().view(torch.uint8).numpy().tobytes()[:1 << 20])
    dig[th] = h.hexdigest()[:16]
    print(f"{name}: MC_SYNTH_THREADS={th}: {dt:.1f} s, {sum(v.numel() for v in sd.values()) / 1e9:.2f} G values, digest {dig[th]}", flush=True)
    del sd
print("identical" if len(set(dig.values())) == 1 else "DIFFERENT")
