"""GPU-box diagnostic: localise where the HIP backbone and oracle/device_path.py part ways.  One decoder layer of a full-width case:
the runtime's workspace keeps layer 0's q|k|v, rotated q, attention output and SwiGLU output after the prefill; each is compared
with the restatement's trace (relative to that tensor's own max)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fullwidth_cases as fc  # noqa: E402
from oracle import device_path, pipeline  # noqa: E402


def al(v, a=256):
    return (v + a - 1) // a * a


def main():
    from modelcompose_amd.model.builder import build_from_state_dict
    name = sys.argv[1] if len(sys.argv) > 1 else "configs1_vision"
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    fc_layers = layers
    meta, sd, ids, mi = fc.build_case(name)
    if layers != 2:
        meta["num_hidden_layers"] = layers
        sd = {k: v for k, v in sd.items() if not (k.startswith("model.layers.") and int(k.split(".")[2]) >= layers)}
    model = build_from_state_dict(meta, sd)
    mid = fc.to_dev(mi)
    feats, _ = model.encode_modal_inputs(mid, model.prefix_tokens, model.suffix_tokens)
    plan = model._plan(ids.cuda(), None, None, mid, feats)
    st = model._prefill(plan, feats, 0, want_hidden=True, want_logits=True)
    torch.cuda.synchronize()
    lay = st["layout"]
    cfg = model.config
    M, B, Lq = lay.M, plan.B, plan.Lmax
    Hd, I, H, D = cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim
    qkvd = 3 * H * D
    ws = st["ws"]
    off = 0

    def take(nbytes, dtype, shape):
        nonlocal off
        t = ws[off:off + nbytes].view(dtype).view(*shape).clone()
        off += al(nbytes)
        return t
    qkv = take(M * qkvd * 2, torch.bfloat16, (M, qkvd))
    qseq = take(B * Lq * H * D * 2, torch.bfloat16, (B * Lq, H * D))
    attn = take(M * Hd * 2, torch.bfloat16, (M, Hd))
    inter = take(M * I * 2, torch.bfloat16, (M, I))
    rs = take(M * 4, torch.float32, (M,))
    # oracle
    sdf = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
    od = pipeline.OracleModel.from_state_dict(sdf, meta, emulate="device")
    fb = {m: f.float().cpu() for m, f in feats.items()}
    am, emb, _, mam = od.prepare(ids, fc.to_f32(mi), feats_blocks=fb)
    tr = {}
    with torch.no_grad():
        logits, kv = device_path.forward(od.device_weights(), device_path.bf(emb.float()), mam, last_only=True, trace=tr)
    # routed row r <-> sequence slot (b, t)
    seq_of_row = torch.from_numpy(lay.order_b.astype(np.int64) * Lq + lay.order_t.astype(np.int64))

    def cmp(tag, dev_rows, ora_seq):
        d = dev_rows.float().cpu()
        o = ora_seq[seq_of_row]
        err = (d - o).abs()
        print(f"{tag:10s} max|err|/max|ref| = {err.max().item() / o.abs().max().item():.3e}   mean|err|/mean|ref| = {err.mean().item() / o.abs().mean().item():.3e}"
              f"   frac elements differing = {(err > 0).float().mean().item():.4f}", flush=True)
    last = layers - 1
    if layers == 1:
        cmp("qkv", qkv, tr["0.qkv"])
        cmp("attn", attn, tr["0.attn"])
        cmp("inter", inter, tr["0.inter"])
        d = qseq.float().cpu()
        o = tr["0.q_rot"].view(B, Lq, H, D).reshape(B * Lq, H * D)
        print(f"q_rot      max|err|/max|ref| = {(d - o).abs().max().item() / o.abs().max().item():.3e}  differing {(d != o).float().mean().item():.4f}")
    hid = st["hidden"].float().cpu()                    # final-norm output of every routed row
    x2 = tr[f"{last}.x2"]
    nl = device_path.bf(x2 * device_path._rs(x2, cfg.rms_norm_eps) * od.device_weights().final_norm)
    cmp("final_norm", hid, nl)
    lg = st["logits"].float().cpu()
    print("logits     rel err", ((lg - logits[:, -1]).abs().max() / logits.abs().max()).item())


if __name__ == "__main__":
    main()
