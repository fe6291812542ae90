"""Full-width parity cases shared by tests/test_fullwidth_parity_gpu.py and tools/screen_fullwidth_seeds.py.

The real Vicuna-7B widths (hidden 4096, 32 heads x 128, FFN 11008, vocab 32000, LoRA r = 128) and the real-size encoders
(CLIP-L/14-336 24 layers, BEATs 12 layers + 32-query Q-Former, LanguageBind-Video 24 layers x 8 frames, PointBERT 8192 points / 512
groups / k = 32) with TWO decoder layers - what the CPU oracle finishes in seconds to a few minutes.  Weights and inputs come from
torch's CPU generator, so the build container (seed screening) and the GPU box (the test) see bit-identical tensors.

Seeds: the logits of a random-init model are close to Gaussian noise over the 32000-entry vocabulary, so the top-2 gap of a step is
small now and then (mean 0.044 of the logit scale, exponentially distributed), while bf16 storage makes any two correct
implementations differ by ~2e-3 of the scale per logit after two layers.  Every batch ROW therefore has its own input
seed, screened on the CPU by tools/screen_fullwidth_seeds.py: a row seed is accepted when BOTH oracles (reference branch form fp32,
and the device-rounding restatement) produce the same 17 ids and every step's top-2 margin is >= MIN_MARGIN of the logit scale.
Rows are independent samples, so screened rows can be batched freely; "ids equal on every row and every step" is then asserted
with no margin gate.  The final pick is confirmed on the GPU (tools/screen_rows_gpu.py: kernels are deterministic, so a row that matches
once matches always).  `extra_rows` are UNSCREENED seeds: for them the tests assert the property that holds for any input - ids equal
up to the first step whose oracle top-2 margin is within the bf16 noise, where the device then picks the oracle's runner-up."""
from __future__ import annotations

import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NEW = 17                 # prefill token + 16 decode steps
MIN_MARGIN = 4e-3          # of max |logit| (screening only; the tests assert plain equality)

SENT = {"vision": -200, "audio": -203, "video": -204, "point": -205}

CASES = {
    # BASELINE configs[1]: vision-only LocalLoRA Vicuna (adapters default + vision)
    "configs1_vision": dict(modals=("vision",), reset=None, inputs=("vision",), seed=11, row_seeds=[449, 470], extra_rows=[100, 101, 103, 104, 106, 107]),
    # configs[2]: online-merge-reset 3-way composed model fed image + audio; the video adapter / encoder are present, the input absent
    "configs2_image_audio_video_absent": dict(modals=("vision", "audio", "video"), reset="default-vision=0.333,default-audio=0.333,default-video=0.333",
                                              inputs=("vision", "audio"), seed=21, row_seeds=[519, 547], extra_rows=[200, 201, 202, 203, 204, 205]),
    # configs[3]: 4-modality composed model, MCUB-4-shaped inputs (spliced length 3337)
    "configs3_mcub4": dict(modals=("vision", "audio", "video", "point"),
                           reset="default-vision=0.25,default-audio=0.25,default-video=0.25,default-point=0.25",
                           inputs=("vision", "audio", "video", "point"), seed=31, row_seeds=[600, 622], extra_rows=[300, 301]),
}


# Full DEPTH: the metric's model - 3-way composed Vicuna-7B, all 32 decoder layers, image + audio + video, spliced length
# 2793 - on two unscreened rows.  The oracle side is a committed fixture (tests/golden/g15_fulldepth_iav.npz, written in the build
# container by `python -m oracle.gen_golden g15`: ~45 GB of fp32 weights and ~15 minutes of CPU, too much for the GPU box's test run);
# `depth8` is the same model cut to 8 layers, cheap enough for the oracle to run next to the test (error growth 2 -> 8 -> 32 layers).
DEPTH_CASES = {
    "fulldepth_iav": dict(modals=("vision", "audio", "video"), reset="default-vision=0.333,default-audio=0.333,default-video=0.333",
                          inputs=("vision", "audio", "video"), seed=41, row_seeds=[700, 701], layers=32),
    # EIGHT unscreened rows of the same model (rows 700 / 701 are the two above), with the logits of BOTH oracles - the fp32
    # branch form and the device-rounding restatement teacher-forced on the fp32 oracle's ids - tests/golden/g17_fulldepth_iav8.npz
    "fulldepth_iav8": dict(modals=("vision", "audio", "video"), reset="default-vision=0.333,default-audio=0.333,default-video=0.333",
                           inputs=("vision", "audio", "video"), seed=41, row_seeds=[700, 701, 702, 703, 704, 705, 706, 707], layers=32),
    "depth8_iav": dict(modals=("vision", "audio", "video"), reset="default-vision=0.333,default-audio=0.333,default-video=0.333",
                       inputs=("vision", "audio", "video"), seed=41, row_seeds=[700], layers=8),
}


def _case(name):
    return CASES[name] if name in CASES else DEPTH_CASES[name]


def _weight_cache_path(c):
    """MC_TEST_WEIGHT_CACHE (set by tests/conftest.py for GPU sessions): a directory shared by the processes of one test session.  The
    32-layer model's 8.5 G synthetic weights take ~85 s of torch's single-threaded CPU generator, and the session draws the SAME ones twice -
    in the pytest process (bf16 library) and in the fp16 child of tests/test_fp16_gpu.py; the second draw becomes a read of the first."""
    d = os.environ.get("MC_TEST_WEIGHT_CACHE")
    if not d or c.get("layers", 2) < 8:
        return None
    return os.path.join(d, f"seed{c['seed']}_layers{c['layers']}_{'-'.join(c['modals'])}.pt")


def build_weights(name: str, lora_b_std: float = 0.01):
    """-> (meta, sd bf16 on the CPU)."""
    from modelcompose_amd import synthetic
    c = _case(name)
    meta = synthetic.vicuna7b_meta(c["modals"], c["reset"], layers=c.get("layers", 2))
    path = _weight_cache_path(c)
    sd = None
    if path and os.path.exists(path):
        try:
            sd = torch.load(path, map_location="cpu", mmap=True, weights_only=True)
        except Exception:                                   # a torn or foreign file: draw the weights as if there were no cache
            sd = None
    if sd is None:
        sd = synthetic.synthetic_state_dict(meta, device="cpu", seed=c["seed"], dtype=torch.bfloat16)
        if path:
            try:
                tmp = f"{path}.{os.getpid()}.tmp"
                torch.save(sd, tmp)
                os.replace(tmp, path)                       # atomic: a reader sees the whole file or none
            except Exception:
                pass
    if lora_b_std != 0.01:
        for k in sd:
            if ".lora_B." in k:
                sd[k] = (sd[k].float() * (lora_b_std / 0.01)).to(torch.bfloat16)
    return meta, sd


def build_rows(name: str, row_seeds=None):
    """-> (input_ids (B, L_text), modal_inputs bf16 on the CPU): row r is drawn from its own generator (seed row_seeds[r])."""
    from modelcompose_amd import synthetic
    c = _case(name)
    row_seeds = c["row_seeds"] if row_seeds is None else row_seeds
    ids, per = [], {m: [] for m in c["inputs"]}
    for rs in row_seeds:
        ids.append(synthetic.synthetic_prompt(1, [SENT[m] for m in c["inputs"]], seed=rs))
        g = torch.Generator().manual_seed(rs + 100000)
        rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
        if "vision" in per:
            per["vision"].append(rnd(1, 3, 336, 336))
        if "audio" in per:
            fb = rnd(1, 1024, 128) * 0.5
            fb[:, 998:] = 0
            per["audio"].append(fb)
        if "video" in per:
            per["video"].append(rnd(1, 3, 8, 224, 224))
        if "point" in per:
            xyz = rnd(1, 8192, 3)
            xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(1, 8192, 1, generator=g) ** (1 / 3)
            per["point"].append(torch.cat([xyz, torch.rand(1, 8192, 3, generator=g)], -1))
    B = len(row_seeds)
    mi = {}
    for m, parts in per.items():
        t = torch.cat(parts, 0).to(torch.bfloat16)
        mi[m] = {"audio_inputs": t, "audio_padding_mask": torch.zeros(B, 1024, dtype=torch.bool)} if m == "audio" else t
    return torch.cat(ids, 0), mi


def build_case(name: str, row_seeds=None, lora_b_std: float = 0.01):
    """-> (meta, sd bf16 on the CPU, input_ids (B, L_text), modal_inputs bf16 on the CPU)."""
    meta, sd = build_weights(name, lora_b_std)
    ids, mi = build_rows(name, row_seeds)
    if "point" in mi:
        meta["fps_start"] = [0] * ids.shape[0]
    return meta, sd, ids, mi


def sd_to_f32_inplace(sd):
    """bf16 -> fp32 one tensor at a time (a 32-layer state dict is 17 GB in bf16 and 35 GB in fp32: never hold both)."""
    for k in list(sd):
        v = sd[k]
        if v.is_floating_point() and v.dtype != torch.float32:
            sd[k] = v.float()
            del v
    return sd


def to_f32(x):
    if isinstance(x, dict):
        return {k: to_f32(v) for k, v in x.items()}
    return x.float() if torch.is_tensor(x) and x.is_floating_point() else x


def to_dev(x, dev="cuda"):
    if isinstance(x, dict):
        return {k: to_dev(v, dev) for k, v in x.items()}
    return x.to(dev) if torch.is_tensor(x) else x


def margins(logits: torch.Tensor) -> torch.Tensor:
    """(B, T, V) -> (B, T) top-2 gap as a fraction of max |logit| over the whole tensor."""
    top2 = logits.float().topk(2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]) / logits.float().abs().max()
