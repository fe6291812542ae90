#!/usr/bin/env python
"""Headline benchmark: samples/s of composed-Vicuna-7B greedy generation on the metric's inputs (BASELINE.json `metric`:
"composed-Vicuna-7B greedy gen, img+audio+video").

A "step" = one pass of the hot path over one batch of synthetic input on every rank:
  CLIP-ViT-L/14-336 + BEATs/Q-Former + LanguageBind-Video encode -> projectors -> splice (2792 tokens per sample) ->
  LocalLoRA-composed Vicuna-7B prefill (3-way online-merge-reset, routed adapters) -> 32 greedy tokens (device-resident loop,
  hipGraph) -> all-gather of the generated ids (N > 1).
Default workload `iav` = the metric's config: BASELINE configs[2]'s model (online-merge-reset vision/audio/video = 0.333) fed
336 px image + 10 s audio + 8-frame video, batch 48 per GPU (the measured-best batch; weak scaling: per-GPU work fixed).  Inputs and weights are
resident in HBM before the timed region.  Other workloads (parity-test configs, not the headline): `vision` = configs[1],
`mcub4` = configs[3], `train` = configs[4].

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under `torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE in the environment) the process IS a rank and
WORLD_SIZE must equal --gpus.  Run bare (`python bench.py --gpus N`, as the reference starts its N workers from one command,
scripts/model_composition/test/MCUB-4.sh:21,42-70) the process is only a launcher: before anything touches the GPU it starts N child
processes of this file with the rank environment set, relays rank 0's JSON line and exits with the children's worst status.

Prints ONE JSON line on rank 0 (contract in the task prompt) with these extra objects, all measured OUTSIDE the timed region
in a separate profiled pass of the same step:
  roofline        – the dominant kernel (gemm_tile256_kernel, MFMA-bound): algorithmic FLOPs per launch / average launch duration
                    from HIP events on the launch stream
  roofline_decode – the decode step (HBM-bound): algorithmic bytes per step (weights once + the KV cache once) / step time of the
                    graph-replayed loop, plus every decode kernel class's own HBM fraction from event-bracketed launches
  stages          – encode / prefill / decode milliseconds per step
  cpu_baseline    – the oracle (CPU port of the reference algorithm, torch fp32, host cores) on a bounded sample of the same workload
                    (one sample, timed with 1 and with 8 of the 32 decoder layers)
  secondary       – N = 1 only: BASELINE configs[1] / [3] / [4] (`vision`, `mcub4`, `train`) for a few steps each in child processes after
                    the headline model has been freed: {value, ms_per_step, config} per config (driver-visible, not part of `value`)
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def pin_rank_cpus(environ=None, apply=True):
    """Per-rank CPU affinity, set in the rank process BEFORE torch is imported (its thread pools size themselves from the affinity mask):
    rank r of the node's W ranks takes the r-th contiguous block of the cores this process may run on.  Every rank issues several hundred
    encoder launches per batch from Python; eight unpinned ranks on one host migrate across sockets and contend for the same cores
    (the reference pins nothing: scripts/model_composition/test/MCUB-4.sh:42-58 just backgrounds one worker per GPU).  MC_BENCH_PIN=0 turns
    it off.  Returns the core list (None = not pinned)."""
    env = os.environ if environ is None else environ
    r, w = env.get("LOCAL_RANK"), env.get("LOCAL_WORLD_SIZE") or env.get("WORLD_SIZE")
    if r is None or w is None or env.get("MC_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_getaffinity"):
        return None
    r, w = int(r), int(w)
    cpus = sorted(os.sched_getaffinity(0))
    per = len(cpus) // max(w, 1)
    if w < 2 or per < 1 or not 0 <= r < w:
        return None
    mine = cpus[r * per:(r + 1) * per]
    if apply:
        os.sched_setaffinity(0, mine)
        os.environ.setdefault("OMP_NUM_THREADS", str(min(per, 32)))
    return mine


PINNED_CPUS = pin_rank_cpus() if __name__ == "__main__" else None

import torch  # noqa: E402

# MC_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL init, barriers, gathers, the MAX all-reduce of the timing) in a world of one
# process: the only way to exercise it on a single-GPU box
DIST = os.environ.get("MC_BENCH_FORCE_DIST", "") == "1"
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0                      # spec HBM3E peak

WORKLOADS = {
    # name: (modalities, sentinels, default per-GPU batch, metric suffix, description)
    # per-GPU batch 48 = the measured-best of the sweep 16 / 24 / 32 / 48 / 64 -> 20.6 / 20.9 / 21.4 / 22.0 / 21.95 samples/s (the decode step's
    # 13.2-GB weight read amortises over more rows; 72 GB of KV cache + 52 GB of composed weights of the 288 GB)
    "iav": (("vision", "audio", "video"), [-200, -203, -204], 48, "img+audio+video",
            "metric config = configs[2]'s model (3-way composed Vicuna-7B, online-merge-reset vision/audio/video = 0.333; routed adapters "
            "default/audio/vision/video) with the metric's inputs: 336 px image + 10 s audio (1024x128 fbank) + 8-frame 224 px video"),
    "vision": (("vision",), [-200], 16, "img",
               "configs[1]: vision-only composed Vicuna-7B (LocalLoRA r128: default+vision adapters), synthetic 336 px images"),
    "mcub4": (("vision", "audio", "video", "point"), [-200, -203, -204, -205], 4, "img+audio+video+point",
              "configs[3]: 4-modality composed Vicuna-7B (online-merge-reset 4 x 0.25), MCUB-4-shaped inputs: 336 px image + 10 s audio + "
              "8-frame video + 8192-point cloud"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (0 = the workload's default: 48 for iav, 16 for vision, 4 for mcub4 / train)")
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer decoder layers (invalidates the number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="time the oracle on ONE sample with all 32 decoder layers (35 GB of fp32 weights, several minutes of host time) instead of "
                         "extrapolating from 1 and 8 layers: cpu_baseline.kind = 'port-measured'")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other BASELINE configs after the timed region")
    ap.add_argument("--no-profile", action="store_true", help="skip the profiled pass (roofline objects become null)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=None,
                    help="the eval loop's throughput mode (default for the iav / vision workloads): two generation pipelines on two HIP streams, the "
                    "decode of batch i - HBM-bound - beside the encoders + prefill of batch i+1 - MFMA-bound "
                    "(MultimodalLlamaForCausalLM.generate_pipelined, eval/model_multimodal_qa_loader.py --pipeline); every step still is one full "
                    "batch, all K batches start and finish inside the timed region")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="one generate() call per step, nothing overlapped")
    ap.add_argument("--no-overlap", action="store_true", help="debug only (train): weight-gradient / rank-projection GEMMs on the main stream")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16"],
                    help="16-bit storage element of the run: bf16 (BASELINE.json's dtype, the headline) or fp16 - the reference's own inference dtype "
                         "(modelcompose/model/builder.py:41, :162, :185); selects libmc_hip_f16.so for the whole process (MC_STORAGE_DTYPE)")
    ap.add_argument("--gather", default="ids", choices=["ids", "logits"],
                    help="what the ranks all-gather per batch: the generated ids (default; what the reference's per-chunk answer files hold, "
                         "MCUB-4.sh:60-70) or, as BASELINE.json's north_star words it, the step LOGITS [B, new_tokens, vocab] fp32 (the decode "
                         "graph is still replayed: runtime option graph_logits)")
    ap.add_argument("--workload", default="iav", choices=["iav", "vision", "generate", "mcub4", "train"],
                    help="iav (default) = the metric's config; vision (alias generate) = configs[1]; mcub4 = configs[3]; train = configs[4], the "
                         "stage-2 finetune step (forward + backward + gradient all-reduce + AdamW)")
    a = ap.parse_args()
    if a.workload == "generate":
        a.workload = "vision"
    return a


# ------------------------------------------------------------------------------------------------------------------- inputs
def synthetic_inputs(modals, B, dev, seed):
    """image N(0,1) (B,3,336,336); BEATs fbank N(0,0.5²) (B,1024,128) with the last 26 frames zero (10 s = 998 frames),
    padding mask all False; video N(0,1) (B,3,8,224,224); points xyz uniform in the unit ball, rgb U[0,1], (B,8192,6)."""
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32)
    mi = {}
    from modelcompose_amd import _lib
    st = _lib.storage_dtype()                       # the library's 16-bit storage element (bf16, or fp16 with --dtype fp16)
    if "vision" in modals:
        mi["vision"] = rnd(B, 3, 336, 336).to(st)
    if "audio" in modals:
        fbank = rnd(B, 1024, 128) * 0.5
        fbank[:, 998:] = 0
        mi["audio"] = {"audio_inputs": fbank.to(st), "audio_padding_mask": torch.zeros(B, 1024, dtype=torch.bool, device=dev)}
    if "video" in modals:
        mi["video"] = rnd(B, 3, 8, 224, 224).to(st)
    if "point" in modals:
        xyz = rnd(B, 8192, 3)
        xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(B, 8192, 1, generator=g, device=dev) ** (1 / 3)
        mi["point"] = torch.cat([xyz, torch.rand(B, 8192, 3, generator=g, device=dev)], -1).to(st)
    return mi


def workload_meta(name, layers):
    from modelcompose_amd import synthetic
    modals = WORKLOADS[name][0]
    reset = None if len(modals) == 1 else ",".join(f"default-{m}={round(1.0 / len(modals), 3)}" for m in modals)
    return synthetic.vicuna7b_meta(modals, reset, layers=layers)


# ------------------------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline_full(workload: str, new_tokens: int):
    """The oracle on one sample of the workload at FULL depth, timed, nothing extrapolated.  Needs ~40 GB of host memory."""
    from modelcompose_amd import synthetic
    from oracle import pipeline
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    modals, sentinels = WORKLOADS[workload][0], WORKLOADS[workload][1]
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"
    meta = workload_meta(workload, 32)
    sd = synthetic.synthetic_state_dict(meta, device=gen_dev, seed=7, dtype=torch.float32)
    sd = {k: v.cpu() for k, v in sd.items()}
    om = pipeline.OracleModel.from_state_dict(sd, meta)
    ids = synthetic.synthetic_prompt(1, sentinels)
    mi = synthetic_inputs(modals, 1, gen_dev, 3)
    mi = {k: ({kk: (vv.float().cpu() if vv.is_floating_point() else vv.cpu()) for kk, vv in v.items()} if isinstance(v, dict) else v.float().cpu())
          for k, v in mi.items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        om.generate(ids, mi, max_new_tokens=new_tokens, ignore_eos=True)
        full = time.perf_counter() - t0
    return {"value": 1.0 / full, "unit": "samples/s", "cores": cores, "kind": "port-measured", "extrapolated": False,
            "sample": f"oracle (torch fp32 CPU port of the reference path, batch 1 as the reference's eval loop) on ONE sample of the same workload "
                      f"({WORKLOADS[workload][3]}, {new_tokens} greedy tokens), all 32 decoder layers, every encoder: {full:.1f} s, timed, no extrapolation"}


def cpu_baseline(workload: str, new_tokens: int):
    """Oracle = CPU port of the reference algorithm (branch-form LocalLoRA on all tokens, mask-sum routing, every encoder), torch fp32.
    Bounded sample of the SAME workload: one sample (batch 1, the reference's eval batch), `new_tokens` greedy tokens, timed with 1 and
    with 8 of the 32 decoder layers (about a minute of CPU); the per-layer cost from the difference is scaled to 32, the fixed cost
    (encoders, projectors, splice, lm_head) is measured in full."""
    from modelcompose_amd import synthetic
    from oracle import pipeline
    # torch's intra-op pool degrades badly far beyond ~32 threads on these op sizes (256 threads measured 70x slower)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    modals, sentinels = WORKLOADS[workload][0], WORKLOADS[workload][1]
    times = {}
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"      # weights are only GENERATED on the GPU, then moved to host fp32
    lo, hi = 1, 8                                                 # the deep point is a quarter of the real depth, not 3 layers
    for nl in (lo, hi):
        meta = workload_meta(workload, nl)
        sd = synthetic.synthetic_state_dict(meta, device=gen_dev, seed=7, dtype=torch.float32)
        sd = {k: v.cpu() for k, v in sd.items()}
        om = pipeline.OracleModel.from_state_dict(sd, meta)
        ids = synthetic.synthetic_prompt(1, sentinels)
        mi = synthetic_inputs(modals, 1, gen_dev, 3)
        mi = {k: ({kk: (vv.float().cpu() if vv.is_floating_point() else vv.cpu()) for kk, vv in v.items()} if isinstance(v, dict) else v.float().cpu())
              for k, v in mi.items()}
        if "point" in mi:
            meta["fps_start"] = [0]
        with torch.no_grad():
            if nl == lo:
                om.generate(ids, mi, max_new_tokens=1, ignore_eos=True)               # untimed warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            om.generate(ids, mi, max_new_tokens=new_tokens, ignore_eos=True)
            times[nl] = time.perf_counter() - t0
        del sd, om
    per_layer = max((times[hi] - times[lo]) / (hi - lo), 1e-9)
    fixed = max(times[lo] - lo * per_layer, 0.0)
    full = fixed + 32 * per_layer
    measured = None
    mpath = os.path.join(ROOT, "profiles", "cpu_baseline_full.json")          # a committed full-depth run of `--cpu-baseline-full` (same oracle, same sample)
    if os.path.exists(mpath):
        try:
            measured = json.load(open(mpath))
        except Exception:
            measured = None
    live = {"value": 1.0 / full, "seconds_per_sample": round(full, 1), "extrapolated": True, "cores": cores,
            "how": f"this run, this host: {lo} and {hi} of 32 decoder layers timed ({times[lo]:.1f}s, {times[hi]:.1f}s), per-layer cost x32 + measured fixed cost"}
    if measured and measured.get("value"):
        # the number of record is the MEASURED full-depth run (nothing extrapolated); the bounded live sample of this run
        # stays beside it as the cross-check that this host is in the same regime (python bench.py --cpu-baseline-full re-measures it: ~3 min)
        return {"value": measured["value"], "unit": "samples/s", "cores": measured.get("cores", cores), "kind": "port", "extrapolated": False,
                "sample": measured.get("sample"), "measured_by": "profiles/cpu_baseline_full.json (python bench.py --cpu-baseline-full on an MI355X box's host)",
                "live_check": live}
    return {"value": 1.0 / full, "unit": "samples/s", "cores": cores, "kind": "port", "extrapolated": True,
            "measured_full_depth_run": measured,
            "sample": f"oracle (torch fp32 CPU port of the reference path, batch 1 as the reference's eval loop) on one sample of the same "
                      f"workload ({WORKLOADS[workload][3]}, {new_tokens} greedy tokens), timed with {lo} and {hi} of 32 decoder layers "
                      f"({times[lo]:.1f}s, {times[hi]:.1f}s): per-layer cost x32 + measured fixed cost (all encoders, projectors, splice, "
                      f"lm_head) = {full:.1f}s per sample; a full 32-layer run of this sample in the build container (8 cores) took 352 s, "
                      f"and two rows x 17 tokens of it 831 s (oracle/gen_golden.py g15)"}


# ------------------------------------------------------------------------------------------------------------------- other configs
def secondary_runs(new_tokens: int):
    """BASELINE configs[1] (vision, B = 16), configs[3] (mcub4, the 4-modality model) and configs[4] (stage-2 finetune step) for a few
    steps each, AFTER the timed region of the headline workload and after its model has been freed: one child process per config
    (`python bench.py --workload X --no-profile --no-cpu-baseline --no-secondary`, i.e. the same timing code path with its own barrier +
    synchronize bracket), whose JSON line is attached in compact form.  Not part of `value`."""
    import subprocess
    out = {}
    # "iav_128": the headline workload at the REFERENCE's decode budget, max_new_tokens=128 (eval/model_multimodal_qa_loader.py:101; SURVEY §8d
    # "report also 128"): the HBM-bound share of a step roughly triples
    # "iav_fp16": the headline workload on the reference's own storage dtype (fp16: builder.py:41, :162, :185), with its roofline objects
    # "iav_b1_128": the reference's eval geometry - batch 1, 128 greedy tokens, one generate() call per sample, nothing overlapped
    #               (eval/model_multimodal_qa_loader.py:49-52, :94-102): time to first token, time per token, the M = 1 decode roofline
    for name, steps, warm in (("iav_fp16", 4, 2), ("iav_b1_128", 4, 2), ("iav_128", 3, 2), ("vision", 6, 2), ("mcub4", 4, 2), ("train", 10, 3)):
        wl, nt = ("iav", 128) if name in ("iav_128", "iav_b1_128") else (("iav", new_tokens) if name == "iav_fp16" else (name, new_tokens))
        extra = {"iav_fp16": ["--dtype", "fp16"], "iav_b1_128": ["--batch", "1", "--no-pipeline"], "train": []}.get(name, ["--no-profile"])
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", str(steps), "--warmup", str(warm),
               "--no-cpu-baseline", "--no-secondary", "--new-tokens", str(nt)] + extra
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MC_BENCH_FORCE_DIST")}
        try:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
            line = next((l for l in reversed(r.stdout.strip().splitlines()) if l.startswith("{")), None)
            if r.returncode != 0 or line is None:
                out[name] = {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
                continue
            j = json.loads(line)
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"],
                         "warmup": j["warmup"], "n_gpus": j["n_gpus"], "dtype": j["dtype"],
                         "config": {k: v for k, v in j["config"].items() if k in ("workload", "per_gpu_batch", "new_tokens", "layers", "spliced_length",
                                                                                 "pipelined", "parallelism", "trainable_params")},
                         "wall_s_incl_model_build": round(time.perf_counter() - t0, 1)}
            for k in ("roofline", "roofline_decode", "stages_ms", "sequential", "pipelined"):
                if j.get(k):
                    out[name][k] = j[k]
            if name == "iav_b1_128" and j.get("stages_ms"):
                stg = j["stages_ms"]
                out[name]["ms_to_first_token"] = round(stg["encode"] + stg["prefill"], 3)
                out[name]["ms_per_token"] = round(stg["decode"] / max(nt - 1, 1), 4)
        except Exception as e:                                           # evidence only: never lose the headline line to it
            out[name] = {"error": repr(e)[:300]}
    return out


# ------------------------------------------------------------------------------------------------------------------- train (configs[4])
def train_step_flops(meta, B, L, n_targets, n_img_tokens):
    """Algorithmic FLOPs of one configs[4] step (what the function needs, not what a particular implementation spends):
    frozen-base forward + input-gradient pass of the 224 decoder linears (no base weight gradients), causal attention forward + backward
    (2.5x), LoRA rank-r branch of the token's OWN adapter forward + input gradient + weight gradient, lm_head forward + input gradient on
    the target rows only, CLIP-L forward (frozen), projector forward + backward."""
    Hd, I, V, Ln, r = meta["hidden_size"], meta["intermediate_size"], meta["vocab_size"], meta["num_hidden_layers"], meta["lora_r"]
    T = B * L
    p_lin = Ln * (4 * Hd * Hd + 3 * Hd * I)
    base = 2 * (2.0 * T * p_lin)
    attn = 3.5 * B * (4.0 * L * L * Hd / 2) * Ln
    lora = 3 * 2.0 * T * r * Ln * (4 * (Hd + Hd) + 3 * (Hd + I))
    head = 2 * 2.0 * B * n_targets * V * Hd
    c = meta.get("clip") or {}
    D, Di, Lc = c.get("hidden_size", 1024), c.get("intermediate_size", 4096), c.get("num_hidden_layers", 24) - 1
    Tc = n_img_tokens + 1
    clip = B * (2.0 * Tc * Lc * (4 * D * D + 2 * D * Di) + 4.0 * Tc * Tc * D * Lc + 2.0 * n_img_tokens * 588 * D)
    proj = B * n_img_tokens * 2.0 * (D * Hd + Hd * Hd) * 3 - B * n_img_tokens * 2.0 * D * Hd      # fwd + wgrad + dgrad, no dgrad into the frozen tower
    return {"base_linears": base, "attention": attn, "lora": lora, "lm_head": head, "clip": clip, "projector": proj,
            "total": base + attn + lora + head + clip + proj}


def _lib_name():
    from modelcompose_amd import _lib
    return _lib.storage_name()


def train_main(args, world, rank, local):
    """BASELINE configs[4]: stage-2 finetune step of the vision LocalLoRA model (adapters default + vision, r=128), per-GPU batch 4
    synthetic image-text pairs (683-token spliced sequence, the last 60 tokens are targets), bf16 compute, fp32 master weights;
    a step = forward + backward + bucketed RCCL gradient all-reduce + AdamW.  value = samples/s over all ranks."""
    from modelcompose_amd import synthetic
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    dev = torch.device("cuda", local)
    meta = synthetic.vicuna7b_meta(("vision",), None, layers=args.layers)
    meta["lora_dropout"] = 0.0
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    model = build_from_state_dict(meta, sd, device=dev)
    st = MultimodalTrainStep(model, lr=2e-4, overlap_wgrad=not args.no_overlap, force_exchange=DIST)
    del sd
    model._raw = {}
    torch.cuda.empty_cache()
    B = args.batch or 4
    ids = synthetic.synthetic_prompt(B, [-200], seed=rank).to(dev)
    labels = ids.clone()
    labels[:, :-60] = -100
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    pixels = torch.randn(B, 3, 336, 336, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)

    def barrier():
        if DIST or world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = st.step(ids, labels, {"vision": pixels})
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = st.step(ids, labels, {"vision": pixels})
    barrier()
    dt = time.perf_counter() - t0
    if DIST or world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        fl = train_step_flops(meta, B, 683, 60, 576)
        roof = {"bound": "mfma", "kernel": "whole step (forward + backward + AdamW; dominant kernel gemm_tile256_kernel)",
                "achieved": round(fl["total"] / (ms_step * 1e-3) / 1e12, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl["total"] / (ms_step * 1e-3) / 1e12 / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                "algorithmic_flops_per_step": {k: float(v) for k, v in fl.items()}}
        if not args.no_profile:
            # the dominant kernel alone: two more steps with every GEMM on the main stream (side-stream launches would overlap the bracketed
            # ones) and one HIP-event bracket per gemm_tile256_kernel launch
            from modelcompose_amd import _lib
            L_ = _lib.lib()
            # (with no side stream forward_backward would leave the tile choice at its default - 192-column tiles for these under-filled
            # launches - while the timed steps, which fill the idle CUs from the side stream, run 256-column tiles: the bracketed launches are
            # kept on the timed steps' tile choice.  The two extra optimizer steps come after final_loss has been taken.)
            final_loss = float(loss.item())
            ws, st._wstream = st._wstream, None
            tile_opt = 0 if ws is not None else 1
            torch.cuda.synchronize()
            L_.mc_gemm_set_option(b"tile192", tile_opt)
            L_.mc_gemm_profile_enable(1)
            try:
                for _ in range(2):
                    st.step(ids, labels, {"vision": pixels})
                torch.cuda.synchronize()
            finally:
                L_.mc_gemm_profile_enable(0)
                L_.mc_gemm_set_option(b"tile192", 1)
                st._wstream = ws
            ms_, fl_, n_ = C.c_double(0), C.c_double(0), C.c_int64(0)
            L_.mc_gemm_profile_read(C.byref(ms_), C.byref(fl_), C.byref(n_))
            ach = (fl_.value / max(ms_.value, 1e-9)) / 1e9
            roof["dominant_kernel"] = {"kernel": "gemm_tile256_kernel", "achieved": round(ach, 2), "unit": "TFLOP/s",
                                       "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "launches_per_step": int(n_.value // 2),
                                       "ms_per_step": round(ms_.value / 2, 3), "flops_per_step_launched": fl_.value / 2,
                                       "tile192_option": tile_opt,
                                       "note": "HIP events per launch, non-overlapped pass (the timed steps overlap weight-gradient GEMMs on a side stream); "
                                               "same tile selection as the timed steps"}
        print(json.dumps({
            "metric": "samples/sec (whole node) stage-2 finetune step, composed Vicuna-7B", "value": round(world * B * args.steps / dt, 4),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": _lib_name(), "data": "synthetic",
            "config": {"workload": "configs[4]: stage-2 finetune step, vision LocalLoRA Vicuna-7B (r128, default+vision adapters), "
                                   f"batch {B} per GPU, 683-token sequences, fwd+bwd+all-reduce+AdamW", "per_gpu_batch": B,
                       "layers": args.layers, "parallelism": f"ddp{world}", "trainable_params": int(st.n_params),
                       "final_loss": final_loss if not args.no_profile else float(loss.item())},
            "roofline": roof}), flush=True)
    if DIST or world > 1:
        torch.distributed.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------- profiled pass
PK_NAMES = ["qkv_gemm", "rope_kv", "attention", "o_gemm", "rms", "gate_up_gemm", "down_gemm", "lm_head_gemm", "other"]


def profiled_pass(model, step_fn, cfg, B, new_tokens, n_steps):
    """Roofline evidence, outside the throughput timing.  Pass A (graph decode, as shipped): GEMM events + stage events.  Pass B (one
    launch per kernel): every decode kernel class bracketed by events."""
    from modelcompose_amd import _lib
    L = _lib.lib()
    # ---- pass A
    L.mc_gemm_profile_enable(1)
    stages = {"encode": 0.0, "prefill": 0.0, "decode": 0.0}
    lens = None
    evs = []
    for _ in range(n_steps):
        ev = {}
        step_fn(stage_events=ev)
        evs.append(ev)
    torch.cuda.synchronize()
    L.mc_gemm_profile_enable(0)
    for ev in evs:
        for k in stages:
            stages[k] += ev[k][0].elapsed_time(ev[k][1]) / n_steps
        lens = ev["spliced_lens"]
    graph_active = model.runtime_option("graph_active")
    ms, fl, n, by = C.c_double(0), C.c_double(0), C.c_int64(0), C.c_double(0)
    L.mc_gemm_profile_read(C.byref(ms), C.byref(fl), C.byref(n))
    L.mc_gemm_profile_read_bytes(C.byref(by))
    achieved = (fl.value / max(ms.value, 1e-9)) / 1e9          # flops/ms -> TFLOP/s
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # HBM bytes per launch from separate rocprofv3 PMC passes (tools/profile_round.sh)
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("workload") == cfg["workload_name"] and tj.get("per_gpu_batch") == B:     # only a pass over THIS workload counts
                traffic = tj.get("gemm_tile256_kernel_bytes_per_launch")
                traffic_src = (f"profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, round {tj.get('round')}, {tj.get('date')}; "
                               f"scope: {tj.get('scope', 'all launches of one prefill')}; algorithmic bytes of the same launches: "
                               f"{tj.get('algorithmic_bytes_per_launch')})")
        except Exception:
            traffic = None
    by_class = {}
    for cname, (k0_, k1_) in (("decoder_layers_K_ge_4096", (4096, 1 << 30)), ("encoder_towers_K_lt_4096", (0, 4095))):
        ms_c, fl_c, n_c = C.c_double(0), C.c_double(0), C.c_int64(0)
        L.mc_gemm_profile_read_range(k0_, k1_, C.byref(ms_c), C.byref(fl_c), C.byref(n_c))
        if n_c.value:
            a_c = fl_c.value / max(ms_c.value, 1e-9) / 1e9
            by_class[cname] = {"achieved": round(a_c, 2), "frac": round(a_c / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "launches": int(n_c.value),
                               "ms_per_step": round(ms_c.value / n_steps, 3)}
    roofline = {"bound": "mfma", "kernel": "gemm_tile256_kernel", "achieved": round(achieved, 2), "peak": MFMA_BF16_DENSE_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launches": int(n.value), "avg_launch_us": round(ms.value / max(n.value, 1) * 1e3, 2),
                "avg_flops_per_launch": fl.value / max(n.value, 1), "avg_algorithmic_bytes_per_launch": by.value / max(n.value, 1),
                "share_of_step": round(ms.value / n_steps / max(sum(stages.values()), 1e-9), 4),
                # the same events split by the launch's reduction length: the Vicuna decoder's four linears (K = 4096 / 11008) and the
                # encoder towers' (K = 768 / 1024: a tile's prologue + store-bound epilogue weigh 28 % there against 7 %)
                "by_class": by_class}
    # ---- decode roofline: algorithmic bytes of one step = every decoder weight + lm_head once, K and V of every cached key once
    Hd, I, V, Ln, H, D = cfg["hidden"], cfg["inter"], cfg["vocab"], cfg["layers"], cfg["heads"], cfg["head_dim"]
    wbytes = {"qkv_gemm": 2.0 * 3 * Hd * Hd, "o_gemm": 2.0 * Hd * Hd, "gate_up_gemm": 2.0 * 2 * I * Hd, "down_gemm": 2.0 * Hd * I,
              "lm_head_gemm": 2.0 * V * Hd}
    n_dec = new_tokens - 1
    # step i (0-based) attends lens + i + 1 keys per sequence
    kv_keys = float(sum(int(l) + i + 1 for l in lens for i in range(n_dec))) / max(n_dec, 1)      # summed over the batch, average per step
    kv_bytes_layer = kv_keys * H * D * 2 * 2.0
    step_bytes = Ln * (wbytes["qkv_gemm"] + wbytes["o_gemm"] + wbytes["gate_up_gemm"] + wbytes["down_gemm"] + kv_bytes_layer) + wbytes["lm_head_gemm"]
    step_ms = stages["decode"] / max(n_dec, 1)
    dec = {"bound": "hbm", "kernel": "decode step (5 launches per layer + lm_head + argmax, hipGraph replay)", "unit": "GB/s", "peak": HBM_PEAK_GBS,
           "achieved": round(step_bytes / max(step_ms, 1e-9) / 1e6, 1), "frac": round(step_bytes / max(step_ms, 1e-9) / 1e6 / HBM_PEAK_GBS, 4),
           "bytes_per_step": step_bytes, "ms_per_step": round(step_ms, 4), "graph_active": bool(graph_active),
           "avg_keys_per_sequence": round(kv_keys / B, 1)}
    # ---- pass B: per-kernel-class durations of the decode step, one launch per kernel with events around it
    _lib.check(L.mc_llm_set_option(model._handle, b"profile", 1), "profile")
    step_fn()
    torch.cuda.synchronize()
    _lib.check(L.mc_llm_set_option(model._handle, b"profile", 0), "profile")
    nk = L.mc_llm_profile_kinds()
    per = {}
    for phase, pname in ((0, "prefill"), (1, "decode")):
        tm, cnt = (C.c_double * nk)(), (C.c_int64 * nk)()
        _lib.check(L.mc_llm_profile_read(model._handle, phase, tm, cnt), "profile_read")
        per[pname] = {PK_NAMES[k]: (tm[k], cnt[k]) for k in range(nk) if cnt[k]}
    kern = {}
    for name, (tms, cnt) in per["decode"].items():
        us = tms / cnt * 1e3
        b = wbytes.get(name, kv_bytes_layer if name == "attention" else None)
        kern[name] = {"avg_us": round(us, 2), "launches": int(cnt)}
        if b is not None:
            kern[name].update(bytes=b, gbs=round(b / us / 1e3, 1), frac=round(b / us / 1e3 / HBM_PEAK_GBS, 4))
    dec["kernels"] = kern
    # fabric-side read traffic of the same kernels from the PMC passes kept under profiles/ (tools/pmc_decode_chain.sh, tools/pmc_decode_attn.sh:
    # FETCH_SIZE with the gfx950 correction, standalone launches at the headline decode shape - 48 rows, 2809 keys), as a ratio to the algorithmic bytes
    try:
        pm = os.path.join(ROOT, "profiles", "r06_pmc")
        ch = json.load(open(os.path.join(pm, "decode_chain_traffic_r06b.json")))["kernels"]
        at = json.load(open(os.path.join(pm, "decode_attn_traffic_r06b.json")))
        dec["traffic_ratio"] = {"qkv_gemm": ch["qkv"]["ratio_to_algorithmic"], "o_gemm": ch["o"]["ratio_to_algorithmic"],
                                "gate_up_gemm": ch["gate_up"]["ratio_to_algorithmic"], "down_gemm": ch["down"]["ratio_to_algorithmic"],
                                "attention": at["ratio_to_algorithmic"],
                                "source": "profiles/r06_pmc/decode_chain_traffic_r06b.json, decode_attn_traffic_r06b.json (rocprofv3 --pmc FETCH_SIZE, separate passes)"}
    except Exception:
        dec["traffic_ratio"] = None
    dec["kernels_note"] = "per-class averages from a pass with one launch per kernel, each bracketed by HIP events on the launch stream (the shipped path replays a graph)"
    pre = {name: {"total_ms": round(tms, 3), "launches": int(cnt)} for name, (tms, cnt) in per["prefill"].items()}
    return roofline, dec, {k: round(v, 3) for k, v in stages.items()}, pre


# ------------------------------------------------------------------------------------------------------------------- generate workloads
def generate_main(args, world, rank, local):
    from modelcompose_amd import _lib, synthetic
    from modelcompose_amd.dist import gather_ids, gather_logits
    from modelcompose_amd.model.builder import build_from_state_dict
    name = args.workload
    modals, sentinels, defB, msuffix, desc = WORKLOADS[name]
    if args.pipeline is None:
        args.pipeline = name in ("iav", "vision")
    dev = torch.device("cuda", local)
    meta = workload_meta(name, args.layers)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    model = build_from_state_dict(meta, sd, device=dev)
    # the load-time composition (the path's "fused AXPY over state_dict tensors"): algorithmic bytes = W read ONCE per linear + one dense W' written
    # per routed adapter + the LoRA factors, over the device time of the composition loop (events around it in finalize(); weights already in HBM)
    compose_roofline = None
    if getattr(model, "compose_kernel_ms", 0) and getattr(model, "compose_bytes", 0):
        gbs = model.compose_bytes / model.compose_kernel_ms / 1e6
        compose_roofline = {"bound": "hbm", "kernel": "compose_tile_kernel (W' = bf16((W + sum s B A) diag(g)) for every routed adapter of every linear; one persistent launch per model)",
                            "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                            "bytes": model.compose_bytes, "ms": round(model.compose_kernel_ms, 3), "launches": int(model.compose_launches),
                            "avg_launch_us": round(model.compose_kernel_ms / max(model.compose_launches, 1) * 1e3, 2),
                            "loop_wall_ms": round(model.compose_ms, 3),
                            "note": "one-time, at load; achieved = algorithmic bytes / sum of the launches' HIP-event durations; loop_wall_ms is the "
                                    "whole composition loop incl. the host-issued transposes of the LoRA factors between the launches"}
    model.use_graph = not args.no_graph
    _lib.check(_lib.lib().mc_llm_set_option(model._handle, b"use_graph", 0 if args.no_graph else 1), "set_option")
    del sd
    model._raw = {}
    torch.cuda.empty_cache()
    B = args.batch or defB
    ids = synthetic.synthetic_prompt(B, sentinels, seed=rank).to(dev)
    mi = synthetic_inputs(modals, B, dev, 100 + rank)
    if "point" in modals:
        model.model.modal_encoders["point"].fps_start = torch.zeros(B, dtype=torch.long)

    want_lg = args.gather == "logits"
    if want_lg:
        _lib.check(_lib.lib().mc_llm_set_option(model._handle, b"graph_logits", 1), "set_option")
    gkw = {"return_step_logits": True} if want_lg else {}

    def gather(out):
        if want_lg:
            out, lg = out
            gather_logits(lg, world, force=DIST, equal_shapes=True)
        return gather_ids(out[:, ids.shape[1]:], world, force=DIST, equal_shapes=True)

    def step(**kw):
        return gather(model.generate(ids, modal_inputs=mi, max_new_tokens=args.new_tokens, ignore_eos=True, **gkw, **kw))

    def step_local(**kw):
        # rank 0's evidence passes run after the other ranks have left: no collective in them
        return model.generate(ids, modal_inputs=mi, max_new_tokens=args.new_tokens, ignore_eos=True, **kw)

    def barrier():
        if DIST or world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        if not args.pipeline:
            for _ in range(n):
                step()
            return
        for out in model.generate_pipelined(((ids, mi) for _ in range(n)), max_new_tokens=args.new_tokens, ignore_eos=True, **gkw):
            gather(out)

    # the pipelined loop alternates two generation pipelines (own KV cache, workspace, decode graph): both must have run once before the
    # timed region, whatever W the caller asked for
    priming = max(0, 2 - args.warmup) if args.pipeline else 0
    run_steps(priming + args.warmup)
    barrier()
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if DIST or world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if DIST or world > 1:
            torch.distributed.destroy_process_group()
        return
    # host time spent issuing the towers' launches for the last timed batch (Python-issued: several hundred launches per batch and rank)
    host_issue_ms = round(float(getattr(model, "last_encode_issue_ms", 0.0)), 2)
    feats, _ = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
    spliced = int(ids.shape[1] - len(sentinels) + sum(f.shape[1] for f in feats.values()))
    value = world * B * args.steps / dt
    line = {
        "metric": f"samples/sec (whole node) composed-Vicuna-7B greedy gen, {msuffix}",
        "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": _lib.storage_name(), "data": "synthetic",
        "config": {"workload": f"{desc}; batch {B} per GPU, {spliced}-token spliced prompt, {args.new_tokens} greedy tokens",
                   "workload_name": name, "per_gpu_batch": B, "new_tokens": args.new_tokens, "layers": args.layers, "spliced_length": spliced,
                   "block_tokens": {m: int(f.shape[1]) for m, f in feats.items()}, "adapters": list(model.modal_names),
                   "parallelism": f"dp{world}", "gathered": args.gather, "decode_graph_requested": not args.no_graph, "pipelined": bool(args.pipeline),
                   "pipeline_priming_steps": priming},
        "roofline": None, "roofline_decode": None, "roofline_compose": compose_roofline,
    }
    del feats
    if not args.no_profile:
        # per-kernel evidence: a separate pass of sequential generate() calls with one HIP-event bracket per launch on the launch stream
        # (overlapped launches of the pipelined loop would inflate each other's durations)
        cfgd = dict(hidden=meta["hidden_size"], inter=meta["intermediate_size"], vocab=meta["vocab_size"], layers=args.layers,
                    heads=meta["num_attention_heads"], head_dim=meta["hidden_size"] // meta["num_attention_heads"], workload_name=name)
        roofline, dec, stages, pre = profiled_pass(model, step_local, cfgd, B, args.new_tokens, min(args.steps, 3))
        line["roofline"], line["roofline_decode"], line["stages_ms"] = roofline, dec, stages
        line["prefill_kernel_classes_ms"] = pre
        line["config"]["decode_graph"] = dec["graph_active"]
    if world == 1 and not args.no_profile and name != "mcub4":
        # informational, outside the timed region and outside the contract's fields: the same K batches through the other eval loop
        was = args.pipeline
        args.pipeline = not was
        run_steps(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        args.pipeline = was
        line["sequential" if was else "pipelined"] = {
            "value": round(B * args.steps / dtp, 4), "unit": "samples/s", "ms_per_step": round(dtp / args.steps * 1e3, 3),
            "note": ("bench.py --no-pipeline: one generate() call per batch, nothing overlapped; same tokens" if was else
                     "bench.py --pipeline: decode of batch i overlapped with encoders + prefill of batch i+1; same tokens")}
    del model
    torch.cuda.empty_cache()
    if world == 1 and not DIST and not args.no_secondary and name == "iav" and args.layers == 32:
        line["secondary"] = secondary_runs(args.new_tokens)
    line["host"] = {"pinned_cpus": len(PINNED_CPUS) if PINNED_CPUS else None, "cpu_count": os.cpu_count(),
                    "encode_issue_ms_last_batch": host_issue_ms}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline_full(name, args.new_tokens) if args.cpu_baseline_full else cpu_baseline(name, args.new_tokens)
    print(json.dumps(line), flush=True)
    if DIST or world > 1:
        torch.distributed.destroy_process_group()


# ------------------------------------------------------------------------------------------------------------------- N-rank launcher
def rank_commands(n: int, argv: list[str], port: int, base_env=None):
    """The N child processes of `python bench.py --gpus N`: [(command, environment)] - one per GPU, the same arguments, the rank
    environment torch.distributed.run would have set (MCUB-4.sh:42-58 starts one worker per GPU chunk the same way)."""
    base = dict(os.environ if base_env is None else base_env)
    out = []
    for r in range(n):
        env = dict(base)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out.append(([sys.executable, os.path.abspath(__file__)] + list(argv), env))
    return out


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv: list[str], deadline_s: float | None = None) -> int:
    """Parent of an N-rank run.  Never touches the GPU (no HIP call, no torch.cuda.is_available()): it only starts the children, passes
    rank 0's stdout through, keeps the other ranks' output for the error case, and returns the worst exit status.  A rank that fails
    takes the others down (they would hang in the next collective).  Clean-up is unconditional: every child runs in its own
    session (a signal aimed at the parent's group does not reach it half-way through a collective), and whatever ends the parent - the
    poll loop's own verdict, SIGTERM / SIGINT / SIGHUP from a caller's timeout, an exception, the wall-clock deadline
    (MC_BENCH_LAUNCH_TIMEOUT seconds, default 7200) - terminates the exact PIDs it started, then kills what is left after a grace period."""
    import signal
    import subprocess
    import tempfile
    port = int(os.environ.get("MASTER_PORT", "0")) or _free_port()
    if deadline_s is None:
        deadline_s = float(os.environ.get("MC_BENCH_LAUNCH_TIMEOUT", "7200"))
    procs, logs = [], []

    def reap(grace=5.0):
        live = [p for p in procs if p.poll() is None]
        for p in live:
            try:
                p.terminate()
            except OSError:
                pass
        t_end = time.time() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    p.kill()
                    p.wait(timeout=5)
                except (OSError, subprocess.TimeoutExpired):
                    pass

    class _Stop(Exception):
        pass

    def on_signal(signum, _frame):
        raise _Stop(signum)

    old_handlers = {}
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            old_handlers[sg] = signal.signal(sg, on_signal)
        except (ValueError, OSError):            # not the main thread: the finally clause below still cleans up
            pass
    rc = 0
    t_start = time.time()
    try:
        for r, (cmd, env) in enumerate(rank_commands(n, argv, port)):
            # ranks > 0 write to a file, not a pipe nobody drains (a rank that fills a 64-KiB pipe with warnings would block inside a collective)
            log = None if r == 0 else tempfile.TemporaryFile(mode="w+")
            logs.append(log)
            procs.append(subprocess.Popen(cmd, env=env, stdout=log, stderr=None if r == 0 else subprocess.STDOUT, text=True,
                                          start_new_session=True))
        alive = set(range(n))
        while alive:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = rc or code
                    if r != 0:
                        logs[r].seek(0)
                        sys.stderr.write(f"[bench.py launcher] rank {r} exited with {code}:\n{logs[r].read()[-2000:]}\n")
                    for o in alive:                      # exact PIDs we started, nothing by pattern
                        procs[o].terminate()
            if alive and time.time() - t_start > deadline_s:
                sys.stderr.write(f"[bench.py launcher] ranks {sorted(alive)} still running after {deadline_s:.0f} s: tearing the job down\n")
                rc = rc or 124
                break
            time.sleep(0.2)
    except _Stop as e:
        sys.stderr.write(f"[bench.py launcher] signal {e.args[0]}: stopping {sum(p.poll() is None for p in procs)} rank(s)\n")
        rc = rc or 128 + int(e.args[0])
    finally:
        for sg in old_handlers:                      # a second signal during the grace period must not abort the clean-up
            try:
                signal.signal(sg, signal.SIG_IGN)
            except (ValueError, OSError):
                pass
        reap()
        for sg, h in old_handlers.items():
            try:
                signal.signal(sg, h)
            except (ValueError, OSError):
                pass
        for log in logs:
            if log is not None:
                log.close()
    return rc


def main():
    args = parse()
    if args.dtype:                                   # before anything imports the package: one storage dtype (one library) per process
        os.environ["MC_STORAGE_DTYPE"] = args.dtype
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or bare, which "
                         f"starts the ranks itself)")
    if os.environ.get("MC_BENCH_LAUNCH_PROBE") == "1":
        # launcher self-test (tests/test_bench_launcher_cpu.py): a rank reports the environment it was started with and leaves, GPU untouched
        print(json.dumps({"probe": True, "rank": rank, "local_rank": local, "world": world, "gpus": args.gpus, "pid": os.getpid(),
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}", "steps": args.steps,
                          "cpus": PINNED_CPUS}), flush=True)
        pf = os.environ.get("MC_BENCH_PROBE_PIDFILE")
        if pf:                                           # launcher clean-up tests: every rank leaves its pid, then waits to be stopped
            with open(f"{pf}.{rank}", "w") as f:
                f.write(str(os.getpid()))
            if os.environ.get("MC_BENCH_PROBE_IGNORE_TERM") == "1":      # a rank stuck in a collective: only the launcher's SIGKILL ends it
                import signal
                signal.signal(signal.SIGTERM, signal.SIG_IGN)
            time.sleep(float(os.environ.get("MC_BENCH_PROBE_SLEEP", "0")))
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # MC_BENCH_SHARE_GPU=1 + MC_BENCH_BACKEND=gloo (functional test only, never a measurement): N ranks on a box with fewer GPUs - rank r
    # uses device r mod #devices and the collectives run over gloo (RCCL refuses two ranks on one device) - so that the N > 1 path of this
    # file (launcher, barriers, MAX all-reduce of the timing, rank-major gather of the ids) runs with N real processes on the 1-GPU test box
    if os.environ.get("MC_BENCH_SHARE_GPU") == "1":
        local = local % torch.cuda.device_count()
    backend = os.environ.get("MC_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if DIST or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.workload == "train":
        return train_main(args, world, rank, local)
    return generate_main(args, world, rank, local)


if __name__ == "__main__":
    main()
