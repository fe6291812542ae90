#!/usr/bin/env python
"""Headline benchmark: samples/s of composed-Vicuna-7B greedy generation (BASELINE.json metric).

A "step" = one pass of the hot path over one batch of synthetic input on every rank:
  CLIP-ViT-L/14-336 encode -> mlp2x_gelu projector -> splice -> LocalLoRA-composed Vicuna-7B prefill ->
  32 greedy tokens (device-resident loop) -> all-gather of the generated ids (N > 1).
Workload at every N = BASELINE.json configs[1]: vision-only Vicuna-7B bf16, batch 16 synthetic 336 px images per GPU
(weak scaling: per-GPU work fixed).  Inputs and weights are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0 (contract in the task prompt) with two extra objects:
  roofline     – the dominant kernel (gemm_tile256_kernel, MFMA-bound): algorithmic FLOPs per launch / average launch
                 duration measured live with HIP events on the launch stream during the timed steps
  cpu_baseline – the oracle (CPU port of the reference algorithm, torch fp32, all host cores) on a bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# MC_BENCH_FORCE_DIST=1 runs the N > 1 code path (RCCL init, barriers, gathers, the MAX all-reduce of the timing) in a world of one
# process: the only way to exercise it on a single-GPU box
DIST = os.environ.get("MC_BENCH_FORCE_DIST", "") == "1"
MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBS = 8000.0                      # spec HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--new-tokens", type=int, default=32)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer decoder layers (invalidates the number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--pipeline", action="store_true", help="generate workload: two generation pipelines on two HIP streams (decode of batch i "
                    "beside the prefill of batch i+1, MultimodalLlamaForCausalLM.generate_pipelined); every step still is one full batch")
    ap.add_argument("--no-overlap", action="store_true", help="debug only (train): weight-gradient / rank-projection GEMMs on the main stream")
    ap.add_argument("--gemm-debug", type=int, default=0, help="debug only: mc_gemm_debug word (A/B of kernel variants; e.g. 2048 = no 192-column tiles)")
    ap.add_argument("--workload", default="generate", choices=["generate", "train", "mcub4", "iav"],
                    help="generate = BASELINE configs[1] (the headline metric); train = configs[4], the stage-2 finetune step "
                         "(forward + backward + gradient all-reduce + AdamW), per-GPU batch 4; mcub4 = configs[3], the 4-modality "
                         "composed model on MCUB-4-shaped inputs (image + 10 s audio + 8-frame video + 8192-point cloud), per-GPU batch 2; "
                         "iav = the metric string's literal inputs (336 px image + 10 s audio + 8-frame video) on configs[2]'s 3-way composed "
                         "model (online-merge-reset 3 x 0.333), per-GPU batch 4")
    return ap.parse_args()


def cpu_baseline(new_tokens: int):
    """Oracle = CPU port of the reference algorithm (branch-form LocalLoRA on all tokens, mask-sum routing), torch fp32.
    Bounded sample: BASELINE config 1 (1 image, ~683-token prompt, `new_tokens` greedy tokens) with 2 and with 4 of the 32
    decoder layers; the per-layer cost from the difference is scaled to 32 layers, fixed costs (CLIP-L, projector, lm_head,
    splice) are measured in full."""
    from modelcompose_amd import synthetic
    from oracle import pipeline
    # torch's intra-op pool degrades badly far beyond ~32 threads on these op sizes (256 threads measured 70x slower)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    times = {}
    gen_dev = "cuda" if torch.cuda.is_available() else "cpu"      # weights are only GENERATED on the GPU, then moved to host fp32
    for nl in (2, 4):
        meta = synthetic.vicuna7b_meta(("vision",), None, layers=nl)
        sd = synthetic.synthetic_state_dict(meta, device=gen_dev, seed=7, dtype=torch.float32)
        sd = {k: v.cpu() for k, v in sd.items()}
        om = pipeline.OracleModel.from_state_dict(sd, meta)
        ids = synthetic.synthetic_prompt(1, [-200])
        px = torch.randn(1, 3, 336, 336)
        with torch.no_grad():
            om.generate(ids, {"vision": px}, max_new_tokens=2, ignore_eos=True)      # untimed warm-up (thread pool, allocator)
            t0 = time.perf_counter()
            om.generate(ids, {"vision": px}, max_new_tokens=new_tokens, ignore_eos=True)
            times[nl] = time.perf_counter() - t0
        del sd, om
    per_layer = max((times[4] - times[2]) / 2.0, 1e-9)
    fixed = max(times[2] - 2 * per_layer, 0.0)
    full = fixed + 32 * per_layer
    return {"value": 1.0 / full, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch fp32 CPU port of the reference path) on BASELINE config 1: 1x336px image, 683-token prompt, "
                      f"{new_tokens} greedy tokens; timed with 2 and 4 of 32 decoder layers ({times[2]:.2f}s, {times[4]:.2f}s), "
                      f"per-layer cost x32 + measured fixed cost (CLIP-L/14-336, projector, lm_head) = {full:.1f}s per sample"}


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
    B = 4 if args.batch == 16 else args.batch
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
        print(json.dumps({
            "metric": "samples/sec (whole node) stage-2 finetune step, composed Vicuna-7B", "value": round(world * B * args.steps / dt, 4),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[4]: stage-2 finetune step, vision LocalLoRA Vicuna-7B (r128, default+vision adapters), "
                                   f"batch {B} per GPU, 683-token sequences, fwd+bwd+all-reduce+AdamW", "per_gpu_batch": B,
                       "layers": args.layers, "parallelism": f"ddp{world}", "trainable_params": int(st.n_params),
                       "final_loss": float(loss.item())}}), flush=True)
    if DIST or world > 1:
        torch.distributed.destroy_process_group()


def mcub4_main(args, world, rank, local):
    """BASELINE configs[3]: vision + audio + video + point composed Vicuna-7B (online-merge-reset, 4 x 0.25), MCUB-4-shaped synthetic
    inputs: 336 px image, 1024x128 fbank (10 s), 8 x 224 px frames, 8192 x 6 point cloud; spliced length 3337; greedy decode.
    --workload iav: the same loop without the point modality on the 3-way composed model of configs[2] (3 x 0.333; spliced length 2790)."""
    from modelcompose_amd import synthetic
    from modelcompose_amd.dist import gather_ids
    from modelcompose_amd.model.builder import build_from_state_dict
    dev = torch.device("cuda", local)
    iav = args.workload == "iav"
    modals = ("vision", "audio", "video") if iav else ("vision", "audio", "video", "point")
    reset = ",".join(f"default-{m}={0.333 if iav else 0.25}" for m in modals)
    meta = synthetic.vicuna7b_meta(modals, reset, layers=args.layers)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    model = build_from_state_dict(meta, sd, device=dev)
    del sd
    model._raw = {}
    torch.cuda.empty_cache()
    B = (4 if iav else 2) if args.batch == 16 else args.batch
    ids = synthetic.synthetic_prompt(B, [-200, -203, -204] + ([] if iav else [-205]), seed=rank).to(dev)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev, dtype=torch.float32)
    fbank = rnd(B, 1024, 128) * 0.5
    fbank[:, 998:] = 0
    xyz = rnd(B, 8192, 3)
    xyz = xyz / xyz.norm(dim=-1, keepdim=True).clamp_min(1e-6) * torch.rand(B, 8192, 1, generator=g, device=dev) ** (1 / 3)
    mi = {"vision": rnd(B, 3, 336, 336).to(torch.bfloat16),
          "audio": {"audio_inputs": fbank.to(torch.bfloat16), "audio_padding_mask": torch.zeros(B, 1024, dtype=torch.bool, device=dev)},
          "video": rnd(B, 3, 8, 224, 224).to(torch.bfloat16),
          "point": torch.cat([xyz, torch.rand(B, 8192, 3, generator=g, device=dev)], -1).to(torch.bfloat16)}
    if iav:
        del mi["point"]
    else:
        model.model.modal_encoders["point"].fps_start = torch.zeros(B, dtype=torch.long)

    def step():
        out = model.generate(ids, modal_inputs=mi, max_new_tokens=args.new_tokens, ignore_eos=True)
        return gather_ids(out[:, ids.shape[1]:], world, force=DIST)

    def barrier():
        if DIST or world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if DIST or world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        feats, _ = model.encode_modal_inputs(mi, model.prefix_tokens, model.suffix_tokens)
        print(json.dumps({
            "metric": "samples/sec (whole node) composed-Vicuna-7B greedy gen, " + ("img+audio+video" if iav else "img+audio+video+point"),
            "value": round(world * B * args.steps / dt, 4),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("configs[2] model with the metric's inputs: 3-way composed Vicuna-7B (online-merge-reset 3 x 0.333; 7 adapters), "
                                    "336 px image + 10 s audio + 8-frame video, " if iav else
                                    "configs[3]: 4-modality composed Vicuna-7B (online-merge-reset 4 x 0.25; 9 adapters), MCUB-4-shaped inputs, ") +
                                   f"batch {B} per GPU, {args.new_tokens} greedy tokens", "per_gpu_batch": B, "new_tokens": args.new_tokens,
                       "layers": args.layers, "parallelism": f"dp{world}",
                       "spliced_length": int(ids.shape[1] - 4 + sum(f.shape[1] for f in feats.values())),
                       "block_tokens": {m: int(f.shape[1]) for m, f in feats.items()}, "ids_shape": list(out.shape)}}), flush=True)
    if DIST or world > 1:
        torch.distributed.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local)
    if DIST or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if args.gemm_debug:
        from modelcompose_amd import _lib as _l
        _l.lib().mc_gemm_debug(args.gemm_debug)
    if args.workload == "train":
        return train_main(args, world, rank, local)
    if args.workload in ("mcub4", "iav"):
        return mcub4_main(args, world, rank, local)
    from modelcompose_amd import _lib, synthetic
    from modelcompose_amd.dist import gather_ids
    from modelcompose_amd.model.builder import build_from_state_dict
    import ctypes as C

    dev = torch.device("cuda", local)
    meta = synthetic.vicuna7b_meta(("vision",), None, layers=args.layers)
    sd = synthetic.synthetic_state_dict(meta, device=dev, seed=1234)
    model = build_from_state_dict(meta, sd, device=dev)
    model.use_graph = not args.no_graph
    _lib.check(_lib.lib().mc_llm_set_option(model._handle, b"use_graph", 0 if args.no_graph else 1), "set_option")
    del sd
    model._raw = {}
    torch.cuda.empty_cache()

    B = args.batch
    ids = synthetic.synthetic_prompt(B, [-200], seed=rank).to(dev)
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    pixels = torch.randn(B, 3, 336, 336, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    modal_inputs = {"vision": pixels}

    def step():
        out = model.generate(ids, modal_inputs=modal_inputs, max_new_tokens=args.new_tokens, ignore_eos=True)
        return gather_ids(out[:, ids.shape[1]:], world, force=DIST)

    def barrier():
        if DIST or world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        if not args.pipeline:
            for _ in range(n):
                step()
            return
        for out in model.generate_pipelined(((ids, modal_inputs) for _ in range(n)), max_new_tokens=args.new_tokens, ignore_eos=True):
            gather_ids(out[:, ids.shape[1]:], world, force=DIST)

    run_steps(args.warmup)
    L = _lib.lib()
    barrier()
    L.mc_gemm_profile_enable(1)
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    L.mc_gemm_profile_enable(0)
    ms, fl, n = C.c_double(0), C.c_double(0), C.c_int64(0)
    L.mc_gemm_profile_read(C.byref(ms), C.byref(fl), C.byref(n))
    alg_bytes = C.c_double(0)
    L.mc_gemm_profile_read_bytes(C.byref(alg_bytes))
    if DIST or world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if DIST or world > 1:
            torch.distributed.destroy_process_group()
        return
    value = world * B * args.steps / dt
    achieved = (fl.value / max(ms.value, 1e-9)) / 1e9          # flops/ms -> TFLOP/s
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")     # HBM bytes per launch from the rocprofv3 PMC passes (see profiles/)
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("gemm_tile256_kernel_bytes_per_launch")
        except Exception:
            traffic = None
    line = {
        "metric": "samples/sec (whole node) composed-Vicuna-7B greedy gen",
        "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "configs[1]: vision-only composed Vicuna-7B (LocalLoRA r128: default+vision adapters), "
                               f"batch {B} synthetic 336px images per GPU, 683-token spliced prompt, {args.new_tokens} greedy tokens",
                   "per_gpu_batch": B, "new_tokens": args.new_tokens, "layers": args.layers, "parallelism": f"dp{world}",
                   "decode_graph": not args.no_graph, "pipelined": bool(args.pipeline)},
        "roofline": {"bound": "mfma", "kernel": "gemm_tile256_kernel", "achieved": round(achieved, 2),
                     "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK_TFLOPS, 4),
                     "traffic": traffic, "launches": int(n.value),
                     "avg_launch_us": round(ms.value / max(n.value, 1) * 1e3, 2),
                     "avg_flops_per_launch": fl.value / max(n.value, 1),
                     "avg_algorithmic_bytes_per_launch": alg_bytes.value / max(n.value, 1)},
    }
    if world == 1 and not args.pipeline:
        # informational, outside the timed region and outside the contract's fields: the same K batches through generate_pipelined (two
        # generation pipelines on two HIP streams).  Kept out of `value` because concurrent launches stretch the per-launch durations the
        # roofline object is computed from.
        args.pipeline = True
        run_steps(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(args.steps)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t0
        args.pipeline = False
        line["pipelined"] = {"value": round(B * args.steps / dtp, 4), "unit": "samples/s", "ms_per_step": round(dtp / args.steps * 1e3, 3),
                             "note": "bench.py --pipeline: decode of batch i overlapped with encoders + prefill of batch i+1; same tokens"}
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args.new_tokens)
    print(json.dumps(line), flush=True)
    if DIST or world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
