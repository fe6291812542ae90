"""Checkpoint composition CLI (file level).

Drop-in for `scripts/model_composition/merge_unimodal_modelcompose.py filepaths... -o OUT --strategy S -K 20`
(reference :28-159) for the strategies on the hot path:

  online-merge-<mode>            tensors unique to one checkpoint are kept; tensors present in several (the shared
  online-merge-reset-k=v,...     '...default...' LoRA keys) are renamed '...default-{modal}...' per source checkpoint
                                 (:94-103); the text after 'reset-' goes to config['reset_scaling_weights'], otherwise
                                 the mode goes to config['merge_default_weights'] (:124-129)
  sum / mean                     element-wise (:105-112)

The config union keeps the first truthy value per key (`a or b`, :117-123) and records
{modal}_lora_alpha / {modal}_lora_r (:131-136).  For online-merge no arithmetic happens on disk — exactly as in the
reference; the arithmetic consequence (W + sum_m coefficient*alpha/r * B_m A_m) is applied once at load time on the
GPU by csrc/compose.hip.

  ties-mean / ties-sum / ties-max   TIES merging of the tensors shared by every checkpoint (ties_merging.py:88-221 through
                                 merge_unimodal_modelcompose.py:78-93): trim each flattened checkpoint to its top-K % magnitudes,
                                 elect a sign per parameter, aggregate the agreeing entries — on the GPU (csrc/merge.hip: exact
                                 radix select + two streaming passes); needs a HIP device, there is no CPU fallback.
convert-* (checkpoints trained with lora_strategy 'same') and convert-drop-* are handled as the reference does (:42-73)."""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List, Sequence, Tuple

import torch

from .checkpoint_io import load_tensors

# config key -> modality, in the reference's lookup order (:15-21)
_MODAL_OF_KEY: Tuple[Tuple[str, str], ...] = (
    ("mm_vision_encoder", "vision"), ("mm_vision_tower", "vision"), ("mm_vision2_encoder", "vision2"),
    ("mm_vision2_tower", "vision2"), ("mm_video_encoder", "video"), ("mm_audio_encoder", "audio"),
    ("mm_point_encoder", "point"))


def get_modal_from_config(config: dict) -> str:
    for key, modal in _MODAL_OF_KEY:
        value = config.get(key)
        if isinstance(value, str) and value:
            return modal
    raise AssertionError("No modality is recognized, please check the config.")


def _read_checkpoint(path: str) -> Tuple[Dict[str, torch.Tensor], dict]:
    tensors_file = os.path.join(path, "adapter_model.bin")
    if not os.path.exists(tensors_file):
        tensors_file = os.path.join(path, "mm_projector.bin")
    with open(os.path.join(path, "config.json")) as f:
        cfg = json.load(f)
    return load_tensors(tensors_file), cfg


def _group_by_name(per_ckpt: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, List[Tuple[int, torch.Tensor]]]:
    """weights_to_merge of the reference (:30-40): name -> [(checkpoint index, tensor)] in order of first appearance."""
    groups: Dict[str, List[Tuple[int, torch.Tensor]]] = {}
    for i, tensors in enumerate(per_ckpt):
        for name, t in tensors.items():
            groups.setdefault(name, []).append((i, t))
    return groups


def _online_merge(groups: Dict[str, List[Tuple[int, torch.Tensor]]], modals: Sequence[str]) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    for name, who in groups.items():
        if len(who) == 1:
            out[name] = who[0][1]
            continue
        if "default" not in name:
            raise AssertionError(f"tensor '{name}' is shared by several checkpoints but is not a 'default' adapter key")
        for i, t in who:
            out[name.replace("default", f"default-{modals[i]}")] = t
    return out


def _elementwise(groups: Dict[str, List[Tuple[int, torch.Tensor]]], mean: bool) -> Dict[str, torch.Tensor]:
    out = {}
    for n, who in groups.items():
        ts = [t for _, t in who]
        out[n] = sum(ts) / len(ts) if mean else sum(ts)
    return out


def _convert_same_to_modal_language(groups, configs) -> Dict[str, List[Tuple[int, torch.Tensor]]]:
    """`convert-*` (merge_unimodal_modelcompose.py:42-59): checkpoints trained with lora_strategy 'same' hold only `.default` adapters;
    each is re-labelled 'modal+language' and every `.default` tensor of checkpoint i gets a copy named after checkpoint i's modality."""
    for cfg in configs:
        if "lora_strategy" in cfg:
            assert cfg["lora_strategy"] == "same"                                           # :47
            cfg["lora_strategy"] = "modal+language"
    modal_types = [get_modal_from_config(c) for c in configs]
    converted: Dict[str, List[Tuple[int, torch.Tensor]]] = {}
    for name, who in groups.items():
        if ".default" in name:
            for i, modal in enumerate(modal_types):
                # the reference indexes the list by checkpoint number (:58): a `.default` tensor missing from a checkpoint is an error
                if i >= len(who) or who[i][0] != i:
                    raise IndexError(f"convert-: '{name}' is missing from checkpoint {i}")
                converted.setdefault(name.replace("default", modal), []).append((i, who[i][1].clone()))
    return converted


_DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _kth_magnitude(flat: torch.Tensor, k: int) -> torch.Tensor:
    """Per row of flat [n, d] (device), the k-th smallest |x| (1-indexed, torch.kthvalue's convention) as an fp32 device tensor [n]:
    exact 3-pass radix select over the float bits of |x| (11 + 11 + 10 bits; the 2048-bin histograms come back to the host between
    passes)."""
    import ctypes as C
    import numpy as np
    from . import _lib
    n, d = flat.shape
    L = _lib.lib()
    code = _DTYPE_CODE[flat.dtype]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = flat.device
    prefix = torch.zeros(n, dtype=torch.int32, device=dev)
    remaining = [k] * n
    pre_host = [0] * n
    for (shift, nbits, pshift) in ((21, 11, 32), (10, 11, 21), (0, 10, 10)):
        hist = torch.zeros(n, 2048, dtype=torch.int32, device=dev)
        _lib.check(L.mc_ties_hist(flat.data_ptr(), code, flat.stride(0), d, n, shift, 1 << nbits, prefix.data_ptr(), pshift, hist.data_ptr(), st),
                   "mc_ties_hist")
        h = hist.cpu().numpy().astype("int64") & 0xFFFFFFFF
        for r in range(n):
            cum = 0
            for b in range(1 << nbits):
                if cum + h[r, b] >= remaining[r]:
                    remaining[r] -= cum
                    pre_host[r] = (pre_host[r] << nbits) | b
                    break
                cum += h[r, b]
            else:
                raise RuntimeError("radix select ran past the histogram (NaN in the checkpoint?)")
        prefix = torch.tensor([p - (1 << 32) if p >= (1 << 31) else p for p in pre_host], dtype=torch.int32, device=dev)
    thr = torch.from_numpy(np.array(pre_host, dtype=np.uint32).view(np.float32).copy()).to(dev)     # k-th smallest |x| per row
    return thr


def ties_merge_vectors(flat: torch.Tensor, K, merge_func: str) -> torch.Tensor:
    """flat [n, d] device tensor (fp32 / bf16 / fp16) -> merged [d] in the same dtype (ties_merging.py:160-179).
    The k-th smallest magnitude of every row is found exactly by a 3-pass radix select over the float bits of |x| (11 + 11 + 10
    bits; the 2048-bin histograms come back to the host between passes), then mc_ties_merge elects signs and aggregates."""
    import ctypes as C
    from . import _lib
    if not flat.is_cuda:
        raise ValueError("ties_merge_vectors needs device (HIP) tensors; this path has no CPU fallback")
    if flat.dtype not in _DTYPE_CODE:
        raise ValueError(f"unsupported checkpoint dtype {flat.dtype}")
    if merge_func not in ("mean", "sum", "max"):
        raise ValueError(f"Merge method {merge_func} is not defined.")                     # ties_merging.py:155
    if K >= 1:
        K = K / 100                                                                         # :89-90
    flat = flat.contiguous()
    n, d = flat.shape
    k = d - int(d * K)                                                                      # :97-98: k-th smallest, 1-indexed
    if not 1 <= k <= d:
        raise RuntimeError(f"kthvalue(): selected number k out of range for dimension {d}")
    thr = _kth_magnitude(flat, k)
    L = _lib.lib()
    code = _DTYPE_CODE[flat.dtype]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = flat.device
    sign = torch.empty(d, dtype=torch.int8, device=dev)
    sign_sum = torch.zeros(1, dtype=torch.int64, device=dev)
    out = torch.empty(d, dtype=flat.dtype, device=dev)
    _lib.check(L.mc_ties_merge(flat.data_ptr(), code, flat.stride(0), d, n, thr.data_ptr(), sign.data_ptr(), sign_sum.data_ptr(),
                               {"mean": 0, "sum": 1, "max": 2}[merge_func], out.data_ptr(), st), "mc_ties_merge")
    return out


def ties_merge_state_dicts(checks: Sequence[Dict[str, torch.Tensor]], K=20, merge_func: str = "mean", device="cuda") -> Dict[str, torch.Tensor]:
    """do_merging (ties_merging.py:182-221): flatten in sorted key order, merge on the GPU, un-flatten (host tensors out).
    mean divides a half-precision sum by an fp32 count, which torch promotes to fp32 before the copy back into the
    checkpoint dtype (:145-148, :217-219); the kernel applies the same roundings."""
    if not torch.cuda.is_available():
        raise RuntimeError("TIES merging runs on the HIP device; no CPU fallback")
    keys = sorted(checks[0])
    for c in checks[1:]:
        if set(c) != set(keys):
            raise ValueError("Differing parameter names in models.")                        # :61-72
    flat = torch.stack([torch.cat([c[k].reshape(-1) for k in keys]).to(device) for c in checks])
    merged = ties_merge_vectors(flat, K, merge_func).cpu()
    out, off = {}, 0
    for k in keys:
        t = checks[0][k]
        out[k] = merged[off:off + t.numel()].view(t.shape).clone()
        off += t.numel()
    return out


def _ties(groups: Dict[str, List[Tuple[int, torch.Tensor]]], func: str, K) -> Dict[str, torch.Tensor]:
    """convert_delta_to_ft (ties_merging.py:224-250) + do_merging: tensors present in every checkpoint are merged, tensors present
    in exactly one are kept; merged tensors first (sorted names, as vector_to_state_dict rebuilds them), then the kept ones (:85-86)."""
    n = max(len(v) for v in groups.values())
    uniques = {}
    shared = [dict() for _ in range(n)]
    for name, who in groups.items():
        if len(who) == n:
            for i in range(n):
                shared[i][name] = who[i][1]
        else:
            assert len(who) == 1, f"tensor '{name}' appears in {len(who)} of {n} checkpoints"   # :246
            uniques[name] = who[0][1]
    out = ties_merge_state_dicts(shared, K, func)
    out.update(uniques)
    return out


def interference_metrics(flat: torch.Tensor, reset_thresh=50) -> Dict[str, float]:
    """Parameter-interference metrics of n >= 2 task vectors flat [n, d] on the device (calculate_metrics.py:26-37, :58-67):
    L2 = ||x0 - x1||, Cosine = 1 - cos(x0, x1), SSD = 1 - mean_j |sum_i x_ij| / sum_i |x_ij| over columns with a non-zero magnitude
    sum, TSSD = SSD of the rows trimmed to their top `reset_thresh` % magnitudes (topk_values_mask, ties_merging.py:88-109).
    One HBM pass (mc_merge_metrics); the per-workgroup double partial sums are added on the host in block order."""
    import ctypes as C
    from . import _lib
    if not flat.is_cuda:
        raise ValueError("interference_metrics needs device (HIP) tensors; this path has no CPU fallback")
    if flat.dtype not in _DTYPE_CODE:
        raise ValueError(f"unsupported checkpoint dtype {flat.dtype}")
    if flat.dim() != 2 or flat.shape[0] < 2:
        raise ValueError("interference_metrics needs at least two task vectors [n, d]")
    K = reset_thresh / 100 if reset_thresh >= 1 else reset_thresh                           # ties_merging.py:89-90
    flat = flat.contiguous()
    n, d = flat.shape
    k = d - int(d * K)
    if not 1 <= k <= d:
        raise RuntimeError(f"kthvalue(): selected number k out of range for dimension {d}")
    thr = _kth_magnitude(flat, k)
    L = _lib.lib()
    partial = torch.empty(L.mc_merge_metrics_blocks(), 8, dtype=torch.float64, device=flat.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.mc_merge_metrics(flat.data_ptr(), _DTYPE_CODE[flat.dtype], flat.stride(0), d, n, thr.data_ptr(), partial.data_ptr(), st),
               "mc_merge_metrics")
    a = partial.cpu().sum(dim=0).tolist()
    cos = a[1] / max((a[2] ** 0.5) * (a[3] ** 0.5), 1e-8)                                   # torch.cosine_similarity's eps clamp
    return {"L2": a[0] ** 0.5, "Cosine": 1 - cos, "SSD": 1 - a[4] / a[5] if a[5] else float("nan"),
            "TSSD": 1 - a[6] / a[7] if a[7] else float("nan")}


def parse_merge_info(file: str):
    """calculate_metrics.py:14-24: (input paths, strategy, output path) out of merge_info.txt, or (None, None, None)."""
    import re
    m = re.search(r"Inputs:\n(.*?)\n\nOutput\((.*?)\):(.*?)$", open(file).read().strip(), re.DOTALL)
    if not m:
        return None, None, None
    return m.group(1).split("\n"), m.group(2), m.group(3)


def calculate_metrics(merged_ckpt: str, reset_thresh=50, device="cuda") -> Dict[str, float]:
    """calculate_metrics.py:41-76: reload the inputs named by <merged_ckpt>/merge_info.txt, flatten the tensors every input shares
    (fp32, sorted key order), compute the four metrics on the GPU, write <merged_ckpt>/merge_metrics.txt in the reference's format."""
    if not torch.cuda.is_available():
        raise RuntimeError("calculate_metrics runs on the HIP device; no CPU fallback")
    filepaths, _, _ = parse_merge_info(os.path.join(merged_ckpt, "merge_info.txt"))
    per_ckpt = [{k: v.float() for k, v in _read_checkpoint(fp)[0].items()} for fp in filepaths]
    groups: Dict[str, List[torch.Tensor]] = {}
    for tensors in per_ckpt:
        for name, t in tensors.items():
            groups.setdefault(name, []).append(t)
    n = max(len(v) for v in groups.values())
    keys = sorted(name for name, ts in groups.items() if len(ts) == n)                      # convert_delta_to_ft keeps the shared ones
    for name, ts in groups.items():
        assert len(ts) in (1, n), f"tensor '{name}' appears in {len(ts)} of {n} checkpoints"
    flat = torch.stack([torch.cat([groups[k][i].reshape(-1) for k in keys]).to(device) for i in range(n)])
    m = interference_metrics(flat, reset_thresh)
    with open(os.path.join(merged_ckpt, "merge_metrics.txt"), "w") as f:
        for name in ("L2", "Cosine", "SSD", "TSSD"):
            f.write(f"{name}: {m[name]}\n")
    return m


def llava_key_to_multimodal_key(key: str):
    """scripts/convert_llava_to_multimodal/convert_checkpoint.py:47-56: LLaVA-format tensor name -> multimodal-format name (the
    'default' LoRA pair becomes the 'vision' adapter, mm_projector / prefix / suffix tokens get the '.vision' slot); None = dropped."""
    if "lora_A.default" in key or "lora_B.default" in key:
        return key.replace("default", "vision")
    for old, new in (("mm_projector", "modal_projectors.vision"), ("prefix_tokens", "prefix_tokens.vision"), ("suffix_tokens", "suffix_tokens.vision")):
        if old in key:
            return key.replace(old, new)
    return None


def convert_llava_checkpoint(llava_checkpoint: str, output_path: str) -> None:
    """convert_checkpoint.py:68-88: read the two-shard LLaVA checkpoint, keep the renamed LoRA tensors in adapter_model.bin and the
    other renamed tensors in non_lora_trainables.bin, copy the tokenizer / config files.  Shards are memory-mapped (no 13 GB
    unpickle into anonymous memory); tensors are written byte-exact."""
    import shutil
    weights = {}
    for shard in ("pytorch_model-00001-of-00002.bin", "pytorch_model-00002-of-00002.bin"):
        weights.update(load_tensors(os.path.join(llava_checkpoint, shard)))
    adapter, other = {}, {}
    for k, v in weights.items():
        nk = llava_key_to_multimodal_key(k)
        if not nk:
            continue
        (adapter if "lora" in nk else other)[nk] = v
    os.makedirs(output_path, exist_ok=True)
    torch.save(adapter, os.path.join(output_path, "adapter_model.bin"))
    torch.save(other, os.path.join(output_path, "non_lora_trainables.bin"))
    for f in os.listdir(llava_checkpoint):
        if f in ("special_tokens_map.json", "tokenizer.model", "tokenizer_config.json", "config.json"):
            shutil.copy(os.path.join(llava_checkpoint, f), os.path.join(output_path, f))


def merge_checkpoints(filepaths: Sequence[str], output_path: str, strategy: str = "sum", K: int = 20):
    loaded = [_read_checkpoint(p) for p in filepaths]
    configs = [c for _, c in loaded]
    groups = _group_by_name([t for t, _ in loaded])
    merged = None
    if strategy.startswith("convert-"):                                                     # :42-73
        strategy = strategy[len("convert-"):]
        converted = _convert_same_to_modal_language(groups, configs)
        if strategy.startswith("drop-"):
            # TIES over the tensors the checkpoints share, the single-owner tensors kept, the per-modality copies on top (:62-71); the
            # strategy string then matches no later branch of the reference ("DO NOTHING") and labels merge_info as it stands
            func = strategy[len("drop-"):]
            merged = _ties(groups, func, K)
            merged.update({name: who[0][1] for name, who in converted.items()})
        else:
            groups.update(converted)
    elif strategy.startswith("drop-"):
        raise NotImplementedError(f"Merge strategy [{strategy}]: drop-* is only defined behind convert- (the reference script fails on it)")
    label = strategy
    extra_cfg = {}
    if merged is not None:
        pass
    elif strategy.startswith("online-merge-"):
        merged = _online_merge(groups, [get_modal_from_config(c) for c in configs])
        label = strategy[len("online-merge-"):]
        if label.startswith("reset-"):
            extra_cfg["reset_scaling_weights"] = label[len("reset-"):]
        else:
            extra_cfg["merge_default_weights"] = label
    elif strategy.startswith("ties-"):
        func = strategy[len("ties-"):]
        assert func in ("sum", "mean", "max")                                               # merge_unimodal_modelcompose.py:80
        merged = _ties(groups, func, K)
        label = f"dis-{func}-{K}"                                                           # :89
    elif strategy in ("sum", "mean"):
        merged = _elementwise(groups, mean=strategy == "mean")
    else:
        raise NotImplementedError(f"Merge strategy [{strategy}] not implemented")
    # union of the configs: first truthy value wins; the merge record is written while visiting the first config,
    # so later configs may still `or` over a falsy value of the same key (reference order of operations)
    union: dict = {}
    for i, cfg in enumerate(configs):
        for k, v in cfg.items():
            union[k] = (union[k] or v) if k in union else v
        if i == 0:
            union.update(extra_cfg)
    for cfg in configs:
        modal = get_modal_from_config(cfg)
        union[f"{modal}_lora_alpha"] = cfg["lora_alpha"]
        union[f"{modal}_lora_r"] = cfg["lora_r"]
    os.makedirs(output_path, exist_ok=True)
    torch.save(merged, os.path.join(output_path, "adapter_model.bin"))
    with open(os.path.join(output_path, "config.json"), "w") as f:
        json.dump(union, f, indent=4)
    with open(os.path.join(output_path, "merge_info.txt"), "w") as f:
        f.write("Inputs:\n" + "\n".join(filepaths) + f"\n\nOutput({label}):{output_path}")
    print(f"Merged checkpoints saved to {output_path}")
    return merged, union


def main(argv=None):
    ap = argparse.ArgumentParser(description="Merge multiple torch checkpoints")
    ap.add_argument("filepaths", nargs="+", help="List of checkpoint file paths to merge")
    ap.add_argument("-o", "--output", default="merged_checkpoint.pth", help="Output file path")
    ap.add_argument("--strategy", default="sum", help="Merge strategy")
    ap.add_argument("-K", default=20, type=int, help="K for ties-merging")
    ns = ap.parse_args(argv)
    merge_checkpoints(ns.filepaths, ns.output, ns.strategy, ns.K)


if __name__ == "__main__":
    main()
