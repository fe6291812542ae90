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
GPU by csrc/compose.hip.  TIES / convert / drop strategies are outside the hot path and raise."""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, List, Sequence, Tuple

import torch

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
    return torch.load(tensors_file, map_location="cpu"), cfg


def _online_merge(per_ckpt: Sequence[Dict[str, torch.Tensor]], modals: Sequence[str]) -> Dict[str, torch.Tensor]:
    owners: Dict[str, List[int]] = {}
    for i, tensors in enumerate(per_ckpt):
        for name in tensors:
            owners.setdefault(name, []).append(i)
    out: Dict[str, torch.Tensor] = {}
    for name, who in owners.items():
        if len(who) == 1:
            out[name] = per_ckpt[who[0]][name]
            continue
        if "default" not in name:
            raise AssertionError(f"tensor '{name}' is shared by several checkpoints but is not a 'default' adapter key")
        for i in who:
            out[name.replace("default", f"default-{modals[i]}")] = per_ckpt[i][name]
    return out


def _elementwise(per_ckpt: Sequence[Dict[str, torch.Tensor]], mean: bool) -> Dict[str, torch.Tensor]:
    groups: Dict[str, List[torch.Tensor]] = {}
    for tensors in per_ckpt:
        for name, t in tensors.items():
            groups.setdefault(name, []).append(t)
    return {n: (sum(ts) / len(ts) if mean else sum(ts)) for n, ts in groups.items()}


def merge_checkpoints(filepaths: Sequence[str], output_path: str, strategy: str = "sum", K: int = 20):
    loaded = [_read_checkpoint(p) for p in filepaths]
    tensors = [t for t, _ in loaded]
    configs = [c for _, c in loaded]
    if strategy.startswith(("ties-", "convert-", "drop-")):
        raise NotImplementedError(f"Merge strategy [{strategy}] (TIES / convert / drop) is not on the hot path; "
                                  f"use the reference script for it.")
    label = strategy
    extra_cfg = {}
    if strategy.startswith("online-merge-"):
        merged = _online_merge(tensors, [get_modal_from_config(c) for c in configs])
        label = strategy[len("online-merge-"):]
        if label.startswith("reset-"):
            extra_cfg["reset_scaling_weights"] = label[len("reset-"):]
        else:
            extra_cfg["merge_default_weights"] = label
    elif strategy in ("sum", "mean"):
        merged = _elementwise(tensors, mean=strategy == "mean")
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
