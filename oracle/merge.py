"""Oracle (TEST INFRASTRUCTURE): checkpoint composition, restating
scripts/model_composition/merge_unimodal_modelcompose.py:28-145 for the
'online-merge-*' strategies (key rename + config union; no arithmetic) and the
plain 'sum' / 'mean' strategies.  Byte-exact tensors."""
from __future__ import annotations

import json
import os
from collections import defaultdict

import torch

# merge_unimodal_modelcompose.py:15-21
MODAL_DICT = {"mm_vision_encoder": "vision", "mm_vision_tower": "vision", "mm_vision2_encoder": "vision2",
              "mm_vision2_tower": "vision2", "mm_video_encoder": "video", "mm_audio_encoder": "audio",
              "mm_point_encoder": "point"}


def get_modal_from_config(config):
    """:22-26 (first recognised key in MODAL_DICT order)."""
    for key in MODAL_DICT:
        if key in config and isinstance(config[key], str) and len(config[key]) > 0:
            return MODAL_DICT[key]
    raise AssertionError("No modality is recognized, please check the config.")


def merge_checkpoints(filepaths, output_path, strategy="sum", K=20):
    configs, weights = [], defaultdict(list)
    for fp in filepaths:                                                     # :31-40
        ap = os.path.join(fp, "adapter_model.bin")
        if not os.path.exists(ap):
            ap = os.path.join(fp, "mm_projector.bin")
        w = torch.load(ap, map_location="cpu")
        configs.append(json.load(open(os.path.join(fp, "config.json"))))
        for k in w:
            weights[k].append(w[k])
    if strategy.startswith("online-merge-"):                                 # :94-103
        merged = {}
        names = [get_modal_from_config(c) for c in configs]
        for k in weights:
            if len(weights[k]) == 1:
                merged[k] = weights[k][0]
            else:
                assert "default" in k
                for n, w in zip(names, weights[k]):
                    merged[k.replace("default", f"default-{n}")] = w
    elif strategy == "sum":                                                  # :105-108
        merged = {k: sum(v) for k, v in weights.items()}
    elif strategy == "mean":                                                 # :109-112
        merged = {k: sum(v) / len(v) for k, v in weights.items()}
    else:
        raise NotImplementedError(f"oracle does not restate strategy '{strategy}'")
    mc = {}
    for c in configs:                                                        # :116-129
        for k in c:
            mc[k] = (mc[k] or c[k]) if k in mc else c[k]
        if strategy.startswith("online-merge-"):
            strategy = strategy.replace("online-merge-", "")
            if strategy.startswith("reset-"):
                mc["reset_scaling_weights"] = strategy.replace("reset-", "")
            else:
                mc["merge_default_weights"] = strategy
    for c in configs:                                                        # :131-136
        n = get_modal_from_config(c)
        mc[f"{n}_lora_alpha"] = c["lora_alpha"]
        mc[f"{n}_lora_r"] = c["lora_r"]
    os.makedirs(output_path, exist_ok=True)                                  # :138-145
    torch.save(merged, os.path.join(output_path, "adapter_model.bin"))
    json.dump(mc, open(os.path.join(output_path, "config.json"), "w"), indent=4)
    with open(os.path.join(output_path, "merge_info.txt"), "w") as f:
        f.write("Inputs:\n{}\n\nOutput({}):{}".format("\n".join(filepaths), strategy, output_path))
    return merged, mc
