"""Oracle (TEST INFRASTRUCTURE): checkpoint composition, restating
scripts/model_composition/merge_unimodal_modelcompose.py:28-145 for the
'online-merge-*' strategies (key rename + config union; no arithmetic) and the
plain 'sum' / 'mean' strategies, and scripts/model_composition/ties_merging.py:88-221 for 'ties-{mean,sum,max}'
(trim to the top-K % magnitudes per checkpoint, elect a sign per parameter, merge the agreeing entries).  Byte-exact tensors."""
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


def ties_merge_vectors(flat: torch.Tensor, K, merge_func: str) -> torch.Tensor:
    """flat [n_checkpoints, d] -> merged [d]  (ties_merging.py:88-179; `merge_func` in {'mean', 'sum', 'max'}).
    topk_values_mask :88-109: K >= 1 is a percentage; keep |x| >= (d - int(d*K))-th smallest magnitude of the row (ties kept);
    resolve_sign :122-125 with zero sums taking the majority sign :112-119; disjoint_merge :128-157."""
    if K >= 1:
        K = K / 100
    n, d = flat.shape
    k = d - int(d * K)
    kth = flat.abs().kthvalue(k, dim=1, keepdim=True).values
    upd = flat * (flat.abs() >= kth)
    sign = torch.sign(upd.sum(dim=0))
    majority = torch.sign(sign.sum())
    sign[sign == 0] = majority
    keep = torch.where(sign.unsqueeze(0) > 0, upd > 0, upd < 0)
    sel = upd * keep
    if merge_func == "mean":
        return sel.sum(dim=0) / torch.clamp((sel != 0).sum(dim=0).float(), min=1)
    if merge_func == "sum":
        return sel.sum(dim=0)
    if merge_func == "max":
        return sel.abs().max(dim=0)[0] * sign
    raise ValueError(f"Merge method {merge_func} is not defined.")


def interference_metrics(flat: torch.Tensor, reset_thresh=50):
    """scripts/model_composition/calculate_metrics.py:26-37 on the stacked fp32 task vectors (:58-67): L2 and cosine distance of rows
    0 and 1, soft sign dissimilarity of all rows, and of the rows trimmed by topk_values_mask(K=reset_thresh) (ties_merging.py:88-109)."""
    def ssd(xy):
        a, s = xy.abs().sum(dim=0), xy.sum(dim=0)
        nz = a != 0
        return float(1 - (s[nz] / a[nz]).abs().mean())
    K = reset_thresh / 100 if reset_thresh >= 1 else reset_thresh
    n, d = flat.shape
    kth = flat.abs().kthvalue(d - int(d * K), dim=1, keepdim=True).values
    trimmed = flat * (flat.abs() >= kth)
    return {"L2": float(torch.sqrt(((flat[0] - flat[1]) ** 2).sum())),
            "Cosine": 1 - torch.cosine_similarity(flat[0].unsqueeze(0), flat[1].unsqueeze(0)).item(),
            "SSD": ssd(flat), "TSSD": ssd(trimmed)}


def ties_merge_state_dicts(checks, K=20, merge_func="mean"):
    """do_merging :182-221: flatten in sorted key order (state_dict_to_vector :22-31), merge, un-flatten."""
    keys = sorted(checks[0])
    flat = torch.vstack([torch.cat([c[k].reshape(-1) for k in keys]) for c in checks])
    merged = ties_merge_vectors(flat, K, merge_func)
    out, off = {}, 0
    for k in keys:
        n = checks[0][k].numel()
        out[k] = merged[off:off + n].view_as(checks[0][k]).clone()
        off += n
    return out


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
    if strategy.startswith("ties-"):                                         # :78-93 (convert_delta_to_ft: ties_merging.py:224-250)
        func = strategy.replace("ties-", "")
        assert func in ("sum", "mean", "max")
        n = max(len(v) for v in weights.values())
        shared = {k: v for k, v in weights.items() if len(v) == n}
        merged = {k: v[0] for k, v in weights.items() if len(v) != n}
        assert all(len(v) == 1 for k, v in weights.items() if len(v) != n)
        merged.update(ties_merge_state_dicts([{k: v[i] for k, v in shared.items()} for i in range(n)], K, func))
        strategy = f"dis-{func}-{K}"
    elif strategy.startswith("online-merge-"):                               # :94-103
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
