"""Stage-2 finetune step on the GPU (BASELINE config 5 in miniature) against the reference's own loss and gradients
(tests/golden/g9_train_step.npz: loss.backward() of the unmodified reference on the same weights and batch)."""
import os

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stepper():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    a, meta, sd = load_golden("g9_train_step")
    meta = dict(meta, lora_dropout=0.0)
    model = build_from_state_dict(meta, sd)
    return MultimodalTrainStep(model, lr=1e-3), a, meta, sd


def test_loss_and_gradients_match_reference(stepper):
    st, a, meta, sd = stepper
    loss = st.forward_backward(a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    # bf16 activations through 2 layers vs the fp32 reference
    assert abs(loss.item() - a["loss"].item()) < 2e-2 * abs(a["loss"].item())
    ref = {k[6:]: v for k, v in a.items() if k.startswith("grad::")}
    got = st.named_gradients()
    assert sorted(got) == sorted(ref)
    worst = 0.0
    for k, g in ref.items():
        x = got[k].float().cpu().reshape(g.shape)
        scale = g.abs().max().item()
        err = (x - g).abs().max().item()
        if scale == 0.0:                       # a parameter the loss does not depend on for this batch
            assert err == 0.0, (k, err)
            continue
        cos = torch.nn.functional.cosine_similarity(x.flatten(), g.flatten(), dim=0).item()
        worst = max(worst, err / scale)
        # gradients flow through bf16 activations / bf16 activation gradients: 6 % of each tensor's scale, direction within 0.5 %
        assert err <= 6e-2 * scale, (k, err, scale)
        assert cos > 0.995, (k, cos)
    print("worst relative gradient error", worst)


def test_optimizer_step_reduces_the_loss_and_is_deterministic(stepper):
    st, a, meta, sd = stepper
    args = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    l0 = st.forward_backward(*args).item()
    g0 = st.G.clone()
    l0b = st.forward_backward(*args).item()
    assert l0 == l0b and torch.equal(g0, st.G)                 # no atomics anywhere: bitwise reproducible
    for _ in range(5):
        st.step(*args)
    l1 = st.forward_backward(*args).item()
    assert l1 < l0 - 0.05, (l0, l1)


def test_forward_with_labels_returns_the_reference_loss():
    """MultimodalLlamaForCausalLM.forward(labels=...) (multimodal_llama.py:722-733): shifted CE on the HIP kernel, mean over the
    kept targets; checked against the reference's loss on the same batch (bf16 forward vs fp32 reference: 2 %)."""
    from modelcompose_amd.model.builder import build_from_state_dict
    a, meta, sd = load_golden("g9_train_step")
    model = build_from_state_dict(dict(meta, lora_dropout=0.0), sd)
    out = model.forward(a["input_ids"].cuda(), labels=a["labels"].cuda(), modal_inputs={"vision": a["pixels"].cuda()})
    assert abs(out.loss.item() - a["loss"].item()) < 2e-2 * abs(a["loss"].item())
    assert out.logits.shape[0] == a["input_ids"].shape[0] and out.logits.dtype == torch.float32
    # all targets ignored -> nan, like torch's cross_entropy with ignore_index
    lab = torch.full_like(a["labels"], -100)
    out = model.forward(a["input_ids"].cuda(), labels=lab.cuda(), modal_inputs={"vision": a["pixels"].cuda()})
    assert torch.isnan(out.loss)


def _check_grads(got, ref, tol=6e-2, min_cos=0.995):
    assert sorted(got) == sorted(ref)
    worst = 0.0
    top = max(g.abs().max().item() for g in ref.values())
    for k, g in ref.items():
        x = got[k].float().cpu().reshape(g.shape)
        scale, err = g.abs().max().item(), (x - g).abs().max().item()
        if scale == 0.0:
            assert err == 0.0, (k, err)
            continue
        if scale < 1e-6 * top:
            # a gradient that is zero in exact arithmetic (the bias of attention KEYS: softmax is invariant to a per-query constant) - the
            # reference holds fp32 cancellation noise there, the device bf16 cancellation noise; neither has a direction to compare
            assert err <= 1e-3 * top, (k, err, top)
            continue
        cos = torch.nn.functional.cosine_similarity(x.flatten(), g.flatten(), dim=0).item()
        worst = max(worst, err / scale)
        assert err <= tol * scale, (k, err, scale)
        assert cos > min_cos, (k, cos)
    return worst


def test_lora_dropout_matches_the_oracle_with_the_same_masks():
    """configs[4] as the reference runs it (run_finetune_vision_damc.sh: lora_dropout 0.05; nn.Dropout on the LoRA input,
    multimodal_llama.py:133-148).  The step's Philox masks are regenerated independently on the CPU (oracle/philox.py), checked against
    the device's own mask bit for bit, and handed to the autograd oracle: same loss, same gradients."""
    import numpy as np
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    from oracle import philox, train as otrain
    a, meta, sd = load_golden("g9_train_step")
    p = 0.25                                                       # far from 0 so that a missing / misplaced mask cannot hide in the tolerance
    meta = dict(meta, lora_dropout=p)
    model = build_from_state_dict(meta, sd)
    st = MultimodalTrainStep(model, lr=1e-3, dropout_seed=1234)
    ids, labels, px = a["input_ids"], a["labels"], a["pixels"]
    loss = st.forward_backward(ids.cuda(), labels.cuda(), {"vision": px.cuda()})
    B = ids.shape[0]
    Hd, I = meta["hidden_size"], meta["intermediate_size"]
    om_probe = otrain.pipeline.OracleModel.from_state_dict({k: v.float() for k, v in sd.items()}, meta)
    _, emb, _, _ = om_probe.prepare(ids, {"vision": px}, None, labels)
    L = emb.shape[1]
    masks = {}
    for layer in range(meta["num_hidden_layers"]):
        for blk, lin, K in (("self_attn", "q_proj", Hd), ("self_attn", "k_proj", Hd), ("self_attn", "v_proj", Hd), ("self_attn", "o_proj", Hd),
                            ("mlp", "gate_proj", Hd), ("mlp", "up_proj", Hd), ("mlp", "down_proj", I)):
            keep = philox.dropout_keep(B * L, K, p, st._seed, st._stream_id(layer, lin))
            dev = st.dropout_keep_scale(layer, lin, B * L, K).cpu()
            assert torch.equal(dev != 0, torch.from_numpy(keep)), (layer, lin)            # Philox on the device == numpy restatement
            masks[f"model.layers.{layer}.{blk}.{lin}"] = (torch.from_numpy(keep).float() / (1.0 - p)).view(B, L, K)
    frac = float(np.mean([m.ne(0).float().mean().item() for m in masks.values()]))
    assert abs(frac - (1 - p)) < 0.01
    ref_loss, _, ref_grads = otrain.loss_and_grads(sd, meta, ids, labels, {"vision": px}, dropout_masks=masks)
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    worst = _check_grads(st.named_gradients(), {k: v for k, v in ref_grads.items()})
    # and the masks matter: the no-dropout gradients are NOT within tolerance of the dropout reference
    ref0_loss, _, ref0 = otrain.loss_and_grads(sd, meta, ids, labels, {"vision": px})
    k = "model.layers.0.mlp.down_proj.lora_A.default.weight"
    assert (ref0[k] - ref_grads[k]).abs().max() > 0.2 * ref_grads[k].abs().max()
    # a new step draws new masks; the same step index reproduces its masks exactly
    g1 = st.G.clone()
    st.forward_backward(ids.cuda(), labels.cuda(), {"vision": px.cuda()})
    assert torch.equal(g1, st.G)
    st.step_count += 1
    st.forward_backward(ids.cuda(), labels.cuda(), {"vision": px.cuda()})
    assert not torch.equal(g1, st.G)
    print("worst relative gradient error with dropout", worst)


def test_padded_ragged_batch_matches_the_oracle():
    """The collator right-pads ids (pad id) / labels (-100) and passes attention_mask = ids != pad (multimodal_dataset.py:148-214); one
    sample is text only, so the spliced lengths are ragged as well (multimodal_arch.py:390-430).  Loss and gradients against the autograd
    oracle on the same padded batch."""
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    from oracle import train as otrain
    a, meta, sd = load_golden("g9_train_step")
    meta = dict(meta, lora_dropout=0.0)
    model = build_from_state_dict(meta, sd)
    st = MultimodalTrainStep(model, lr=1e-3)
    V, pad = -200, meta.get("pad_token_id", 0) or 0
    g = torch.Generator().manual_seed(11)
    r = lambda n: torch.randint(3, meta["vocab_size"] - 1, (n,), generator=g).tolist()
    rows = [[1] + r(5) + [V, 13] + r(9), [1] + r(3) + [V, 13] + r(4), [1] + r(12)]          # image, shorter + image, text only
    Lt = max(len(x) for x in rows)
    ids = torch.tensor([x + [pad] * (Lt - len(x)) for x in rows])
    am = torch.tensor([[True] * len(x) + [False] * (Lt - len(x)) for x in rows])
    labels = ids.clone()
    labels[~am] = -100
    labels[:, :4] = -100
    labels[ids == V] = -100
    px = torch.cat([a["pixels"][:1], a["pixels"][:1] * 0.7])
    loss = st.forward_backward(ids.cuda(), labels.cuda(), {"vision": px.cuda()}, attention_mask=am.cuda())
    ref_loss, _, ref_grads = otrain.loss_and_grads(sd, meta, ids, labels, {"vision": px}, attention_mask=am)
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    _check_grads(st.named_gradients(), ref_grads)
    with pytest.raises(NotImplementedError):
        st.forward_backward(ids.cuda(), labels.cuda(), {"vision": px.cuda()}, attention_mask=am.flip(1).cuda())


def test_absent_modality_gets_no_gradient_and_no_optimizer_update():
    """a text-only batch after an image batch must not re-apply the image batch's projector / prefix / suffix gradients
    (the reference leaves .grad None and AdamW skips those tensors)."""
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    a, meta, sd = load_golden("g9_train_step")
    model = build_from_state_dict(dict(meta, lora_dropout=0.0), sd)
    st = MultimodalTrainStep(model, lr=1e-3)
    st.step(a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    lo = st.aux_lo
    assert st.G[lo:].abs().max().item() > 0
    p_before, m_before = st.P[lo:].clone(), st.m1[lo:].clone()
    g = torch.Generator().manual_seed(2)
    ids = torch.cat([torch.ones(2, 1, dtype=torch.long), torch.randint(3, meta["vocab_size"] - 1, (2, 11), generator=g)], 1)
    labels = ids.clone()
    labels[:, :3] = -100
    st.step(ids.cuda(), labels.cuda(), {})
    assert st.G[lo:].abs().max().item() == 0.0
    assert torch.equal(st.P[lo:], p_before) and torch.equal(st.m1[lo:], m_before)
    assert st._aux_steps["vision"] == 1 and st.step_count == 2


def _torch_adamw(params, grads, lr_of, wd_of, steps=1, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.AdamW (what HF Trainer builds from llava_trainer.py's groups) on CPU fp32 copies."""
    ps = {k: torch.nn.Parameter(v.detach().float().cpu().clone()) for k, v in params.items()}
    opt = torch.optim.AdamW([{"params": [ps[k]], "lr": lr_of(k), "weight_decay": wd_of(k)} for k in ps], betas=betas, eps=eps)
    for _ in range(steps):
        for k in ps:
            ps[k].grad = grads[k].detach().float().cpu().reshape(ps[k].shape).clone()
        opt.step()
    return {k: v.detach() for k, v in ps.items()}


def test_finetune_step_at_the_real_widths_matches_the_autograd_oracle():
    """configs[4] at the widths bench.py's `train` secondary times - Vicuna-7B hidden 4096 / 32 x 128 / FFN
    11008 / vocab 32000 / LoRA r = 128 (default + vision), real-size CLIP-L/14-336 + mlp2x_gelu projector, two decoder layers, two
    683-token image-text rows, the last 60 tokens are targets - against the autograd oracle on the host cores (oracle/train.py: the
    reference's training graph, multimodal_llama.py:676-745, differentiated by torch).  Same bounds as the tiny reference fixture g9."""
    import time
    import fullwidth_cases as fc
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    from oracle import train as otrain
    meta, sd, ids, mi = fc.build_case("configs1_vision", [449, 470])
    meta = dict(meta, lora_dropout=0.0)
    labels = ids.clone()
    labels[:, :-60] = -100
    model = build_from_state_dict(meta, sd)
    st = MultimodalTrainStep(model, lr=2e-4)
    loss = st.forward_backward(ids.cuda(), labels.cuda(), fc.to_dev(mi))
    torch.cuda.synchronize()
    t0 = time.time()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref_loss, _, ref_grads = otrain.loss_and_grads(sd, meta, ids, labels, fc.to_f32(mi))
    print(f"[real-width train] oracle {time.time() - t0:.0f}s  loss HIP {loss.item():.5f} oracle {ref_loss.item():.5f}")
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    worst = _check_grads(st.named_gradients(), ref_grads)
    print(f"[real-width train] worst gradient error {worst:.3e} of its tensor's scale over {len(ref_grads)} tensors")
    # and the step itself runs: parameters move, the bf16 copy follows
    p0 = st.P.clone()
    st.optimizer_step()
    assert (st.P - p0).abs().max().item() > 0 and torch.equal(st.P16, st.P.to(torch.bfloat16))


def test_two_rate_adamw_follows_the_reference_parameter_groups():
    """VERDICT r3 missing #2: llava_trainer.py:210-290 - with --mm_projector_lr the `modal_projectors` tensors take that rate, with
    --mm_language_lr ALSO set the `lora_A.default` / `lora_B.default` tensors join that group (at mm_projector_lr: the reference never reads
    mm_language_lr's value), everything else takes --learning_rate (run_finetune_vision_damc.sh:28).  The default adapter's A rows / B columns
    are slices of stacked tensors here: one launch, the rate chosen per element.  Checked against torch.optim.AdamW over the same gradients
    with the groups built by NAME, as the reference builds them."""
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    a, meta, sd = load_golden("g9_train_step")
    model = build_from_state_dict(dict(meta, lora_dropout=0.0), sd)
    LR, PLR = 1e-3, 1e-4
    args = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    for lang, wd in ((5e-5, 0.0), (None, 0.1)):
        st = MultimodalTrainStep(model, lr=LR, mm_projector_lr=PLR, mm_language_lr=lang, weight_decay=wd)

        def lr_ref(name):
            if "modal_projectors" in name:
                return PLR
            if lang is not None and ("lora_A.default" in name or "lora_B.default" in name):
                return PLR
            return LR
        wd_ref = lambda name: 0.0 if name.endswith("bias") else wd
        st.forward_backward(*args)
        p0 = {k: v.clone() for k, v in st.named_parameters().items()}
        g0 = {k: v.clone() for k, v in st.named_gradients().items()}
        assert any("lora_A.default" in k for k in p0) and any("lora_B.vision" in k for k in p0) and any("modal_projectors" in k for k in p0)
        st.optimizer_step()
        want = _torch_adamw(p0, g0, lr_ref, wd_ref)
        got = st.named_parameters()
        moved = {}
        for k in p0:
            x, w = got[k].float().cpu().reshape(want[k].shape), want[k]
            assert (x - w).abs().max().item() <= 2e-7 + 1e-6 * w.abs().max().item(), (k, lang, (x - w).abs().max().item())
            moved[k] = (x - p0[k].float().cpu().reshape(x.shape)).abs().max().item()
            assert st.lr_of(k) == lr_ref(k), k
        # the first AdamW step moves every element with a non-zero gradient by ~lr: the two groups are 10x apart
        a_def = max(v for k, v in moved.items() if "lora_A.default" in k)
        a_vis = max(v for k, v in moved.items() if "lora_A.vision" in k)
        if lang is not None:
            assert a_vis > 5 * a_def > 0, (a_vis, a_def)
        else:
            assert 0.5 < a_vis / a_def < 2.0, (a_vis, a_def)
        # the bf16 working copy was refreshed by the same launch
        assert torch.equal(st.P16, st.P.to(torch.bfloat16))


def test_gradient_accumulation_sums_micro_batches_and_steps_on_their_mean():
    """VERDICT r3 missing #2: --gradient_accumulation_steps (run_finetune_vision_damc.sh:45).  accumulate=True micro-batches add their
    gradients to a pending sum, the closing call leaves the SUM in G (exchanged once), AdamW steps on sum / micro-batches."""
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    a, meta, sd = load_golden("g9_train_step")
    model = build_from_state_dict(dict(meta, lora_dropout=0.0), sd)
    st = MultimodalTrainStep(model, lr=1e-3)
    mb1 = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    mb2 = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": (a["pixels"] * 0.5).cuda()})
    g = torch.Generator().manual_seed(3)
    ids3 = torch.cat([torch.ones(2, 1, dtype=torch.long), torch.randint(3, meta["vocab_size"] - 1, (2, 9), generator=g)], 1)
    lab3 = ids3.clone()
    lab3[:, :2] = -100
    mb3 = (ids3.cuda(), lab3.cuda(), {})                                 # text only: no projector gradient from this one
    l, G = [], []
    for mb in (mb1, mb2, mb3):
        l.append(st.forward_backward(*mb).item())
        G.append(st.G.clone())
    p0 = {k: v.clone() for k, v in st.named_parameters().items()}
    la = st.forward_backward(*mb1, accumulate=True).item()
    with pytest.raises(RuntimeError):
        st.optimizer_step()                                              # a step is closed by an accumulate=False call
    lb = st.forward_backward(*mb2, accumulate=True).item()
    lc = st.forward_backward(*mb3).item()
    assert (la, lb, lc) == tuple(l)
    assert torch.equal(st.G, (G[0] + G[1]) + G[2])                       # fp32 adds in this order, nothing else touches the sum
    gsum = {k: v.clone() for k, v in st.named_gradients().items()}
    st.optimizer_step()
    want = _torch_adamw(p0, {k: v / 3 for k, v in gsum.items()}, lambda k: 1e-3, lambda k: 0.0)
    got = st.named_parameters()
    for k in p0:
        x, w = got[k].float().cpu().reshape(want[k].shape), want[k]
        assert (x - w).abs().max().item() <= 2e-7 + 1e-6 * w.abs().max().item(), k
    assert st._aux_steps["vision"] == 1 and st.step_count == 1
    # the convenience loop: same three micro-batches from the same start -> same parameters
    st2 = MultimodalTrainStep(model, lr=1e-3)
    mean_loss = st2.step_accumulated([mb1, mb2, mb3]).item()
    assert abs(mean_loss - sum(l) / 3) < 1e-6
    assert torch.equal(st2.P, st.P)
    # and a plain step afterwards is a plain step again
    st2.step(*mb1)
    assert st2._micro == 1 and st2._accum_n == 0


def test_gradient_exchange_and_id_gather_over_rccl_in_a_world_of_one(tmp_path):
    """The N > 1 code path - RCCL process group, bucketed gradient all-reduce overlapped with the backward, all-gather of generated ids -
    forced in a world of one rank so that a single-GPU box exercises it (a separate process: the process group is global state)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import torch.distributed as dist
from conftest import load_golden
from modelcompose_amd.dist import gather_ids
from modelcompose_amd.model.builder import build_from_state_dict
from modelcompose_amd.train import MultimodalTrainStep
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a, meta, sd = load_golden("g9_train_step")
model = build_from_state_dict(dict(meta, lora_dropout=0.0), sd)
args = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
ref = MultimodalTrainStep(model, lr=1e-3)
l0 = ref.forward_backward(*args); g0 = ref.G.clone()
st = MultimodalTrainStep(model, lr=1e-3, force_exchange=True, bucket_layers=1)
assert st._exchange and len(st._buckets) >= 2
l1 = st.forward_backward(*args)
torch.cuda.synchronize()
assert l0.item() == l1.item() and torch.equal(g0, st.G), "all-reduce over one rank must be the identity"
ids = model.generate(a["input_ids"].cuda(), modal_inputs={"vision": a["pixels"].cuda()}, max_new_tokens=4, ignore_eos=True)[:, a["input_ids"].shape[1]:]
out = gather_ids(ids, 1, force=True)
assert torch.equal(out, ids)
dist.destroy_process_group()
print("RCCL_WORLD_OF_ONE_OK")
''' % (root, root)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert "RCCL_WORLD_OF_ONE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_audio_recipe_qformer_projector_backward_matches_the_oracle():
    """Audio stage-2 recipe (run_finetune_audio_damc.sh:37-38: BEATs encoder frozen, qformer projector + LoRA trained): loss and every
    gradient - incl. all Q-Former parameters (queries, position embedding, LayerNorms, self- / cross-attention, FFN, output projection) -
    against the autograd oracle.  Base weights, BEATs and the Q-Former come from the reference fixture g8; the LoRA pair is new (r = 32)."""
    import math
    from modelcompose_amd.model.builder import build_from_state_dict
    from modelcompose_amd.train import MultimodalTrainStep
    from oracle import train as otrain
    a, meta8, sd8 = load_golden("g8_e2e_4modal")
    keep = lambda k: not (".lora_" in k or k.startswith("model.modal_encoders.") and not k.startswith("model.modal_encoders.audio.")
                          or k.startswith("model.modal_projectors.") and not k.startswith("model.modal_projectors.audio.")
                          or (k.startswith("prefix_tokens.") or k.startswith("suffix_tokens.")) and not k.endswith(".audio"))
    sd = {k: v for k, v in sd8.items() if keep(k)}
    meta = {k: v for k, v in meta8.items() if k not in ("clip", "video", "point", "mm_vision_encoder", "mm_video_encoder", "mm_point_encoder")}
    meta.update(modal_names=["default", "audio"], reset_scaling_weights=None, lora_r=32, lora_alpha=64, lora_dropout=0.0)
    g = torch.Generator().manual_seed(5)
    Hd, I = meta["hidden_size"], meta["intermediate_size"]
    for l in range(meta["num_hidden_layers"]):
        for blk, lin, n, k in (("self_attn", "q_proj", Hd, Hd), ("self_attn", "k_proj", Hd, Hd), ("self_attn", "v_proj", Hd, Hd), ("self_attn", "o_proj", Hd, Hd),
                               ("mlp", "gate_proj", I, Hd), ("mlp", "up_proj", I, Hd), ("mlp", "down_proj", Hd, I)):
            for ad in ("default", "audio"):
                p = f"model.layers.{l}.{blk}.{lin}"
                sd[f"{p}.lora_A.{ad}.weight"] = (torch.rand(32, k, generator=g) * 2 - 1) / math.sqrt(k)
                sd[f"{p}.lora_B.{ad}.weight"] = torch.randn(n, 32, generator=g) * 0.02
    model = build_from_state_dict(meta, sd)
    st = MultimodalTrainStep(model, lr=1e-3, dropout_seed=77)
    qt = st.qformers["audio"]
    assert qt.p_hidden == 0.1 and qt.p_attn == 0.1                # BertConfig defaults (the reference builds a default config)
    A = -203
    r = lambda n: torch.randint(3, meta["vocab_size"] - 1, (n,), generator=g).tolist()
    ids = torch.tensor([[1] + r(4) + [A, 13] + r(7), [1] + r(2) + [A, 13] + r(9)])
    labels = ids.clone()
    labels[:, :5] = -100
    labels[ids == A] = -100
    mi = {"audio": {"audio_inputs": a["fbank"], "audio_padding_mask": a["padding_mask"]}}
    mid = {"audio": {"audio_inputs": a["fbank"].cuda(), "audio_padding_mask": a["padding_mask"].cuda()}}
    from oracle import encoders_extra as oex
    from oracle import philox
    import numpy as np
    # (1) as model.eval() would run the projector: no dropout inside it
    qt.training = False
    loss = st.forward_backward(ids.cuda(), labels.cuda(), mid)
    ref_loss, _, ref_grads = otrain.loss_and_grads(sd, meta, ids, labels, mi)
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    got = st.named_gradients()
    qf = [k for k in ref_grads if k.startswith("model.modal_projectors.audio.")]
    assert len(qf) >= 40 and all(k in got for k in qf)
    worst = _check_grads({k: got[k] for k in ref_grads}, ref_grads, tol=8e-2, min_cos=0.99)
    print("worst relative gradient error (audio recipe, no dropout)", worst)
    loss_nodrop = loss.item()
    # (2) as the reference trains it: dropout 0.1 on the embeddings, the attention probabilities and the three output dense
    # layers of every layer (Qformer.py:108, :259, :288, :374).  The oracle applies the step's own Philox masks, regenerated in numpy.
    qt.training = True
    loss = st.forward_backward(ids.cuda(), labels.cuda(), mid)
    seed = st._seed
    pj = qt.proj
    sites = {"self.out": qt.SITE_SELF_OUT, "cross.out": qt.SITE_CROSS_OUT, "ffn.out": qt.SITE_FFN_OUT, "self.probs": qt.SITE_SELF_PROBS,
             "cross.probs": qt.SITE_CROSS_PROBS}
    used = []

    def qdrop(tag, x):
        layer, site = (0, qt.SITE_EMB) if tag == "emb" else (tag[0], sites[tag[1]])
        sid = qt.stream_id(layer, site)
        if x.dim() == 4:                                           # probabilities (B, H, Lq, S): element ((b H + h) Lq + q) S + key
            b_, h_, lq, s_ = x.shape
            keep = philox.dropout_keep(b_ * h_ * lq, s_, qt.p_attn, seed, sid).reshape(x.shape)
            p_ = qt.p_attn
        else:                                                      # hidden states (B, N, Dm): element (b N + n) Dm + j
            b_, n_, d_ = x.shape
            keep = philox.dropout_keep(b_ * n_, d_, qt.p_hidden, seed, sid).reshape(x.shape)
            p_ = qt.p_hidden
        used.append((tag, float(keep.mean())))
        return x * torch.from_numpy(keep.astype(np.float32) / (1.0 - p_))
    oex.QFORMER_DROPOUT = qdrop
    try:
        ref_loss_d, _, ref_grads_d = otrain.loss_and_grads(sd, meta, ids, labels, mi)
    finally:
        oex.QFORMER_DROPOUT = None
    assert len(used) == 1 + 5 * pj.nl and all(0.8 < m < 0.97 for _, m in used), used
    assert abs(ref_loss_d.item() - ref_loss.item()) > 1e-4          # the masks change the function
    assert abs(loss.item() - ref_loss_d.item()) < 2e-2 * abs(ref_loss_d.item())
    assert abs(loss.item() - ref_loss_d.item()) < abs(loss_nodrop - ref_loss_d.item()) + 2e-3 * abs(ref_loss_d.item())
    got = st.named_gradients()
    worst = _check_grads({k: got[k] for k in ref_grads_d}, ref_grads_d, tol=8e-2, min_cos=0.99)
    print("worst relative gradient error (audio recipe, dropout 0.1 inside the Q-Former)", worst)
    # the masks are the step's: another step draws other masks, the same step the same ones
    l1 = st.forward_backward(ids.cuda(), labels.cuda(), mid).item()
    assert l1 == loss.item()
    st.step(ids.cuda(), labels.cuda(), mid)                       # the optimizer path over the Q-Former's parameter group
    assert st._aux_steps["audio"] == 1


def test_the_reference_train_call_sequence_runs_on_the_hip_classes(tmp_path):
    """VERDICT r5 #7 / SURVEY §8(b): train_multimodal.train's model-facing calls, in its order, against this repo's classes -
    `MultimodalLlamaForCausalLM.from_pretrained(base, lora kwargs, mm_vision_encoder=...)` (train_multimodal.py:307-325),
    `model.get_model().initialize_multimodal_modules(model_args, fsdp)` (:396-399), `get_modal_encoders().to(...)`, the requires_grad
    selection of lora_strategy 'modal+language' (:436-465) - then the stage-2 step.  The LoRA factors of a from_pretrained model start as
    peft's reset does (B = 0): the first loss equals the un-adapted model's; one step moves selected tensors only; a projector frozen
    by the caller keeps its values; a selection the step has no backward for is refused."""
    import json
    from types import SimpleNamespace
    from modelcompose.model import MultimodalLlamaForCausalLM
    from modelcompose_amd.train import MultimodalTrainStep
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    a, meta, sd = load_golden("g9_train_step")
    clip_dir, base = tmp_path / "clip-tiny", tmp_path / "vicuna-tiny"
    clip_dir.mkdir(); base.mkdir()
    pre = "model.modal_encoders.vision.vision_tower."
    torch.save({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}, clip_dir / "pytorch_model.bin")
    json.dump(dict(meta["clip"], model_type="clip_vision_model"), open(clip_dir / "config.json", "w"))
    json.dump({"crop_size": 28, "size": 28, "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]},
              open(clip_dir / "preprocessor_config.json", "w"))
    is_extra = lambda k: ".lora_" in k or k.startswith("prefix_tokens") or k.startswith("suffix_tokens") or k.startswith("model.modal_projectors.") or k.startswith(pre)
    torch.save({k: v for k, v in sd.items() if not is_extra(k)}, base / "pytorch_model.bin")
    plain = {k: v for k, v in meta.items() if k not in ("clip", "modal_names") and not k.startswith("mm_") and not k.startswith("lora_") and not k.startswith("local_")}
    json.dump(plain, open(base / "config.json", "w"))
    proj_file = tmp_path / "mm_projector.bin"
    torch.save({k: v for k, v in sd.items() if k.startswith("model.modal_projectors.")}, proj_file)

    def build(freeze_proj=False, strategy="modal+language"):
        training_args = SimpleNamespace(lora_strategy=strategy, lora_r=meta["lora_r"], lora_alpha=meta["lora_alpha"], lora_dropout=0.0, fsdp=None, bf16=True,
                                        device="cuda", freeze_mm_mlp_adapter=freeze_proj, cache_dir=None)
        model_args = SimpleNamespace(model_name_or_path=str(base), mm_vision_encoder=str(clip_dir), mm_audio_encoder=None, mm_video_encoder=None,
                                     mm_point_encoder=None, mm_projector_type=meta.get("mm_projector_type", "linear"), mm_vision_select_layer=-2,
                                     mm_vision_select_feature="patch", pretrain_mm_mlp_adapter=str(proj_file), tune_mm_mlp_adapter=False,
                                     local_prefix_tokens=meta.get("local_prefix_tokens", 0), local_suffix_tokens=meta.get("local_suffix_tokens", 0),
                                     layer_local_tokens=False, seperate_layernorm=False, freeze_backbone=False)
        # ---- train_multimodal.py:307-325
        model = MultimodalLlamaForCausalLM.from_pretrained(
            model_args.model_name_or_path, cache_dir=training_args.cache_dir, lora_strategy=training_args.lora_strategy, lora_r=training_args.lora_r,
            lora_alpha=training_args.lora_alpha, lora_dropout=training_args.lora_dropout, local_prefix_tokens=model_args.local_prefix_tokens,
            local_suffix_tokens=model_args.local_suffix_tokens, layer_local_tokens=model_args.layer_local_tokens,
            seperate_layernorm=model_args.seperate_layernorm, mm_vision_encoder=model_args.mm_vision_encoder, mm_audio_encoder=model_args.mm_audio_encoder,
            mm_video_encoder=model_args.mm_video_encoder, mm_point_encoder=model_args.mm_point_encoder)
        model.config.use_cache = False
        # ---- :396-465
        model.get_model().initialize_multimodal_modules(model_args=model_args, fsdp=training_args.fsdp)
        model.get_modal_encoders().to(dtype=torch.bfloat16, device=training_args.device)
        assert set(model.get_modal_processors()) == {"vision"}
        if training_args.freeze_mm_mlp_adapter:                                   # :431-434 (the lora_strategy block below re-enables them: reference behaviour)
            for p in model.get_modal_projectors().parameters():
                p.requires_grad = False
        if training_args.lora_strategy is not None:
            model.requires_grad_(False)
            for n, p in model.named_parameters():
                if "prefix_tokens" in n or "suffix_tokens" in n:
                    p.requires_grad = True
            for p in model.get_modal_projectors().parameters():
                p.requires_grad = True
            for n, p in model.get_model().named_parameters():
                if "lora" not in n:
                    continue
                if training_args.lora_strategy == "modal+language":
                    p.requires_grad = True
                elif training_args.lora_strategy == "same" and ("lora_A.default" in n or "lora_B.default" in n):
                    p.requires_grad = True
        if freeze_proj:                                                          # a caller that really wants the projector fixed freezes it AFTER the block
            for p in model.get_modal_projectors().parameters():
                p.requires_grad = False
        return model

    model = build()
    names = model.trainable_names()
    assert any(".lora_A.vision." in n for n in names) and any(".lora_B.default." in n for n in names) and any("modal_projectors.vision" in n for n in names)
    assert not any(n.endswith("q_proj.weight") for n in names)                      # the base stays frozen
    assert all(float(v.abs().max()) == 0.0 for k, v in model._raw.items() if ".lora_B." in k)      # peft's reset: B = 0
    st = MultimodalTrainStep(model, lr=1e-3)
    args = (a["input_ids"].cuda(), a["labels"].cuda(), {"vision": a["pixels"].cuda()})
    l0 = st.forward_backward(*args).item()
    assert l0 == l0 and l0 > 0
    p_before = st.P.clone()
    for _ in range(3):
        st.step(*args)
    assert st.forward_backward(*args).item() < l0
    assert not torch.equal(st.P, p_before)
    # projector tensors frozen by the caller: they keep their values (learning rate 0), the LoRA factors move
    model2 = build(freeze_proj=True)
    st2 = MultimodalTrainStep(model2, lr=1e-3)
    proj = [p_ for n, p_ in st2.params.items() if n.startswith("model.modal_projectors.")]
    assert proj and st2._frozen_names
    before = st2.P.clone()
    st2.step(*args)
    for p_ in proj:
        assert torch.equal(st2.P[p_.off:p_.off + p_.n], before[p_.off:p_.off + p_.n]), p_.name
    assert not torch.equal(st2.P, before)
    # a selection without a backward here ('same': the default adapter only) is refused, not silently trained as 'modal+language'
    model3 = build(strategy="same")
    with pytest.raises(NotImplementedError):
        MultimodalTrainStep(model3, lr=1e-3)
