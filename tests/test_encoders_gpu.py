"""GPU parity of the audio / video / point encoders and the Q-Former projector against fixtures produced by the
reference (tests/golden/g5_*.npz).  Tolerances: bf16 storage through a few transformer layers vs the fp32 reference,
stated as a fraction of the feature scale; index work (FPS centres, kNN sets) bit-exact."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    return (got.float().cpu() - ref.float()).abs().max().item() / ref.float().abs().max().item()


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_beats_encoder_matches_reference():
    from modelcompose_amd.model.encoders_extra import BeatsConfig, HipBeatsAudioEncoder
    a, cfg, sd = load_golden("g5_beats")
    enc = HipBeatsAudioEncoder(None, None, config=BeatsConfig(cfg))
    enc.load_state_dict(sd)
    f, mask = enc(a["fbank"].cuda(), a["padding_mask"].cuda())
    assert enc._pending_mask_check and not HipBeatsAudioEncoder(None, None, config=BeatsConfig(cfg))._pending_mask_check      # a device mask defers its check
    assert torch.equal(mask.cpu(), ~a["pooled_mask"])
    valid = ~a["pooled_mask"]
    # padded positions are don't-care (never attended, ignored by the caller: multimodal_arch.py:233-235)
    assert rel_err(f.cpu()[valid], a["features"][valid]) < 2 ** -5
    f2, _ = enc(a["fbank"].cuda(), None)
    assert rel_err(f2, a["features_nopad"]) < 2 ** -5
    # round 4: a DEVICE mask is analysed on the device (no host round trip: one launch pools the frame mask, writes the clip lengths,
    # zeroes the padded rows); a HOST mask on the host.  Same function: bitwise equal features and masks.
    fh, mh = enc(a["fbank"].cuda(), a["padding_mask"])                   # host mask (every forward first collects a pending device check)
    assert not enc._pending_mask_check
    enc.check_pending()
    assert torch.equal(fh, f) and torch.equal(mh, mask)
    # padding that is not a suffix: refused at once for a host mask, at the deferred check for a device mask
    bad = a["padding_mask"].clone()
    bad[:, : bad.shape[1] // 4] = True
    bad[:, bad.shape[1] // 4:] = False
    with pytest.raises(NotImplementedError):
        enc(a["fbank"].cuda(), bad)
    enc(a["fbank"].cuda(), bad.cuda())
    with pytest.raises(NotImplementedError):
        enc.check_pending()
    enc(a["fbank"].cuda(), a["padding_mask"].cuda())
    enc.check_pending()                                                 # the flag was cleared: a good mask passes again
    # a standalone user who never calls check_pending() still hears about it - at the encoder's next forward
    enc(a["fbank"].cuda(), bad.cuda())
    with pytest.raises(NotImplementedError):
        enc(a["fbank"].cuda(), None)
    enc(a["fbank"].cuda(), None)


def test_qformer_projector_matches_reference():
    from modelcompose_amd.model.encoders_extra import HipQformerProjector
    a, cfg, sd = load_golden("g5_qformer")
    proj = HipQformerProjector(cfg["num_query_token"], cfg["encoder_width"], cfg["num_hidden_layers"], 64, cfg["hidden_size"],
                               cfg["num_attention_heads"], cfg["intermediate_size"], cfg["layer_norm_eps"])
    proj.load_state_dict(sd)
    y = proj(a["x"].cuda())
    assert y.shape == a["y"].shape
    assert rel_err(y, a["y"]) < 2 ** -5


def test_languagebind_video_tower_matches_reference():
    from modelcompose_amd.model.encoders_extra import HipLanguageBindVideoTower, VideoConfig
    a, cfg, sd = load_golden("g5_video")
    tower = HipLanguageBindVideoTower(None, None, delay_load=True, config=VideoConfig(**cfg))
    tower.load_state_dict(sd)
    f = tower(a["video"].cuda())
    assert f.shape == a["hs_m2"].shape
    assert rel_err(f, a["hs_m2"]) < 2 ** -5
    assert rel_err(tower.hidden_state(a["video"].cuda(), 0), a["hs0"]) < 2 ** -6


def test_pointbert_encoder_matches_reference():
    from modelcompose_amd.model.encoders_extra import HipPointEncoder, PointConfig
    from oracle import encoders_extra as ex
    a, cfg, sd = load_golden("g5_point")
    enc = HipPointEncoder(None, None, delay_load=True, config=PointConfig(**cfg))
    enc.load_state_dict(sd)
    enc.fps_start = a["fps_start"]
    pts = a["points"].to(torch.bfloat16)
    y, cidx, nidx, centers = enc.forward(pts.cuda(), return_aux=True)
    # the oracle on the same bf16-valued points: FPS centres and kNN neighbour sets are integer work -> bit-exact
    yo, cidx_o, nidx_o, center_o = ex.pointbert_encode(pts.float(), sd, cfg, a["fps_start"], return_aux=True)
    assert torch.equal(cidx.cpu().long(), cidx_o)
    assert torch.equal(centers.cpu(), center_o)
    assert torch.equal(nidx.cpu().long().sort(-1).values, nidx_o.sort(-1).values)
    assert rel_err(y, yo) < 2 ** -5
    # and against the reference's own output (fp32 points): only the bf16 rounding of the input coordinates differs
    assert rel_err(y, a["features"]) < 2 ** -4


def test_imagebind_audio_branch_matches_reference():
    from modelcompose_amd.model.imagebind_audio import HipImageBindAudioEncoder
    a, cfg, sd = load_golden("g5_imagebind")
    enc = HipImageBindAudioEncoder(None, None, delay_load=True, config=cfg)
    enc.load_state_dict(sd)
    cls, y = enc.forward(a["x"].cuda(), return_cls=True)
    assert y.shape == a["out"].shape
    assert rel_err(cls, a["cls_feature"]) < 2 ** -5
    assert rel_err(y, a["out"]) < 2 ** -5


def test_beats_audio_processor_fbank_matches_oracle():
    """Kaldi log-mel front-end + BEATs normalisation + padding on the GPU against the oracle restatement (itself cross-checked with
    an independent implementation; torchaudio, the reference's dependency, is absent -> parity unpinned by the reference).
    Tolerance: fp32 FFT / summation order on values of magnitude ~1 after normalisation."""
    import numpy as np
    from modelcompose_amd.model.audio_processor import HipBeatsAudioProcessor
    from oracle import fbank
    rng = np.random.default_rng(5)
    proc = HipBeatsAudioProcessor()
    for T in (16000 * 10, 16000 * 3 + 77, 399, 400):
        wav = (rng.standard_normal(T) * 0.1 + 0.01).astype(np.float32)
        ref, mask_ref = fbank.beats_process_waveform(wav)
        got, mask = proc(torch.from_numpy(wav))
        assert got.shape == ref.shape == (1024, 128) and mask.shape == (1024,) and not bool(mask.any())
        m = 1 + (T - 400) // 160 if T >= 400 else 0
        assert torch.count_nonzero(got[m:]) == 0
        f32 = proc.fbank(torch.from_numpy(wav).cuda().view(1, -1), torch.tensor([T], dtype=torch.int32), 1024, out_dtype=torch.float32)[0]
        assert np.abs(f32.cpu().numpy() - ref).max() < 2e-3
        assert (got.float().cpu() - torch.from_numpy(ref)).abs().max().item() < 2 ** -6        # bf16 storage of values within ~[-2, 2]
    pe = HipBeatsAudioProcessor(is_eval=True)
    wav = (rng.standard_normal(16000 * 7) * 0.1).astype(np.float32)
    ref, _ = fbank.beats_process_waveform(wav, is_eval=True)
    got, _ = pe(torch.from_numpy(wav))
    assert got.shape == ref.shape
    assert (got.float().cpu() - torch.from_numpy(ref)).abs().max().item() < 2 ** -6


def test_clip_image_processor_bit_exact_vs_oracle():
    """expand2square + PIL bicubic resize + centre crop on the GPU: the uint8 image is bit-identical to the oracle (= PIL), the
    normalised float32 output equals the oracle's (= transformers CLIPImageProcessor) exactly."""
    import numpy as np
    from modelcompose_amd.model.image_processor import HipCLIPImageProcessor
    from modelcompose_amd.mm_utils import process_images
    from oracle import image as oi
    rng = np.random.default_rng(2)
    proc = HipCLIPImageProcessor(336)
    for (h, w) in ((480, 640), (700, 500), (336, 336), (90, 200), (1200, 1600)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for pad in (True, False):
            ref = oi.clip_preprocess(img, 336, pad)
            bg = tuple(int(x * 255) for x in proc.image_mean) if pad else None
            got, u8 = proc._one(img, bg, out_u8=True)
            sq = oi.expand2square(img, bg) if pad else img
            hh, ww = sq.shape[:2]
            short, long_ = (ww, hh) if ww <= hh else (hh, ww)
            oh, ow = (int(336 * long_ / short), 336) if ww <= hh else (336, int(336 * long_ / short))
            r8 = oi.resize_bicubic_u8(sq, oh, ow)
            r8 = r8[(oh - 336) // 2:(oh - 336) // 2 + 336, (ow - 336) // 2:(ow - 336) // 2 + 336]
            assert np.array_equal(u8.cpu().numpy(), r8), (h, w, pad)
            assert np.array_equal(got.cpu().numpy(), ref), (h, w, pad)
    cfg = type("C", (), {"image_aspect_ratio": "pad"})()
    batch = process_images([rng.integers(0, 256, (300, 400, 3), dtype=np.uint8) for _ in range(2)], proc, cfg)
    assert batch.shape == (2, 3, 336, 336)


def test_languagebind_video_processor_matches_torch_transform():
    """/255 + normalise + bilinear short-side resize + centre crop (+ flip) on the GPU against the oracle, which applies the same torch
    primitives the reference's torchvision / pytorchvideo transforms call.  fp32 within 2e-6 of the value scale (FMA contraction)."""
    from modelcompose_amd.model.video_processor import HipLanguageBindVideoProcessor, sample_frame_ids
    from oracle import video as ov
    import numpy as np
    assert np.array_equal(sample_frame_ids(100, 8), ov.sample_frame_ids(100, 8))
    g = torch.Generator().manual_seed(8)
    proc = HipLanguageBindVideoProcessor(out_dtype=torch.float32)
    for (T, H, W) in ((8, 240, 320), (8, 360, 202), (2, 224, 224), (3, 100, 500)):
        fr = torch.randint(0, 256, (T, H, W, 3), generator=g, dtype=torch.uint8)
        for flip in (False, True):
            ref = ov.video_transform(fr, 224, flip)
            got = proc.transform(fr, flip).cpu()
            assert got.shape == ref.shape == (3, T, 224, 224)
            assert (got - ref).abs().max().item() < 5e-6 * ref.abs().max().item(), (T, H, W, flip)
    out = HipLanguageBindVideoProcessor()(images=[torch.randint(0, 256, (8, 120, 160, 3), generator=g, dtype=torch.uint8)] * 2)
    assert out["pixel_values"].shape == (2, 3, 8, 224, 224) and out["pixel_values"].dtype == torch.bfloat16


def test_fused_add_layernorm_and_in_place_temporal_attention_are_bit_identical():
    """Round 4 (LanguageBind-Video temporal branch, video/modeling_video.py:105-130): `h + temporal_embedding` + temporal_layer_norm1 in one
    pass equals add_rows followed by layernorm bit for bit; the temporal attention addressed in place over the (b t) n d layout
    (mc_attn_mask.b_inner) equals the attention over a permuted (b n) t d copy bit for bit; a launch the tiny kernel does not take
    refuses the split instead of mis-addressing."""
    from modelcompose_amd import _lib, ops
    g = torch.Generator().manual_seed(5)
    B, T, n, D, H = 3, 8, 37, 1024, 16
    d = D // H
    M = B * T * n
    h = torch.randn(M, D, generator=g).to(torch.bfloat16).cuda()
    temb = (torch.randn(T, D, generator=g) * 0.3).to(torch.bfloat16).cuda()
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
    b = (0.1 * torch.randn(D, generator=g)).to(torch.bfloat16).cuda()
    t_idx = ((torch.arange(M) // n) % T).to(torch.int32).cuda()
    s0 = ops.add_rows(h, temb, t_idx)
    n0 = ops.layernorm(s0, w, b, 1e-5)
    s1, n1 = ops.add_layernorm(h, temb, t_idx, w, b, 1e-5)
    assert torch.equal(s0, s1) and torch.equal(n0, n1)
    # attention: (b n) sequences of T tokens
    qkv = torch.randn(M, 3 * D, generator=g).to(torch.bfloat16).cuda()
    import numpy as np
    bb, nn_, t2 = np.meshgrid(np.arange(B), np.arange(n), np.arange(T), indexing="ij")
    perm = torch.from_numpy(((bb * T + t2) * n + nn_).reshape(-1).astype(np.int32)).cuda()
    qp = torch.empty_like(qkv)
    ops.copy_rows(qkv, qp, M, perm, None)                                   # the permuted copy the old path attended over
    a0 = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    st_t = (T * 3 * D, 3 * D, d)
    ops.attn_prefill(qp, qp[:, D:], qp[:, 2 * D:], a0, B * n, H, H, T, T, d, st_t, st_t, st_t, D, False, 0, scale=d ** -0.5, out_map=perm)
    a1 = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    st_tn = (T * n * 3 * D, n * 3 * D, d)
    ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a1, B * n, H, H, T, T, d, st_tn, st_tn, st_tn, D, False, 0, scale=d ** -0.5, out_map=perm,
                     batch_split=(n, 3 * D))
    assert torch.equal(a0, a1) and a1.float().abs().max().item() > 0
    # the split is an argument of ITS launch (mc_attn_mask): the next launch is an ordinary one
    a2 = torch.zeros(M, D, dtype=torch.bfloat16, device="cuda")
    ops.attn_prefill(qp, qp[:, D:], qp[:, 2 * D:], a2, B * n, H, H, T, T, d, st_t, st_t, st_t, D, False, 0, scale=d ** -0.5, out_map=perm)
    assert torch.equal(a2, a0)
    # and it is refused where the tiny kernel does not run (64 tokens per sequence)
    with pytest.raises(ValueError):
        ops.attn_prefill(qkv, qkv[:, D:], qkv[:, 2 * D:], a2, B, H, H, 64, 64, d, (64 * 3 * D, 3 * D, d), (64 * 3 * D, 3 * D, d), (64 * 3 * D, 3 * D, d), D,
                         False, 0, scale=d ** -0.5, batch_split=(n, 3 * D))
