"""TIES merging on the GPU (csrc/merge.hip) against the reference's outputs (tests/golden/g10_ties.npz) and, at sizes the
fixtures cannot hold, against the oracle restatement — bit-exact: it is compare / select / small-sum work."""
import json

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_ties_golden_tensor_and_file_level(tmp_path):
    from modelcompose_amd import compose
    a, meta, _ = load_golden("g10_ties")
    keys = meta["shared_keys"]
    cks = [{k: a[f"in::{i}::{k}"] for k in keys} for i in range(3)]
    for func in ("mean", "sum", "max"):
        for K in (20, 50):
            got = compose.ties_merge_state_dicts(cks, K, func)
            for k in keys:
                assert torch.equal(got[k], a[f"out::{func}::{K}::{k}"]), (func, K, k)
    demo = compose.ties_merge_state_dicts([{"x": torch.Tensor([1, 2, 3]), "y": torch.Tensor([4, 5, 6])},
                                           {"x": torch.Tensor([-1, 2, 3]), "y": torch.Tensor([0, 0, 0])}], 0.9, "mean")
    assert torch.equal(demo["x"], a["demo::x"]) and torch.equal(demo["y"], a["demo::y"])
    paths = []
    for i, modal in enumerate(meta["order"]):
        d = tmp_path / f"ckpt-{modal}"
        d.mkdir()
        w = dict(cks[i])
        w.update({k.split("::", 2)[2]: v for k, v in a.items() if k.startswith(f"fin::{modal}::")})
        torch.save(w, d / "adapter_model.bin")
        json.dump(meta["in_configs"][modal], open(d / "config.json", "w"))
        paths.append(str(d))
    out = tmp_path / "merged"
    compose.merge_checkpoints(paths, str(out), "ties-mean", K=20)
    got = torch.load(out / "adapter_model.bin")
    exp = {k[6:]: v for k, v in a.items() if k.startswith("fout::")}
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert torch.equal(got[k], exp[k]), k
    assert json.load(open(out / "config.json")) == meta["out_config"]
    assert open(out / "merge_info.txt").read().replace(str(tmp_path), "<TMP>") == meta["merge_info"]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n", [2, 3, 5])
def test_ties_large_vectors_bit_exact_vs_oracle(dtype, n):
    """1.5 M-element task vectors with many repeated magnitudes (bf16 has few distinct values: heavy ties at the threshold)."""
    from modelcompose_amd import compose
    from oracle import merge as omerge
    g = torch.Generator().manual_seed(7 + n)
    d = 1_500_003
    flat = (torch.randn(n, d, generator=g) * 0.02).to(dtype)
    flat[:, :1000] = 0
    for func in ("mean", "sum", "max"):
        ref = omerge.ties_merge_vectors(flat.clone(), 20, func)
        got = compose.ties_merge_vectors(flat.cuda(), 20, func).cpu()
        # the reference copies the merged vector back into tensors of the checkpoint dtype (vector_to_parameters, :217-219)
        assert torch.equal(got.float(), ref.to(dtype).float()), (dtype, n, func)


def test_interference_metrics_golden_tensor_and_file_level(tmp_path):
    """calculate_metrics.py on the GPU: one-pass double-precision partial sums vs the reference's fp32 torch reductions (tolerance
    1e-5 relative, the error of the reference's own fp32 sums), then the file-level flow writing merge_metrics.txt."""
    from modelcompose_amd import compose
    a, meta, _ = load_golden("g11_metrics")
    for case in meta["cases"]:
        got = compose.interference_metrics(a[f"flat::{case['n']}"].cuda(), 50)
        for k, v in case["expected"].items():
            assert abs(got[k] - v) <= 1e-5 * max(1.0, abs(v)), (case["n"], k, got[k], v)
    paths = []
    for modal in meta["order"]:
        d = tmp_path / f"ckpt-{modal}"
        d.mkdir()
        torch.save({k.split("::", 2)[2]: v for k, v in a.items() if k.startswith(f"fin::{modal}::")}, d / "adapter_model.bin")
        json.dump(meta["in_configs"][modal], open(d / "config.json", "w"))
        paths.append(str(d))
    out = tmp_path / "merged"
    compose.merge_checkpoints(paths, str(out), "ties-mean", K=20)
    compose.calculate_metrics(str(out))
    exp = dict(line.split(": ") for line in meta["merge_metrics"].strip().split("\n"))
    got = dict(line.split(": ") for line in open(out / "merge_metrics.txt").read().strip().split("\n"))
    assert list(got) == list(exp) == ["L2", "Cosine", "SSD", "TSSD"]
    for k in exp:
        assert abs(float(got[k]) - float(exp[k])) <= 1e-5 * max(1.0, abs(float(exp[k]))), (k, got[k], exp[k])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_interference_metrics_large_vs_oracle(dtype):
    from modelcompose_amd import compose
    from oracle import merge as omerge
    g = torch.Generator().manual_seed(23)
    flat = (torch.randn(3, 3_000_001, generator=g) * 0.02).to(dtype)
    flat[:, :5000] = 0
    ref = omerge.interference_metrics(flat.double(), 50)          # float64 oracle: the GPU sums are double too
    got = compose.interference_metrics(flat.cuda(), 50)
    for k, v in ref.items():
        assert abs(got[k] - v) <= 2e-6 * max(1.0, abs(v)), (k, got[k], v)


def test_convert_ties_and_drop_strategies_match_the_reference_script(tmp_path):
    """merge_unimodal_modelcompose.py:42-73 with the TIES arithmetic on the GPU: convert-ties-mean, convert-drop-mean, convert-drop-sum."""
    from conftest import run_g16_cases
    run_g16_cases(tmp_path, ["ties", "drop", "dropsum"])
