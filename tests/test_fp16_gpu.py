"""The fp16 instantiation of the library (libmc_hip_f16.so: the same kernels on IEEE-half storage, fp32 accumulation) - the reference's own
inference dtype (modelcompose/model/builder.py:41, :162, :185; eval/model_multimodal_qa_loader.py:57-58) - as the PARITY INSTRUMENT: with an
8x finer mantissa the distance from the reference's fp32 outputs must shrink by about that factor.  If it did not, the bf16 error would not be
rounding.  One storage dtype per process: each dtype runs tools/fp16_parity.py in a child process; bf16 stays the headline (BASELINE.json)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(dtype, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fp16_parity.py"), "--dtype", dtype, "--case"] + list(cases),
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_fp16_storage_is_about_eight_times_closer_to_the_reference_than_bf16():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    f16 = _run("fp16", ["g4", "g8"])
    b16 = _run("bf16", ["g4", "g8"])
    assert f16["library"] == "libmc_hip_f16.so" and b16["library"] == "libmc_hip.so"
    rep = {"fp16": f16["cases"], "bf16": b16["cases"]}
    print(json.dumps(rep))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump(rep, open(os.path.join(out, "fp16_vs_bf16_tiny.json"), "w"), indent=1)
    for case in ("g4", "g8"):
        f, b = f16["cases"][case], b16["cases"][case]
        assert f["ids_equal"] and b["ids_equal"], (case, f, b)
        # north_star's tolerance (1e-3 of the logit scale) against the REFERENCE's fp32 outputs; 2e-3 for the four-encoder model, whose
        # feature blocks pass through three more networks
        bound = 1e-3 if case == "g4" else 2e-3
        assert f["prefill_logits_err"] < bound and f["step_logits_err"] < bound, (case, f)
        # and the bf16 error is rounding: it shrinks by about the ratio of the mantissas (8x; at least 3x asserted)
        assert f["step_logits_err"] * 3 < b["step_logits_err"], (case, f, b)
