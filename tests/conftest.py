import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """GPU sessions share the drawn 8- / 32-layer synthetic weights between their processes (tests/fullwidth_cases.py: _weight_cache_path):
    a per-session directory in /dev/shm (17 GB for the 32-layer model; the GPU box has terabytes of RAM) or the temp dir, removed at the end.
    Never on a box without a GPU: the CPU suite does not build those models, and this container's /dev/shm is its memory."""
    if os.environ.get("MC_TEST_WEIGHT_CACHE") or os.environ.get("MC_TEST_WEIGHT_CACHE_OFF") == "1":
        return
    try:
        if torch.cuda.device_count() < 1:                 # (does not initialise the GPU)
            return
        # the synthetic weights of the full-width cases are drawn by 16 threads (bit-identical: modelcompose_amd/synthetic.py; 165 -> 20 s
        # for the 32-layer model on the GPU box's host); child processes inherit both settings
        os.environ.setdefault("MC_SYNTH_THREADS", str(max(1, min(16, (os.cpu_count() or 4) // 4))))
        import shutil
        import tempfile
        base = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > (64 << 30) else tempfile.gettempdir()
        if shutil.disk_usage(base).free < (40 << 30):
            return
        d = tempfile.mkdtemp(prefix="mc_test_weights_", dir=base)
        os.environ["MC_TEST_WEIGHT_CACHE"] = d
        session.config._mc_weight_cache = d
    except Exception:
        pass


def pytest_sessionfinish(session, exitstatus):
    d = getattr(session.config, "_mc_weight_cache", None)
    if d:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
        os.environ.pop("MC_TEST_WEIGHT_CACHE", None)


def load_golden(name):
    """Returns (arrays: dict[str, torch.Tensor], meta: dict|None, sd: dict[str, Tensor])."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrays, sd, meta = {}, {}, None
    for k in z.files:
        if k == "meta":
            meta = json.loads(str(z[k]))
            continue
        t = torch.from_numpy(z[k])
        if k.startswith("sd::"):
            sd[k[4:]] = t
        else:
            arrays[k] = t
    return arrays, meta, sd


@pytest.fixture(scope="session")
def golden():
    return load_golden


def run_g16_cases(tmp_path, tags):
    """merge_checkpoints(convert-*) on the inputs of tests/golden/g16_merge_convert.npz: tensors, key ORDER, config.json and merge_info.txt
    must equal what the reference's own script wrote for the same inputs."""
    import json
    import torch
    from modelcompose_amd import compose
    a, meta, _ = load_golden("g16_merge_convert")
    paths = []
    for modal in meta["order"]:
        d = tmp_path / f"ckpt-{modal}"
        d.mkdir(exist_ok=True)
        torch.save({k.split("::", 2)[2]: v for k, v in a.items() if k.startswith(f"in::{modal}::")}, d / "adapter_model.bin")
        json.dump(meta["in_configs"][modal], open(d / "config.json", "w"))
        paths.append(str(d))
    for tag in tags:
        case = meta["cases"][tag]
        out = tmp_path / f"merged-{tag}"
        compose.merge_checkpoints(paths, str(out), case["strategy"], K=20)
        got = torch.load(out / "adapter_model.bin")
        assert list(got) == case["keys"], tag
        for k in case["keys"]:
            assert torch.equal(got[k], a[f"out::{tag}::{k}"]), (tag, k)
        assert json.load(open(out / "config.json")) == case["out_config"], tag
        assert open(out / "merge_info.txt").read().replace(str(tmp_path), "<TMP>") == case["merge_info"], tag
