"""bench.py's N > 1 code path on the one-GPU box: MC_BENCH_FORCE_DIST=1 runs RCCL init, the barriers, the id all-gather and the MAX
all-reduce of the timing in a world of one; `--gpus` must equal the world."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(argv, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=900, env=env)


def test_bench_world_of_one_over_rccl():
    r = _bench(["--gpus", "1", "--layers", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--new-tokens", "4", "--no-profile",
                "--no-cpu-baseline", "--no-secondary"], {"MC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["value"] > 0 and j["config"]["parallelism"] == "dp1"


def test_bench_rejects_a_world_that_is_not_gpus():
    r = _bench(["--gpus", "2", "--layers", "2", "--steps", "1"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr)


def test_bare_launch_of_two_ranks_runs_the_n_rank_path_with_two_processes():
    """`python bench.py --gpus 2` bare: the launcher starts two rank processes.  On this one-GPU box both share the device and
    the collectives run over gloo (MC_BENCH_SHARE_GPU / MC_BENCH_BACKEND: functional only) - barriers, the MAX all-reduce of the timing and the
    rank-major gather of the ids execute across two real processes, and rank 0's line reports the whole job: n_gpus 2, value = 2 x B x K / t."""
    r = _bench(["--gpus", "2", "--layers", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--new-tokens", "4", "--no-profile",
                "--no-cpu-baseline", "--no-secondary"], {"MC_BENCH_SHARE_GPU": "1", "MC_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                          # rank 0's line only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "dp2" and j["steps"] == 2
    assert abs(j["value"] - 2 * 4 * 2 / (j["ms_per_step"] * 2 / 1e3)) < 1e-2 * j["value"]


def test_two_rank_finetune_step_all_reduces_the_gradients_across_two_processes():
    """configs[4] with two real ranks (same functional set-up): the bucketed gradient all-reduce of the train step runs between two
    processes, the line reports ddp2 and a finite loss."""
    r = _bench(["--gpus", "2", "--workload", "train", "--layers", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--no-profile",
                "--no-cpu-baseline", "--no-secondary"], {"MC_BENCH_SHARE_GPU": "1", "MC_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["config"]["parallelism"] == "ddp2" and j["value"] > 0
    assert j["config"]["final_loss"] == j["config"]["final_loss"] and 0 < j["config"]["final_loss"] < 20       # finite
