"""`python bench.py --gpus N` starts N ranks itself (the reference starts its N workers from one command,
scripts/model_composition/test/MCUB-4.sh:21,42-70).  CPU tier: the launcher's command / environment construction and a real N-process
launch in probe mode (every rank reports the environment it was given and leaves before touching a GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_commands_build_one_child_per_gpu():
    import bench
    cmds = bench.rank_commands(4, ["--gpus", "4", "--steps", "3"], 29517, base_env={"PATH": "/usr/bin", "WORLD_SIZE_IGNORED": "x"})
    assert len(cmds) == 4
    for r, (cmd, env) in enumerate(cmds):
        assert cmd[0] == sys.executable and cmd[1] == os.path.join(ROOT, "bench.py")
        assert cmd[2:] == ["--gpus", "4", "--steps", "3"]
        assert env["RANK"] == str(r) and env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4"
        assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29517"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"          # dmabuf IPC only on this pool: RCCL needs it in every rank


def _run(argv, extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=300, env=env)


def test_bare_launch_starts_n_ranks_and_relays_rank0():
    r = _run(["--gpus", "3", "--steps", "7"], {"MC_BENCH_LAUNCH_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # only rank 0's line reaches stdout
    j = lines[0]
    assert j["rank"] == 0 and j["world"] == 3 and j["gpus"] == 3 and j["steps"] == 7
    assert j["master"].startswith("127.0.0.1:")


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "8"], {"MC_BENCH_LAUNCH_PROBE": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_failing_rank_fails_the_launch():
    # no GPU in this container: every rank stops at "bench.py needs an MI355X" -> the launcher must return non-zero, not hang
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--no-cpu-baseline"], {})
    assert r.returncode != 0


def _alive(pid):
    try:
        os.kill(pid, 0)
    except OSError:
        return False
    # a zombie still answers kill(0): it is gone for our purposes once its state is Z
    try:
        return open(f"/proc/{pid}/stat").read().split(")")[-1].split()[0] != "Z"
    except OSError:
        return False


def _wait_pids(prefix, n, timeout=120):
    import time
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(os.path.exists(f"{prefix}.{r}") and open(f"{prefix}.{r}").read().strip() for r in range(n)):
            return [int(open(f"{prefix}.{r}").read()) for r in range(n)]
        time.sleep(0.2)
    raise AssertionError("the ranks never started")


def test_launcher_takes_its_ranks_down_when_it_is_terminated(tmp_path):
    """a caller's timeout kills only the parent - the ranks (in their own sessions) must not stay behind in a collective."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    prefix = str(tmp_path / "pid")
    env.update(MC_BENCH_LAUNCH_PROBE="1", MC_BENCH_PROBE_PIDFILE=prefix, MC_BENCH_PROBE_SLEEP="600")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        pids = _wait_pids(prefix, 3)
        assert all(_alive(q) for q in pids)
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=60)
        assert p.returncode == 128 + signal.SIGTERM
        t0 = time.time()
        while any(_alive(q) for q in pids) and time.time() - t0 < 20:
            time.sleep(0.2)
        assert not any(_alive(q) for q in pids), "rank processes survived their launcher"
    finally:
        if p.poll() is None:
            p.kill()


def test_a_second_signal_during_the_grace_period_does_not_abort_the_clean_up(tmp_path):
    """ranks that ignore SIGTERM keep the launcher in its 5 s grace period; a repeated TERM then must not leave them behind."""
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    prefix = str(tmp_path / "pid")
    env.update(MC_BENCH_LAUNCH_PROBE="1", MC_BENCH_PROBE_PIDFILE=prefix, MC_BENCH_PROBE_SLEEP="600", MC_BENCH_PROBE_IGNORE_TERM="1")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        pids = _wait_pids(prefix, 2)
        time.sleep(0.5)                                   # the ranks have installed SIG_IGN by now
        p.send_signal(signal.SIGTERM)
        time.sleep(1.5)                                   # inside the grace period
        assert p.poll() is None
        p.send_signal(signal.SIGTERM)
        p.send_signal(signal.SIGINT)
        p.wait(timeout=60)
        assert p.returncode == 128 + signal.SIGTERM
        assert not any(_alive(q) for q in pids), "rank processes survived their launcher"
    finally:
        if p.poll() is None:
            p.kill()
        for q in (pids if "pids" in dir() else []):
            if _alive(q):
                os.kill(q, signal.SIGKILL)


def test_launcher_deadline_tears_a_hung_job_down(tmp_path):
    """rank 0 may exit 0 while another rank hangs: the wall-clock deadline ends the job with a non-zero status."""
    prefix = str(tmp_path / "pid")
    r = _run(["--gpus", "2"], {"MC_BENCH_LAUNCH_PROBE": "1", "MC_BENCH_PROBE_PIDFILE": prefix, "MC_BENCH_PROBE_SLEEP": "600",
                               "MC_BENCH_LAUNCH_TIMEOUT": "25"})
    assert r.returncode == 124, (r.returncode, r.stderr[-500:])
    pids = [int(open(f"{prefix}.{k}").read()) for k in range(2)]
    assert not any(_alive(q) for q in pids)


def test_ranks_get_disjoint_core_blocks():
    import bench
    if not hasattr(os, "sched_getaffinity") or len(os.sched_getaffinity(0)) < 4:
        import pytest
        pytest.skip("needs >= 4 cores")
    seen = []
    for r in range(4):
        mine = bench.pin_rank_cpus({"LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": "4"}, apply=False)
        assert mine and not (set(mine) & set(sum(seen, [])))
        seen.append(mine)
    assert bench.pin_rank_cpus({"LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "4", "MC_BENCH_PIN": "0"}, apply=False) is None
    assert bench.pin_rank_cpus({}, apply=False) is None                       # a single process keeps every core
    # and the real thing: the ranks of a bare launch report the block they were pinned to before importing torch
    r = _run(["--gpus", "2"], {"MC_BENCH_LAUNCH_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["cpus"] == sorted(os.sched_getaffinity(0))[:len(os.sched_getaffinity(0)) // 2]
