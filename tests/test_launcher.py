"""CPU: the rank launcher + watchdog behind `python bench.py --gpus N` (oldoceananigans.jl_amd/launcher.py; the reference's
benchmark/distributed_nonhydrostatic_model.jl:17-57 starts its ranks with mpiexec). Stub ranks, no GPU: the environment every rank
gets, rank 0's output coming back, and the three ways a job is ended with a non-zero status -- a rank that fails, a rank that leaves
early while its peers wait, a job that stalls -- each inside its limit and with no process left behind."""
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launcher():
    # by path, like bench.py's parent process does: the launcher must not need the package (and so never the HIP extension)
    spec = importlib.util.spec_from_file_location("ocn_launcher", os.path.join(ROOT, "oldoceananigans.jl_amd", "launcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return True
    try:                                  # a zombie still answers signal 0
        with open(f"/proc/{pid}/stat") as f:
            return f.read().split(") ")[1][0] != "Z"
    except OSError:
        return False


STUB = r"""
import json, os, sys, time
mode, piddir = sys.argv[1], sys.argv[2]
rank = int(os.environ["RANK"])
open(os.path.join(piddir, f"pid{rank}"), "w").write(str(os.getpid()))
env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OCN_LAUNCH_HEARTBEAT")}
if mode == "ok":
    print(json.dumps(env)); sys.exit(0)
if mode == "fail" and rank == 1:
    n, t0 = int(os.environ["WORLD_SIZE"]), time.time()      # fail once every peer is up (a loaded machine starts them slowly): the test counts their pids
    while time.time() - t0 < 20 and not all(os.path.exists(os.path.join(piddir, f"pid{r}")) for r in range(n)):
        time.sleep(0.05)
    print("rank 1 about to fail", file=sys.stderr); sys.exit(3)
if mode == "leave" and rank == 0:
    sys.exit(0)
if mode == "beat":
    for _ in range(8):
        time.sleep(0.5)
        open(env["OCN_LAUNCH_HEARTBEAT"], "a").write("tick\n")
    print("done"); sys.exit(0)
time.sleep(600)          # "fail" peers, "leave" peers and "stall": wait for something that never comes
"""


def _run(tmp_path, mode, n, **limits):
    L = _launcher()
    stub = tmp_path / "stub.py"
    stub.write_text(STUB)
    import io
    rep = io.StringIO()
    t0 = time.time()
    res = L.launch_ranks([str(stub), mode, str(tmp_path)], n, log_dir=str(tmp_path / "logs"), report=rep, **limits)
    pids = [int((tmp_path / f"pid{r}").read_text()) for r in range(n) if (tmp_path / f"pid{r}").exists()]
    return res, rep.getvalue(), time.time() - t0, pids


def test_every_rank_gets_its_environment_and_rank_output_comes_back(tmp_path):
    res, rep, _, _ = _run(tmp_path, "ok", 3)
    assert res.returncode == 0 and res.rank_codes == [0, 0, 0] and rep == ""
    envs = [json.loads(s) for s in res.stdout]
    assert [e["RANK"] for e in envs] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["LOCAL_WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) == res.port
    assert len({e["OCN_LAUNCH_HEARTBEAT"] for e in envs}) == 3


def test_a_failing_rank_ends_the_job_non_zero_and_nothing_is_left_running(tmp_path):
    res, rep, wall, pids = _run(tmp_path, "fail", 3, time_limit_s=120, stall_limit_s=120)
    assert res.returncode == 3 and "rank 1 exited with status 3" in res.reason
    assert wall < 30, wall                                  # not the peers' 600 s, not the limits
    assert "rank 1 about to fail" in rep and "FAILED" in rep
    assert len(pids) == 3 and not any(_alive(p) for p in pids)


def test_a_rank_that_leaves_early_is_caught_by_the_straggler_limit(tmp_path):
    res, rep, wall, pids = _run(tmp_path, "leave", 2, time_limit_s=120, stall_limit_s=120, straggler_limit_s=2.0)
    assert res.returncode == 124 and "after a peer had exited" in res.reason
    assert wall < 30 and not any(_alive(p) for p in pids)


def test_a_stalled_job_is_killed_at_the_stall_limit_and_heartbeats_keep_a_slow_one_alive(tmp_path):
    res, rep, wall, pids = _run(tmp_path, "stall", 2, time_limit_s=120, stall_limit_s=2.0)
    assert res.returncode == 124 and "no rank has shown progress" in res.reason
    assert wall < 30 and not any(_alive(p) for p in pids)
    (tmp_path / "b").mkdir()
    res, rep, wall, _ = _run(tmp_path / "b", "beat", 2, time_limit_s=120, stall_limit_s=2.0)    # 4 s of work, a beat every 0.5 s
    assert res.returncode == 0 and res.stdout[0].strip() == "done", (res, rep)


def test_the_time_limit_ends_a_job_that_keeps_beating(tmp_path):
    res, rep, wall, pids = _run(tmp_path, "beat", 2, time_limit_s=1.5, stall_limit_s=60)
    assert res.returncode == 124 and "time limit" in res.reason and not any(_alive(p) for p in pids)


def _bench(args, env_extra, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_n_starts_n_ranks_itself_and_prints_rank_0s_line():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts two rank processes (OCN_BENCH_STUB_RANKS=1 replaces the GPU
    work of a rank by a stub line -- this container has no GPU; tests/test_gpu_launcher.py runs the real thing on the card)"""
    res = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"OCN_BENCH_STUB_RANKS": "1"})
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["launcher"]["ranks_started"] == 2 and out["config"]["launcher"]["self_launched"] is True
    assert out["config"]["communicator"]["world"] == 2


def test_bench_refuses_a_world_size_that_contradicts_gpus_also_for_one_rank():
    res = _bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in res.stderr
    res = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "--gpus 1 but WORLD_SIZE=2" in res.stderr


def test_bench_exits_non_zero_with_the_rank_logs_when_a_rank_dies():
    res = _bench(["--gpus", "2"], {"OCN_BENCH_STUB_RANKS": "1", "OCN_BENCH_FAIL_RANK": "1", "OCN_BENCH_FAIL_AT": "warmup"})
    assert res.returncode != 0 and res.stdout.strip() == ""
    assert "rank 1 exited with status" in res.stderr and "OCN_BENCH_FAIL_RANK" in res.stderr
