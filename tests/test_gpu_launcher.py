"""GPU: `python bench.py --gpus N` starts its ranks ITSELF (oldoceananigans.jl_amd/launcher.py) -- on the one-card box as a rehearsal: two
REAL rank processes share card 0 and run the product's partitioned time-step (library orchestration) over the host-staged gloo transport
(tests/host_staged.py; RCCL refuses two ranks on one device). What is checked is the launch path the first multi-GPU lease will take:
children spawned by a parent that never touches the GPU, the environment, rank 0's one JSON line with n_gpus = N and the communicator's own
report, and the watchdog: a rank that dies takes the job down with a non-zero status inside the limit, nothing left running."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)
    return res, time.time() - t0


def test_self_launched_two_rank_rehearsal_prints_one_line_with_n_gpus_2():
    res, _ = _bench(["--gpus", "2", "--size", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {"OCN_REHEARSE_ON_ONE_GPU": "1"})
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and "REHEARSAL" in out["data"]
    cfg = out["config"]
    assert cfg["launcher"] == {"self_launched": True, "ranks_started": 2, "by": "oldoceananigans.jl_amd.launcher"}
    assert cfg["communicator"]["world"] == 2 and cfg["communicator"]["comm_ranks"] == 2 and cfg["communicator"]["rank"] == 0
    assert cfg["max_abs_divergence_after_run"] < 5e-8
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["ms_per_step_median"] > 0
    assert "64x32x32" in cfg["workload"]                     # weak scaling: 32^3 per rank, global (32 * 2) x 32 x 32


def test_a_rank_that_dies_takes_the_job_down_non_zero_within_the_limit():
    """rank 1 exits after the warm-up; rank 0 is left waiting in a receive. The launcher must notice, stop rank 0 and exit non-zero"""
    res, wall = _bench(["--gpus", "2", "--size", "32", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       {"OCN_REHEARSE_ON_ONE_GPU": "1", "OCN_BENCH_FAIL_RANK": "1", "OCN_BENCH_FAIL_AT": "warmup",
                        "OCN_LAUNCH_TIME_LIMIT_S": "300", "OCN_LAUNCH_STALL_LIMIT_S": "120"})
    assert res.returncode != 0 and res.stdout.strip() == "", (res.returncode, res.stdout)
    assert "rank 1 exited with status 17" in res.stderr and "every rank was stopped" in res.stderr
    assert wall < 200, wall


def test_one_rank_self_loop_line_reports_what_rccl_saw():
    """OCN_SELF_LOOP=1: the N > 1 code path on one rank over the library's own RCCL communicator -- `communicator` comes from
    ncclCommCount / ncclCommUserRank"""
    res, _ = _bench(["--gpus", "1", "--size", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {"OCN_SELF_LOOP": "1"})
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    com = out["config"]["communicator"]
    assert out["n_gpus"] == 1 and com["transport"] == "rccl" and com["comm_ranks"] == 1 and com["comm_rank"] == 0 and com["self_loop"] is True
    assert out["config"]["launcher"]["self_launched"] is False


def test_gpus_flag_and_world_size_must_agree_also_for_one_rank():
    res, _ = _bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert res.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in res.stderr
