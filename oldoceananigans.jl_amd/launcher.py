"""One process per GPU on ONE node, started and watched by a parent that never touches the GPU.

What `mpiexec -np R julia distributed_nonhydrostatic_model_mpi.jl ...` does for the reference's benchmark launcher
(benchmark/distributed_nonhydrostatic_model.jl:17-57): start R rank processes, wait for all of them, fail when one fails. Here:

* every rank is a CHILD process started before anything in the parent has initialised the GPU (the parent imports only the standard
  library -- it never loads libocn_mi355x.so or torch; nothing is ever exec'ed from a process that touched the card);
* each child gets RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT (what
  `python -m torch.distributed.run` exports), plus OCN_LAUNCH_HEARTBEAT = a file the rank touches between its phases;
* a watchdog ends the whole job -- every child process group this launcher started, by its exact id -- and returns NON-ZERO with
  the rank logs when (i) a rank exits with a non-zero status, (ii) the job exceeds `time_limit_s`, (iii) no rank has shown progress
  (heartbeat or log growth) for `stall_limit_s`, or (iv) some rank has exited cleanly while others are still running after
  `straggler_limit_s` (a peer that left early leaves the others waiting in ncclRecv for ever).

Standard library only: importable without the HIP extension, testable without a GPU (tests/test_launcher.py)."""
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

__all__ = ["launch_ranks", "heartbeat", "free_port", "LaunchResult"]


class LaunchResult:
    """what `launch_ranks` returns: `returncode` (0 only when every rank exited 0 inside the limits), `reason` (text), `stdout[r]` /
    `stderr[r]` (each rank's captured output), `elapsed_s`, `port`"""

    def __init__(self, returncode, reason, stdout, stderr, elapsed_s, port, rank_codes):
        self.returncode, self.reason, self.stdout, self.stderr = returncode, reason, stdout, stderr
        self.elapsed_s, self.port, self.rank_codes = elapsed_s, port, rank_codes

    def __repr__(self):
        return f"LaunchResult(returncode={self.returncode}, reason={self.reason!r}, rank_codes={self.rank_codes})"


def free_port():
    """a TCP port that is free now on 127.0.0.1 together with the 16 above it (the unique-id exchange listens on MASTER_PORT + 1 .. + 16)"""
    for _ in range(64):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        if port > 65535 - 20:
            continue
        ok = True
        for k in range(1, 17):
            t = socket.socket()
            try:
                t.bind(("127.0.0.1", port + k))
            except OSError:
                ok = False
            finally:
                t.close()
            if not ok:
                break
        if ok:
            return port
    return port


def heartbeat(phase=""):
    """called by a rank between its phases (library loaded, communicator up, model built, warm-up done, ...): tells the launcher's
    watchdog that this rank is alive. A no-op outside `launch_ranks`."""
    path = os.environ.get("OCN_LAUNCH_HEARTBEAT")
    if not path:
        return
    try:
        with open(path, "a") as f:
            f.write(f"{time.time():.3f} {phase}\n")
    except OSError:
        pass


def _kill_group(proc, sig):
    """signal the process group of ONE child this launcher started (start_new_session made the child its leader)"""
    try:
        os.killpg(proc.pid, sig)
    except (ProcessLookupError, PermissionError):
        pass


def _stop_all(procs, grace_s=3.0):
    for p in procs:
        if p.poll() is None:
            _kill_group(p, signal.SIGTERM)
    deadline = time.time() + grace_s
    while time.time() < deadline and any(p.poll() is None for p in procs):
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:
            _kill_group(p, signal.SIGKILL)
    for p in procs:
        try:
            p.wait(timeout=10.0)
        except subprocess.TimeoutExpired:
            pass
        _kill_group(p, signal.SIGKILL)        # grandchildren that outlived the rank's main process


def _tail(path, nbytes=4000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - nbytes))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def _progress_stamp(paths):
    stamp = []
    for p in paths:
        try:
            st = os.stat(p)
            stamp.append((st.st_mtime_ns, st.st_size))
        except OSError:
            stamp.append(None)
    return stamp


def launch_ranks(argv, nranks, time_limit_s=900.0, stall_limit_s=240.0, straggler_limit_s=60.0, env=None, log_dir=None,
                 report=None, poll_s=0.1):
    """start `nranks` copies of `[sys.executable] + argv`, one per GPU, and watch them (module docstring). Blocks until the job has
    ended one way or the other; never raises for a failing rank -- the result says what happened. `report`: a text stream (default
    stderr) that receives the reason and the rank logs of a failed job."""
    if nranks < 1:
        raise ValueError("nranks must be >= 1")
    report = report if report is not None else sys.stderr
    own_dir = log_dir is None
    log_dir = log_dir or tempfile.mkdtemp(prefix="ocn_launch_")
    os.makedirs(log_dir, exist_ok=True)
    port = free_port()
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC for RCCL on this pool
    base.update(WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                OCN_LAUNCHED_BY="oldoceananigans.jl_amd.launcher")
    procs, outs, errs, beats, files = [], [], [], [], []
    t0 = time.time()
    try:
        for r in range(nranks):
            out, err, hb = (os.path.join(log_dir, f"rank{r}.{s}") for s in ("out", "err", "heartbeat"))
            open(hb, "w").close()
            fo, fe = open(out, "wb"), open(err, "wb")
            files += [fo, fe]
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0", OCN_LAUNCH_HEARTBEAT=hb)
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=fo, stderr=fe, stdin=subprocess.DEVNULL,
                                          start_new_session=True))
            outs.append(out); errs.append(err); beats.append(hb)

        reason, code = "", 0
        last_stamp, last_change = _progress_stamp(outs + errs + beats), time.time()
        first_clean_exit = None
        while True:
            codes = [p.poll() for p in procs]
            now = time.time()
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                reason, code = f"rank {r} exited with status {c}", (c if 0 < c < 256 else 1)
                break
            if all(c == 0 for c in codes):
                break
            if any(c == 0 for c in codes):
                first_clean_exit = first_clean_exit or now
                if now - first_clean_exit > straggler_limit_s:
                    left = [r for r, c in enumerate(codes) if c is None]
                    reason, code = (f"ranks {left} still running {straggler_limit_s:.0f} s after a peer had exited "
                                    "(a rank that leaves early leaves its neighbours waiting)"), 124
                    break
            if now - t0 > time_limit_s:
                reason, code = f"time limit of {time_limit_s:.0f} s exceeded", 124
                break
            stamp = _progress_stamp(outs + errs + beats)
            if stamp != last_stamp:
                last_stamp, last_change = stamp, now
            elif now - last_change > stall_limit_s:
                reason, code = f"no rank has shown progress (heartbeat or output) for {stall_limit_s:.0f} s", 124
                break
            time.sleep(poll_s)
    finally:
        _stop_all(procs)
        for f in files:
            f.close()

    elapsed = time.time() - t0
    stdout = [open(p, "rb").read().decode("utf-8", "replace") for p in outs]
    stderr = [open(p, "rb").read().decode("utf-8", "replace") for p in errs]
    rank_codes = [p.returncode for p in procs]
    if code:
        print(f"[launcher] FAILED after {elapsed:.1f} s: {reason}; every rank was stopped. Rank exit codes: {rank_codes}", file=report)
        for r in range(nranks):
            last = _tail(beats[r], 200).strip().splitlines()
            print(f"[launcher] ---- rank {r} (exit {rank_codes[r]}; last heartbeat: {last[-1] if last else 'none'}) stderr tail ----\n"
                  f"{_tail(errs[r])}\n[launcher] ---- rank {r} stdout tail ----\n{_tail(outs[r], 1000)}", file=report)
    if own_dir and not code:
        for p in outs + errs + beats:
            try:
                os.remove(p)
            except OSError:
                pass
        try:
            os.rmdir(log_dir)
        except OSError:
            pass
    return LaunchResult(code, reason, stdout, stderr, elapsed, port, rank_codes)
