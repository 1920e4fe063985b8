#!/usr/bin/env python3
"""bench.py -- headline benchmark: ms/time-step and cell-updates/s of the NonhydrostaticModel RK3 time-step
(WENO(order=5), tracers (T, S), FFT pressure solve) -- BASELINE.json's metric, measured like the reference measures it
(benchmark/benchmarkable_nonhydrostatic_model.jl:23-27: device-synchronised `time_step!(model, Δt)` after warm-up).

    python bench.py --gpus N --steps K --warmup W [--size 256] [--no-cpu-baseline]

N = 1: 256^3 triply periodic on one MI355X (BASELINE.json configs[1]).
N > 1: one rank process per GPU; x-slab decomposition, WEAK scaling (256^3 per GPU, i.e. global (256 N) x 256 x 256). The library
       owns the RCCL communicator (ocn_dist_create; the ncclUniqueId travels over a one-shot TCP exchange on MASTER_PORT + 1) and
       runs the partitioned step itself: halo send / recv on its communication stream, one small all-gather per pressure solve. No
       torch in the process. Two ways to start it, same line printed:
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
         python bench.py --gpus N ...          (WORLD_SIZE unset: bench.py starts the N ranks ITSELF -- oldoceananigans.jl_amd/launcher.py:
                                                children spawned before anything touches the GPU, a watchdog that stops every rank and
                                                exits non-zero with the rank logs when one dies, leaves early or stalls)
       --gpus N with a WORLD_SIZE that says otherwise is refused.

Prints ONE JSON line on rank 0 (contract in the task statement): value = whole-job cell-updates/s with all inputs
resident in HBM; "roofline" = the dominant kernel (fused WENO tendency evaluation) from HIP events recorded on the
launch stream inside the timed region; "cpu_baseline" = the CPU oracle (a port, not the Julia reference) on the host
cores over a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL on this pool (already exported on the boxes)
import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
TENDENCY_BYTES_PER_CELL = 80.0    # (5 fields read + 5 tendencies written) x 8 B
# When the RK3 substep of stages 2 and 3 is fused into the tendency launch that precedes it, that launch also reads the 5
# previous tendencies and writes the 5 updated fields: 80 + 80 B/cell (SURVEY.md 8(d) prices the separate substep at 160:
# the fused form saves re-reading U and Gn). Two of the three launches per time-step are of this kind.
FUSED_SUBSTEP_EXTRA_BYTES_PER_CELL = 80.0
V100_PUBLISHED_CELL_UPDATES = 256 ** 3 / 56.444e-3   # 256^3 F64 WENO 56.444 ms on a V100 (v0.58.8)


PROFILE_TAG = {1: "r01", 2: "r03"}      # committed rocprofv3 summaries of the all-fields kernel (round 1) and the role kernel
PROFILE_TAG_ARITHMETIC = {1: "r03_contract"}    # the opt-in contracted arithmetic mode of the role kernel has its own instruction counts


def profile_path(tendency_impl, arithmetic, what):
    tag = PROFILE_TAG_ARITHMETIC.get(arithmetic) if (arithmetic and tendency_impl == 2) else PROFILE_TAG.get(tendency_impl, "none")
    path = os.path.join(ROOT, "profiles", f"{tag}_tendency_{what}.json")
    if not os.path.exists(path) and tendency_impl == 2 and not arithmetic:      # the role kernel's round-2 passes
        path = os.path.join(ROOT, "profiles", f"r02_tendency_{what}.json")
    return path


def profile_launch_ms(tag_path):
    """the average launch duration of the tendency kernel in the kernel-trace summary that was collected with the PMC passes"""
    stats = os.path.join(os.path.dirname(tag_path), os.path.basename(tag_path).split("_tendency_")[0] + "_kernel_stats_256cubed.csv")
    if not os.path.exists(stats):
        return None
    import csv
    tot = n = 0.0
    try:
        for row in csv.DictReader(ln for ln in open(stats) if not ln.startswith("#")):
            if "tendency_kernel" in row.get("Name", ""):
                tot += float(row["TotalDurationNs"])
                n += float(row["Calls"])
    except (OSError, ValueError, KeyError):
        return None
    return 1e-6 * tot / n if n else None


def measured_traffic(tendency_impl, N, arithmetic=0):
    """HBM bytes per tendency launch from the COMMITTED rocprofv3 PMC passes (profiles/, FETCH_SIZE and WRITE_SIZE collected
    in separate runs and calibrated as MI355X_MICROARCH.md prescribes) -- not measured in this run; None when no profile matches"""
    path = profile_path(tendency_impl, 0, "traffic")
    if N != 256 or not os.path.exists(path):
        return None, None
    try:
        return float(json.load(open(path))["hbm_bytes_per_launch"]), path
    except (OSError, ValueError, KeyError):
        return None, None


# Measured issue cost of the instruction classes of the WENO flux on MI355X (tools/valu_rates.hip, profiles/r02_valu_rates.txt):
# ns per wave64 instruction per SIMD with 4 waves resident, every SIMD of the chip busy -- i.e. at the clock the chip HOLDS under
# that load (1.8-2.3 GHz by class), not at the 2.4 GHz of the data sheet. FP64 mul / fma / add issue every 4 cycles.
VALU_NS = {"SQ_INSTS_VALU_FMA_F64": 2.27, "SQ_INSTS_VALU_MUL_F64": 2.32, "SQ_INSTS_VALU_ADD_F64": 1.99, "SQ_INSTS_VALU_CVT": 1.75,
           "SQ_INSTS_VALU_FMA_F32": 1.19, "SQ_INSTS_VALU_INT32": 1.11, "SQ_INSTS_VALU_TRANS_F32": 3.44, "SQ_INSTS_VALU_TRANS_F64": 6.94}
VALU_NS_OTHER = 1.5               # selects, 64-bit moves, compares (1.1-1.9 measured)
# the same classes priced from the DATA SHEET: FP64 vector 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 FMA lanes x 2.4 GHz, i.e. one wave64
# FP64 mul / fma / add per SIMD every 4 cycles of 2.4 GHz = 1.667 ns (the FP64 transcendental at quarter rate: 16 cycles); the data sheet
# has no per-class figure for the rest, which keep their measured cost
FP64_DATASHEET_NS = 4.0 / 2.4
VALU_NS_DATASHEET = dict(VALU_NS, SQ_INSTS_VALU_FMA_F64=FP64_DATASHEET_NS, SQ_INSTS_VALU_MUL_F64=FP64_DATASHEET_NS,
                         SQ_INSTS_VALU_ADD_F64=FP64_DATASHEET_NS, SQ_INSTS_VALU_TRANS_F64=4 * FP64_DATASHEET_NS)


def measured_valu(tendency_impl, N, t_launch, arithmetic=0):
    """Issue-side roofline of the same kernel (its binding roof): the VALU wave-instructions per launch by class
    (COMMITTED SQ_INSTS_VALU* passes -- not collected in this run) priced (i) with the measured per-class issue cost -> the time the
    instruction stream needs on 1024 fully busy SIMDs at the clock the chip holds, frac = that floor / the LIVE launch time; and (ii)
    with the data-sheet FP64 rate (4 cycles at 2.4 GHz) -> frac_datasheet"""
    path = profile_path(tendency_impl, arithmetic, "valu")
    if N != 256 or not os.path.exists(path) or not t_launch:
        return None
    try:
        c = json.load(open(path))["counters_per_launch"]
        n = float(c["SQ_INSTS_VALU"])
    except (OSError, ValueError, KeyError):
        return None
    classed = {k: float(c[k]) for k in VALU_NS if k in c}
    floor_ds = None
    if classed:
        rest = n - sum(classed.values())
        floor_s = (sum(v * VALU_NS[k] for k, v in classed.items()) + rest * VALU_NS_OTHER) * 1e-9 / 1024
        floor_ds = (sum(v * VALU_NS_DATASHEET[k] for k, v in classed.items()) + rest * VALU_NS_OTHER) * 1e-9 / 1024
        weighting = "measured FP64 / FP32 / conversion / transcendental / integer mix x measured issue cost per class"
    else:                          # round-1 profile: total only -- every instruction at the FP64 rate
        floor_s = n * 2.3e-9 / 1024
        weighting = "all instructions priced as FP64 (no class counters in this profile)"
    fp64 = sum(classed.get(k, 0.0) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
    return {"wave_instructions_per_launch": n, "per_cell": n * 64 / float(N) ** 3, "fp64_per_cell": fp64 * 64 / float(N) ** 3 if classed else None,
            "issue_floor_ms": 1e3 * floor_s,
            "achieved": n / t_launch / 1e9, "peak": n / floor_s / 1e9, "unit": "G wave-instr/s", "frac": floor_s / t_launch,
            "issue_floor_datasheet_ms": 1e3 * floor_ds if floor_ds else None, "frac_datasheet": floor_ds / t_launch if floor_ds else None,
            "frac_datasheet_note": "FP64 classes at the data-sheet rate (one wave64 instruction per SIMD per 4 cycles of 2.4 GHz = 78.6 TFLOP/s), "
                                   "the other classes at their measured cost; `frac` prices FP64 at the clock the chip holds under this load (1.8-2.1 GHz)",
            "weighting": weighting, "from": "committed profile (instruction counts); launch time live",
            "profile_launch_ms": profile_launch_ms(path),
            "source": os.path.relpath(path, ROOT) + " + profiles/r02_valu_rates.txt"}


def initial_state(ocn, model, seed=1234):
    from helpers import smooth_state
    g = model.grid
    flds = model.fields()
    nodes = {n: g.nodes(f.loc) for n, f in flds.items()}
    return smooth_state(nodes, seed)


def cpu_baseline(size, dt, budget_s=20.0, one_thread_budget_s=12.0):
    """time the CPU oracle (oracle/, a port of the reference algorithm -- NOT the Julia reference, which cannot run
    here) on the same workload, all host cores, bounded to ~budget_s of CPU work"""
    from helpers import smooth_state
    from oracle import oracle as O
    cores = O.available_cpus()
    O.lib().oro_set_num_threads(cores)
    # estimate the rate on a 64^3 probe, then pick the largest power-of-two sample <= size that fits the budget
    n = min(64, size)
    g = O.Grid((n, n, n))
    m = O.Model(g, 2)
    m.time_step(dt)
    t0 = time.perf_counter()
    m.time_step(dt)
    rate = n ** 3 / (time.perf_counter() - t0)
    sample = size
    while sample > 64 and 3 * sample ** 3 / rate > budget_s:
        sample //= 2
    g = O.Grid((sample, sample, sample))
    m = O.Model(g, 2)
    names = {"u": "u", "v": "v", "w": "w", "T": "c0", "S": "c1"}
    locs = {"u": (1, 0, 0), "v": (0, 1, 0), "w": (0, 0, 1), "T": (0, 0, 0), "S": (0, 0, 0)}
    d = 1.0 / sample
    nodes = {}
    for k, loc in locs.items():
        ax = []
        for dim in range(3):
            shape = [1, 1, 1]
            shape[dim] = sample
            ax.append((d * (np.arange(sample) + (0.0 if loc[dim] else 0.5))).reshape(shape))
        nodes[k] = ax
    vals = smooth_state(nodes, 1234)
    m.set(**{names[k]: v for k, v in vals.items()})
    m.time_step(dt)                        # warm-up (also the iteration-0 update_state!)
    nsteps = 2
    t0 = time.perf_counter()
    for _ in range(nsteps):
        m.time_step(dt)
    el = time.perf_counter() - t0
    out = {"value": sample ** 3 * nsteps / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
           "sample": f"{nsteps} RK3 steps of the same model at {sample}^3 (oracle/ C restatement, OpenMP)",
           "ms_per_step": 1e3 * el / nsteps}
    # ONE thread, beside the reference's published single-core numbers (docs/src/appendix/benchmarks.md:117-120: 64^3 288.2 ms,
    # 128^3 2.326 s, 256^3 19.561 s per time-step on a Xeon Silver 4216 -- an older version, context only): the largest cube that
    # fits ~one_thread_budget_s at the rate measured on a 32^3 probe
    O.lib().oro_set_num_threads(1)
    g = O.Grid((32, 32, 32))
    m1 = O.Model(g, 2)
    m1.time_step(dt)
    t0 = time.perf_counter()
    m1.time_step(dt)
    rate1 = 32 ** 3 / (time.perf_counter() - t0)
    n1 = 128 if 2 * 128 ** 3 / rate1 <= one_thread_budget_s else 64
    n1 = min(n1, size)
    g = O.Grid((n1, n1, n1))
    m1 = O.Model(g, 2)
    d1 = 1.0 / n1
    nodes = {}
    for k, loc in locs.items():
        ax = []
        for dim in range(3):
            shape = [1, 1, 1]
            shape[dim] = n1
            ax.append((d1 * (np.arange(n1) + (0.0 if loc[dim] else 0.5))).reshape(shape))
        nodes[k] = ax
    vals = smooth_state(nodes, 1234)
    m1.set(**{names[k]: v for k, v in vals.items()})
    dt1 = 0.1 * d1 / 0.6
    m1.time_step(dt1)
    t0 = time.perf_counter()
    m1.time_step(dt1)
    el1 = time.perf_counter() - t0
    O.lib().oro_set_num_threads(cores)
    published = {64: 288.154, 128: 2326.0, 256: 19561.0}
    out["one_thread"] = {"value": n1 ** 3 / el1, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                         "sample": f"1 RK3 step of the same model at {n1}^3 after one warm-up step (oracle/ C restatement, 1 thread)",
                         "ms_per_step": 1e3 * el1,
                         "reference_published_ms_per_step_same_size": published.get(n1),
                         "reference_published_note": "Julia reference, 1 core of a Xeon Silver 4216, Oceananigans v0.58.8 "
                                                     "(docs/src/appendix/benchmarks.md:117-120) -- other hardware and an older version: context only"}
    return out


def workload_physics(ocn, workload):
    """keyword arguments of the model for the non-default workloads (see --workload)"""
    F = ocn.FieldBoundaryConditions
    if workload == "ppb_physics":
        return dict(closure=ocn.ScalarDiffusivity(ν=1e-4, κ=1e-4), buoyancy=ocn.SeawaterBuoyancy(),
                    boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-4)),
                                         "T": F(top=ocn.FluxBoundaryCondition(1e-4), bottom=ocn.GradientBoundaryCondition(0.01))})
    if workload == "ppb_amd":
        return dict(closure=ocn.AnisotropicMinimumDissipation(),
                    buoyancy=ocn.SeawaterBuoyancy(ocn.LinearEquationOfState(thermal_expansion=2e-4, haline_contraction=8e-4)),
                    boundary_conditions={"u": F(top=ocn.FluxBoundaryCondition(-1e-4)),
                                         "T": F(top=ocn.FluxBoundaryCondition(5e-5), bottom=ocn.GradientBoundaryCondition(0.01)),
                                         "S": F(top=ocn.FluxBoundaryCondition(ocn.LinearFieldFlux(b=-1e-3 / 3600.0), field_dependencies="S"))})
    return {}


def load_launcher():
    """oldoceananigans.jl_amd/launcher.py loaded BY PATH: standard library only, so the parent of a self-launched run imports neither
    the package nor the HIP extension nor torch -- it never touches the GPU"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ocn_launcher", os.path.join(ROOT, "oldoceananigans.jl_amd", "launcher.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def launch_self(args):
    """`python bench.py --gpus N` with WORLD_SIZE unset: start N rank processes of this same command line (the reference's launcher
    runs `mpiexec -np R julia ...`, benchmark/distributed_nonhydrostatic_model.jl:43), wait under the watchdog, print rank 0's JSON
    line. Returns the exit status: 0 only when every rank finished and rank 0 printed its line."""
    L = load_launcher()
    limit = float(os.environ.get("OCN_LAUNCH_TIME_LIMIT_S", "900"))
    stall = float(os.environ.get("OCN_LAUNCH_STALL_LIMIT_S", "240"))
    straggler = float(os.environ.get("OCN_LAUNCH_STRAGGLER_LIMIT_S", "60"))
    env = dict(os.environ, OCN_BENCH_SELF_LAUNCHED="1")
    res = L.launch_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, time_limit_s=limit, stall_limit_s=stall,
                         straggler_limit_s=straggler, env=env)
    for r in range(1, args.gpus):              # the other ranks' diagnostics (rank 0's follow) -- stderr only
        if res.returncode == 0 and res.stderr[r].strip():
            print(f"[bench] rank {r} stderr:\n{res.stderr[r].rstrip()}", file=sys.stderr)
    if res.returncode:
        return res.returncode
    sys.stderr.write(res.stderr[0])
    line = None
    for ln in res.stdout[0].splitlines():
        try:
            if isinstance(json.loads(ln), dict):
                line = ln
        except ValueError:
            pass
    if line is None:
        print(f"[bench] every rank exited 0 but rank 0 printed no JSON line; its stdout was:\n{res.stdout[0][-2000:]}", file=sys.stderr)
        return 1
    print(line)
    return 0


def launcher_info(world):
    return {"self_launched": os.environ.get("OCN_BENCH_SELF_LAUNCHED") == "1", "ranks_started": world,
            "by": os.environ.get("OCN_LAUNCHED_BY") or ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "environment")}


def stub_rank(args, rank, world, phase):
    """OCN_BENCH_STUB_RANKS=1 -- tests/test_launcher.py on a machine WITHOUT a GPU: a rank goes through its phases (heartbeats, the
    OCN_BENCH_FAIL_* hook) and rank 0 prints a line of the right shape with no measurement in it. Never a result."""
    for name in ("communicator", "model", "warmup", "timed"):
        time.sleep(0.05)
        phase(name)
    if rank == 0:
        print(json.dumps({"metric": "cell_updates_per_s", "value": None, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f64", "data": "STUB: no GPU work was done (OCN_BENCH_STUB_RANKS=1, launcher test) -- not a measurement",
                          "config": {"workload": "none", "launcher": launcher_info(world),
                                     "communicator": {"world": world, "transport": "none (stub)"}}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=256, help="cells per side per GPU")
    ap.add_argument("--global-size", type=int, default=0,
                    help="fixed GLOBAL grid G^3 split into x-slabs over the ranks (strong scaling; 512 = BASELINE.json configs[3] at "
                         "--gpus 8). Default 0: weak scaling, --size^3 cells per GPU, global (size*N) x size x size")
    ap.add_argument("--tendency-impl", type=int, default=2,
                    help="2: one field per workgroup (default); 1: all-fields kernel of round 1; 0: per-field kernels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arithmetic", type=int, default=0,
                    help="0: the reference's IEEE operation sequence, bit-identical to the oracle (default, what every parity test runs); "
                         "1: the opt-in contracted mode of the WENO flux (FMA-contracted sub-stencil sums and weights, one normalisation, "
                         "reciprocal without the divide fix-up) -- inside north_star's 1e-12, not bit-identical; named in config.arithmetic")
    ap.add_argument("--workload", default="ppp", choices=["ppp", "ppb_stretched", "ppb_physics", "ppb_amd"],
                    help="ppp: BASELINE.json configs[1] (the metric's configuration, default); ppb_stretched: configs[2], "
                         "256x256x128 (Periodic, Periodic, Bounded) with tanh-stretched z (Fourier-tridiagonal solver), single GPU; "
                         "ppb_physics: the same grid with the SURVEY 8f.1 physics switched on -- ScalarDiffusivity, linear "
                         "SeawaterBuoyancy (hydrostatic pressure anomaly), surface Flux / bottom Gradient boundary conditions; "
                         "ppb_amd: the physics of BASELINE.json configs[4] (ocean_wind_mixing_and_convection) on that grid -- "
                         "AnisotropicMinimumDissipation, linear SeawaterBuoyancy, wind-stress / heat-flux / bottom-gradient conditions "
                         "and the example's S-dependent evaporation flux J = -rate S")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_self(args))                  # parent: starts the ranks, never touches the GPU itself
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with `python bench.py --gpus {args.gpus}` (WORLD_SIZE "
                         f"unset) or with torch.distributed.run --nproc-per-node {args.gpus}")
    launcher = load_launcher()
    launcher.heartbeat("rank process started")
    fail_at = os.environ.get("OCN_BENCH_FAIL_AT") if os.environ.get("OCN_BENCH_FAIL_RANK") == str(rank) else None

    def phase(name):
        """heartbeat for the launcher's watchdog; OCN_BENCH_FAIL_RANK / OCN_BENCH_FAIL_AT (tests of the kill path): this rank dies here"""
        launcher.heartbeat(name)
        if fail_at == name:
            print(f"[bench] rank {rank}: OCN_BENCH_FAIL_RANK asked this rank to die at phase '{name}'", file=sys.stderr, flush=True)
            os._exit(17)

    if os.environ.get("OCN_BENCH_STUB_RANKS") == "1":
        return stub_rank(args, rank, world, phase)

    # OCN_REHEARSE_ON_ONE_GPU=1 (test harness): all ranks share card 0 and run the PRODUCT's partitioned step (the library's
    # orchestration) over a host-staged gloo transport plugged into ocn_dist_create_transport (tests/host_staged.py) -- RCCL refuses two
    # ranks on one device. Exercises a multi-process run on a one-GPU box; not a measurement
    rehearsal = os.environ.get("OCN_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        import torch  # noqa: F401  -- must precede the first load of libocn_mi355x.so (tests/conftest.py explains)
    import oldoceananigans_jl_amd as ocn
    N = args.global_size if (args.global_size and world == 1) else args.size
    distributed = world > 1 or os.environ.get("OCN_FORCE_DISTRIBUTED") == "1" or os.environ.get("OCN_SELF_LOOP") == "1"
    if distributed:
        from oldoceananigans_jl_amd import distributed as dist
        # OCN_SELF_LOOP=1 (one rank): the rank is its own west / east neighbour -- the complete N > 1 code path, RCCL send / recv to
        # itself included: the LOCAL cost of the partitioned path, measurable on a one-GPU box
        self_loop = os.environ.get("OCN_SELF_LOOP") == "1" and world == 1
        if rehearsal:
            import torch.distributed as td
            from host_staged import HostStagedCollectives
            from oldoceananigans_jl_amd import _lib
            td.init_process_group("gloo")
            ctx = dist.Distributed.transport(ocn.GPU(0), HostStagedCollectives(torch, td, _lib.lib(), rank, world), world, rank)
        else:
            # the product path: the library owns the RCCL communicator and runs the partitioned step itself (no torch in the process)
            ctx = dist.Distributed.from_environment(local_rank, self_loop=self_loop)
        phase("communicator")
        arch = ctx.arch
        if args.global_size:
            G = args.global_size
            grid = dist.DistributedRectilinearGrid(ctx, size=(G, G, G), extent=(1.0, 1.0, 1.0))
        elif args.workload != "ppp":
            from helpers import tanh_faces
            grid = dist.DistributedRectilinearGrid(ctx, size=(N * world, N, N // 2), x=(0.0, float(world)), y=(0.0, 1.0), z=tanh_faces(N // 2),
                                                   topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
        else:
            grid = dist.DistributedRectilinearGrid(ctx, size=(N * world, N, N), extent=(float(world), 1.0, 1.0))
        model = dist.LibraryDistributedModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"), **workload_physics(ocn, args.workload))
        model.set_option("fuse_substep", int(os.environ.get("OCN_FUSE_SUBSTEP", "1") != "0"))
        step = lambda dt: ocn.time_step(model, dt)            # noqa: E731
        ocn.set_model(model, **dist.local_initial_state(model, initial_state))
        barrier = ctx.barrier
    else:
        arch = ocn.GPU(local_rank % max(1, ocn.ndevices()))
        physics = {}
        if args.workload in ("ppb_stretched", "ppb_physics", "ppb_amd"):
            from helpers import tanh_faces
            grid = ocn.RectilinearGrid(arch, size=(N, N, N // 2), x=(0.0, 1.0), y=(0.0, 1.0), z=tanh_faces(N // 2),
                                       topology=(ocn.Periodic, ocn.Periodic, ocn.Bounded))
            physics = workload_physics(ocn, args.workload)
        else:
            grid = ocn.RectilinearGrid(arch, size=(N, N, N), extent=(1, 1, 1))
        model = ocn.NonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"), **physics)
        step = lambda dt: ocn.time_step(model, dt)            # noqa: E731
        barrier = lambda: None                                # noqa: E731
        ocn.set_model(model, **initial_state(ocn, model))
    model.set_option("tendency_impl", args.tendency_impl)
    arithmetic = int(os.environ.get("OCN_ARITHMETIC", str(args.arithmetic)))
    if arithmetic:
        model.set_option("arithmetic", arithmetic)
    for kv in filter(None, os.environ.get("OCN_MODEL_OPTIONS", "").split(",")):      # A/B experiments: "early_exchange=0,fused_step=0"
        k, v = kv.split("=")
        model.set_option(k.strip(), int(v))
    phase("model")
    dt = 0.1 * (1.0 / (args.global_size or N)) / 0.6                                # Δt = 0.1 Δx / max|u|

    # two untimed initialisation steps regardless of --warmup: the first time-step also evaluates the initial tendencies, creates
    # the RCCL point-to-point communicators (N > 1) and brings the clocks up; the W warm-up steps and the K timed steps follow
    for _ in range(2):
        step(dt)
    phase("first steps")
    for _ in range(args.warmup):
        step(dt)
    phase("warmup")
    model.set_option("profile", 1)
    if distributed:                       # warm the collectives the bracket uses (first use loads RCCL kernels: ~20 ms)
        for _ in range(2):
            barrier()
            ctx.allreduce_max(0.0)
    # the host layer allocates small ctypes argument arrays on every call; a generation-2 collection of a process that has torch
    # loaded costs ~30 ms when it happens to fall into the timed steps (observed), so collect now and pause the collector like
    # `timeit` does
    import gc
    gc.collect()
    gc.disable()
    barrier()
    ocn.synchronize()
    t0 = time.perf_counter()
    debug = os.environ.get("OCN_BENCH_DEBUG") == "1"      # per-step wall times (adds a sync per step: diagnostics only)
    for n in range(args.steps):
        ts = time.perf_counter()
        step(dt)
        if debug:
            ocn.synchronize()
            print(f"[bench] step {n}: {1e3 * (time.perf_counter() - ts):.3f} ms", file=sys.stderr)
    ocn.synchronize()
    t_sync = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed and rank == 0:
        print(f"[bench] closing barrier took {1e3 * (time.perf_counter() - t_sync):.3f} ms of the {1e3 * elapsed:.1f} ms timed region",
              file=sys.stderr)
    tend_ms, tend_n = model.profile_read()
    model.set_option("profile", 0)
    phase("timed")
    # the distribution of single steps (the reference reports min / median / mean of BenchmarkTools samples,
    # benchmark/benchmarkable_nonhydrostatic_model.jl:23-27): K more steps, each bracketed by its own device synchronisation (and, N > 1,
    # a barrier) -- OUTSIDE the timed region above, whose mean is `ms_per_step`
    samples = []
    for _ in range(args.steps):
        barrier()
        ocn.synchronize()
        ts = time.perf_counter()
        step(dt)
        ocn.synchronize()
        samples.append(time.perf_counter() - ts)
    gc.enable()
    median = float(np.median(samples)) if samples else None
    fastest = float(np.min(samples)) if samples else None
    div = ocn.max_abs_divergence(model)
    fused_substep = model.get_option("fuse_substep_active") == 1
    in_kernel = model.get_option("substep_in_tendency_kernel") == 1           # (read here: a partitioned model is closed before the line is built)
    dead_store_skipped = in_kernel and model.get_option("skip_dead_tendency_store") == 1
    arithmetic_active = model.get_option("arithmetic")

    communicator = None
    if distributed:
        elapsed = ctx.allreduce_max(elapsed)
        median, fastest = ctx.allreduce_max(median), ctx.allreduce_max(fastest)
        communicator = ctx.info()
        ctx.barrier()
        model.close()
        ctx.close()
        if rehearsal:
            td.destroy_process_group()
    phase("done")
    if rank != 0:
        return
    cells = float(N) ** 3 * world * (1.0 if args.workload == "ppp" else 0.5)
    gshape = f"{N * world}x{N}x{N}"
    if args.global_size:
        cells, gshape = float(args.global_size) ** 3, f"{args.global_size}x{args.global_size}x{args.global_size}"
    ms = 1e3 * elapsed / args.steps
    value = cells * args.steps / elapsed
    t_launch = 1e-3 * tend_ms / max(tend_n, 1)
    # bytes the tendency launches of a time-step move, averaged over the three: 80 B/cell each; + 80 on the two that carry the next stage's
    # substep (only when it rides in the advection kernel itself: with physics it rides in the epilogue pass); - 40 on the second of those,
    # whose tendency G(U2) feeds that substep only and is not stored (option skip_dead_tendency_store)
    bytes_per_cell = (TENDENCY_BYTES_PER_CELL + (2.0 / 3.0) * FUSED_SUBSTEP_EXTRA_BYTES_PER_CELL * in_kernel
                      - (1.0 / 3.0) * 40.0 * dead_store_skipped)
    cells_per_gpu = cells / world
    achieved = bytes_per_cell * cells_per_gpu / t_launch / 1e9 if tend_n else None
    traffic, traffic_path = measured_traffic(args.tendency_impl, N) if world == 1 and args.workload == "ppp" else (None, None)
    out = {
        "metric": "cell_updates_per_s", "value": value, "unit": "cell-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "ms_per_step_median": 1e3 * median if median else None,
        "ms_per_step_min": 1e3 * fastest if fastest else None,
        "ms_per_step_note": "ms_per_step = mean over the K steps of the timed region (one synchronisation at each end); median / min = K further "
                            "steps synchronised one by one (max over ranks)",
        "higher_is_better": True, "scaling": "strong" if args.global_size else "weak",
        "vs_baseline": value / V100_PUBLISHED_CELL_UPDATES if world == 1 and N == 256 and args.workload == "ppp" and not distributed else None,
        "dtype": "f64", "data": "synthetic; SELF-LOOP: one rank running the N > 1 code path, its RCCL transfers going to itself" if (distributed and os.environ.get("OCN_SELF_LOOP") == "1" and world == 1)
        else "synthetic" if not (distributed and os.environ.get("OCN_REHEARSE_ON_ONE_GPU") == "1")
        else "synthetic; REHEARSAL on one card over gloo + host staging: not a measurement",
        "config": {"workload": (("BASELINE.json configs[3] grid: " if args.global_size == 512 else
                                 "fixed global grid: " if args.global_size else
                                 "BASELINE.json configs[1]" + ("" if world == 1 else f" per GPU ({N}^3 cells on each of {world} x-slabs; "
                                                               "at 8 GPUs the cell count of configs[3]'s 512^3)") + ": ") +
                                f"{gshape} triply-periodic NonhydrostaticModel, WENO(order=5), tracers (T,S), "
                                "RK3, FFT Poisson solve, closure/buoyancy/coriolis = nothing")
                   if args.workload == "ppp" else
                   ("BASELINE.json configs[2]: " + f"{N * world}x{N}x{N // 2} (Periodic, Periodic, Bounded) tanh-stretched z, WENO(order=5), "
                    "tracers (T,S), RK3, Fourier-tridiagonal Poisson solve" +
                    ("; + ScalarDiffusivity, linear SeawaterBuoyancy, Flux / Gradient boundary conditions (SURVEY 8f.1 physics)"
                     if args.workload == "ppb_physics" else
                     "; + AnisotropicMinimumDissipation, linear SeawaterBuoyancy, Flux / Gradient boundary conditions (the physics of "
                     "BASELINE.json configs[4], evaporation flux -rate S included)" if args.workload == "ppb_amd" else "")),
                   "parallelism": "single GPU" if world == 1 else
                   (f"x-slab Partition({world}): RCCL send/recv halos, " +
                    ("substructured x solve (one all-gather of 2 complex per mode per solve)" if args.workload == "ppp" else
                     "distributed Fourier-tridiagonal solve (two all-to-all transposes)")),
                   "dt": dt, "max_abs_divergence_after_run": div,
                   "arithmetic": ("reference operation sequence (IEEE, bit-identical to the oracle)" if not arithmetic_active else
                                  "contracted WENO flux (opt-in, option arithmetic = 1): within 1e-12 of the oracle, not bit-identical"),
                   "tolerance_note": "1e-12 relative vs the oracle holds for u, v, w, T; NOT for a tracer with a large offset (S = 35 + ...) "
                                     "on directions of N >~ 128 points in ANY implementation -- the WENO smoothness indicators are sums of "
                                     "products of values, not of differences (weno_interpolants.jl:204-216); measured bound in DESIGN.md 3",
                   "launcher": launcher_info(world), "communicator": communicator,
                   "vs_baseline_note": "published 56.444 ms on V100 (Oceananigans v0.58.8, docs/src/appendix/"
                                       "benchmarks.md:128); older version without RK3/2 tracers -- context only"},
        "roofline": {"kernel": ("role_tendency_kernel: " if args.tendency_impl == 2 else "fused_tendency_kernel: ") +
                     "flux-sharing WENO-5 tendency evaluation (Gu, Gv, Gw, GT, GS), one launch [+ RK3 substep of the next stage on 2 of 3 launches]"
                     if args.tendency_impl in (1, 2)
                     else "per-field WENO-5 tendency kernels (5 launches)",
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS if achieved else None,
                     # the strict accounting of SURVEY.md 8(d): 80 B/cell on every launch, substep traffic not counted
                     "frac_80B": TENDENCY_BYTES_PER_CELL * cells_per_gpu / t_launch / 1e9 / HBM_PEAK_GBS if tend_n else None,
                     "traffic": traffic, "traffic_from": "committed profile" if traffic else None,
                     "traffic_profile_launch_ms": profile_launch_ms(traffic_path) if traffic else None,
                     "traffic_source": (os.path.relpath(traffic_path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)") if traffic else None,
                     "algorithmic_bytes_per_launch": bytes_per_cell * cells_per_gpu,
                     "algorithmic_bytes_note": ("average over the 3 launches of a time-step: 80 B/cell (tendencies) + 80 B/cell on the 2 "
                                                "launches that carry the fused RK3 substep of the next stage" +
                                                (" - 40 B/cell on the second of them (its tendency feeds that substep only and is not stored)"
                                                 if dead_store_skipped else "")) if in_kernel
                     else "80 B/cell: 5 fields read + 5 tendencies written",
                     "valu": measured_valu(args.tendency_impl, N, t_launch, arithmetic_active) if world == 1 and args.workload == "ppp" else None,
                     "avg_launch_ms": 1e3 * t_launch, "launches_timed": tend_n,
                     "share_of_step": tend_ms / (1e3 * elapsed) if elapsed else None},
    }
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the bench contract)
        out["cpu_baseline"] = cpu_baseline(N, dt)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
