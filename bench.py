#!/usr/bin/env python3
"""bench.py -- headline benchmark: ms/time-step and cell-updates/s of the NonhydrostaticModel RK3 time-step
(WENO(order=5), tracers (T, S), FFT pressure solve) -- BASELINE.json's metric, measured like the reference measures it
(benchmark/benchmarkable_nonhydrostatic_model.jl:23-27: device-synchronised `time_step!(model, Δt)` after warm-up).

    python bench.py --gpus N --steps K --warmup W [--size 256] [--no-cpu-baseline]

N = 1: 256^3 triply periodic on one MI355X (BASELINE.json configs[1]).
N > 1: launched by torch.distributed.run, one rank per GPU; x-slab decomposition, WEAK scaling (256^3 per GPU, i.e.
       global (256 N) x 256 x 256). The library owns the RCCL communicator (ocn_dist_create; the ncclUniqueId travels over a
       one-shot TCP exchange on MASTER_PORT + 1) and runs the partitioned step itself: halo send / recv on its communication
       stream, one small all-gather per pressure solve. No torch in the process.

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


PROFILE_TAG = {1: "r01", 2: "r02"}      # committed rocprofv3 summaries of the all-fields kernel (round 1) and the role kernel


def measured_traffic(tendency_impl, N):
    """HBM bytes per tendency launch from the committed rocprofv3 PMC passes (profiles/, FETCH_SIZE and WRITE_SIZE collected
    in separate runs and calibrated as MI355X_MICROARCH.md prescribes); None when no profile matches this configuration"""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG.get(tendency_impl, 'none')}_tendency_traffic.json")
    if N != 256 or not os.path.exists(path):
        return None
    try:
        return float(json.load(open(path))["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


# Measured issue cost of the instruction classes of the WENO flux on MI355X (tools/valu_rates.hip, profiles/r02_valu_rates.txt):
# ns per wave64 instruction per SIMD with 4 waves resident, every SIMD of the chip busy -- i.e. at the clock the chip HOLDS under
# that load (1.8-2.3 GHz by class), not at the 2.4 GHz of the data sheet. FP64 mul / fma / add issue every 4 cycles.
VALU_NS = {"SQ_INSTS_VALU_FMA_F64": 2.27, "SQ_INSTS_VALU_MUL_F64": 2.32, "SQ_INSTS_VALU_ADD_F64": 1.99, "SQ_INSTS_VALU_CVT": 1.75,
           "SQ_INSTS_VALU_FMA_F32": 1.19, "SQ_INSTS_VALU_INT32": 1.11, "SQ_INSTS_VALU_TRANS_F32": 3.44, "SQ_INSTS_VALU_TRANS_F64": 6.94}
VALU_NS_OTHER = 1.5               # selects, 64-bit moves, compares (1.1-1.9 measured)


def measured_valu(tendency_impl, N, t_launch):
    """Issue-side roofline of the same kernel (its binding roof): the VALU wave-instructions per launch by class
    (committed SQ_INSTS_VALU* passes) priced with the measured per-class issue cost -> the time the instruction stream needs on
    1024 fully busy SIMDs; frac = that floor / the live launch time"""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG.get(tendency_impl, 'none')}_tendency_valu.json")
    if N != 256 or not os.path.exists(path) or not t_launch:
        return None
    try:
        c = json.load(open(path))["counters_per_launch"]
        n = float(c["SQ_INSTS_VALU"])
    except (OSError, ValueError, KeyError):
        return None
    classed = {k: float(c[k]) for k in VALU_NS if k in c}
    if classed:
        floor_s = (sum(v * VALU_NS[k] for k, v in classed.items()) + (n - sum(classed.values())) * VALU_NS_OTHER) * 1e-9 / 1024
        weighting = "measured FP64 / FP32 / conversion / transcendental / integer mix x measured issue cost per class"
    else:                          # round-1 profile: total only -- every instruction at the FP64 rate
        floor_s = n * 2.3e-9 / 1024
        weighting = "all instructions priced as FP64 (no class counters in this profile)"
    return {"wave_instructions_per_launch": n, "per_cell": n * 64 / float(N) ** 3, "issue_floor_ms": 1e3 * floor_s,
            "achieved": n / t_launch / 1e9, "peak": n / floor_s / 1e9, "unit": "G wave-instr/s", "frac": floor_s / t_launch,
            "weighting": weighting, "source": os.path.relpath(path, ROOT) + " + profiles/r02_valu_rates.txt"}


def initial_state(ocn, model, seed=1234):
    from helpers import smooth_state
    g = model.grid
    flds = model.fields()
    nodes = {n: g.nodes(f.loc) for n, f in flds.items()}
    return smooth_state(nodes, seed)


def cpu_baseline(size, dt, budget_s=25.0):
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
    return {"value": sample ** 3 * nsteps / el, "unit": "cell-updates/s", "cores": cores, "kind": "port",
            "sample": f"{nsteps} RK3 steps of the same model at {sample}^3 (oracle/ C restatement, OpenMP)",
            "ms_per_step": 1e3 * el / nsteps}


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
    ap.add_argument("--workload", default="ppp", choices=["ppp", "ppb_stretched", "ppb_physics", "ppb_amd"],
                    help="ppp: BASELINE.json configs[1] (the metric's configuration, default); ppb_stretched: configs[2], "
                         "256x256x128 (Periodic, Periodic, Bounded) with tanh-stretched z (Fourier-tridiagonal solver), single GPU; "
                         "ppb_physics: the same grid with the SURVEY 8f.1 physics switched on -- ScalarDiffusivity, linear "
                         "SeawaterBuoyancy (hydrostatic pressure anomaly), surface Flux / bottom Gradient boundary conditions; "
                         "ppb_amd: the physics of BASELINE.json configs[4] (ocean_wind_mixing_and_convection) on that grid -- "
                         "AnisotropicMinimumDissipation, linear SeawaterBuoyancy, wind-stress / heat-flux / bottom-gradient conditions "
                         "and the example's S-dependent evaporation flux J = -rate S")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # OCN_REHEARSE_ON_ONE_GPU=1 (test harness): all ranks share card 0, collectives staged through the host over gloo by the
    # host-orchestrated form of the step (torch.distributed) -- exercises a multi-process run on a one-GPU box; not a measurement
    rehearsal = os.environ.get("OCN_REHEARSE_ON_ONE_GPU") == "1"
    if rehearsal:
        import torch  # noqa: F401  -- must precede the first load of libocn_mi355x.so (see distributed.init_process_group)
    import oldoceananigans_jl_amd as ocn
    N = args.global_size if (args.global_size and world == 1) else args.size
    distributed = world > 1 or os.environ.get("OCN_FORCE_DISTRIBUTED") == "1" or os.environ.get("OCN_SELF_LOOP") == "1"
    if distributed:
        from oldoceananigans_jl_amd import distributed as dist
        # OCN_SELF_LOOP=1 (one rank): the rank is its own west / east neighbour -- the complete N > 1 code path, RCCL send / recv to
        # itself included: the LOCAL cost of the partitioned path, measurable on a one-GPU box
        self_loop = os.environ.get("OCN_SELF_LOOP") == "1" and world == 1
        if rehearsal:
            ctx = dist.init_process_group(local_rank, rehearse_on_one_gpu=True)
        else:
            # the product path: the library owns the RCCL communicator and runs the partitioned step itself (no torch in the process)
            ctx = dist.Distributed.from_environment(local_rank, self_loop=self_loop)
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
        if rehearsal:
            model = dist.DistributedNonhydrostaticModel(grid=grid, advection=ocn.WENO(), tracers=("T", "S"),
                                                        **workload_physics(ocn, args.workload))
            model.fuse_substep = os.environ.get("OCN_FUSE_SUBSTEP", "1") != "0"
            step = lambda dt: dist.time_step(model, dt)          # noqa: E731
            dist.set_model(model, **dist.local_initial_state(model, initial_state))
        else:
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
    dt = 0.1 * (1.0 / (args.global_size or N)) / 0.6                                # Δt = 0.1 Δx / max|u|

    # two untimed initialisation steps regardless of --warmup: the first time-step also evaluates the initial tendencies, creates
    # the RCCL point-to-point communicators (N > 1) and brings the clocks up; the W warm-up steps and the K timed steps follow
    for _ in range(2):
        step(dt)
    for _ in range(args.warmup):
        step(dt)
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
    gc.enable()
    if distributed and rank == 0:
        print(f"[bench] closing barrier took {1e3 * (time.perf_counter() - t_sync):.3f} ms of the {1e3 * elapsed:.1f} ms timed region",
              file=sys.stderr)
    tend_ms, tend_n = model.profile_read()
    model.set_option("profile", 0)
    div = dist.max_abs_divergence(model) if (distributed and rehearsal) else ocn.max_abs_divergence(model)
    fused_substep = (model.fuse_substep_active() if (distributed and rehearsal) else model.get_option("fuse_substep_active") == 1)

    if distributed:
        elapsed = ctx.allreduce_max(elapsed)
        ctx.barrier()
        if rehearsal:
            model.backend.close()
            ctx.dist.destroy_process_group()
        else:
            model.close()
            ctx.close()
    if rank != 0:
        return
    cells = float(N) ** 3 * world * (1.0 if args.workload == "ppp" else 0.5)
    gshape = f"{N * world}x{N}x{N}"
    if args.global_size:
        cells, gshape = float(args.global_size) ** 3, f"{args.global_size}x{args.global_size}x{args.global_size}"
    ms = 1e3 * elapsed / args.steps
    value = cells * args.steps / elapsed
    t_launch = 1e-3 * tend_ms / max(tend_n, 1)
    bytes_per_cell = TENDENCY_BYTES_PER_CELL + (2.0 / 3.0) * FUSED_SUBSTEP_EXTRA_BYTES_PER_CELL * fused_substep
    cells_per_gpu = cells / world
    achieved = bytes_per_cell * cells_per_gpu / t_launch / 1e9 if tend_n else None
    out = {
        "metric": "cell_updates_per_s", "value": value, "unit": "cell-updates/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if args.global_size else "weak",
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
                     "traffic": measured_traffic(args.tendency_impl, N) if world == 1 and args.workload == "ppp" else None,
                     "traffic_source": f"profiles/{PROFILE_TAG.get(args.tendency_impl, 'none')}_tendency_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
                     "algorithmic_bytes_per_launch": bytes_per_cell * cells_per_gpu,
                     "algorithmic_bytes_note": ("average over the 3 launches of a time-step: 80 B/cell (tendencies) + 80 B/cell on the 2 "
                                                "launches that carry the fused RK3 substep of the next stage") if fused_substep
                     else "80 B/cell: 5 fields read + 5 tendencies written",
                     "valu": measured_valu(args.tendency_impl, N, t_launch) if world == 1 and args.workload == "ppp" else None,
                     "avg_launch_ms": 1e3 * t_launch, "launches_timed": tend_n,
                     "share_of_step": tend_ms / (1e3 * elapsed) if elapsed else None},
    }
    if not args.no_cpu_baseline and world == 1:          # rank 0 at N = 1 only (the bench contract)
        out["cpu_baseline"] = cpu_baseline(N, dt)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
