// ocn_dist.h -- the communicator and the partitioned time-step INSIDE the library (included by ocn_api.hip behind the model code).
//
// Reference: src/DistributedComputations/ -- `Distributed(arch; partition = Partition(R))` (distributed_architectures.jl:166-302),
// send / recv of halo buffers with the two x neighbours (halo_communication.jl:170-187,300,326), MPI.Alltoallv! of the transposes
// (distributed_transpose.jl:185-191), the distributed solvers' solve! (distributed_fft_based_poisson_solver.jl:141-178,
// distributed_fft_tridiagonal_solver.jl:153-257) and the interior / buffer split of
// Models/interleave_communication_and_computation.jl:9-67.
//
// One process per GPU. The product transport is RCCL over xGMI: the library owns the communicator (ncclCommInitRank, the unique id
// comes from the caller's bootstrap), halo transfers run on the library's communication stream between two events, so kernels
// launched afterwards on the compute stream overlap them; the small collectives of the pressure solve run on the compute stream
// itself. librccl is opened at run time (dlopen) the first time a communicator is asked for: single-GPU users never load it.
// A second transport takes the collectives as caller-supplied function pointers (ocn_transport_t): what an MPI.jl binder would plug
// in, and what the tests use to run R ranks on one card (threads) or over gloo.
#pragma once
#include <dlfcn.h>

// ---- RCCL, resolved at run time (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 bytes) ----
typedef struct { char internal[128]; } ocn_nccl_id;
struct RcclApi {
    void *handle = nullptr;
    int (*GetUniqueId)(ocn_nccl_id *) = nullptr;
    int (*CommInitRank)(void **, int, ocn_nccl_id, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;         // optional (ocn_dist_comm_info): what the communicator itself reports
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*CommCuDevice)(void *, int *) = nullptr;
};
static RcclApi g_rccl;
enum { OCN_NCCL_FLOAT64 = 8, OCN_NCCL_MAX = 2 };      // ncclDouble, ncclMax (rccl.h)

static int rccl_load() {
    if (g_rccl.handle) return OCN_OK;
    const char *names[] = {getenv("OCN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return fail(OCN_ESTATE, "librccl could not be opened (%s); set OCN_RCCL_LIB", dlerror());
#define OCN_RCCL_SYM(field, name) \
    if (!(*(void **)(&g_rccl.field) = dlsym(h, name))) { dlclose(h); return fail(OCN_ESTATE, "librccl lacks %s", name); }
    OCN_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    OCN_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    OCN_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    OCN_RCCL_SYM(Send, "ncclSend");
    OCN_RCCL_SYM(Recv, "ncclRecv");
    OCN_RCCL_SYM(GroupStart, "ncclGroupStart");
    OCN_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    OCN_RCCL_SYM(AllGather, "ncclAllGather");
    OCN_RCCL_SYM(AllReduce, "ncclAllReduce");
    OCN_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef OCN_RCCL_SYM
    *(void **)(&g_rccl.CommCount) = dlsym(h, "ncclCommCount");
    *(void **)(&g_rccl.CommUserRank) = dlsym(h, "ncclCommUserRank");
    *(void **)(&g_rccl.CommCuDevice) = dlsym(h, "ncclCommCuDevice");
    g_rccl.handle = h;
    return OCN_OK;
}
#define NCCL_TRY(call)                                                                                           \
    do {                                                                                                         \
        int _r = (call);                                                                                         \
        if (_r != 0) return fail(1000 + _r, "%s: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
    } while (0)
// inside ncclGroupStart / ncclGroupEnd: a failing call must not leave the group open (the communicator would swallow every later call
// into it): close the group, then report the FIRST error
#define NCCL_TRY_IN_GROUP(call)                                                                                  \
    do {                                                                                                         \
        int _r = (call);                                                                                         \
        if (_r != 0) {                                                                                           \
            g_rccl.GroupEnd();                                                                                   \
            return fail(1000 + _r, "%s: %s", #call, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");    \
        }                                                                                                        \
    } while (0)

struct ocn_dist_s {
    int world = 1, rank = 0, west = 0, east = 0;
    int Rx = 1, Ry = 1, south = 0, north = 0;    // pencil layout (ocn_dist_set_layout): rank = ix * Ry + iy
    int kind = 0;                       // 0 RCCL, 1 caller-supplied transport
    void *comm = nullptr;               // ncclComm_t
    hipStream_t comm_stream = nullptr;  // halo transfers (overlap with kernels on the compute stream)
    hipEvent_t ready = nullptr, done = nullptr;
    ocn_transport_t tr = {};
    double *scalar = nullptr;           // device scratch of the scalar reductions
    bool in_flight = false;
    bool self_loop = false;             // ONE rank that is its own west and east neighbour (see ocn_dist_set_self_loop)
};

extern "C" int ocn_dist_unique_id(void *id128) {
    if (!id128) return fail(OCN_EINVAL, "NULL argument");
    int rc = rccl_load();
    if (rc) return rc;
    ocn_nccl_id id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return OCN_OK;
}

extern "C" int ocn_dist_destroy(ocn_dist_t d) {
    if (!d) return OCN_OK;
    if (d->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(d->comm);
    if (d->ready) hipEventDestroy(d->ready);
    if (d->done) hipEventDestroy(d->done);
    if (d->comm_stream) hipStreamDestroy(d->comm_stream);
    hipFree(d->scalar);
    delete d;
    return OCN_OK;
}

static int dist_common_init(ocn_dist_s *d, int world, int rank) {
    d->world = world; d->rank = rank;
    d->west = (rank - 1 + world) % world;          // periodic wrap of the x neighbours (distributed_architectures.jl:391-434)
    d->east = (rank + 1) % world;
    d->Rx = world; d->Ry = 1; d->south = d->north = rank;
    HIP_TRY(hipEventCreateWithFlags(&d->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&d->done, hipEventDisableTiming));
    HIP_TRY(dev_alloc((void **)&d->scalar, 2 * sizeof(double)));
    return OCN_OK;
}

// Distributed(GPU(); partition = Partition(world)) with the library's own RCCL communicator. `id128`: the 128-byte ncclUniqueId made by
// ocn_dist_unique_id on rank 0 and carried to every rank by the caller's bootstrap (MPI_Bcast, a TCP store, a file).
extern "C" int ocn_dist_create(ocn_dist_t *dist, const void *id128, int world, int rank) {
    NEED_INIT();
    if (!dist || !id128 || world < 1 || rank < 0 || rank >= world) return fail(OCN_EINVAL, "invalid argument");
    int rc = rccl_load();
    if (rc) return rc;
    ocn_dist_s *d = new ocn_dist_s();
    d->kind = 0;
    if ((rc = dist_common_init(d, world, rank))) { ocn_dist_destroy(d); return rc; }
    hipError_t e = hipStreamCreateWithFlags(&d->comm_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { ocn_dist_destroy(d); return fail((int)e, "hipStreamCreate: %s", hipGetErrorString(e)); }
    ocn_nccl_id id;
    memcpy(&id, id128, sizeof(id));
    int r = g_rccl.CommInitRank(&d->comm, world, id, rank);
    if (r != 0) { ocn_dist_destroy(d); return fail(1000 + r, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    *dist = d;
    return OCN_OK;
}

// the same architecture object over a caller-supplied transport (MPI.jl in a Julia binder; threads or gloo in the tests)
extern "C" int ocn_dist_create_transport(ocn_dist_t *dist, const ocn_transport_t *transport, int world, int rank) {
    NEED_INIT();
    if (!dist || !transport || world < 1 || rank < 0 || rank >= world) return fail(OCN_EINVAL, "invalid argument");
    if (!transport->exchange_start || !transport->exchange_wait || !transport->all_to_all || !transport->all_gather || !transport->allreduce_max)
        return fail(OCN_EINVAL, "every entry of ocn_transport_t must be set");
    ocn_dist_s *d = new ocn_dist_s();
    d->kind = 1;
    d->tr = *transport;
    int rc = dist_common_init(d, world, rank);
    if (rc) { ocn_dist_destroy(d); return rc; }
    *dist = d;
    return OCN_OK;
}

// MEASUREMENT / TEST ONLY: a communicator of ONE rank treats x as partitioned with itself as both neighbours, so the complete N > 1
// code path (FullyConnected x, pack / exchange / unpack, thin exchanges, gathered interface solve or transposes) runs -- and its
// local cost can be timed -- on a one-GPU box. The result equals the one-rank Periodic run. Set before the model is created.
extern "C" int ocn_dist_set_self_loop(ocn_dist_t d, int enabled) {
    if (!d) return fail(OCN_EINVAL, "NULL argument");
    if (enabled && d->world != 1) return fail(OCN_EINVAL, "self_loop needs a communicator of one rank");
    d->self_loop = enabled != 0;
    return OCN_OK;
}

// What the TRANSPORT itself reports (bench.py prints it next to n_gpus): kind 0 = the library's RCCL communicator -- comm_ranks /
// comm_rank / device from ncclCommCount / ncclCommUserRank / ncclCommCuDevice, i.e. the ranks RCCL really connected, not the numbers the
// caller passed in; kind 1 = caller-supplied collectives (world / rank as given, device = the library's current device).
extern "C" int ocn_dist_comm_info(ocn_dist_t d, int *kind, int *comm_ranks, int *comm_rank, int *device) {
    if (!d) return fail(OCN_EINVAL, "NULL argument");
    int n = d->world, r = d->rank, dev = -1;
    hipGetDevice(&dev);
    if (d->kind == 0 && d->comm) {
        if (g_rccl.CommCount) NCCL_TRY(g_rccl.CommCount(d->comm, &n));
        if (g_rccl.CommUserRank) NCCL_TRY(g_rccl.CommUserRank(d->comm, &r));
        if (g_rccl.CommCuDevice) NCCL_TRY(g_rccl.CommCuDevice(d->comm, &dev));
    }
    if (kind) *kind = d->kind;
    if (comm_ranks) *comm_ranks = n;
    if (comm_rank) *comm_rank = r;
    if (device) *device = dev;
    return OCN_OK;
}

extern "C" int ocn_dist_info(ocn_dist_t d, int *world, int *rank, int *west, int *east) {
    if (!d) return fail(OCN_EINVAL, "NULL argument");
    if (world) *world = d->world;
    if (rank) *rank = d->rank;
    if (west) *west = d->west;
    if (east) *east = d->east;
    return OCN_OK;
}

// Partition(Rx, Ry): rank2index / index2rank with the LAST index fastest (distributed_architectures.jl:354-389) and periodic wrap of
// the four neighbours (:391-434). Call before the model is created.
extern "C" int ocn_dist_set_layout(ocn_dist_t d, int Rx, int Ry) {
    if (!d || Rx < 1 || Ry < 1 || Rx * Ry != d->world) return fail(OCN_EINVAL, "Rx * Ry must equal the number of ranks");
    d->Rx = Rx; d->Ry = Ry;
    const int ix = d->rank / Ry, iy = d->rank % Ry;
    d->west = ((ix - 1 + Rx) % Rx) * Ry + iy;
    d->east = ((ix + 1) % Rx) * Ry + iy;
    d->south = ix * Ry + (iy - 1 + Ry) % Ry;
    d->north = ix * Ry + (iy + 1) % Ry;
    return OCN_OK;
}

// one exchange with an explicit pair of peers, ordered on the compute stream (pencil partitions: the x and the y hop of a fill):
// what leaves through the low side arrives in peer_lo's HIGH halo
static int dist_exchange_pair(ocn_dist_t d, int peer_lo, int peer_hi, const double *lo_send, const double *hi_send, double *lo_recv,
                              double *hi_recv, size_t count) {
    if (d->kind == 1) {
        if (!d->tr.exchange_peers) return fail(OCN_ENOTSUP, "this transport has no exchange_peers entry (needed by pencil partitions)");
        int rc = d->tr.exchange_peers(d->tr.user, peer_lo, peer_hi, lo_send, hi_send, lo_recv, hi_recv, count, (void *)g_stream);
        return rc ? fail(rc, "transport exchange_peers failed") : OCN_OK;
    }
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY_IN_GROUP(g_rccl.Send(lo_send, count, OCN_NCCL_FLOAT64, peer_lo, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(hi_recv, count, OCN_NCCL_FLOAT64, peer_hi, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Send(hi_send, count, OCN_NCCL_FLOAT64, peer_hi, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(lo_recv, count, OCN_NCCL_FLOAT64, peer_lo, d->comm, g_stream));
    NCCL_TRY(g_rccl.GroupEnd());
    return OCN_OK;
}

// MPI.Isend / Irecv! with both x neighbours (halo_communication.jl:300,326): `count` doubles per side. The transfers are ordered
// behind everything already on the compute stream and run on the communication stream; kernels launched next overlap them.
extern "C" int ocn_dist_exchange_start(ocn_dist_t d, const double *west_send, const double *east_send, double *west_recv, double *east_recv,
                                       size_t count) {
    NEED_INIT();
    if (!d || !west_send || !east_send || !west_recv || !east_recv) return fail(OCN_EINVAL, "NULL argument");
    if (d->in_flight) return fail(OCN_ESTATE, "an exchange is already in flight");
    if (d->kind == 1) {
        int rc = d->tr.exchange_start(d->tr.user, west_send, east_send, west_recv, east_recv, count, (void *)g_stream);
        if (rc) return fail(rc, "transport exchange_start failed");
        d->in_flight = true;
        return OCN_OK;
    }
    HIP_TRY(hipEventRecord(d->ready, g_stream));
    HIP_TRY(hipStreamWaitEvent(d->comm_stream, d->ready, 0));
    // one group: the pairing is unambiguous even when both neighbours are the same rank (R = 2) or this rank itself (R = 1) --
    // what leaves through the west side arrives in the west neighbour's EAST halo
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY_IN_GROUP(g_rccl.Send(west_send, count, OCN_NCCL_FLOAT64, d->west, d->comm, d->comm_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(east_recv, count, OCN_NCCL_FLOAT64, d->east, d->comm, d->comm_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Send(east_send, count, OCN_NCCL_FLOAT64, d->east, d->comm, d->comm_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(west_recv, count, OCN_NCCL_FLOAT64, d->west, d->comm, d->comm_stream));
    NCCL_TRY(g_rccl.GroupEnd());
    HIP_TRY(hipEventRecord(d->done, d->comm_stream));
    d->in_flight = true;
    return OCN_OK;
}

// MPI.Waitall (halo_communication.jl:164-165): later work on the compute stream waits for the transfers; the host does not
extern "C" int ocn_dist_exchange_wait(ocn_dist_t d) {
    NEED_INIT();
    if (!d) return fail(OCN_EINVAL, "NULL argument");
    if (!d->in_flight) return OCN_OK;
    d->in_flight = false;
    if (d->kind == 1) {
        int rc = d->tr.exchange_wait(d->tr.user, (void *)g_stream);
        return rc ? fail(rc, "transport exchange_wait failed") : OCN_OK;
    }
    HIP_TRY(hipStreamWaitEvent(g_stream, d->done, 0));
    return OCN_OK;
}

// the same exchange for data the very next kernel reads (the one-column exchanges of the pressure step): nothing could overlap it, so it
// runs on the COMPUTE stream itself -- no event hop to the communication stream and back (two cross-queue dependencies, ~15 us each)
static int dist_exchange_inline(ocn_dist_t d, const double *west_send, const double *east_send, double *west_recv, double *east_recv, size_t count) {
    if (d->in_flight) return fail(OCN_ESTATE, "an exchange is already in flight");
    if (d->kind == 1) {
        int rc = d->tr.exchange_start(d->tr.user, west_send, east_send, west_recv, east_recv, count, (void *)g_stream);
        if (rc) return fail(rc, "transport exchange_start failed");
        rc = d->tr.exchange_wait(d->tr.user, (void *)g_stream);
        return rc ? fail(rc, "transport exchange_wait failed") : OCN_OK;
    }
    NCCL_TRY(g_rccl.GroupStart());
    NCCL_TRY_IN_GROUP(g_rccl.Send(west_send, count, OCN_NCCL_FLOAT64, d->west, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(east_recv, count, OCN_NCCL_FLOAT64, d->east, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Send(east_send, count, OCN_NCCL_FLOAT64, d->east, d->comm, g_stream));
    NCCL_TRY_IN_GROUP(g_rccl.Recv(west_recv, count, OCN_NCCL_FLOAT64, d->west, d->comm, g_stream));
    NCCL_TRY(g_rccl.GroupEnd());
    return OCN_OK;
}

// MPI.Alltoallv! with equal counts (distributed_transpose.jl:185-191): piece r of `send` (count doubles) goes to rank r
extern "C" int ocn_dist_all_to_all(ocn_dist_t d, const double *send, double *recv, size_t count_per_rank) {
    NEED_INIT();
    if (!d || !send || !recv) return fail(OCN_EINVAL, "NULL argument");
    if (d->kind == 1) {
        int rc = d->tr.all_to_all(d->tr.user, send, recv, count_per_rank, (void *)g_stream);
        return rc ? fail(rc, "transport all_to_all failed") : OCN_OK;
    }
    NCCL_TRY(g_rccl.GroupStart());
    for (int r = 0; r < d->world; ++r) {
        NCCL_TRY_IN_GROUP(g_rccl.Send(send + (size_t)r * count_per_rank, count_per_rank, OCN_NCCL_FLOAT64, r, d->comm, g_stream));
        NCCL_TRY_IN_GROUP(g_rccl.Recv(recv + (size_t)r * count_per_rank, count_per_rank, OCN_NCCL_FLOAT64, r, d->comm, g_stream));
    }
    NCCL_TRY(g_rccl.GroupEnd());
    return OCN_OK;
}

// MPI.Allgather of equal pieces: rank r's `count` doubles land at recv[r * count] on every rank
extern "C" int ocn_dist_all_gather(ocn_dist_t d, const double *send, double *recv, size_t count) {
    NEED_INIT();
    if (!d || !send || !recv) return fail(OCN_EINVAL, "NULL argument");
    if (d->kind == 1) {
        int rc = d->tr.all_gather(d->tr.user, send, recv, count, (void *)g_stream);
        return rc ? fail(rc, "transport all_gather failed") : OCN_OK;
    }
    NCCL_TRY(g_rccl.AllGather(send, recv, count, OCN_NCCL_FLOAT64, d->comm, g_stream));
    return OCN_OK;
}

// all_reduce(max, value) over the ranks; synchronises the host (diagnostics, CFL, the bench's timing)
extern "C" int ocn_dist_allreduce_max(ocn_dist_t d, double *value) {
    NEED_INIT();
    if (!d || !value) return fail(OCN_EINVAL, "NULL argument");
    if (d->kind == 1) {
        int rc = d->tr.allreduce_max(d->tr.user, value);
        return rc ? fail(rc, "transport allreduce_max failed") : OCN_OK;
    }
    HIP_TRY(hipMemcpyAsync(d->scalar, value, sizeof(double), hipMemcpyHostToDevice, g_stream));
    NCCL_TRY(g_rccl.AllReduce(d->scalar, d->scalar + 1, 1, OCN_NCCL_FLOAT64, OCN_NCCL_MAX, d->comm, g_stream));
    HIP_TRY(hipMemcpyAsync(value, d->scalar + 1, sizeof(double), hipMemcpyDeviceToHost, g_stream));
    HIP_TRY(hipStreamSynchronize(g_stream));
    return OCN_OK;
}

extern "C" int ocn_dist_barrier(ocn_dist_t d) {
    double v = 0.0;
    return ocn_dist_allreduce_max(d, &v);
}

#include "ocn_transpose.h"

// ---------------------------------------------------------------------------------------------------------------------
// the partitioned model: NonhydrostaticModel on Distributed(GPU(); partition = Partition(R)) (x-slabs)
// ---------------------------------------------------------------------------------------------------------------------
// Irregular partition: the gathered solve (see gather_assemble_kernel). Memory and traffic grow with the GLOBAL grid on every rank --
// the fallback for slab sizes the transposing / substructured solvers do not take, not the scaling path.
struct GatheredSolve {
    ocn_grid_s *ggrid = nullptr;        // the global grid, x Periodic
    ocn_poisson_s *solver = nullptr;    // the single-GPU solver on it
    double *loc = nullptr, *all = nullptr, *gp = nullptr;   // own source term (nmax, Ny, Nz); all ranks' (.., R); global haloed solution
    SlabTable table = {};
    int nmax = 0, Nxg = 0, nymax = 0, Nyg = 0;
};

struct DistModel {
    ocn_dist_t dist = nullptr;
    ocn_dist_poisson_t solver = nullptr;
    GatheredSolve *gs = nullptr;
    PencilSolve *ps = nullptr;          // pencil partition of a triply Periodic regular grid: the reference's transposing FFT solver
    double *ws = nullptr, *es = nullptr, *wr = nullptr, *er = nullptr;     // halo buffers: Hx columns of every prognostic field per side
    size_t slab_total = 0;
    double *p2 = nullptr;                                                   // the solver's raw solution (p dt); the correction passes write p / dt into the model's pressure
    double *buf_a = nullptr, *buf_b = nullptr;                              // the solver's payload / gathered or send / recv buffers
    size_t payload = 0, nbuf = 0;
    bool halos_in_flight = false;
    int async_halos = -1;       // -1 automatic (slab wide enough for whole-tile strips), 0 off, 1 on
    int thin_halos = 1;         // pressure step: exchange the ONE column that is read (u[Nx+1], p[0]) instead of Hx columns
    int early_exchange = 1;     // start update_state!'s exchange from make_pressure_correction!
    int strip_width = 0;        // 0 automatic
    bool bounded_x = false;     // the partitioned direction is Bounded: Right / LeftConnected end ranks, no wrap-around neighbour
    bool pencil = false;        // Partition(Rx, Ry) with Ry > 1: a second hop along y per fill (corners ride along)
    double *ss = nullptr, *ns = nullptr, *sr = nullptr, *nr = nullptr;     // y-halo buffers: Hy rows of every prognostic field per side
    bool plain_y = true;        // the non-partitioned y direction is Periodic (what the partitioned solvers transform); Bounded / Flat y: gathered solve
    // Round 3 -- the pressure step of (connected, Periodic, Periodic) x-slabs without the passes between its stages (ocn_kernels.h,
    // "Round 3"): one-column exchanges of u and p through dense (Ny, Nz) buffers that the source-term / correction kernels read directly,
    // the solution left z-fastest, the early exchange packed from wrapped interior indices
    bool fused_step = false;
    double *cws = nullptr, *ces = nullptr, *cwr = nullptr, *cer = nullptr;   // column buffers: Ny * Nz doubles each
    bool fused() const { return fused_step && thin_halos != 0 && partitioned(); }
    bool general() const { return bounded_x || pencil || !plain_y; }                    // no overlap / thin exchanges on such partitions
    bool partitioned() const { return dist->world > 1 || dist->self_loop; }
};

static void dist_model_free(DistModel *dm) {
    if (!dm) return;
    ocn_dist_poisson_destroy(dm->solver);
    pencil_solve_free(dm->ps);
    if (dm->gs) {
        ocn_poisson_destroy(dm->gs->solver);
        ocn_grid_destroy(dm->gs->ggrid);
        hipFree(dm->gs->loc); hipFree(dm->gs->all); hipFree(dm->gs->gp);
        delete dm->gs;
    }
    hipFree(dm->ws); hipFree(dm->es); hipFree(dm->wr); hipFree(dm->er); hipFree(dm->p2); hipFree(dm->buf_a);
    hipFree(dm->ss); hipFree(dm->ns); hipFree(dm->sr); hipFree(dm->nr);
    hipFree(dm->cws); hipFree(dm->ces); hipFree(dm->cwr); hipFree(dm->cer);
    if (dm->buf_b != dm->buf_a) hipFree(dm->buf_b);
    delete dm;
}

static size_t dist_slab(const DGrid &g, const int loc[3], int depth) {
    int P[3];
    parent_size(g, loc, P);
    return (size_t)depth * P[1] * P[2];
}

static size_t dist_rows(const DGrid &g, const int loc[3], int depth) {
    int P[3];
    parent_size(g, loc, P);
    return (size_t)depth * P[0] * P[2];
}

static int y_halo_buffers(const DGrid &g, double *const *fields, const int (*locs)[3], int n, double *south, double *north, bool pack) {
    FieldList fl;
    RowList rl;
    fl.n = n;
    long off = 0, maxt = 0;
    for (int f = 0; f < n; ++f) {
        int P[3];
        parent_size(g, locs[f], P);
        fl.p[f] = fields[f];
        rl.off[f] = off; rl.p0[f] = P[0]; rl.p1[f] = P[1]; rl.p2[f] = P[2];
        const long cnt = (long)g.Hy * P[0] * P[2];
        off += cnt;
        maxt = std::max(maxt, cnt);
    }
    const int nb = (int)((maxt + 255) / 256);
    // a wall side (Right / LeftConnected y) keeps what the local boundary fill left there: the ring still wraps, the data is dropped
    if (pack) hipLaunchKernelGGL(y_halo_buffer_kernel<true>, dim3(nb), dim3(256), 0, g_stream, fl, rl, g.Ny, g.Hy, g.Hy, south, north, true, true);
    else      hipLaunchKernelGGL(y_halo_buffer_kernel<false>, dim3(nb), dim3(256), 0, g_stream, fl, rl, g.Ny, g.Hy, g.Hy, south, north,
                                 !wall_lo(g.ty), !wall_hi(g.ty));
    KERNEL_CHECK();
    return OCN_OK;
}

// fill_send_buffers! (communication_buffers.jl:281-289) reading the y / z halo rows at their wrapped interior source: what a local fill
// followed by x_halo_buffers(pack) would put into the buffers, without the fill. y, z Periodic only.
static int x_halo_pack_wrapped(const DGrid &g, double *const *fields, const int (*locs)[3], int n, double *west, double *east) {
    if (n > OCN_MAX_FIELDS) return fail(OCN_EINVAL, "at most %d fields per call", OCN_MAX_FIELDS);
    FieldList fl;
    SlabList sl;
    fl.n = n;
    const long rows = (long)(g.Ny + 2 * g.Hy) * (g.Nz + 2 * g.Hz);
    for (int f = 0; f < n; ++f) {
        int P[3];
        parent_size(g, locs[f], P);
        if ((long)P[1] * P[2] != rows) return fail(OCN_ESTATE, "x_halo_pack_wrapped: fields of different parent shapes");
        fl.p[f] = fields[f]; sl.p0[f] = P[0]; sl.rows[f] = rows; sl.off[f] = (long)f * g.Hx * rows;
    }
    const long threads = (long)g.Hx * rows;
    hipLaunchKernelGGL(x_halo_pack_wrapped_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, g_stream, fl, sl, g.Nx, g.Hx, g.Hx, g.Ny, g.Hy,
                       g.Nz, g.Hz, west, east);
    KERNEL_CHECK();
    return OCN_OK;
}

// fill_halo_regions! of partitioned fields (halo_communication.jl:87-110): local boundary conditions first
// (boundary_condition_ordering.jl: the communication condition last), then the x exchange of `nx` leading fields, `depth` columns
static int dist_fill_halo_regions(ocn_model_s *m, double *const *fields, const int (*locs)[3], int n, bool fill_open,
                                  const ocn_bc_t (*bcs)[6], int nx = -1, int depth = 0) {
    DistModel *dm = m->dm;
    int rc = fill_halo_regions(m->grid, fields, locs, n, fill_open, bcs);
    if (rc || !dm->partitioned()) return rc;
    const DGrid &g = m->grid->d;
    if (nx < 0 || dm->pencil) nx = n;
    if (depth <= 0 || dm->pencil) depth = g.Hx;
    size_t count = 0;
    for (int f = 0; f < nx; ++f) count += dist_slab(g, locs[f], depth);
    if (dm->dist->Rx > 1 || dm->dist->self_loop) {
        if ((rc = x_halo_buffers(g, fields, locs, nx, dm->ws, dm->es, true, depth))) return rc;
        if (dm->pencil) {
            if ((rc = dist_exchange_pair(dm->dist, dm->dist->west, dm->dist->east, dm->ws, dm->es, dm->wr, dm->er, count))) return rc;
        } else {
            if ((rc = ocn_dist_exchange_start(dm->dist, dm->ws, dm->es, dm->wr, dm->er, count))) return rc;
            if ((rc = ocn_dist_exchange_wait(dm->dist))) return rc;
        }
        if ((rc = x_halo_buffers(g, fields, locs, nx, dm->wr, dm->er, false, depth))) return rc;
    }
    if (!dm->pencil) return OCN_OK;
    // second hop: Hy rows over the whole x extent -- the x halos received above included, which carries the corners
    // (fill_corners!, halo_communication.jl:137-162, as two one-dimensional hops)
    size_t county = 0;
    for (int f = 0; f < n; ++f) county += dist_rows(g, locs[f], g.Hy);
    if ((rc = y_halo_buffers(g, fields, locs, n, dm->ss, dm->ns, true))) return rc;
    if ((rc = dist_exchange_pair(dm->dist, dm->dist->south, dm->dist->north, dm->ss, dm->ns, dm->sr, dm->nr, county))) return rc;
    return y_halo_buffers(g, fields, locs, n, dm->sr, dm->nr, false);
}

// solve_for_pressure! + solve!(::DistributedFFTBasedPoissonSolver | ::DistributedFourierTridiagonalPoissonSolver)
static int dist_solve_for_pressure(ocn_model_s *m) {
    DistModel *dm = m->dm;
    int rc;
    if (dm->ps) {
        // pencil partition, triply Periodic: source term into the z-local complex field, the transposing solve, real part into p dt
        const DGrid &g = m->grid->d;
        if ((rc = source_term(g, m->U[0], m->U[1], m->U[2], dm->ps->tf->zfield, false))) return rc;
        if ((rc = pencil_solve(dm->ps))) return rc;
        hipLaunchKernelGGL(copy_real_kernel, grid3(g.Nx, g.Ny, g.Nz, BLK), BLK, 0, g_stream, g, make_view(g, dm->p2, LOC_C),
                           (const double2 *)dm->ps->tf->zfield, 1.0, false, (const double2 *)nullptr);
        KERNEL_CHECK();
        return OCN_OK;
    }
    if (dm->gs) {
        // irregular partition: every rank gathers the whole source term and runs the single-GPU solver on the global grid
        GatheredSolve *q = dm->gs;
        const DGrid &g = m->grid->d, &G = q->ggrid->d;
        ocn_poisson_s *ps = q->solver;
        const size_t piece = (size_t)q->nmax * q->nymax * g.Nz;
        if ((rc = source_term(g, m->U[0], m->U[1], m->U[2], q->loc, ps->kind == 1, true, q->nmax, (long)q->nmax * q->nymax))) return rc;
        if ((rc = ocn_dist_all_gather(dm->dist, q->loc, q->all, piece))) return rc;
        const bool real_path = g_real_fft && !ps->general;
        if (!real_path && (rc = ensure_complex(ps))) return rc;
        if (real_path)
            hipLaunchKernelGGL(gather_assemble_kernel<false>, grid3(G.Nx, G.Ny, G.Nz, BLK), BLK, 0, g_stream, (const double *)q->all, (void *)ps->rrhs,
                               q->table, q->nmax, q->nymax, G.Nx, G.Ny, G.Nz);
        else
            hipLaunchKernelGGL(gather_assemble_kernel<true>, grid3(G.Nx, G.Ny, G.Nz, BLK), BLK, 0, g_stream, (const double *)q->all,
                               (void *)(ps->kind == 0 ? ps->storage : ps->source), q->table, q->nmax, q->nymax, G.Nx, G.Ny, G.Nz);
        KERNEL_CHECK();
        if ((rc = real_path ? poisson_solve_real(ps, q->gp) : poisson_solve(ps, q->gp))) return rc;
        hipLaunchKernelGGL(slab_extract_kernel, grid3(g.Nx, g.Ny, g.Nz, BLK), BLK, 0, g_stream, g, make_view(g, dm->p2, LOC_C),
                           make_view(G, q->gp, LOC_C), q->table.first[dm->dist->rank / dm->dist->Ry], q->table.firsty[dm->dist->rank % dm->dist->Ry]);
        KERNEL_CHECK();
        return OCN_OK;
    }
    ocn_dist_poisson_s *s = dm->solver;
    if ((rc = ocn_dist_poisson_source_term(s, m->U[0], m->U[1], m->U[2]))) return rc;
    if (s->sub) {
        // z Periodic: substructured solve along the partitioned direction -- local transforms and sweeps, one small all-gather
        if ((rc = ocn_dist_poisson_forward_local(s))) return rc;
        if (dm->partitioned()) {
            if ((rc = ocn_dist_all_gather(dm->dist, dm->buf_a, dm->buf_b, 2 * dm->payload))) return rc;
        } else
            HIP_TRY(hipMemcpyAsync(dm->buf_b, dm->buf_a, 2 * dm->payload * sizeof(double), hipMemcpyDeviceToDevice, g_stream));
        return ocn_dist_poisson_backward_local(s, dm->p2);
    }
    const size_t per_rank = 2 * dm->nbuf / (size_t)dm->dist->world;
    if ((rc = ocn_dist_poisson_forward_yz(s))) return rc;
    if (dm->buf_a != dm->buf_b && (rc = ocn_dist_all_to_all(dm->dist, dm->buf_a, dm->buf_b, per_rank))) return rc;      // transpose_y_to_x!
    if ((rc = ocn_dist_poisson_solve_x(s))) return rc;
    if (dm->buf_a != dm->buf_b && (rc = ocn_dist_all_to_all(dm->dist, dm->buf_a, dm->buf_b, per_rank))) return rc;      // transpose_x_to_y!
    return ocn_dist_poisson_backward_yz(s, dm->p2);
}

// compute_pressure_correction! (pressure_correction.jl:8-20). Of the x halos only ONE column is read before update_state! fills
// everything again: u[Nx+1] by the divergence, p[0] by the correction. The reference's generic fills move Hx columns of u, v, w and of
// p here; `thin_halos` exchanges the one column of u and of p -- identical results in every cell that is read.
// where the solver leaves its dense solution and how cell (i, j, k) is addressed in it: (i-1) + sj (j-1) + sk (k-1)
struct DenseSolution { const double *p; long sj, sk; };
static DenseSolution dist_dense_solution(const ocn_dist_poisson_s *s, const DGrid &g) {
    if (s->xfast) return {(const double *)s->rx, (long)g.Nx, (long)g.Nx * g.Ny};
    return {(const double *)s->zfield, (long)s->Nxe * s->Nz, (long)s->Nxe};          // transposing solvers: paired-column layout (x, z, y)
}

// after a failed step: forget a started exchange (the error already on record stays the one reported)
static void dist_abandon_exchange(ocn_model_s *m) {
    DistModel *dm = m->dm;
    if (!dm) return;
    dm->halos_in_flight = false;
    if (dm->dist && dm->dist->in_flight) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof g_err);
        (void)ocn_dist_exchange_wait(dm->dist);
        memcpy(g_err, keep, sizeof g_err);
    }
}

static int dist_compute_pressure_correction_fused(ocn_model_s *m) {
    // fill_halo_regions!(velocities) reduces to ONE column of u going west (the divergence at i = Nx reads u[Nx+1]; y / z neighbours are
    // read at wrapped interior indices) -- plus, on a Bounded z, the local fill that sets the wall faces of w --; solve;
    // fill_halo_regions!(pNHS) reduces to ONE column of p dt going east (the correction at i = 1 reads p[0]). Both columns travel as
    // dense (Ny, Nz) buffers the consumer kernels read directly.
    DistModel *dm = m->dm;
    const DGrid &g = m->grid->d;
    ocn_dist_poisson_s *s = dm->solver;
    const size_t col = (size_t)g.Ny * g.Nz;
    int rc;
    if (g.tz != OCN_PERIODIC && (rc = fill_halo_regions(m->grid, m->U, m->loc, 3, true, m->any_bc ? m->bcs : nullptr))) return rc;
    hipLaunchKernelGGL(column_pack_kernel, dim3((g.Ny + 255) / 256, g.Nz), dim3(256), 0, g_stream, g, make_view(g, m->U[0], LOC_U), 1, g.Nx,
                       dm->cws, dm->ces);
    KERNEL_CHECK();
    if ((rc = dist_exchange_inline(dm->dist, dm->cws, dm->ces, dm->cwr, dm->cer, col))) return rc;
    if ((rc = dist_poisson_source_term_wrapped(s, m->U[0], m->U[1], m->U[2], dm->cer))) return rc;     // from the east neighbour: its u[1]
    if (s->sub) {
        if ((rc = ocn_dist_poisson_forward_local(s))) return rc;
        if ((rc = ocn_dist_all_gather(dm->dist, dm->buf_a, dm->buf_b, 2 * dm->payload))) return rc;
        if ((rc = dist_poisson_backward_local(s, nullptr, /*keep_zfast=*/true))) return rc;
    } else {
        const size_t per_rank = 2 * dm->nbuf / (size_t)dm->dist->world;
        if ((rc = ocn_dist_poisson_forward_yz(s))) return rc;
        if (dm->buf_a != dm->buf_b && (rc = ocn_dist_all_to_all(dm->dist, dm->buf_a, dm->buf_b, per_rank))) return rc;      // transpose_y_to_x!
        if ((rc = ocn_dist_poisson_solve_x(s))) return rc;
        if (dm->buf_a != dm->buf_b && (rc = ocn_dist_all_to_all(dm->dist, dm->buf_a, dm->buf_b, per_rank))) return rc;      // transpose_x_to_y!
        if ((rc = dist_poisson_backward_yz(s, nullptr, /*keep_dense=*/true))) return rc;
    }
    if (s->zfirst)
        hipLaunchKernelGGL(column_pack_zfast_kernel, dim3((g.Nz + 255) / 256, g.Ny), dim3(256), 0, g_stream, g.Nx, g.Ny, g.Nz, (const double *)s->rreal,
                           dm->cws, dm->ces);
    else {
        const DenseSolution d = dist_dense_solution(s, g);
        hipLaunchKernelGGL(column_pack_dense_kernel, dim3((g.Ny + 255) / 256, g.Nz), dim3(256), 0, g_stream, g.Nx, g.Ny, g.Nz, d.p, dm->cws, dm->ces, d.sj, d.sk);
    }
    KERNEL_CHECK();
    return dist_exchange_inline(dm->dist, dm->cws, dm->ces, dm->cwr, dm->cer, col);                      // cwr: the west neighbour's p[Nx] = our p[0]
}

static int dist_compute_pressure_correction(ocn_model_s *m) {
    DistModel *dm = m->dm;
    if (dm->fused()) return dist_compute_pressure_correction_fused(m);
    const bool thin = dm->thin_halos != 0 && !dm->pencil;
    int rc = dist_fill_halo_regions(m, m->U, m->loc, 3, true, m->any_bc ? m->bcs : nullptr, thin ? 1 : 3, thin ? 1 : 0);
    if (rc) return rc;
    if ((rc = dist_solve_for_pressure(m))) return rc;
    double *pp[1] = {dm->p2};
    const int pl[1][3] = {{OCN_CENTER, OCN_CENTER, OCN_CENTER}};
    return dist_fill_halo_regions(m, pp, pl, 1, true, nullptr, 1, thin ? 1 : 0);
}

// make_pressure_correction! (pressure_correction.jl:40-53); the correction passes read the solver's raw solution (p dt, second array) and
// write p / dt into the model's pressure field -- no divide pass, no pointer swap (ocn_model_field pointers stay valid).
// start_halo_exchange (tendencies are evaluated next): correct the two Hx-wide boundary strips first, fill their y / z halos, pack
// them and START the x exchange of the coming update_state!; the interior correction runs while the halos are in flight.
static int dist_make_pressure_correction(ocn_model_s *m, double dt, bool start_halo_exchange, bool keep_p = true) {
    DistModel *dm = m->dm;
    const DGrid &g = m->grid->d;
    const double dtp = std::fmax(2.220446049250313e-16, dt);
    int rc;
    const bool early = start_halo_exchange && dm->partitioned() && dm->early_exchange && dm->async_halos != 0 && g.Nx > 2 * g.Hx && !dm->general();
    if (dm->fused()) {
        // corrections straight from the z-fastest solution (p[0] from the received column): both strips in one launch, pack from wrapped
        // interior indices (no local fill of the strips), start the exchange, then the interior
        const bool zf = dm->solver->zfirst, zb = g.tz != OCN_PERIODIC;
        const DenseSolution d = dist_dense_solution(dm->solver, g);
        const double *pd = zf ? (const double *)dm->solver->rreal : d.p, *pw = (const double *)dm->cwr;
        const FView vu = make_view(g, m->U[0], LOC_U), vv = make_view(g, m->U[1], LOC_V), vw = make_view(g, m->U[2], LOC_W), vp = make_view(g, m->p, LOC_C);
        auto pcz = [&](int ia, int ib) {
            if (!zf)         // the solution is dense and x-fastest: the single-GPU path's dense correction on a column range
                hipLaunchKernelGGL(pressure_correction_dense_slab_kernel, grid3(ib - ia + 1, g.Ny, g.Nz, BLK), BLK, 0, g_stream, g, vu, vv, vw, pd, pw, vp, dtp, ia, ib,
                                   d.sj, d.sk, zb, keep_p);
            else
                hipLaunchKernelGGL(pressure_correction_zfast_kernel, dim3((ib - ia + 32) / 32, (g.Nz + 31) / 32, (g.Ny + OCN_ZC_JB - 1) / OCN_ZC_JB), dim3(32, 8), 0,
                                   g_stream, g, vu, vv, vw, pd, pw, vp, dtp, ia, ib);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? OCN_OK : fail((int)e, "pressure correction (partitioned slab): %s", hipGetErrorString(e));
        };
        if (!early) return pcz(1, g.Nx);
        if (!zf)
            hipLaunchKernelGGL(pressure_correction_dense_strips_kernel, dim3(1, (g.Ny + 31) / 32, g.Nz), dim3(8, 32), 0, g_stream, g, vu, vv, vw, pd, pw, vp, dtp, g.Hx,
                               d.sj, d.sk, zb, keep_p);
        else
            hipLaunchKernelGGL(pressure_correction_zfast_strips_kernel, dim3(1, (g.Ny + 31) / 32, g.Nz), dim3(8, 32), 0, g_stream, g, vu, vv, vw, pd, pw, vp, dtp, g.Hx);
        KERNEL_CHECK();
        if (zb) {            // a Bounded z: the strips' z halos come from their boundary conditions -- local fill, then the plain pack
            if ((rc = fill_halo_regions(m->grid, m->U, m->loc, m->nf, false, m->any_bc ? m->bcs : nullptr))) return rc;
            if ((rc = x_halo_buffers(g, m->U, m->loc, m->nf, dm->ws, dm->es, true))) return rc;
        } else if ((rc = x_halo_pack_wrapped(g, m->U, m->loc, m->nf, dm->ws, dm->es))) return rc;
        if ((rc = ocn_dist_exchange_start(dm->dist, dm->ws, dm->es, dm->wr, dm->er, dm->slab_total))) return rc;
        dm->halos_in_flight = true;
        return pcz(g.Hx + 1, g.Nx - g.Hx);
    }
    auto pc = [&](const int *range) { return pressure_correction(g, m->U[0], m->U[1], m->U[2], dm->p2, range, m->p, dtp); };
    if (!early) return pc(nullptr);
    const int west[6] = {1, g.Hx, 1, g.Ny, 1, g.Nz}, east[6] = {g.Nx - g.Hx + 1, g.Nx, 1, g.Ny, 1, g.Nz};
    const int mid[6] = {g.Hx + 1, g.Nx - g.Hx, 1, g.Ny, 1, g.Nz};
    if ((rc = pc(west)) || (rc = pc(east))) return rc;
    if ((rc = fill_halo_regions(m->grid, m->U, m->loc, m->nf, false, m->any_bc ? m->bcs : nullptr))) return rc;   // the strips' y / z halos
    if ((rc = x_halo_buffers(g, m->U, m->loc, m->nf, dm->ws, dm->es, true))) return rc;
    if ((rc = ocn_dist_exchange_start(dm->dist, dm->ws, dm->es, dm->wr, dm->er, dm->slab_total))) return rc;
    dm->halos_in_flight = true;
    return pc(mid);
}

static int update_state_tail(ocn_model_s *m, bool compute_tend, const FusedSubstep *sub, const int *amd_range);

// update_state! (update_nonhydrostatic_model_state.jl:20-56) with the interior / buffer split of
// interleave_communication_and_computation.jl:9-67 when the exchange overlaps the interior tendencies
static int dist_update_state(ocn_model_s *m, bool compute_tend, const FusedSubstep *sub) {
    DistModel *dm = m->dm;
    const DGrid &g = m->grid->d;
    const ocn_bc_t(*bcs)[6] = m->any_bc ? m->bcs : nullptr;
    // eddy diffusivities at i = 0 and Nx + 1 are evaluated by the rank itself from the exchanged halos (the numbers a serial
    // Periodic grid's fill copies from the other side): no exchange of the diffusivity fields
    // (on a wall side the halo of the diffusivities comes from their boundary condition, like on a serial Bounded grid)
    const int ext = dm->partitioned() ? 1 : 0;
    const int ey = dm->pencil ? 1 : 0;                 // ... and along y on pencils
    const int amd_range[6] = {1 - (wall_lo(g.tx) ? 0 : ext), g.Nx + (wall_hi(g.tx) ? 0 : ext), 1 - (wall_lo(g.ty) ? 0 : ey), g.Ny + (wall_hi(g.ty) ? 0 : ey),
                              1, g.Nz};
    int rc;
    if (dm->halos_in_flight) {
        // the x exchange was started by make_pressure_correction!: finish the local fills (all columns are final now), take the
        // halos, then everything in one piece
        dm->halos_in_flight = false;
        if ((rc = fill_halo_regions(m->grid, m->U, m->loc, m->nf, false, bcs))) return rc;
        if ((rc = ocn_dist_exchange_wait(dm->dist))) return rc;
        if ((rc = x_halo_buffers(g, m->U, m->loc, m->nf, dm->wr, dm->er, false))) return rc;
        return update_state_tail(m, compute_tend, sub, amd_range);
    }
    const bool physics = has_physics(m) || m->any_flux_bc || m->any_linear_flux;
    const bool overlap = dm->async_halos < 0 ? g.Nx >= 3 * 64 : dm->async_halos != 0;
    if (!compute_tend || !dm->partitioned() || !overlap || g.Nx <= 2 * g.Hx || physics || dm->general()) {
        if ((rc = dist_fill_halo_regions(m, m->U, m->loc, m->nf, false, bcs))) return rc;
        return update_state_tail(m, compute_tend, sub, amd_range);
    }
    // start the exchange, compute the interior that does not depend on x halos, finish, compute the two strips. The reference's strips
    // are Hx wide; the tendency kernel works on 64-lane tiles, so the strips are one tile wide whenever that leaves a tile of interior.
    int W = dm->strip_width > 0 ? dm->strip_width : (g.Nx >= 3 * 64 ? 64 : g.Hx);
    if (!(g.Hx <= W && 2 * W < g.Nx)) return fail(OCN_EINVAL, "strip width %d must satisfy Hx <= W < Nx / 2", W);
    if ((rc = fill_halo_regions(m->grid, m->U, m->loc, m->nf, false, bcs))) return rc;
    if ((rc = x_halo_buffers(g, m->U, m->loc, m->nf, dm->ws, dm->es, true))) return rc;
    if ((rc = ocn_dist_exchange_start(dm->dist, dm->ws, dm->es, dm->wr, dm->er, dm->slab_total))) return rc;
    const int interior[6] = {W + 1, g.Nx - W, 1, g.Ny, 1, g.Nz}, ws[6] = {1, W, 1, g.Ny, 1, g.Nz}, es[6] = {g.Nx - W + 1, g.Nx, 1, g.Ny, 1, g.Nz};
    auto tend = [&](const int *range) {
        return compute_tendencies(g, m->U[0], m->U[1], m->U[2], m->U + 3, m->ntr, m->Gn[0], m->Gn[1], m->Gn[2], m->Gn + 3, range,
                                  m->tendency_impl, sub);
    };
    if ((rc = tend(interior))) return rc;                                  // ... while the halos fly (:27-67)
    if ((rc = ocn_dist_exchange_wait(dm->dist))) return rc;                // synchronize_communication! (distributed_fields.jl:71-88)
    if ((rc = x_halo_buffers(g, m->U, m->loc, m->nf, dm->wr, dm->er, false))) return rc;
    if ((rc = tend(ws)) || (rc = tend(es))) return rc;                     // compute_buffer_tendencies!
    return OCN_OK;
}

static int dist_pressure_step(ocn_model_s *m, double dt, bool tendencies_follow, bool keep_p) {
    int rc = dist_compute_pressure_correction(m);
    if (rc) return rc;
    return dist_make_pressure_correction(m, dt, tendencies_follow, keep_p);
}

// NonhydrostaticModel(grid::DistributedRectilinearGrid; ...) -- `local_grid`: the rank's slab with x topology FullyConnected
// (OCN_CONNECTED) when the direction is partitioned; `Lx_global`: extent of the global domain along x (the solver's eigenvalues)
static int dist_model_create(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global, const int *local_sizes,
                             int global_x_topology, double Ly_global = 0.0, const int *sizes_y = nullptr, int global_y_topology = OCN_PERIODIC);
extern "C" int ocn_dist_model_create(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global) {
    return dist_model_create(model, local_grid, ntracers, dist, Lx_global, nullptr, OCN_PERIODIC);
}
// Partition(Rx, Ry) pencils (f.4 of SURVEY.md 8): rank = ix * Ry + iy, the local grid is connected in x (Rx > 1) and in y (Ry > 1):
// FullyConnected where the global direction is Periodic, Right / Fully / LeftConnected along a Bounded one (insert_connected_topology,
// distributed_grids.jl:339-346); sizes_x[Rx], sizes_y[Ry] list the slab widths (NULL: equal). Every fill makes two hops -- x, then y over
// the whole x extent, so the corners arrive without corner messages (halo_communication.jl:137-162) --; the pressure solve is the
// gathered one (the reference's pencil transposes, distributed_transpose.jl:12-15, are not built).
extern "C" int ocn_dist_model_create_pencil(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                            double Ly_global, int Rx, int Ry, const int *sizes_x, const int *sizes_y, int global_x_topology,
                                            int global_y_topology) {
    if (!dist) return fail(OCN_EINVAL, "NULL argument");
    if (global_x_topology != OCN_PERIODIC && global_x_topology != OCN_BOUNDED) return fail(OCN_EINVAL, "the x direction is Periodic or Bounded");
    if (global_y_topology != OCN_PERIODIC && global_y_topology != OCN_BOUNDED) return fail(OCN_EINVAL, "the y direction is Periodic or Bounded");
    int rc = ocn_dist_set_layout(dist, Rx, Ry);
    if (rc) return rc;
    return dist_model_create(model, local_grid, ntracers, dist, Lx_global, sizes_x, global_x_topology, Ly_global, sizes_y, global_y_topology);
}
// the same for an irregular partition: local_sizes[r] = Nx of rank r (local_size, distributed_grids.jl:44-58: N ÷ R cells per rank and
// the remainder on the last one; or any `Sizes`). Equal sizes take the solvers above; otherwise the pressure solve gathers the source
// term on every rank and runs the single-GPU solver on the global grid (correct for every slab layout, but its memory and traffic
// grow with the global grid: a fallback, not the scaling path).
extern "C" int ocn_dist_model_create_sizes(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                           const int *local_sizes) {
    if (!local_sizes) return fail(OCN_EINVAL, "NULL argument");
    return dist_model_create(model, local_grid, ntracers, dist, Lx_global, local_sizes, OCN_PERIODIC);
}
// ... and for a Bounded partitioned direction (global_x_topology = OCN_BOUNDED): insert_connected_topology (distributed_grids.jl:339-346)
// makes the first rank's local grid RightConnected (wall on its west side), the last one's LeftConnected, the others FullyConnected;
// walls get their boundary conditions and the advection scheme's fallbacks on the wall side only, the ring has no wrap-around
// neighbour. local_sizes as above (NULL: equal slabs). The pressure solve takes the gathered form on the global Bounded grid.
extern "C" int ocn_dist_model_create_partition(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global,
                                               const int *local_sizes, int global_x_topology) {
    if (global_x_topology != OCN_PERIODIC && global_x_topology != OCN_BOUNDED) return fail(OCN_EINVAL, "the partitioned direction is Periodic or Bounded");
    return dist_model_create(model, local_grid, ntracers, dist, Lx_global, local_sizes, global_x_topology);
}

static int gathered_solve_create(DistModel *dm, ocn_grid_t local_grid, double Lx_global, const int *sizes, int global_tx, double Ly_global,
                                 const int *sizes_y, int global_ty) {
    const DGrid &g = local_grid->d;
    const int R = dm->dist->Rx, Ry = dm->dist->Ry, ix = dm->dist->rank / Ry, iy = dm->dist->rank % Ry;
    if (dm->dist->world > OCN_MAX_RANKS) return fail(OCN_ENOTSUP, "gathered solves take at most %d ranks", OCN_MAX_RANKS);
    GatheredSolve *q = new GatheredSolve();
    dm->gs = q;
    q->table.R = R; q->table.Ry = Ry;
    for (int r = 0; r < R; ++r) {
        if (R > 1 && sizes[r] < g.Hx) return fail(OCN_EINVAL, "x slab %d holds %d columns, fewer than the halo %d", r, sizes[r], g.Hx);
        q->table.first[r] = q->Nxg;
        q->Nxg += sizes[r];
        q->nmax = std::max(q->nmax, sizes[r]);
    }
    q->table.first[R] = q->Nxg;
    for (int r = 0; r < Ry; ++r) {
        const int ny = sizes_y ? sizes_y[r] : g.Ny;
        if (Ry > 1 && ny < g.Hy) return fail(OCN_EINVAL, "y slab %d holds %d rows, fewer than the halo %d", r, ny, g.Hy);
        q->table.firsty[r] = q->Nyg;
        q->Nyg += ny;
        q->nymax = std::max(q->nymax, ny);
    }
    q->table.firsty[Ry] = q->Nyg;
    if (sizes[ix] != g.Nx) return fail(OCN_EINVAL, "sizes_x[%d] = %d but the local grid has Nx = %d", ix, sizes[ix], g.Nx);
    if ((sizes_y ? sizes_y[iy] : g.Ny) != g.Ny) return fail(OCN_EINVAL, "sizes_y[%d] = %d but the local grid has Ny = %d", iy, sizes_y[iy], g.Ny);
    const int N[3] = {q->Nxg, q->Nyg, g.Nz}, H[3] = {g.Hx, g.Hy, g.Hz}, topo[3] = {global_tx, Ry > 1 ? global_ty : g.ty, g.tz};
    const double L[3] = {Lx_global, Ry > 1 ? Ly_global : local_grid->L[1], local_grid->L[2]};
    const bool zr = local_grid->z_regular;
    int rc = ocn_grid_create(&q->ggrid, N, H, topo, L, Lx_global / (double)q->Nxg, Ry > 1 ? Ly_global / (double)q->Nyg : g.dy, local_grid->h_dzc[g.Hz],
                             zr ? nullptr : local_grid->h_dzc.data(), zr ? nullptr : local_grid->h_dzf.data());
    if (rc) return rc;
    if ((rc = ocn_poisson_create(&q->solver, q->ggrid, -1))) return rc;
    int P[3];
    parent_size(q->ggrid->d, LOC_C, P);
    const size_t piece = (size_t)q->nmax * q->nymax * g.Nz;
    HIP_TRY(dev_alloc((void **)&q->loc, piece * sizeof(double)));
    HIP_TRY(dev_alloc((void **)&q->all, piece * (size_t)dm->dist->world * sizeof(double)));
    HIP_TRY(dev_alloc((void **)&q->gp, (size_t)P[0] * P[1] * P[2] * sizeof(double)));
    HIP_TRY(hipMemsetAsync(q->loc, 0, piece * sizeof(double), g_stream));
    HIP_TRY(hipMemsetAsync(q->all, 0, piece * (size_t)dm->dist->world * sizeof(double), g_stream));
    HIP_TRY(hipMemsetAsync(q->gp, 0, (size_t)P[0] * P[1] * P[2] * sizeof(double), g_stream));
    return OCN_OK;
}

// the pressure step without fills / copies between its stages (ocn_kernels.h "Round 3"): a partitioned (connected, Periodic, *) slab with one
// of the accelerated solvers -- the substructured ones (z Periodic: every field shares one parent shape, which the wrapped pack relies on)
// or the transposing ones (z Periodic or Bounded: the solution stays in their dense paired-column array)
static int dist_enable_fused_step(DistModel *dm, const DGrid &g, bool part) {
    if (!(g_dist_fused_step && part && g.ty == OCN_PERIODIC && g.Nx >= 2 && (g.tz == OCN_PERIODIC || g.tz == OCN_BOUNDED))) return OCN_OK;
    if (dm->solver->sub && (g.tz != OCN_PERIODIC || !(dm->solver->zfirst || dm->solver->xfast))) return OCN_OK;    // (the paired-column substructured layout keeps the unfused step)
    const size_t col = (size_t)g.Ny * g.Nz * sizeof(double);
    double **cb[4] = {&dm->cws, &dm->ces, &dm->cwr, &dm->cer};
    for (auto b : cb) {
        hipError_t e = dev_alloc((void **)b, col);
        if (e != hipSuccess) return fail((int)e, "dev_alloc(column buffers): %s", hipGetErrorString(e));
        hipMemsetAsync(*b, 0, col, g_stream);
    }
    dm->fused_step = true;
    return OCN_OK;
}

static int dist_model_create(ocn_model_t *model, ocn_grid_t local_grid, int ntracers, ocn_dist_t dist, double Lx_global, const int *local_sizes,
                             int global_x_topology, double Ly_global, const int *sizes_y, int global_y_topology) {
    NEED_INIT();
    if (!model || !local_grid || !dist) return fail(OCN_EINVAL, "NULL argument");
    const DGrid &g = local_grid->d;
    const bool pencil = dist->Ry > 1;
    const int Rx = dist->Rx, ix = dist->rank / dist->Ry;
    const bool part = Rx > 1 || dist->self_loop;         // the x direction is partitioned
    const bool bounded_x = global_x_topology == OCN_BOUNDED;
    // insert_connected_topology (distributed_grids.jl:339-346)
    const int expect = !part ? global_x_topology
                             : (!bounded_x ? OCN_CONNECTED : (ix == 0 ? OCN_RIGHT_CONNECTED : (ix == Rx - 1 ? OCN_LEFT_CONNECTED : OCN_CONNECTED)));
    if (bounded_x && dist->self_loop) return fail(OCN_EINVAL, "self_loop closes a Periodic direction");
    if (g.tx != expect)
        return fail(OCN_EINVAL, "rank %d of %d: the local grid's x topology code is %d, insert_connected_topology gives %d", dist->rank, dist->world, g.tx, expect);
    const int iy = dist->rank % dist->Ry, Ry = dist->Ry;
    const int expect_y = !pencil ? g.ty : (global_y_topology != OCN_BOUNDED ? OCN_CONNECTED
                                           : (iy == 0 ? OCN_RIGHT_CONNECTED : (iy == Ry - 1 ? OCN_LEFT_CONNECTED : OCN_CONNECTED)));
    if (pencil && g.ty != expect_y)
        return fail(OCN_EINVAL, "rank %d of %d: the local grid's y topology code is %d, insert_connected_topology gives %d", dist->rank, dist->world, g.ty, expect_y);
    if (!pencil && (g.ty == OCN_CONNECTED || g.ty == OCN_RIGHT_CONNECTED || g.ty == OCN_LEFT_CONNECTED))
        return fail(OCN_EINVAL, "a connected y direction needs Ry > 1 (ocn_dist_model_create_pencil)");
    if (pencil && dist->self_loop) return fail(OCN_EINVAL, "self_loop has one rank");
    int rc = model_create(model, local_grid, ntracers, /*with_solver=*/false);
    if (rc) return rc;
    ocn_model_s *m = *model;
    DistModel *dm = new DistModel();
    m->dm = dm;
    dm->dist = dist;
    auto bail = [&](int code) { ocn_model_destroy(m); *model = nullptr; return code; };
    for (int f = 0; f < m->nf; ++f) dm->slab_total += dist_slab(g, m->loc[f], g.Hx);
    double **bufs[4] = {&dm->ws, &dm->es, &dm->wr, &dm->er};
    for (auto b : bufs) {
        hipError_t e = dev_alloc((void **)b, dm->slab_total * sizeof(double));
        if (e != hipSuccess) return bail(fail((int)e, "dev_alloc(halo buffers): %s", hipGetErrorString(e)));
        hipMemsetAsync(*b, 0, dm->slab_total * sizeof(double), g_stream);
    }
    {
        int P[3];
        parent_size(g, LOC_C, P);
        const size_t bytes = (size_t)P[0] * P[1] * P[2] * sizeof(double);
        hipError_t e = dev_alloc((void **)&dm->p2, bytes);
        if (e != hipSuccess) return bail(fail((int)e, "dev_alloc(p2): %s", hipGetErrorString(e)));
        hipMemsetAsync(dm->p2, 0, bytes, g_stream);
    }
    bool irregular = false;
    if (local_sizes)
        for (int r = 0; r < Rx; ++r) irregular = irregular || local_sizes[r] != local_sizes[0];
    dm->bounded_x = bounded_x && part;
    dm->pencil = pencil;
    dm->plain_y = pencil || g.ty == OCN_PERIODIC;
    if (pencil) {
        size_t rows = 0;
        for (int f = 0; f < m->nf; ++f) rows += dist_rows(g, m->loc[f], g.Hy);
        double **yb[4] = {&dm->ss, &dm->ns, &dm->sr, &dm->nr};
        for (auto b : yb) {
            hipError_t e = dev_alloc((void **)b, rows * sizeof(double));
            if (e != hipSuccess) return bail(fail((int)e, "dev_alloc(y halo buffers): %s", hipGetErrorString(e)));
            hipMemsetAsync(*b, 0, rows * sizeof(double), g_stream);
        }
    }
    // Partition(Rx, Ry) pencils of a triply Periodic regular grid with equal blocks that the reference's transposes can move
    // (distributed_fft_based_poisson_solver.jl:213-226): its DistributedFFTBasedPoissonSolver (z / y / x transforms with two transposes
    // each way); every other pencil keeps the gathered solve
    if (pencil && g_dist_pencil_transposes && !irregular && global_x_topology == OCN_PERIODIC && global_y_topology == OCN_PERIODIC &&
        g.tz == OCN_PERIODIC && local_grid->z_regular) {
        bool equal_y = true;
        if (sizes_y)
            for (int r = 0; r < Ry; ++r) equal_y = equal_y && sizes_y[r] == sizes_y[0];
        const int N[3] = {g.Nx * Rx, g.Ny * Ry, g.Nz};
        if (equal_y && N[2] % Ry == 0 && N[1] % Rx == 0) {
            const double L[3] = {Lx_global, Ly_global, local_grid->L[2]};
            if ((rc = pencil_solve_create(&dm->ps, dist, N, L))) return bail(rc);
            return OCN_OK;
        }
    }
    if (irregular || dm->general()) {
        if (dist->self_loop) return bail(fail(OCN_EINVAL, "self_loop has one slab"));
        std::vector<int> equal((size_t)Rx, g.Nx);
        if ((rc = gathered_solve_create(dm, local_grid, Lx_global, local_sizes ? local_sizes : equal.data(), global_x_topology, Ly_global, sizes_y, global_y_topology)))
            return bail(rc);
        return OCN_OK;
    }
    if ((rc = ocn_dist_poisson_create(&dm->solver, local_grid, dist->world, dist->rank, Lx_global))) return bail(rc);
    size_t n = 0;
    ocn_dist_poisson_payload_size(dm->solver, &n);
    if (n) {
        dm->payload = n;
        hipError_t e = dev_alloc((void **)&dm->buf_a, 2 * n * sizeof(double));
        if (e == hipSuccess) e = dev_alloc((void **)&dm->buf_b, 2 * n * sizeof(double) * (size_t)dist->world);
        if (e != hipSuccess) return bail(fail((int)e, "dev_alloc(gather buffers): %s", hipGetErrorString(e)));
        hipMemsetAsync(dm->buf_a, 0, 2 * n * sizeof(double), g_stream);
        hipMemsetAsync(dm->buf_b, 0, 2 * n * sizeof(double) * (size_t)dist->world, g_stream);
        if ((rc = ocn_dist_poisson_set_gather_buffers(dm->solver, dm->buf_a, dm->buf_b))) return bail(rc);
        if ((rc = dist_enable_fused_step(dm, g, part))) return bail(rc);
    } else {
        ocn_dist_poisson_buffer_size(dm->solver, &n);
        dm->nbuf = n;
        hipError_t e = dev_alloc((void **)&dm->buf_a, 2 * n * sizeof(double));
        // one rank: the "transposes" are the identity -- alias the buffers instead of copying
        if (!part) dm->buf_b = dm->buf_a;
        else if (e == hipSuccess) e = dev_alloc((void **)&dm->buf_b, 2 * n * sizeof(double));
        if (e != hipSuccess) return bail(fail((int)e, "dev_alloc(transpose buffers): %s", hipGetErrorString(e)));
        hipMemsetAsync(dm->buf_a, 0, 2 * n * sizeof(double), g_stream);
        if (dm->buf_b != dm->buf_a) hipMemsetAsync(dm->buf_b, 0, 2 * n * sizeof(double), g_stream);
        if ((rc = ocn_dist_poisson_set_buffers(dm->solver, dm->buf_a, dm->buf_b))) return bail(rc);
        if ((rc = dist_enable_fused_step(dm, g, part))) return bail(rc);
    }
    return OCN_OK;
}

static int dist_model_set_option(ocn_model_s *m, const char *key, int value) {
    DistModel *dm = m->dm;
    if (!dm) return -1;
    if (!strcmp(key, "async_halos")) { dm->async_halos = value; return OCN_OK; }
    if (!strcmp(key, "thin_halos")) { dm->thin_halos = value; return OCN_OK; }
    if (!strcmp(key, "early_exchange")) { dm->early_exchange = value; return OCN_OK; }
    if (!strcmp(key, "strip_width")) { dm->strip_width = value; return OCN_OK; }
    if (!strcmp(key, "fused_step")) {
        if (value && !dm->cws) return fail(OCN_ENOTSUP, "fused_step needs a (connected, Periodic, Periodic) slab with the z-fastest substructured solver");
        dm->fused_step = value != 0;
        return OCN_OK;
    }
    return -1;
}

static int dist_model_get_option(const ocn_model_s *m, const char *key, int *value) {
    const DistModel *dm = m->dm;
    if (!dm) return -1;
    // which distributed pressure solver the model runs: ocn_dist_poisson_layout's code (4 x-fastest, 1..3 z-fastest, 0 paired columns,
    // -1 transposing), -2 = the gathered solve on the global grid, -3 = the pencil transposes (TransposableField)
    if (!strcmp(key, "dist_poisson_layout")) { *value = dm->gs ? -2 : (dm->ps ? -3 : -1); return dm->solver ? ocn_dist_poisson_layout(dm->solver, value) : OCN_OK; }
    if (!strcmp(key, "fused_step")) { *value = dm->fused() ? 1 : 0; return OCN_OK; }
    if (!strcmp(key, "async_halos")) { *value = dm->async_halos; return OCN_OK; }
    if (!strcmp(key, "thin_halos")) { *value = dm->thin_halos; return OCN_OK; }
    if (!strcmp(key, "early_exchange")) { *value = dm->early_exchange; return OCN_OK; }
    return -1;
}

// global maximum of |div u| (test helper of the partitioned model)
extern "C" int ocn_dist_model_max_abs_divergence(ocn_model_t m, double *value) {
    NEED_INIT();
    if (!m || !m->dm || !value) return fail(OCN_EINVAL, "not a distributed model");
    int rc = dist_fill_halo_regions(m, m->U, m->loc, 3, true, m->any_bc ? m->bcs : nullptr);
    if (rc) return rc;
    if ((rc = ocn_max_abs_divergence(m->grid, m->U[0], m->U[1], m->U[2], value))) return rc;
    return ocn_dist_allreduce_max(m->dm->dist, value);
}
