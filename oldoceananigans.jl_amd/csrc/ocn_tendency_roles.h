// ocn_tendency_roles.h -- flux-sharing WENO-5 tendency kernel, one FIELD per workgroup ("roles").
//
// Same contract and the same IEEE operation sequence per flux as ocn_tendency_fused.h (compute_nonhydrostatic_tendencies.jl:49-163,
// upwind_biased_advective_fluxes.jl:23-121): every face flux is evaluated once and handed to the neighbouring cell, a workgroup owns a
// 64 x TY column tile and marches along z with the z-window of its field in registers, low-side x / y fluxes go through LDS.
//
// What changes is the decomposition of the work. The all-fields kernel keeps the z-windows of 5 fields (60 VGPRs) plus the state
// of 15 reconstructions per thread: 206 VGPRs, 2 waves per SIMD, 78 spilled SGPRs (22 FP64 constants + 44 pointers). Here a
// workgroup evaluates the three fluxes of ONE field (role u, v, w or tracer t) on its tile: one 6-deep z-window (two more 4-deep
// ones for w), 3 reconstructions per cell and plane, < 128 VGPRs => two 8-wave workgroups per CU, 4 waves per SIMD, 15.6 KB of
// LDS per workgroup and a hot loop of ~5 KB of code per role. The (3 + ntracers) role workgroups of one (tile, z-chunk) pair are
// dispatched back to back onto the same XCD (blocks b and b + 8 share an XCD under the observed round-robin placement; a pure
// speed choice), so the velocity planes all of them read are fetched into that XCD's L2 once.
//
// Addressing: a plane's base address is a scalar (SGPR pair, advanced by one plane per iteration on the scalar unit); a thread
// carries loop-invariant 32-bit byte offsets of its six y-window rows; x neighbours are immediate offsets. No vector instruction is
// spent on addresses inside the loop.
#pragma once
#include "ocn_tendency_fused.h"

#ifndef OCN_ROLE_WAVES
#define OCN_ROLE_WAVES 6      // waves per SIMD the register allocation must allow (8-wave workgroups: 4 = two per CU, 6 = three)
#endif
#ifndef OCN_ROLE_TY
#define OCN_ROLE_TY 7         // rows of a tile = row waves of a workgroup (+ 1 edge wave)
#endif
#ifndef OCN_ROLE_PF
#define OCN_ROLE_PF 1         // planes the two cold streams (new z-window level, previous tendency) are fetched ahead of their use
#endif
#ifndef OCN_ROLE_ABLATE
#define OCN_ROLE_ABLATE 0     // timing experiments only (WRONG RESULTS when non-zero): 1 hot z-window load, 2 no previous-tendency load,
#endif                        // 4 no stores, 8 x / y windows from registers, 16 no barrier, 32 idle edge wave
#ifndef OCN_ROLE_STORE_AUX
#define OCN_ROLE_STORE_AUX 2   // cache policy bits of the tendency / next-stage stores (1 sc0, 2 nt, 16 sc1)
#endif
#ifndef OCN_ROLE_GM_AUX
#define OCN_ROLE_GM_AUX 0      // cache policy bits of the previous-tendency load (read once)
#endif
enum { ROLE_U = 0, ROLE_V = 1, ROLE_W = 2, ROLE_C = 3 };

template <int NF>
struct RoleArgs {
    const double *U[NF];       // u, v, w, tracers
    double *G[NF];
    double *Un[NF];            // fused RK3 substep of the next stage (see FusedArgs)
    const double *Gm[NF];
    int s1;                    // x-row stride (elements)
    unsigned s2;               // plane stride (elements)
    long off;                  // (Hx-1) + s1 (Hy-1)
    Range6 r;                  // cell range of the launch
    int wk0;                   // first level whose w tendency is stored (exclude_periphery on Bounded z)
    int kchunk, ntile_x, ntile, npair, band;
    int has_zeta;
    int store_G;               // 0: the tendency is consumed by the fused substep only (FusedSubstep::store_G)
    double dt, gamma, zeta;
};

// Buffer addressing (MUBUF): descriptor of the whole parent array (4 SGPRs) + scalar plane offset + per-thread 32-bit byte offset
// + 12-bit immediate: buffer_load_dwordx2 v, voff, s[rsrc], soff offen offset:IMM. The per-thread offsets are biased by -24 B so
// that the six x-neighbours are the immediates 0 .. 40.
typedef __amdgpu_buffer_rsrc_t Rsrc;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
// num_records = 2 GiB (parent arrays are smaller: checked on the host): a per-thread offset with bit 31 set is OUT OF RANGE, the
// hardware drops such a store and returns 0 for such a load. Lanes that must not access memory get that bit (ROLE_OOB) instead
// of a branch around the instruction -- the instruction stream between two waits stays free of control flow, which is what lets
// the compiler count the younger operations exactly (s_waitcnt vmcnt(n)).
// per-level metric tables are read-only for the kernel's lifetime: addressed through the constant address space, their
// (wave-uniform) elements come in through the scalar cache instead of as vector loads that occupy VGPRs and the vmcnt queue
typedef const double __attribute__((address_space(4))) *KTab;
__device__ __forceinline__ KTab ktab(const double *p) { return (KTab)p; }
#define ROLE_OOB 0x80000000u
__device__ __forceinline__ Rsrc make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)ROLE_OOB, 0x00020000);   // raw buffer
}
template <int IMM> __device__ __forceinline__ double ldb(Rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(voff + (unsigned)IMM), (int)soff, 0));
}
template <int IMM> __device__ __forceinline__ double ldb_once(Rsrc r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)(voff + (unsigned)IMM), (int)soff, OCN_ROLE_GM_AUX));
}
template <int IMM> __device__ __forceinline__ void stb(Rsrc r, unsigned voff, unsigned soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, (int)(voff + (unsigned)IMM), (int)soff, OCN_ROLE_STORE_AUX);
}

// per-thread loop-invariant offset: c = byte offset of (i - 3, j) inside a plane; sr[n] = scalar byte offset of row j - 3 + n
// relative to row j, folded into the scalar offset of the access (n = 3 is the own row). The per-thread state of all six rows of a
// y-window is ONE VGPR; the row displacement rides on the scalar unit.
struct ColOff { unsigned c; unsigned s1; };     // s1 = row stride in bytes (wave-uniform)

// x-direction windows around column i: immediates 0 .. 40 (centre 24)
template <bool OWN> __device__ __forceinline__ Win6 xwin6(Rsrc b, unsigned so, const ColOff &o, double centre) {
    Win6 w;
    w.s[0] = ldb<0>(b, o.c, so); w.s[1] = ldb<8>(b, o.c, so); w.s[2] = ldb<16>(b, o.c, so);
    w.s[3] = OWN ? centre : ldb<24>(b, o.c, so);
    w.s[4] = ldb<32>(b, o.c, so); w.s[5] = ldb<40>(b, o.c, so);
    return w;
}
__device__ __forceinline__ Win6 xwin4(Rsrc b, unsigned so, const ColOff &o) {
    Win6 w;
    w.s[0] = 0.0; w.s[5] = 0.0;
    w.s[1] = ldb<8>(b, o.c, so); w.s[2] = ldb<16>(b, o.c, so); w.s[3] = ldb<24>(b, o.c, so); w.s[4] = ldb<32>(b, o.c, so);
    return w;
}
template <bool OWN> __device__ __forceinline__ Win6 ywin6(Rsrc b, unsigned so, const ColOff &o, double centre) {
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = (OWN && n == 3) ? centre : ldb<24>(b, o.c, so + (unsigned)(n - 3) * o.s1);
    return w;
}
__device__ __forceinline__ Win6 ywin4(Rsrc b, unsigned so, const ColOff &o) {
    Win6 w;
    w.s[0] = 0.0; w.s[5] = 0.0;
#pragma unroll
    for (int n = 1; n < 5; ++n) w.s[n] = ldb<24>(b, o.c, so + (unsigned)(n - 3) * o.s1);
    return w;
}
// z-direction 4-window (levels k-2 .. k+1) straight from memory (edge wave of role w only)
__device__ __forceinline__ Win6 zwin4(Rsrc b, unsigned so, const ColOff &o, unsigned s2) {
    Win6 w;
    w.s[0] = 0.0; w.s[5] = 0.0;
    w.s[1] = ldb<24>(b, o.c, so - 2u * s2); w.s[2] = ldb<24>(b, o.c, so - s2); w.s[3] = ldb<24>(b, o.c, so); w.s[4] = ldb<24>(b, o.c, so + s2);
    return w;
}

// per-plane scalar state of a workgroup
struct PlaneCtx {
    Rsrc u, v, w, q;           // u, v, w and the role's own field
    unsigned so;               // byte offset of plane k
    unsigned s2;               // plane stride in bytes
    double axk, ayk;
    const double *axz, *ayz;   // Ax, Ay at levels k-2 ..
};

// ---- loads and arithmetic of the three low-side fluxes of field ROLE at (i, j, k) ----
// z-flux inputs besides the own z-window: the advecting w along x (role u) / along y (role v), indices 1 .. 4; w at the face (tracers)
template <int ROLE> __device__ __forceinline__ Win6 load_zin(const PlaneCtx &p, unsigned so, const ColOff &o) {
    if (ROLE == ROLE_U) return xwin4(p.w, so, o);
    if (ROLE == ROLE_V) return ywin4(p.w, so, o);
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = 0.0;
    if (ROLE == ROLE_C) w.s[3] = ldb<24>(p.w, o.c, so);
    return w;
}
template <int ROLE, int ARITH>
__device__ __forceinline__ double z_flux(const DGrid &g, const Win6 &zin, int i, int j, int k, const Win6 &qz) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    const double az = g.az;
    if (ROLE == ROLE_U) {
        const double wt = sym4<ARITH>(zin, az, bx, i, false, g.Nx);                         // advective_momentum_flux_Wu :39-45
        return wt * bias6<ARITH>(qz, wt > 0, bz, k, false, g.Nz);
    } else if (ROLE == ROLE_V) {
        const double wt = sym4<ARITH>(zin, az, by, j, false, g.Ny);                         // Wv :63-69
        return wt * bias6<ARITH>(qz, wt > 0, bz, k, false, g.Nz);
    } else if (ROLE == ROLE_W) {
        const double wt = sym4<ARITH>(qz, az, bz, k - 1, true, g.Nz);                       // Ww :87-93
        return wt * bias6<ARITH>(qz, wt > 0, bz, k - 1, true, g.Nz);
    } else {
        const double w0 = zin.s[3];                                                  // advective_tracer_flux_z :115-121
        return az * w0 * bias6<ARITH>(qz, w0 > 0, bz, k, false, g.Nz);
    }
}
// x-flux: own field along x (qx) + the advecting u: along y (role v, indices 1 .. 4), along z (role w), at the face (tracers)
template <int ROLE> __device__ __forceinline__ Win6 load_xaux(const PlaneCtx &p, const ColOff &o) {
    if (ROLE == ROLE_V) return ywin4(p.u, p.so, o);
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = 0.0;
    if (ROLE == ROLE_C) w.s[3] = ldb<24>(p.u, o.c, p.so);
    return w;
}
template <int ROLE, int ARITH>
__device__ __forceinline__ double x_flux(const DGrid &g, const PlaneCtx &p, const Win6 &qx, const Win6 &aux, int i, int j, int k) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    if (ROLE == ROLE_U) {
        const double ut = sym4<ARITH>(qx, p.axk, bx, i - 1, true, g.Nx);                    // advective_momentum_flux_Uu :23-29
        return ut * bias6<ARITH>(qx, ut > 0, bx, i - 1, true, g.Nx);
    } else if (ROLE == ROLE_V) {
        const double ut = sym4<ARITH>(aux, p.axk, by, j, false, g.Ny);                      // Uv :47-53
        return ut * bias6<ARITH>(qx, ut > 0, bx, i, false, g.Nx);
    } else if (ROLE == ROLE_W) {
        const double ut = sym4z(aux, p.axz, bz, k, false, g.Nz);                     // Uw :71-77
        return ut * bias6<ARITH>(qx, ut > 0, bx, i, false, g.Nx);
    } else {
        const double u0 = aux.s[3];                                                  // advective_tracer_flux_x :99-105
        return p.axk * u0 * bias6<ARITH>(qx, u0 > 0, bx, i, false, g.Nx);
    }
}
// y-flux: own field along y (qy) + the advecting v: along x (role u), along z (role w), at the face (tracers)
template <int ROLE> __device__ __forceinline__ Win6 load_yaux(const PlaneCtx &p, const ColOff &o) {
    if (ROLE == ROLE_U) return xwin4(p.v, p.so, o);
    Win6 w;
#pragma unroll
    for (int n = 0; n < 6; ++n) w.s[n] = 0.0;
    if (ROLE == ROLE_C) w.s[3] = ldb<24>(p.v, o.c, p.so);
    return w;
}
template <int ROLE, int ARITH>
__device__ __forceinline__ double y_flux(const DGrid &g, const PlaneCtx &p, const Win6 &qy, const Win6 &aux, int i, int j, int k) {
    const bool bx = g.tx != 0, by = g.ty != 0, bz = g.tz != 0;
    if (ROLE == ROLE_U) {
        const double vt = sym4<ARITH>(aux, p.ayk, bx, i, false, g.Nx);                      // advective_momentum_flux_Vu :31-37
        return vt * bias6<ARITH>(qy, vt > 0, by, j, false, g.Ny);
    } else if (ROLE == ROLE_V) {
        const double vt = sym4<ARITH>(qy, p.ayk, by, j - 1, true, g.Ny);                    // Vv :55-61
        return vt * bias6<ARITH>(qy, vt > 0, by, j - 1, true, g.Ny);
    } else if (ROLE == ROLE_W) {
        const double vt = sym4z(aux, p.ayz, bz, k, false, g.Nz);                     // Vw :79-85
        return vt * bias6<ARITH>(qy, vt > 0, by, j, false, g.Ny);
    } else {
        const double v0 = aux.s[3];                                                  // advective_tracer_flux_y :107-113
        return p.ayk * v0 * bias6<ARITH>(qy, v0 > 0, by, j, false, g.Ny);
    }
}

// Workgroup barrier for the LDS flux exchange only: __syncthreads() carries a release fence that drains the vector-memory queue
// (s_waitcnt vmcnt(0)) -- exactly the stores and prefetches that are meant to stay in flight across it. Nothing the workgroup's
// waves exchange goes through global memory, so the LDS counter is all this barrier has to wait for.
__device__ __forceinline__ void vmov(double &dst, const double &src) { asm volatile("v_mov_b64 %0, %1" : "=v"(dst) : "v"(src)); }
__device__ __forceinline__ void role_barrier() {
    if (!(OCN_ROLE_ABLATE & 16)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One workgroup = TY row waves + one edge wave, all working on field `fidx`. A row wave owns one row of the tile (two rows per wave
// were implemented and measured slower: 1.70 vs 1.52 ms); the edge wave evaluates the y-fluxes of the row above the tile and the
// x-fluxes of the column right of it (one lane per row).
//
// Order of the memory operations of a row wave within one plane. gfx950 counts loads AND stores with one in-order counter (vmcnt):
// a wait for any load also waits for every older operation. The operations with long latencies -- the first touch of plane k + 3
// of the own field, the previous tendency, the stores of the cell just closed -- are therefore issued together as the YOUNGEST
// operations of the plane, right behind the loads of the y-window, so that no wait of this plane includes them: they have the
// arithmetic of the y-flux, the barrier and the first loads of the next plane to complete. The new element of the z-window
// arrives one plane ahead of its use (qn), the previous tendency of the cell closed in the next plane likewise (gmn).
template <int ROLE, int TY, bool SUB, bool BZ, int ARITH, typename Args>
__device__ __forceinline__ void role_march(const DGrid &g, const Args &a, const int fidx, const int i0, const int j0, const int kc0,
                                           const int kc1, double (*FX)[TY][66], double (*FY)[TY + 1][64]) {
    // Bounded z: only the planes within reach of a wall need the fallback logic of the scheme (every z-stencil test of
    // topologically_conditional_interpolation.jl:46-52 holds for 4 <= k <= Nz - 2). Those planes run a copy of the plane body compiled
    // with the wall logic; all others run the copy compiled for a Periodic z -- the same operations on the same operands -- so the hot
    // loop of a Bounded-z launch is the Periodic one (it was 6.5 % slower per cell: more code in the loop, conditions per plane).
    DGrid gP = g, gB = g;
    gP.tz = 0; gB.tz = BZ ? 1 : 0;
    auto near_wall = [&](int k) { return BZ && (k < 4 || k > g.Nz - 2); };
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool edge = wave == TY;
    const int i = i0 + lane;
    const int row0 = wave;                            // tile row of this wave (edge wave: the row above the tile)
    const int j = j0 + row0;
    const bool flux_r = !edge && i <= a.r.i1 + 1 && j <= a.r.j1 + 1, cell_r = !edge && i <= a.r.i1 && j <= a.r.j1;
    const bool edge_y = edge && i <= a.r.i1 && j <= a.r.j1 + 1;
    const int ie = i0 + 64, je = j0 + lane;
    const bool edge_x = edge && lane < TY && ie <= a.r.i1 + 1 && je <= a.r.j1;

    const unsigned s2 = 8u * a.s2;                    // plane stride in bytes
    const int Hz = g.Hz;
    ColOff o, e;                                      // (i - 3, j); the edge wave's second job: column i0 + 64 of row j0 + lane
    o.s1 = e.s1 = 8u * (unsigned)a.s1;
    o.c = 8u * (unsigned)(a.off + i - 3 + (long)a.s1 * j);
    e.c = 8u * (unsigned)(a.off + ie - 3 + (long)a.s1 * je);

    // The descriptors of a workgroup start at the lowest plane it can touch (level kc0 - 3), not at the array: every offset below is
    // relative to that plane and stays under 2^31 for parent arrays of any size (role_tendency_supported: (kchunk + 8) planes).
    const long pb = max(0L, (long)(kc0 - 1 + Hz) - 3);
    const long sh = pb * (long)a.s2;
    PlaneCtx p;
    p.u = make_rsrc(a.U[0] + sh); p.v = make_rsrc(a.U[1] + sh); p.w = make_rsrc(a.U[2] + sh);
    p.q = ROLE == ROLE_U ? p.u : (ROLE == ROLE_V ? p.v : (ROLE == ROLE_W ? p.w : make_rsrc(a.U[fidx] + sh)));
    p.s2 = s2;
    p.so = s2 * (unsigned)((long)(kc0 - 1 + Hz) - pb);   // plane of level kc0
    // the streams touched once per cell: tendency out, previous tendency in (a.Gm is the tendency array itself when the substep has
    // no zeta: loaded and not used), next-stage field out
    const Rsrc rG = make_rsrc(a.G[fidx] + sh), rUn = make_rsrc((SUB ? a.Un[fidx] : a.G[fidx]) + sh), rGm = make_rsrc((SUB ? a.Gm[fidx] : a.G[fidx]) + sh);
    const unsigned cell_off = cell_r ? 0u : ROLE_OOB;
    const unsigned g_oob = (SUB && !a.store_G) ? ROLE_OOB : 0u;

    double fz_prev = 0, qn = 0, gmn = 0;
    double qnn = 0, gmnn = 0;                         // OCN_ROLE_PF == 2: the same streams one more plane ahead
    Win6 qz;                                          // own 6-deep z-window (role w reads the z-windows of u, v from memory:
                                                      // kept in registers they cost 20 VGPRs and spilled)
#pragma unroll
    for (int n = 0; n < 6; ++n) qz.s[n] = 0;
    if (flux_r) {
#pragma unroll
        for (int n = 0; n < 5; ++n) qz.s[n + 1] = ldb<24>(p.q, o.c, p.so + (unsigned)(n - 3) * s2);   // levels kc0-3 .. kc0+1
        qn = ldb<24>(p.q, o.c, p.so + 2u * s2);                                                        // level kc0+2
        if (OCN_ROLE_PF == 2) qnn = ldb<24>(p.q, o.c, p.so + 3u * s2);                                 // level kc0+3 (inside the halo)
        if (SUB && OCN_ROLE_PF == 2 && !edge) gmnn = ldb_once<24>(rGm, o.c | cell_off, p.so);          // level kc0: closed in plane kc0+1
    }

    // The edge wave and the row waves run SEPARATE loops over the same planes (one barrier per plane in each): loop-carried values
    // of one kind of wave never meet the other's in a control-flow join, where the register allocator would place copies that
    // wait for the prefetches.
    if (edge) {
        for (int k = kc0; k <= kc1; ++k) {
            const int buf = k & 1;
            const long pk = (long)(k - 1 + Hz);
            p.axk = ktab(g.ax)[pk]; p.ayk = ktab(g.ay)[pk];
            p.axz = g.ax + pk - 2; p.ayz = g.ay + pk - 2;
            asm volatile("" : "+v"(o.c));
            auto edge_plane = [&](const DGrid &gg) {
                if (edge_y && !(OCN_ROLE_ABLATE & 32)) {
                    const Win6 qy = ywin6<false>(p.q, p.so, o, 0.0);
                    const Win6 aux = ROLE == ROLE_W ? zwin4(p.v, p.so, o, s2) : load_yaux<ROLE>(p, o);
                    FY[buf][TY][lane] = y_flux<ROLE, ARITH>(gg, p, qy, aux, i, j, k);
                }
                if (edge_x && !(OCN_ROLE_ABLATE & 32)) {
                    asm volatile("" : "+v"(e.c));
                    const Win6 qxe = xwin6<false>(p.q, p.so, e, 0.0);
                    const Win6 aux = ROLE == ROLE_W ? zwin4(p.u, p.so, e, s2) : load_xaux<ROLE>(p, e);
                    FX[buf][lane][64] = x_flux<ROLE, ARITH>(gg, p, qxe, aux, ie, je, k);
                }
            };
            if (ROLE == ROLE_W && near_wall(k)) edge_plane(gB); else edge_plane(gP);     // only role w interpolates along z here
            p.so += s2;
            role_barrier();
        }
        return;
    }
    if (j > a.r.j1 + 1) {                             // a row wave entirely above the range (wave-uniform): barriers only
        for (int k = kc0; k <= kc1; ++k) role_barrier();
        return;
    }
    // Lanes right of the range get the out-of-range bit instead of a branch: their loads return 0, their stores are dropped,
    // their fluxes land in LDS columns no cell reads. The plane loop below has no divergent control flow around loop-carried values.
    if (!flux_r) o.c |= ROLE_OOB;

    // z-flux through the bottom face of cell k, then the closing of cell k - 1 (tendency + the next stage's substep): values only,
    // the stores are placed by the caller
    double Gn = 0, Uv = 0;
    bool store = false;
    double gm = 0;
    Win6 zin;
    auto z_loads = [&]() {
        // The z-windows slide by one level, the prefetched level enters at the top. Written as explicit moves at THIS point of the
        // instruction stream (behind the barrier): left to the register allocator the rotation of the loop-carried window is
        // resolved at the end of the previous plane, where the copy of the prefetched value waits for it (vmcnt(0)) before the barrier.
#pragma unroll
        for (int n = 0; n < 5; ++n) vmov(qz.s[n], qz.s[n + 1]);
        vmov(qz.s[5], qn);
        if (OCN_ROLE_PF == 2) vmov(qn, qnn);
        if (SUB) vmov(gm, gmn);
        if (SUB && OCN_ROLE_PF == 2) vmov(gmn, gmnn);
        zin = load_zin<ROLE>(p, p.so, o);
    };
    auto z_compute = [&](const DGrid &gg, const int k, const int buf, const long pk) {
        const double fz = z_flux<ROLE, ARITH>(gg, zin, i, j, k, qz);
        const int pb = buf ^ 1;
        const long pkm = pk - 1;
        const double vinv = ROLE == ROLE_W ? ktab(g.vinv_f)[pkm] : ktab(g.vinv_c)[pkm];
        const double dx = FX[pb][row0][lane + 1] - FX[pb][row0][lane];
        const double dy = FY[pb][row0 + 1][lane] - FY[pb][row0][lane];
        const double div = vinv * ((dx + dy) + (fz - fz_prev));
        Gn = -div + 0.0;
        store = cell_r && k > kc0 && (ROLE != ROLE_W || k - 1 >= a.wk0) && (!(OCN_ROLE_ABLATE & 4) || Gn == 1.2345);
        if (SUB) {
            // rk3_substep_field! of the next stage on the cell just closed (runge_kutta_3.jl:212-226)
            Uv = qz.s[2];
            if (a.has_zeta) Uv += a.dt * (a.gamma * Gn + a.zeta * gm);
            else            Uv += a.dt * a.gamma * Gn;
        }
        fz_prev = fz;
    };
    auto stores = [&]() {
        const unsigned st = o.c | (store ? 0u : ROLE_OOB);
        stb<24>(rG, st | g_oob, (OCN_ROLE_ABLATE & 64) ? s2 * 3u : p.so - s2, Gn);                     // level k - 1 (dropped by the hardware when nothing reads it)
        if (SUB) stb<24>(rUn, st, (OCN_ROLE_ABLATE & 64) ? s2 * 3u : p.so - s2, Uv);
    };

    for (int k = kc0; k <= kc1; ++k) {
        const int buf = k & 1;
        const long pk = (long)(k - 1 + Hz);
        p.axk = ktab(g.ax)[pk]; p.ayk = ktab(g.ay)[pk];
        p.axz = g.ax + pk - 2; p.ayz = g.ay + pk - 2;
        // keep the 32-bit offsets opaque per iteration, so that `offset + constant` stays inside the loop and folds into the immediate
        asm volatile("" : "+v"(o.c));
        // Issue order of a plane: [z inputs] z-flux + closing arithmetic [x-window, stores] x-flux [y-window, prefetches] y-flux,
        // barrier. (Issuing the x- / y-window loads one flux earlier was tried: 16 more VGPRs, no gain at six waves per SIMD.)
        z_loads();
        Win6 qx, xaux, qy, yaux;
        auto x_loads = [&]() {
            qx = (OCN_ROLE_ABLATE & 8) ? qz : xwin6<true>(p.q, p.so, o, qz.s[3]);
            xaux = ROLE == ROLE_W ? zwin4(p.u, p.so, o, s2) : load_xaux<ROLE>(p, o);
        };
        auto y_loads = [&]() {
            qy = (OCN_ROLE_ABLATE & 8) ? qz : ywin6<true>(p.q, p.so, o, qz.s[3]);
            yaux = ROLE == ROLE_W ? zwin4(p.v, p.so, o, s2) : load_yaux<ROLE>(p, o);
        };
        auto prefetches = [&]() {
            if (OCN_ROLE_PF == 2) {
                // level k + 4 exists only while k + 1 <= kc1 can still use it (k + 4 <= Nz + Hz); the last plane re-reads level k + 3
                qnn = ldb<24>(p.q, o.c, (OCN_ROLE_ABLATE & 1) ? p.so : p.so + (k < kc1 ? 4u : 3u) * s2);
                if (SUB && !(OCN_ROLE_ABLATE & 2)) gmnn = ldb_once<24>(rGm, o.c | cell_off, p.so + s2);   // level k + 1: closed in plane k + 2
                return;
            }
            qn = ldb<24>(p.q, o.c, (OCN_ROLE_ABLATE & 1) ? p.so : p.so + 3u * s2);
            if (SUB && !(OCN_ROLE_ABLATE & 2)) gmn = ldb_once<24>(rGm, o.c | cell_off, p.so);          // level k: closed in plane k + 1
        };
        auto plane = [&](const DGrid &gg) {
            __builtin_amdgcn_sched_barrier(0);
            z_compute(gg, k, buf, pk);
            __builtin_amdgcn_sched_barrier(0);
            x_loads();
            stores();
            __builtin_amdgcn_sched_barrier(0);
            FX[buf][row0][lane] = x_flux<ROLE, ARITH>(gg, p, qx, xaux, i, j, k);
            __builtin_amdgcn_sched_barrier(0);
            y_loads();
            prefetches();
            __builtin_amdgcn_sched_barrier(0);
            FY[buf][row0][lane] = y_flux<ROLE, ARITH>(gg, p, qy, yaux, i, j, k);
        };
        if (near_wall(k)) plane(gB); else plane(gP);
        p.so += s2;
        role_barrier();
    }
    // peeled plane kc1 + 1: only the z-fluxes that close the cells of level kc1
    {
        const int k = kc1 + 1;
        z_loads();
        if (near_wall(k)) z_compute(gB, k, k & 1, (long)(k - 1 + Hz)); else z_compute(gP, k, k & 1, (long)(k - 1 + Hz));
        stores();
    }
}

template <int NTR, int TY, bool BZ, bool SUB, int ARITH = 0>
__global__ void __launch_bounds__(64 * (TY + 1), OCN_ROLE_WAVES) role_tendency_kernel(DGrid gin, RoleArgs<3 + NTR> a) {
    constexpr int NF = 3 + NTR;
    __shared__ double FX[2][TY][66];                // low-side x-fluxes of columns 0..64 (65 used, padded)
    __shared__ double FY[2][TY + 1][64];            // low-side y-fluxes of rows 0..TY R

    DGrid g = gin;
    g.tx = 0; g.ty = 0; g.tz = BZ ? 1 : 0;          // compile-time topology (x, y Periodic / FullyConnected is a launch precondition)

    // Blocks d and d + 8 share an XCD (observed round-robin placement). Each XCD works through a contiguous band of (tile, chunk)
    // pairs -- neighbouring tiles, whose halo columns / rows overlap, meet in one L2 -- and the NF roles of a pair take
    // consecutive slots of that XCD, so the planes all roles read arrive in that L2 once.
    const unsigned d = blockIdx.x, xcd = d & 7u, slot = d >> 3;
    const unsigned role = slot % NF, pair = xcd * (unsigned)a.band + slot / NF;
    if (slot / NF >= (unsigned)a.band || pair >= (unsigned)a.npair) return;
    const unsigned tile = pair % (unsigned)a.ntile, chunk = pair / (unsigned)a.ntile;
    const int i0 = a.r.i0 + (int)(tile % (unsigned)a.ntile_x) * 64, j0 = a.r.j0 + (int)(tile / (unsigned)a.ntile_x) * TY;
    const int kc0 = a.r.k0 + (int)chunk * a.kchunk;
    const int kc1 = min(kc0 + a.kchunk - 1, a.r.k1);
    if (role == 0) role_march<ROLE_U, TY, SUB, BZ, ARITH>(g, a, 0, i0, j0, kc0, kc1, FX, FY);
    else if (role == 1) role_march<ROLE_V, TY, SUB, BZ, ARITH>(g, a, 1, i0, j0, kc0, kc1, FX, FY);
    else if (role == 2) role_march<ROLE_W, TY, SUB, BZ, ARITH>(g, a, 2, i0, j0, kc0, kc1, FX, FY);
    else role_march<ROLE_C, TY, SUB, BZ, ARITH>(g, a, (int)role, i0, j0, kc0, kc1, FX, FY);
}

// The role kernel addresses memory with a 31-bit byte offset (bit 31 is the out-of-range flag of its buffer descriptors, whose num_records
// is 2 GiB) relative to the lowest plane of the workgroup's chunk: (kchunk + 3 below + 4 of look-ahead + 1) planes + the row / column offsets
// must stay below 2^31 -- a limit on the PLANE size (29 MB at the largest chunk: 1900 x 1900 points), not on the array (a single-GPU
// 1024 x 1024 x 256 grid, 2.2 GB per field, took the slower all-fields kernel before round 3's per-workgroup descriptor base).
static int g_role_kchunk = 0;      // 0: automatic
static inline bool role_tendency_supported(const DGrid &g) {
    const double plane = 8.0 * (g.Nx + 2.0 * g.Hx) * (g.Ny + 2.0 * g.Hy);
    return plane * ((g_role_kchunk > 0 ? g_role_kchunk : 64) + 8.0) < 2147483648.0;
}

static int g_arithmetic = 0;       // 0: the reference's operation sequence (bit-identical to the oracle); 1: contracted WENO flux (ocn_device.h)
static int g_role_ldspad = 0;      // experiments: extra dynamic LDS per workgroup (bytes) to limit the workgroups per CU

// Work per launch in plane-iterations: blocks x (kchunk + 1) spread over the resident workgroups (3 per CU at <= 80 VGPRs); smaller chunks balance the
// tail, every chunk pays one extra z-flux plane and the window priming loads.
static inline int pick_role_kchunk(long tiles_roles, int nz) {
    int best = nz;
    double best_cost = -1;
    const long slots = (long)((4 * OCN_ROLE_WAVES) / (OCN_ROLE_TY + 1)) * g_num_cus;      // resident workgroups of the chip
    for (int kc = 8; kc <= 64; ++kc) {
        if (kc > nz && kc != 8) break;
        const int nchunk = (nz + kc - 1) / kc;
        const int kce = (nz + nchunk - 1) / nchunk;            // balanced chunk length
        const long blocks = tiles_roles * nchunk;
        const double per_block = kce + 1.5;
        const double cost = (double)blocks * per_block / slots + per_block;   // perfectly spread work + one block of tail
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = kce; }
    }
    return best;
}

template <int NTR, int TY>
static int launch_roles_t(const DGrid &g, hipStream_t stream, RoleArgs<3 + NTR> &a, bool sub) {
    constexpr int NF = 3 + NTR;
    const int nx = a.r.i1 - a.r.i0 + 1, ny = a.r.j1 - a.r.j0 + 1, nz = a.r.k1 - a.r.k0 + 1;
    if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
    a.ntile_x = (nx + 63) / 64;
    a.ntile = a.ntile_x * ((ny + TY - 1) / TY);
    a.kchunk = g_role_kchunk > 0 ? g_role_kchunk : pick_role_kchunk((long)a.ntile * NF, nz);
    const int nchunk = (nz + a.kchunk - 1) / a.kchunk;
    a.npair = a.ntile * nchunk;
    a.band = (a.npair + 7) / 8;
    const unsigned nblocks = (unsigned)a.band * 8u * NF;
    const dim3 blk(64 * (TY + 1));
#define OCN_LAUNCH_ROLES(BZV, SUBV, AR) hipLaunchKernelGGL((role_tendency_kernel<NTR, TY, BZV, SUBV, AR>), dim3(nblocks), blk, (size_t)g_role_ldspad, stream, g, a)
    if (g_arithmetic == 1) {          // the opt-in contracted arithmetic of the WENO flux (ocn_device.h; option "arithmetic")
        if (g.tz != 0) { if (sub) OCN_LAUNCH_ROLES(true, true, 1); else OCN_LAUNCH_ROLES(true, false, 1); }
        else           { if (sub) OCN_LAUNCH_ROLES(false, true, 1); else OCN_LAUNCH_ROLES(false, false, 1); }
    } else {
        if (g.tz != 0) { if (sub) OCN_LAUNCH_ROLES(true, true, 0); else OCN_LAUNCH_ROLES(true, false, 0); }
        else           { if (sub) OCN_LAUNCH_ROLES(false, true, 0); else OCN_LAUNCH_ROLES(false, false, 0); }
    }
#undef OCN_LAUNCH_ROLES
    return 0;
}

template <int NTR>
static int launch_roles_n(const DGrid &g, hipStream_t stream, const double *u, const double *v, const double *w, const double *const *tr,
                          double *Gu, double *Gv, double *Gw, double *const *Gc, const int *range, const FusedSubstep *sub) {
    constexpr int NF = 3 + NTR;
    RoleArgs<NF> a;
    a.U[0] = u; a.U[1] = v; a.U[2] = w; a.G[0] = Gu; a.G[1] = Gv; a.G[2] = Gw;
    for (int t = 0; t < NTR; ++t) { a.U[3 + t] = tr[t]; a.G[3 + t] = Gc[t]; }
    a.has_zeta = sub ? sub->has_zeta : 0;
    a.store_G = (sub && !sub->store_G) ? 0 : 1;
    a.dt = sub ? sub->dt : 0.0; a.gamma = sub ? sub->gamma : 0.0; a.zeta = sub ? sub->zeta : 0.0;
    for (int f = 0; f < NF; ++f) { a.Un[f] = sub ? sub->Un[f] : nullptr; a.Gm[f] = sub ? sub->Gm[f] : nullptr; }
    const int Px = g.Nx + 2 * g.Hx, Py = g.Ny + 2 * g.Hy;
    a.s1 = Px;
    a.s2 = (unsigned)((long)Px * Py);
    a.off = (g.Hx - 1) + (long)Px * (g.Hy - 1);
    if (range) {
        a.r = Range6{range[0], range[1], range[2], range[3], range[4], range[5]};
        a.wk0 = a.r.k0;                                   // KernelParameters launches ignore exclude_periphery
    } else {
        a.r = Range6{1, g.Nx, 1, g.Ny, 1, g.Nz};
        a.wk0 = (g.tz != 0 && g.Nz > 1) ? 2 : 1;          // exclude_periphery: w tendencies start at k = 2 on Bounded z
    }
    return launch_roles_t<NTR, OCN_ROLE_TY>(g, stream, a, sub != nullptr);
}

static inline int launch_role_tendency(const DGrid &g, hipStream_t stream, const double *u, const double *v, const double *w,
                                       const double *const *tr, int ntr, double *Gu, double *Gv, double *Gw, double *const *Gc,
                                       const int *range, const FusedSubstep *sub = nullptr) {
    switch (ntr) {
        case 0: return launch_roles_n<0>(g, stream, u, v, w, tr, Gu, Gv, Gw, Gc, range, sub);
        case 1: return launch_roles_n<1>(g, stream, u, v, w, tr, Gu, Gv, Gw, Gc, range, sub);
        case 2: return launch_roles_n<2>(g, stream, u, v, w, tr, Gu, Gv, Gw, Gc, range, sub);
        case 3: return launch_roles_n<3>(g, stream, u, v, w, tr, Gu, Gv, Gw, Gc, range, sub);
        default: return -2;
    }
}
