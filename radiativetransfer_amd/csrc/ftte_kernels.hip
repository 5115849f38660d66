// ftte_kernels.hip -- CDNA4 (gfx950) kernels of the diffuse long-characteristics sweep.
//
// What is computed (per direction, per frequency group): the reference's cell transfer,
//   transportRoutinesModule.f90:587-961  /  equiSources.f90:1580-1788,
// on a uniform grid.  How it is organised has nothing in common with the reference's serial
// pointer walk:
//
//  * After the izone rotation (rotateIndicesModule.f90) the march axis is sweep-i and every cell
//    of a layer shares one ray pattern, so a physical ray is a chain of 1-3 segments per layer
//    that drifts by at most one cell per layer along sweep-j and sweep-k, identically for all
//    rays of the direction.  Chains never interact: the only coupling is that a cell's mean
//    collects the segments of up to three different rays.
//  * One wavefront owns a tile of 64 x ROWS rays: 64 lanes along u, the storage-contiguous axis
//    of this direction's layout (every kappa/J access is a 512-byte coalesced row), ROWS rays
//    per lane along v.  The ray intensities never leave registers for the whole march.
//  * A cell's mean needs the 2nd/3rd segments of the rays one step lower in u and/or v: from
//    lane-1 through a DPP wave shift, from the previous row through registers.  Lane 0 and row
//    0 of every tile are therefore a read-only halo, recomputed by the neighbouring tile: no
//    LDS, no barrier, no atomics, no inter-wave communication of any kind, and every cell of
//    the grid is owned by exactly one lane of one wave per direction, which makes J a plain,
//    deterministic read-modify-write.
//  * The host turns each direction into a 32-byte-per-layer table (segment lengths, chain
//    class, cumulative drift) that the wave reads with scalar loads; per-layer control flow is
//    wave-uniform.
//
// Roofline: 24 algorithmic bytes per cell.direction.frequency update (kappa 8 B read, J 8 B
// read + 8 B write), ~2 segments of ~45 fp64 VALU instructions each: HBM-bound by design,
// with the fp64 pipe at 60-80 % when HBM saturates.
#include <hip/hip_runtime.h>

#include "ftte_internal.h"
#include "ftte_kernels.h"
#include "ftte_math.h"

namespace ftte {

// Pointers that went through uniform() have lost their address space as far as the compiler can tell, and a generic pointer
// is read with flat_load, which also occupies the LDS path and counts on lgkmcnt: say that they are global memory.
using gcbyte = const __attribute__((address_space(1))) char;
using gbyte = __attribute__((address_space(1))) char;
using gcdouble = const __attribute__((address_space(1))) double;
using gdouble = __attribute__((address_space(1))) double;

// value held by lane-1 (lane 0 receives its own value back; it is a halo lane and never uses it)
__device__ __forceinline__ double from_lane_below(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // DPP wave_shr:1 -- a full-rate VALU move, no LDS crossbar
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// Values that are the same in every lane but reach the wave through vector loads (the compiler cannot prove the
// tables are not written by the kernel, so it will not use scalar loads for them): pin them into SGPRs, so that
// everything derived from them -- plane and row addresses, segment lengths, branch conditions -- is scalar.
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ long uniform(long x)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x & 0xffffffffl));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)x >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ double uniform(double x) { return __longlong_as_double(uniform((long)__double_as_longlong(x))); }
template <typename T> __device__ __forceinline__ T *uniform(T *p) { return reinterpret_cast<T *>(uniform(reinterpret_cast<long>(p))); }

// One segment of one ray.  EDGE: the cell may lie outside the domain, where the ray is
// re-initialised with the inflow (transportRoutinesModule.f90:594-597) and adds nothing.
// EMIT: 0 no emission (the reference as shipped), 1 `x` is the reference's emissivity eta, 2 `x` is a source function S
// (ftte_math.h: ftte_segment_emit, ftte_segment_source).
template <bool EDGE, int EMIT>
__device__ __forceinline__ double segment(const ftte_consts &K, double &I, double kap, double x, double dpath, bool inside,
                                          double uvb)
{
    double It = I, m;
    if (EMIT == 0) m = ftte_segment(&K, &It, kap * dpath);
    else if (EMIT == 2) m = ftte_segment_source(&K, K.c[9], &It, kap * dpath, x);
    else m = ftte_segment_emit(&K, &It, kap * dpath, x, 0.0);
    if (EDGE) {
        I = inside ? It : uvb;
        return inside ? m : 0.0;
    }
    I = It;
    return m;
}

// One layer of one tile.  RC: chain class (ftte_internal.h).  I[r]: intensity of ray r of this
// lane on entry to the layer / on exit.
//   kplane / jplane : byte pointers to the virtual element (row 0, column position 0) of this layer's
//                     kappa / J plane; rows are sv elements apart; a lane's column enters as a
//                     non-negative position (mirrored columns are folded into the position by the
//                     caller) so that every access is  uniform 64-bit base + 32-bit lane offset.
//   STACK           : wavefronts of the workgroup stacked along v.  The cell of row 0 needs the 2nd/3rd
//                     segments of the row below, which belongs to the wavefront below: with STACK = 1 row 0
//                     is a read-only halo (the tile below recomputes it); with STACK > 1 only the lowest
//                     wavefront has a halo row, the others receive the two numbers per lane they need
//                     through LDS (`xchg`, one s_barrier per layer that has such segments) and own row 0.
template <int ROWS, int RC, bool EDGE, int STACK, int EMIT>
__device__ __forceinline__ void layer_step(const ftte_consts &K, double (&I)[ROWS], gcbyte *__restrict__ kplane,
                                           gcbyte *__restrict__ xplane, gbyte *__restrict__ jplane, int cv0, int cu,
                                           int n, int sv, bool mirror_u,
                                           double d0, double d1, double d2, double w, double uvb, bool first,
                                           bool lane_owned, int lane, int wid, double *xchg)
{
    constexpr int SHAPE = (RC == RC_THREE_U_SWAP) ? RC_THREE_U : (RC == RC_THREE_V_SWAP) ? RC_THREE_V : RC;
    constexpr bool THIRD_FIRST = (RC == RC_THREE_U_SWAP || RC == RC_THREE_V_SWAP);
    constexpr bool HAS_U = (SHAPE == RC_TWO_U || SHAPE == RC_THREE_U || SHAPE == RC_THREE_V); // touches column u+1
    constexpr bool HAS_V = (SHAPE == RC_TWO_V || SHAPE == RC_THREE_U || SHAPE == RC_THREE_V); // touches row v+1
    constexpr int NSEG = (SHAPE == RC_ONE) ? 1 : (SHAPE <= RC_TWO_V ? 2 : 3);
    const bool row0_owned = STACK > 1 && wid > 0; // wave-uniform
    constexpr int R0 = (STACK > 1) ? 0 : 1;       // first row that may be owned

    // lane-varying column positions, clamped into the domain for EDGE tiles
    const int c0 = EDGE ? clampi(cu, 1, n) : cu;
    const int c1 = EDGE ? clampi(cu + 1, 1, n) : cu + 1;
    const unsigned off0 = 8u * (unsigned)(mirror_u ? n + 1 - c0 : c0);
    const unsigned off1 = 8u * (unsigned)(mirror_u ? n + 1 - c1 : c1);
    const bool in_u0 = !EDGE || (cu >= 1 && cu <= n);
    const bool in_u1 = !EDGE || (cu + 1 >= 1 && cu + 1 <= n);
    const long row_bytes = 8l * sv;

    // ---- issue every load of the layer up front ------------------------------------------------
    // kappa through the caches (halo rows and straddling lines are shared with neighbouring tiles); J is touched once
    // per direction, by this lane only: loaded and stored non-temporally, so that it streams past the L2 instead of
    // evicting kappa (measured: -6.5 % time; only the load or only the store non-temporal: none, or worse)
    double K0[ROWS + 1], K1[ROWS + 1], X0[EMIT ? ROWS + 1 : 1], X1[EMIT ? ROWS + 1 : 1], Jacc[ROWS];
#pragma unroll
    for (int r = 0; r <= ROWS; ++r) {
        const int row = EDGE ? clampi(cv0 + r, 1, n) : cv0 + r;
        gcbyte *rp = kplane + row * row_bytes;
        const bool need0 = (r < ROWS) || (SHAPE == RC_TWO_V || SHAPE == RC_THREE_V);
        const bool need1 = HAS_U && ((SHAPE == RC_TWO_U) ? (r < ROWS) : (SHAPE == RC_THREE_U) ? true : (r >= 1));
        if (need0) K0[r] = *(gcdouble *)(rp + off0);
        if (need1) K1[r] = *(gcdouble *)(rp + off1);
        if (EMIT) {
            gcbyte *xp = xplane + row * row_bytes;
            if (need0) X0[r] = *(gcdouble *)(xp + off0);
            if (need1) X1[r] = *(gcdouble *)(xp + off1);
        }
    }
    constexpr int XM = EMIT ? ~0 : 0; // index mask: without emission the X arrays have one (unused) element
    const bool own_lane = lane_owned && in_u0;
#pragma unroll
    for (int r = R0; r < ROWS; ++r) Jacc[r] = 0.0;
    if (!first && own_lane) {
#pragma unroll
        for (int r = R0; r < ROWS; ++r) {
            const int row = cv0 + r;
            if ((r > 0 || row0_owned) && (!EDGE || (row >= 1 && row <= n)))
                Jacc[r] = __builtin_nontemporal_load((gcdouble *)(jplane + row * row_bytes + off0));
        }
    }

    // ---- march the rays of this lane through the layer, row by row -----------------------------
    double prev1 = 0.0, prev2 = 0.0; // 2nd / 3rd segment means of the previous row (for HAS_V gathers)
    double own0 = 0.0, side0 = 0.0;  // row 0: its own xy mean, and what lane-1 of the same row gives it
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int row = cv0 + r;
        const bool in_v0 = !EDGE || (row >= 1 && row <= n);
        const bool in_v1 = !EDGE || (row + 1 >= 1 && row + 1 <= n);

        double m0, m1 = 0.0, m2 = 0.0;
        m0 = segment<EDGE, EMIT>(K, I[r], K0[r], X0[r & XM], d0, in_v0 && in_u0, uvb);
        if (SHAPE == RC_TWO_U) m1 = segment<EDGE, EMIT>(K, I[r], K1[r], X1[r & XM], d1, in_v0 && in_u1, uvb);
        if (SHAPE == RC_TWO_V) m1 = segment<EDGE, EMIT>(K, I[r], K0[r + 1], X0[(r + 1) & XM], d1, in_v1 && in_u0, uvb);
        if (SHAPE == RC_THREE_U) {
            m1 = segment<EDGE, EMIT>(K, I[r], K1[r], X1[r & XM], d1, in_v0 && in_u1, uvb);
            m2 = segment<EDGE, EMIT>(K, I[r], K1[r + 1], X1[(r + 1) & XM], d2, in_v1 && in_u1, uvb);
        }
        if (SHAPE == RC_THREE_V) {
            m1 = segment<EDGE, EMIT>(K, I[r], K0[r + 1], X0[(r + 1) & XM], d1, in_v1 && in_u0, uvb);
            m2 = segment<EDGE, EMIT>(K, I[r], K1[r + 1], X1[(r + 1) & XM], d2, in_v1 && in_u1, uvb);
        }

        // the cell (row, cu) collects: its own xy segment, and the 2nd / 3rd segments that end
        // up in it, which belong to the rays one step lower in u and/or v
        if (r >= 1) {
            double g1 = 0.0, g2 = 0.0;
            if (SHAPE == RC_TWO_U) g1 = from_lane_below(m1);
            if (SHAPE == RC_TWO_V) g1 = prev1;
            if (SHAPE == RC_THREE_U) { g1 = from_lane_below(m1); g2 = from_lane_below(prev2); }
            if (SHAPE == RC_THREE_V) { g1 = prev1; g2 = from_lane_below(prev2); }
            double acc = m0;
            if (NSEG == 2) acc += g1;
            if (NSEG == 3) {
                // reference order: xy + xz + yz (transportRoutinesModule.f90:695-941)
                acc += THIRD_FIRST ? g2 : g1;
                acc += THIRD_FIRST ? g1 : g2;
            }
            Jacc[r] += ftte_cell_mean(acc, NSEG, w);
            // computed here, for every lane: otherwise the whole mean (selects, products) is sunk into the
            // lane-predicated store block after the last row, with every row's operands kept alive until then
            asm volatile("" : "+v"(Jacc[r]));
        } else if (STACK > 1) {
            own0 = m0;
            if (SHAPE == RC_TWO_U || SHAPE == RC_THREE_U) side0 = from_lane_below(m1);
        }
        if (HAS_V) { prev1 = m1; prev2 = m2; }
        // keep the rows in program order: their arithmetic is independent, and left alone the scheduler
        // interleaves all of them, which multiplies the live registers by the row count
        __builtin_amdgcn_sched_barrier(0);
    }

    if (STACK > 1) {
        // row 0 of every wavefront but the lowest: the row below lives in the wavefront below
        double below1 = 0.0, below2 = 0.0;
        if (HAS_V) {
            double *mine = xchg + (wid * 2) * 64;
            mine[lane] = prev1;      // my top row's 2nd-segment mean
            mine[64 + lane] = prev2; // and 3rd
            __syncthreads();
            if (row0_owned) {
                const double *theirs = xchg + ((wid - 1) * 2) * 64;
                below1 = theirs[lane];
                below2 = theirs[64 + (lane > 0 ? lane - 1 : 0)]; // the 3rd segment comes from one lane lower as well
            }
        }
        double g1 = 0.0, g2 = 0.0;
        if (SHAPE == RC_TWO_U) g1 = side0;
        if (SHAPE == RC_TWO_V) g1 = below1;
        if (SHAPE == RC_THREE_U) { g1 = side0; g2 = below2; }
        if (SHAPE == RC_THREE_V) { g1 = below1; g2 = below2; }
        double acc = own0;
        if (NSEG == 2) acc += g1;
        if (NSEG == 3) {
            acc += THIRD_FIRST ? g2 : g1;
            acc += THIRD_FIRST ? g1 : g2;
        }
        Jacc[0] += ftte_cell_mean(acc, NSEG, w);
    }

    if (own_lane) {
#pragma unroll
        for (int r = R0; r < ROWS; ++r) {
            const int row = cv0 + r;
            if ((r > 0 || row0_owned) && (!EDGE || (row >= 1 && row <= n)))
                __builtin_nontemporal_store(Jacc[r], (gdouble *)(jplane + row * row_bytes + off0));
        }
    }
}

template <int ROWS, bool EDGE, int STACK, int EMIT>
__device__ __forceinline__ void layer_dispatch(const ftte_consts &K, double (&I)[ROWS], int rc, gcbyte *kplane,
                                               gcbyte *xplane, gbyte *jplane, int cv0, int cu, int n, int sv,
                                               bool mirror_u, double d0,
                                               double d1, double d2, double w, double uvb, bool first, bool lane_owned,
                                               int lane, int wid, double *xchg)
{
#define FTTE_CASE(C)                                                                                                     \
    case C:                                                                                                              \
        layer_step<ROWS, C, EDGE, STACK, EMIT>(K, I, kplane, xplane, jplane, cv0, cu, n, sv, mirror_u, d0, d1, d2, w,   \
                                               uvb, first, lane_owned, lane, wid, xchg);                                 \
        break;
    switch (rc) {
        FTTE_CASE(RC_ONE)
        FTTE_CASE(RC_TWO_U)
        FTTE_CASE(RC_TWO_V)
        FTTE_CASE(RC_THREE_U)
        FTTE_CASE(RC_THREE_V)
        FTTE_CASE(RC_THREE_U_SWAP)
    default:
        layer_step<ROWS, RC_THREE_V_SWAP, EDGE, STACK, EMIT>(K, I, kplane, xplane, jplane, cv0, cu, n, sv, mirror_u, d0, d1,
                                                             d2, w, uvb, first, lane_owned, lane, wid, xchg);
        break;
    }
#undef FTTE_CASE
}

// grid: nitems * nnu workgroups of STACK wavefronts; the frequency group is the fastest index, so that
// (for nnu = 8) all tiles of one group run on one XCD.
template <int ROWS, int WAVES, int STACK, int EMIT>
__global__ void __launch_bounds__(64 * STACK, WAVES) sweep_kernel(const LaunchRec L)
{
    __shared__ double xchg_lds[STACK > 1 ? 2 * STACK * 2 * 64 : 1]; // [parity][wavefront][2nd|3rd][lane]
    extern __shared__ double occupancy_pad[];                        // diagnostic: dynamic LDS only limits residency
    const int nnu = L.nnu;
    const int nu = blockIdx.x % nnu;
    const WorkItem *ip = L.items + blockIdx.x / nnu;
    const int slot = uniform((int)ip->slot), tu = uniform((int)ip->tu), tv = uniform((int)ip->tv);
    const int i_first = uniform((int)ip->i_first), i_last = uniform((int)ip->i_last);
    const DirRec &D = L.dir[slot];
    const int lane = threadIdx.x & 63;
    const int wid = STACK > 1 ? uniform((int)(threadIdx.x >> 6)) : 0;
    const int n = L.n;

    const double uvb = uniform(L.uvb[nu]);
    const double w = uniform(D.w);
    const int sv = uniform(D.sv), si = uniform(D.si);
    const bool mirror_u = uniform(D.su) < 0;
    const bool first = uniform(D.first) != 0;
    const long org = uniform((long)D.org);
    gcbyte *kbase = (gcbyte *)(uniform(D.kappa) + (long)nu * L.group_stride + org);
    gbyte *jbase = (gbyte *)(uniform(D.J) + (long)nu * L.group_stride + org);
    gcbyte *xbase = EMIT ? (gcbyte *)(uniform(D.emis) + (long)nu * L.group_stride + org) : nullptr;
    const int u_lo = uniform(D.u_lo), v_lo = uniform(D.v_lo);

    // labels of this lane's rays: u label of the lane (lane 0: halo), v label of this wavefront's row 0
    // (row 0 of wavefront 0: halo)
    const int ul = u_lo + 63 * tu + lane - 1;
    const int vl0 = v_lo + (STACK * ROWS - 1) * tv - 1 + wid * ROWS;
    const bool lane_owned = lane != 0;

    double I[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) I[r] = uvb;

    const __attribute__((address_space(1))) LayerRec *layers = (const __attribute__((address_space(1))) LayerRec *)uniform(D.layers);
    int parity = 0;
    for (int i = i_first; i <= i_last; ++i) {
        const __attribute__((address_space(1))) LayerRec *rp = layers + (i - 1);
        const double d0 = uniform(rp->dpath[0]), d1 = uniform(rp->dpath[1]), d2 = uniform(rp->dpath[2]);
        const int rc = uniform(rp->info) & 7;
        const int drift = uniform(rp->drift);
        const int du = (int)(short)(drift & 0xffff), dv = drift >> 16;
        const int cu = ul + du;
        const int cv0 = vl0 + dv;
        gcbyte *kplane = kbase + 8l * i * si;
        gbyte *jplane = jbase + 8l * i * si;
        gcbyte *xplane = EMIT ? xbase + 8l * i * si : nullptr;
        double *xchg = xchg_lds + parity * (STACK * 2 * 64);
        const bool has_v = rc != RC_ONE && rc != RC_TWO_U; // this layer passes segments from row to row

        // wave-uniform position tests on every cell any lane of this wavefront may touch (halo and +1 offsets included)
        const int u_min = u_lo + 63 * tu - 1 + du, v_min = cv0;
        const bool interior = u_min >= 1 && u_min + 64 <= n && v_min >= 1 && v_min + ROWS <= n;
        const bool outside = STACK > 1 && (u_min > n || u_min + 64 < 1 || v_min > n || v_min + ROWS < 1);
        if (outside) {
            // nothing of this wavefront is in the domain at this layer: its rays keep (or, having left, no longer need)
            // the inflow value; it only has to keep the workgroup's exchange protocol going
            if (has_v) {
                xchg[(wid * 2) * 64 + lane] = 0.0;
                xchg[(wid * 2 + 1) * 64 + lane] = 0.0;
                __syncthreads();
            }
        } else if (interior)
            layer_dispatch<ROWS, false, STACK, EMIT>(L.math, I, rc, kplane, xplane, jplane, cv0, cu, n, sv, mirror_u, d0, d1,
                                                     d2, w, uvb, first, lane_owned, lane, wid, xchg);
        else
            layer_dispatch<ROWS, true, STACK, EMIT>(L.math, I, rc, kplane, xplane, jplane, cv0, cu, n, sv, mirror_u, d0, d1,
                                                    d2, w, uvb, first, lane_owned, lane, wid, xchg);
        if (STACK > 1 && has_v) parity ^= 1;
    }
}

// WAVES = waves per SIMD the register allocation is held to (512 / WAVES VGPRs per lane)
static int g_lds_pad = 0; // bytes of dynamic LDS per workgroup (diagnostic knob "ldspad")
void set_lds_pad(int bytes) { g_lds_pad = bytes; }
int lds_pad() { return g_lds_pad; }

template <int ROWS, int STACK>
static int launch_variant(const LaunchRec &L, int waves, dim3 grid, hipStream_t stream)
{
    const dim3 block(64 * STACK);
    switch (waves) {
    case 2: hipLaunchKernelGGL((sweep_kernel<ROWS, 2, STACK, 0>), grid, block, g_lds_pad, stream, L); break;
    case 3: hipLaunchKernelGGL((sweep_kernel<ROWS, 3, STACK, 0>), grid, block, g_lds_pad, stream, L); break;
    case 4: hipLaunchKernelGGL((sweep_kernel<ROWS, 4, STACK, 0>), grid, block, g_lds_pad, stream, L); break;
    case 5: hipLaunchKernelGGL((sweep_kernel<ROWS, 5, STACK, 0>), grid, block, g_lds_pad, stream, L); break;
    case 6: hipLaunchKernelGGL((sweep_kernel<ROWS, 6, STACK, 0>), grid, block, g_lds_pad, stream, L); break;
    default: return -1;
    }
    return 0;
}

int launch_sweep(const LaunchRec &L, int rows, int waves, int stack, int nnu, hipStream_t stream)
{
    if (L.nitems <= 0 || nnu != L.nnu) return L.nitems <= 0 ? 0 : -1;
    const dim3 grid((unsigned)L.nitems * (unsigned)nnu);
    int rc = -1;
    if (L.emit) { // emission: one built shape (8 rows, single wavefront, 3 waves per SIMD)
        if (rows != 8 || stack != 1) return -1;
        if (L.emit == 1) hipLaunchKernelGGL((sweep_kernel<8, 3, 1, 1>), grid, dim3(64), g_lds_pad, stream, L);
        else hipLaunchKernelGGL((sweep_kernel<8, 3, 1, 2>), grid, dim3(64), g_lds_pad, stream, L);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (rows == 4 && stack == 1) rc = launch_variant<4, 1>(L, waves, grid, stream);
    else if (rows == 4 && stack == 4) rc = launch_variant<4, 4>(L, waves, grid, stream);
    else if (rows == 4 && stack == 8) rc = launch_variant<4, 8>(L, waves, grid, stream);
    else if (rows == 8 && stack == 1) rc = launch_variant<8, 1>(L, waves, grid, stream);
    else if (rows == 8 && stack == 2) rc = launch_variant<8, 2>(L, waves, grid, stream);
    else if (rows == 8 && stack == 4) rc = launch_variant<8, 4>(L, waves, grid, stream);
    else if (rows == 16 && stack == 1) rc = launch_variant<16, 1>(L, waves, grid, stream);
    if (rc) return rc;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Layouts.  Cell-array order is [ic][jc][kc] (kc fastest).  A direction whose march axis is
// storage-j reads layout 1 = [jc][ic][kc]; storage-k reads layout 2 = [kc][ic][jc], so that the
// march axis is always the slowest and a sweep plane is always made of contiguous rows.
// ------------------------------------------------------------------------------------------------

// Brick order (BrickLaunch::tiled): element (a, b, c) of a frame [a][b][c] -- a the march axis, c contiguous -- when the eight rows
// b of a brick's layer are kept in one piece of 8 x 64 doubles, the pieces of a layer in the order of the bricks (along b, then
// along c).  n a multiple of 64 (and so of kBrickRows).
// chunk > 0: the `chunk` layers of a brick follow each other too (the brick in one piece; n a multiple of chunk).
__device__ __forceinline__ long tiled_index(int n, int a, int b, int c, int chunk = 0)
{
    const long piece = 64 * kBrickRows, in_piece = (b % kBrickRows) * 64 + c % 64;
    if (chunk > 0)
        return ((((long)(a / chunk) * (n / kBrickRows) + b / kBrickRows) * (n / 64) + c / 64) * chunk + a % chunk) * piece + in_piece;
    return (((long)a * (n / kBrickRows) + b / kBrickRows) * (n / 64) + c / 64) * piece + in_piece;
}

// dst[jc][ic][kc] = src[ic][jc][kc]   (rows of n doubles move as they are); layout 0: the rows stay where they are and only the
// brick order (tiled) moves them
template <bool tiled>
__global__ void __launch_bounds__(256) to_layout1_kernel(const double *__restrict__ src, double *__restrict__ dst, int n,
                                                         long group_stride, int layout, int tchunk)
{
    const long g = blockIdx.z;
    const int ic = blockIdx.y / n, jc = blockIdx.y % n;
    const double *s = src + g * group_stride + ((long)ic * n + jc) * n;
    const int a = layout == 1 ? jc : ic, b = layout == 1 ? ic : jc;
    double *d = dst + g * group_stride;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x)
        d[tiled ? tiled_index(n, a, b, k, tchunk) : ((long)a * n + b) * n + k] = s[k];
}

// dst[kc][ic][jc] = src[ic][jc][kc]   (per ic plane, a 32x32 tiled transpose through LDS)
template <bool tiled>
__global__ void __launch_bounds__(256) to_layout2_kernel(const double *__restrict__ src, double *__restrict__ dst, int n,
                                                         long group_stride, int tchunk)
{
    __shared__ double tile[32][33];
    const long g = blockIdx.z / n;
    const int ic = blockIdx.z % n;
    const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const double *s = src + g * group_stride + (long)ic * n * n;
    double *d = dst + g * group_stride;
    for (int r = ty; r < 32; r += 8)
        if (j0 + r < n && k0 + tx < n) tile[r][tx] = s[(long)(j0 + r) * n + k0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < n && j0 + tx < n) d[tiled ? tiled_index(n, k0 + r, ic, j0 + tx, tchunk) : ((long)(k0 + r) * n + ic) * n + j0 + tx] = tile[tx][r];
}

// The caller's opacities (cell-array order) into the library's three copies in ONE pass: dst0 as they come, dst1[jc][ic][kc] the
// same rows elsewhere, dst2[kc][ic][jc] each ic plane transposed through LDS.  Read once, written three times (a copy and two
// transposes read them three times).
__global__ void __launch_bounds__(256) set_layouts_kernel(const double *__restrict__ src, double *__restrict__ dst0, double *__restrict__ dst1,
                                                          double *__restrict__ dst2, int n, long group_stride)
{
    __shared__ double tile[32][33];
    const long g = blockIdx.z / n;
    const int ic = blockIdx.z % n;
    const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
    const long base = g * group_stride;
    for (int r = ty; r < 32; r += 8)
        if (j0 + r < n && k0 + tx < n) {
            const double v = src[base + ((long)ic * n + j0 + r) * n + k0 + tx];
            tile[r][tx] = v;
            dst0[base + ((long)ic * n + j0 + r) * n + k0 + tx] = v;
            dst1[base + ((long)(j0 + r) * n + ic) * n + k0 + tx] = v;
        }
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (k0 + r < n && j0 + tx < n) dst2[base + ((long)(k0 + r) * n + ic) * n + j0 + tx] = tile[tx][r];
}

int launch_set_layouts(const double *src, double *dst0, double *dst1, double *dst2, int n, int nnu, long group_stride, hipStream_t stream)
{
    const dim3 grid((n + 31) / 32, (n + 31) / 32, n * nnu);
    hipLaunchKernelGGL(set_layouts_kernel, grid, dim3(256), 0, stream, src, dst0, dst1, dst2, n, group_stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_to_layout(int layout, const double *src, double *dst, int n, int nnu, long group_stride, hipStream_t stream, bool tiled, int tchunk)
{
    if (tiled && (n % 64 != 0 || n % kBrickRows != 0 || (tchunk > 0 && n % tchunk != 0))) return -1;
    if (layout == 1 || (layout == 0 && tiled)) {
        const dim3 grid((n + 255) / 256, n * n, nnu);
        if (tiled) hipLaunchKernelGGL(to_layout1_kernel<true>, grid, dim3(256), 0, stream, src, dst, n, group_stride, layout, tchunk);
        else hipLaunchKernelGGL(to_layout1_kernel<false>, grid, dim3(256), 0, stream, src, dst, n, group_stride, layout, tchunk);
    } else if (layout == 2) {
        const dim3 grid((n + 31) / 32, (n + 31) / 32, n * nnu);
        if (tiled) hipLaunchKernelGGL(to_layout2_kernel<true>, grid, dim3(256), 0, stream, src, dst, n, group_stride, tchunk);
        else hipLaunchKernelGGL(to_layout2_kernel<false>, grid, dim3(256), 0, stream, src, dst, n, group_stride, tchunk);
    } else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// J[ic][jc][kc] = sum over the accumulators in list order (layout 0 first, then 1, then 2; within
// a layout by slot).  One block per (32 jc x 32 kc) tile of one ic plane of one group.
struct MergeRec {
    const double *acc[3 * kMaxAcc];
    int layout[3 * kMaxAcc];
    int count;
};

// accumulate != 0: J += the accumulators (the additions continue the sequence of an earlier partial merge)
// leaf_of_base != nullptr (hybrid sweep of a refined cell array): the accumulators are arrays over the BASE cells (group stride
// n^3) and J is in cell-array order (group stride j_stride): element (ic, jc, kc) goes to its leaf, refined base cells are skipped.
// tiled: the accumulators are in brick order (tiled_index)
template <bool tiled>
__global__ void __launch_bounds__(256) merge_kernel(const MergeRec M, double *__restrict__ J, int n, long group_stride, int accumulate,
                                                    const int32_t *__restrict__ leaf_of_base, long j_stride, int tchunk)
{
    __shared__ double tile[32][33];
    const long g = blockIdx.z / n;
    const int ic = blockIdx.z % n;
    const int j0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double sum[4] = {0.0, 0.0, 0.0, 0.0};
    bool have = false;
    if (accumulate && !leaf_of_base) {
        for (int q = 0; q < 4; ++q) {
            const int jc = j0 + ty + 8 * q, kc = k0 + tx;
            if (jc < n && kc < n) sum[q] = __builtin_nontemporal_load(&J[g * group_stride + ((long)ic * n + jc) * n + kc]);
        }
        have = true;
    }
    // the accumulators that need no transpose (layouts 0 and 1: they come first in the list), eight at a time: all their loads in
    // flight, then the sums in list order
    int a0 = 0;
    while (a0 < M.count && M.layout[a0] != 2) {
        int nb = 0;
        while (nb < 8 && a0 + nb < M.count && M.layout[a0 + nb] != 2) ++nb;
        double v[8][4];
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b >= nb) continue;
            const double *s = M.acc[a0 + b] + g * group_stride;
            const bool one = M.layout[a0 + b] == 1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jc = j0 + ty + 8 * q, kc = k0 + tx;
                v[b][q] = 0.0;
                if (jc >= n || kc >= n) continue;
                const int x = one ? jc : ic, y = one ? ic : jc;
                v[b][q] = __builtin_nontemporal_load(&s[tiled ? tiled_index(n, x, y, kc, tchunk) : ((long)x * n + y) * n + kc]);
            }
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (b >= nb) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) sum[q] = have ? sum[q] + v[b][q] : v[b][q];
            have = true;
        }
        a0 += nb;
    }
    for (int a = a0; a < M.count; ++a) {
        const double *s = M.acc[a] + g * group_stride;
        if (M.layout[a] == 2) {
            // element (ic, jc, kc) sits at [kc][ic][jc]: read rows along jc, transpose through LDS
            __syncthreads();
            for (int q = 0; q < 4; ++q) {
                const int r = ty + 8 * q; // kc offset
                if (k0 + r < n && j0 + tx < n)
                    tile[r][tx] = __builtin_nontemporal_load(&s[tiled ? tiled_index(n, k0 + r, ic, j0 + tx, tchunk) : ((long)(k0 + r) * n + ic) * n + j0 + tx]);
            }
            __syncthreads();
        }
        for (int q = 0; q < 4; ++q) {
            const int r = ty + 8 * q; // jc offset
            const int jc = j0 + r, kc = k0 + tx;
            if (jc >= n || kc >= n) continue;
            double v;
            if (M.layout[a] == 0) v = __builtin_nontemporal_load(&s[tiled ? tiled_index(n, ic, jc, kc, tchunk) : ((long)ic * n + jc) * n + kc]);
            else if (M.layout[a] == 1) v = __builtin_nontemporal_load(&s[tiled ? tiled_index(n, jc, ic, kc, tchunk) : ((long)jc * n + ic) * n + kc]);
            else v = tile[tx][r];
            sum[q] = have ? sum[q] + v : v;
        }
        have = true;
    }
    for (int q = 0; q < 4; ++q) {
        const int jc = j0 + ty + 8 * q, kc = k0 + tx;
        if (jc >= n || kc >= n) continue;
        const long b = ((long)ic * n + jc) * n + kc;
        if (!leaf_of_base) __builtin_nontemporal_store(sum[q], &J[g * group_stride + b]);
        else if (leaf_of_base[b] >= 0) { // += : the forest has already added the directions in whose box this cell lies
            double *dst = &J[g * j_stride + leaf_of_base[b]];
            *dst = accumulate ? *dst + sum[q] : sum[q];
        }
    }
}

int launch_merge(const double *const *acc, const int *layout, int count, double *J, int n, int nnu, long group_stride,
                 bool accumulate, hipStream_t stream, const int32_t *leaf_of_base, long j_stride, bool tiled, int tchunk)
{
    if (count > 3 * kMaxAcc) return -1;
    MergeRec M;
    M.count = count;
    for (int a = 0; a < count; ++a) { M.acc[a] = acc[a]; M.layout[a] = layout[a]; }
    const dim3 grid((n + 31) / 32, (n + 31) / 32, n * nnu);
    if (tiled) hipLaunchKernelGGL(merge_kernel<true>, grid, dim3(256), 0, stream, M, J, n, group_stride, accumulate ? 1 : 0, leaf_of_base, j_stride, tchunk);
    else hipLaunchKernelGGL(merge_kernel<false>, grid, dim3(256), 0, stream, M, J, n, group_stride, accumulate ? 1 : 0, leaf_of_base, j_stride, tchunk);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Refined cell arrays.  The host planner (ftte_amr.cpp) has turned the reference's per-direction
// neighbour links into a forest over segments, ordered by depth; one launch processes one depth of
// up to kAmrBatch directions: a thread per (segment, frequency group) takes the intensity its upstream
// segment left (or the inflow, or the mean of two segments: transportRoutinesModule.f90:594-649),
// crosses its own segment (ftte_math.h), and stores the outgoing intensity and the segment's mean.
// ------------------------------------------------------------------------------------------------
// grid: x over the (segment, frequency group) pairs of the fullest direction, y = direction of the batch.
// NNU_SHIFT >= 0: nnu = 1 << NNU_SHIFT (no division); -1: any nnu.
// segment `e` of the level that starts at `begin` in direction D's list, frequency group nu
__device__ __forceinline__ void amr_segment(const AmrLevelRec &A, const AmrDirRec &D, int64_t begin, unsigned e, unsigned nu, unsigned nnu)
{
    const SegRec R = D.rec[begin + e];
    const int seg = R.seg, up = R.up, up2 = R.up2;
    const size_t at = (size_t)(unsigned)seg * nnu + nu;
    double I;
    if (up == -3) I = D.faces[(size_t)nu * A.face_stride + (size_t)R.at]; // a ray handed over by a brick (hybrid sweep)
    else if (up < 0) I = A.uvb[nu];
    else {
        I = D.Iout[(size_t)(unsigned)up * nnu + nu];
        if (up2 >= 0) I = 0.5 * (I + D.Iout[(size_t)(unsigned)up2 * nnu + nu]);
    }
    const unsigned cell = (unsigned)seg / 3u;
    const size_t kat = (size_t)nu * A.group_stride + (size_t)cell * A.cell_stride;
    const double kap = A.kappa[kat];
    double m;
    if (A.emit == 0) m = ftte_segment(&A.math, &I, kap * R.dpath);
    else {
        const double x = A.emis[kat];
        m = A.emit == 2 ? ftte_segment_source(&A.math, A.math.c[9], &I, kap * R.dpath, x) : ftte_segment_emit(&A.math, &I, kap * R.dpath, x, 0.0);
    }
    D.Iout[at] = I;                              // read again by the segments downstream
    __builtin_nontemporal_store(m, &D.mean[at]); // read once, by the combine kernel
}

template <int NNU_SHIFT>
__global__ void __launch_bounds__(256) amr_level_kernel(const AmrLevelRec A)
{
    const int d = blockIdx.y;
    const unsigned count = (unsigned)A.count[d];
    const unsigned nnu = (unsigned)A.nnu;
    const unsigned t = blockIdx.x * 256u + threadIdx.x;
    const unsigned e = NNU_SHIFT >= 0 ? t >> (NNU_SHIFT >= 0 ? NNU_SHIFT : 0) : t / nnu;
    const unsigned nu = NNU_SHIFT >= 0 ? t & (nnu - 1u) : t - e * nnu;
    if (e >= count) return;
    amr_segment(A, A.dir[d], A.begin[d], e, nu, nnu);
}

// A run of THIN levels in one launch: one workgroup per direction walks levels depth0 .. depth0 + ndepth - 1 of its forest, a
// barrier between two levels (what a level reads of the one before was written by this workgroup, on this CU: visible behind the
// barrier).  A level of a few hundred segments is a launch of a few microseconds that the next one has to wait for, and a forest
// around a refined patch is a hundred of them: here they cost a barrier each.  tables: [depth][count[ndir], begin[ndir]].
template <int NNU_SHIFT>
__global__ void __launch_bounds__(1024) amr_levels_kernel(const AmrLevelRec A, const int64_t *__restrict__ tables, int ndepth)
{
    const int d = blockIdx.x;
    const unsigned nnu = (unsigned)A.nnu;
    const AmrDirRec &D = A.dir[d];
    for (int k = 0; k < ndepth; ++k) {
        const int64_t *T = tables + (size_t)k * 2 * (size_t)A.ndir;
        const unsigned items = (unsigned)T[d] * nnu;
        const int64_t begin = T[A.ndir + d];
        for (unsigned t = threadIdx.x; t < items; t += 1024u) {
            const unsigned e = NNU_SHIFT >= 0 ? t >> (NNU_SHIFT >= 0 ? NNU_SHIFT : 0) : t / nnu;
            const unsigned nu = NNU_SHIFT >= 0 ? t & (nnu - 1u) : t - e * nnu;
            amr_segment(A, D, begin, e, nu, nnu);
        }
        __syncthreads();
    }
}

// J[g][leaf] += (w / nseg) * (mean_xy + mean_xz + mean_yz), one direction after the other in list order
// (transportRoutinesModule.f90:953-955)
__global__ void __launch_bounds__(256) amr_combine_kernel(const AmrLevelRec A, double *__restrict__ J, int zero_first)
{
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nnu = A.nnu;
    const bool pow2 = (nnu & (nnu - 1)) == 0; // wave-uniform: a shift instead of a 64-bit division
    const long slot = pow2 ? tid >> (31 - __builtin_clz(nnu)) : tid / nnu;
    const int nu = (int)(tid - slot * nnu);
    if (slot >= (A.cells ? A.ncells : A.ncell)) return;
    const long cell = A.cells ? A.cells[slot] : slot; // where the leaf sits in J; everything of the forest is numbered by `slot`
    double acc_J = zero_first ? 0.0 : J[(long)nu * A.ncell + cell];
    for (int d = 0; d < A.ndir; ++d) {
        const AmrDirRec &D = A.dir[d];
        const int active = D.active[slot];
        if (active & 4) continue; // outside this direction's region: a brick holds the cell's contribution
        double acc = __builtin_nontemporal_load(&D.mean[(3 * slot) * nnu + nu]);
        int nseg = 1;
        if (active & 1) { acc += __builtin_nontemporal_load(&D.mean[(3 * slot + 1) * nnu + nu]); ++nseg; }
        if (active & 2) { acc += __builtin_nontemporal_load(&D.mean[(3 * slot + 2) * nnu + nu]); ++nseg; }
        acc_J += ftte_cell_mean(acc, nseg, D.w);
    }
    J[(long)nu * A.ncell + cell] = acc_J;
}

// the rays that leave the forest's region: into the face buffers the bricks behind it read (hybrid sweep)
__global__ void __launch_bounds__(256) amr_export_kernel(const AmrLevelRec A)
{
    const AmrDirRec &D = A.dir[blockIdx.y];
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nnu = A.nnu;
    const long e = t / nnu;
    const int nu = (int)(t - e * nnu);
    // this launch's share of the direction's rays (a pass of the hybrid sweep), or all of them
    const long count = A.count ? (long)A.count[blockIdx.y] : (long)D.nexports, first = A.begin ? (long)A.begin[blockIdx.y] : 0l;
    if (e >= count) return;
    const AmrExport X = D.exports[first + e];
    D.faces[(size_t)nu * A.face_stride + (size_t)X.at] = D.Iout[(size_t)(unsigned)X.seg * nnu + nu];
}

// the rays that enter a fine block's own brick sweep from the forest around it (AmrImport)
__global__ void __launch_bounds__(256) amr_fine_import_kernel(const AmrLevelRec A)
{
    const AmrDirRec &D = A.dir[blockIdx.y];
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nnu = A.nnu;
    const long e = t / nnu;
    const int nu = (int)(t - e * nnu);
    if (e >= (long)D.nimports) return;
    const AmrImport X = D.imports[e];
    double I;
    if (X.up < 0) I = A.uvb[nu];
    else {
        I = D.Iout[(size_t)(unsigned)X.up * nnu + nu];
        if (X.up2 >= 0) I = 0.5 * (I + D.Iout[(size_t)(unsigned)X.up2 * nnu + nu]);
    }
    D.faces[(size_t)nu * A.face_stride + (size_t)X.at] = I;
}

int launch_amr_fine_import(const AmrLevelRec &A, int64_t most, hipStream_t stream)
{
    if (most <= 0) return 0;
    const dim3 grid((unsigned)((most * A.nnu + 255) / 256), (unsigned)A.ndir);
    hipLaunchKernelGGL(amr_fine_import_kernel, grid, dim3(256), 0, stream, A);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_amr_export(const AmrLevelRec &A, int64_t most, hipStream_t stream)
{
    if (most <= 0) return 0;
    const dim3 grid((unsigned)((most * A.nnu + 255) / 256), (unsigned)A.ndir);
    hipLaunchKernelGGL(amr_export_kernel, grid, dim3(256), 0, stream, A);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// kappa of the base cells of a refined cell array, storage order [ic][jc][kc], from the leaf-ordered array (a refined base cell,
// which no brick reads, gets 0): what the bricks of the hybrid sweep march through
__global__ void __launch_bounds__(256) base_cells_kernel(const double *__restrict__ leaf_values, const int32_t *__restrict__ leaf_of_base,
                                                         double *__restrict__ base_values, long nbase, long ncell)
{
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nbase) return;
    const int32_t leaf = leaf_of_base[b];
    const long g = blockIdx.y;
    base_values[g * nbase + b] = leaf >= 0 ? leaf_values[g * ncell + leaf] : 0.0;
}

int launch_base_cells(const double *leaf_values, const int32_t *leaf_of_base, double *base_values, long nbase, long ncell, int nnu,
                      hipStream_t stream)
{
    hipLaunchKernelGGL(base_cells_kernel, dim3((unsigned)((nbase + 255) / 256), (unsigned)nnu), dim3(256), 0, stream, leaf_values,
                       leaf_of_base, base_values, nbase, ncell);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// dst[cell][g] = src[g][cell]: the forest path reads all groups of a cell together, one 8 * nnu byte run per segment.
// cells != nullptr: dst[position][g] = src[g][cells[position]] for the `count` leaves of the list (src has `ncell` per group)
__global__ void __launch_bounds__(256) cell_major_kernel(const double *__restrict__ src, double *__restrict__ dst, long ncell, int nnu,
                                                         const int32_t *__restrict__ cells, long count)
{
    extern __shared__ double tile[]; // [nnu][64 + 1]
    const long c0 = (long)blockIdx.x * 64;
    for (int i = threadIdx.x; i < nnu * 64; i += 256) {
        const int g = i >> 6, c = i & 63;
        if (c0 + c < count) tile[g * 65 + c] = src[(long)g * ncell + (cells ? (long)cells[c0 + c] : c0 + c)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nnu * 64; i += 256) {
        const int c = i / nnu, g = i - c * nnu;
        if (c0 + c < count) dst[(c0 + c) * nnu + g] = tile[g * 65 + c];
    }
}

int launch_cell_major(const double *src, double *dst, long ncell, int nnu, hipStream_t stream, const int32_t *cells, long count)
{
    if (!cells) count = ncell;
    if (count <= 0) return 0;
    hipLaunchKernelGGL(cell_major_kernel, dim3((unsigned)((count + 63) / 64)), dim3(256), (size_t)nnu * 65 * sizeof(double), stream, src, dst,
                       ncell, nnu, cells, count);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_amr_level(const AmrLevelRec &A, hipStream_t stream)
{
    if (A.most <= 0) return 0;
    const int64_t threads = A.most * A.nnu;
    if (threads >= ((int64_t)1 << 32)) return -1;
    const dim3 grid((unsigned)((threads + 255) / 256), (unsigned)A.ndir);
    switch (A.nnu) {
    case 1: hipLaunchKernelGGL(amr_level_kernel<0>, grid, dim3(256), 0, stream, A); break;
    case 2: hipLaunchKernelGGL(amr_level_kernel<1>, grid, dim3(256), 0, stream, A); break;
    case 4: hipLaunchKernelGGL(amr_level_kernel<2>, grid, dim3(256), 0, stream, A); break;
    case 8: hipLaunchKernelGGL(amr_level_kernel<3>, grid, dim3(256), 0, stream, A); break;
    case 16: hipLaunchKernelGGL(amr_level_kernel<4>, grid, dim3(256), 0, stream, A); break;
    default: hipLaunchKernelGGL(amr_level_kernel<-1>, grid, dim3(256), 0, stream, A); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_amr_levels(const AmrLevelRec &A, const int64_t *tables, int ndepth, hipStream_t stream)
{
    if (ndepth <= 0 || A.ndir <= 0) return 0;
    const dim3 grid((unsigned)A.ndir);
    switch (A.nnu) {
    case 1: hipLaunchKernelGGL(amr_levels_kernel<0>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    case 2: hipLaunchKernelGGL(amr_levels_kernel<1>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    case 4: hipLaunchKernelGGL(amr_levels_kernel<2>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    case 8: hipLaunchKernelGGL(amr_levels_kernel<3>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    case 16: hipLaunchKernelGGL(amr_levels_kernel<4>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    default: hipLaunchKernelGGL(amr_levels_kernel<-1>, grid, dim3(1024), 0, stream, A, tables, ndepth); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_amr_combine(const AmrLevelRec &A, double *J, bool zero_first, hipStream_t stream)
{
    const long threads = (A.cells ? A.ncells : A.ncell) * A.nnu;
    if (threads <= 0) return 0;
    hipLaunchKernelGGL(amr_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, A, J,
                       zero_first ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Point sources.  The reference follows each source's 12 base HEALPix rays depth-first, splitting a ray
// into its four daughter pixels when it has travelled rmax(level) cells (startNewLongRay,
// equiSources.f90:3120-3385).  Here the recursion is unrolled breadth-first by pixel level: one launch
// per level, a thread per ray; a ray that must split appends its state to a queue that seeds the next
// launch (four threads per record).  The tree walk (find/zoom??Neighbour, :2647-2960) uses node indices
// instead of the reference's call sequences: a node's position inside its parent is its index offset.
// Rates are deposited with hardware fp64 atomics (the only order-dependent step of the whole library:
// sums over rays differ in the last bits from run to run).
// ------------------------------------------------------------------------------------------------

// The look-up interpolates the logarithms of the tables.  They are kept as pairs (number rate, heating rate) of one
// reaction side by side, i1 fastest: the two i1 neighbours of both tables are 32 contiguous bytes.
//   logtab[reaction][idust][i3][i2][i1][2]
__device__ __forceinline__ size_t logtab_index(int reaction0, int flat) { return ((size_t)reaction0 * kTableSize + flat) * 2; }

// stellarBetaTable's accumulation over frequency bins, one thread per depth tuple (stellarBetaTable.f90:217-285)
__global__ void __launch_bounds__(256) rate_table_kernel(const FreqBin *__restrict__ bins, int nbins, double *__restrict__ tables,
                                                         double *__restrict__ logtab)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kTableSize) return;
    const int i1 = t % 11, i2 = (t / 11) % 11, i3 = (t / 121) % 11, id = t / 1331;
    // float(idepth)/float(ndepth)*maxOpticalDepth: a single-precision quotient widened, :236-242
    const double d1 = (double)((float)i1 / 10.f) * 10.0, d2 = (double)((float)i2 / 10.f) * 10.0;
    const double d3 = (double)((float)i3 / 10.f) * 10.0, dd = (double)((float)id / 10.f) * 10.0;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nbins; ++b) {
        const FreqBin B = bins[b];
        const double t1 = B.r24 * d1, t2 = B.r26 * d2, t3 = B.r25 * d3, td = B.rdust * dd;
        const double a = B.dtmp * exp(-(t1 + t2 + t3 + td));
#pragma unroll
        for (int r = 0; r < 3; ++r)
            if (B.excess[r] >= 0.0) { acc[r] = acc[r] + a; acc[3 + r] = acc[3 + r] + B.excess[r] * a; }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        tables[r * kTableSize + t] = acc[r];
        tables[(3 + r) * kTableSize + t] = acc[3 + r];
        logtab[logtab_index(r, t)] = log(acc[r]);
        logtab[logtab_index(r, t) + 1] = log(acc[3 + r]);
    }
}

__global__ void __launch_bounds__(256) log_table_kernel(const double *__restrict__ tables, double *__restrict__ logtab)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * kTableSize) return;
    const int r = t / kTableSize, flat = t - r * kTableSize;
    logtab[logtab_index(r, flat)] = log(tables[r * kTableSize + flat]);
    logtab[logtab_index(r, flat) + 1] = log(tables[(3 + r) * kTableSize + flat]);
}

// getRatesHydrogenHelium, equiSources.f90:4157-4311, on the table of logarithms: number and heating rate of one reaction.
// The reference forms, for the lower dust slab q = idust and the upper one,
//   v_q = c1 ((1-c3)(1-c2) T(i1+1,i2,i3) + c3 (1-c2) T(i1+1,i2,i3+1) + c2 (1-c3) T(i1+1,i2+1,i3) + c3 c2 T(i1+1,i2+1,i3+1))
//       + (1-c1) (the same four with i1)
// and returns exp((1-cd) v_0 + cd v_1); the same operations in the same order here.  Without dust cd = 0 and the upper slab
// contributes 0 * v_1 = 0 exactly: it is not read.
__device__ __forceinline__ void lookup_rates(const double *__restrict__ logtab, int dust, int reaction, double tau1, double tau2,
                                             double tau3, double taud, double &number_rate, double &heating_rate)
{
    if (tau1 > 10.0 || tau2 > 10.0 || tau3 > 10.0 || taud > 10.0) { number_rate = 0.0; heating_rate = 0.0; return; }
    const int i1 = (int)(tau1 / 10.0 * 10.0), i2 = (int)(tau2 / 10.0 * 10.0), i3 = (int)(tau3 / 10.0 * 10.0);
    const double c1 = tau1 * 10.0 / 10.0 - (double)i1, c2 = tau2 * 10.0 / 10.0 - (double)i2, c3 = tau3 * 10.0 / 10.0 - (double)i3;
    int id = 0;
    double cd = 0.0;
    if (dust != 0) { id = (int)(taud / 10.0 * 10.0); cd = taud * 10.0 / 10.0 - (double)id; }
    // at tau == 10 exactly the reference reads one past the table with weight 0; stay inside
    const int s1 = i1 < 10 ? 2 : 0, s2 = i2 < 10 ? 11 : 0, s3 = i3 < 10 ? 121 : 0;
    const double w00 = (1. - c3) * (1. - c2), w01 = c3 * (1. - c2), w10 = c2 * (1. - c3), w11 = c3 * c2;
    const int nslab = (dust != 0 && id < 10) ? 2 : 1;
    double vn[2] = {0.0, 0.0}, ve[2] = {0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (q < nslab) {
            const double *P = logtab + logtab_index(reaction - 1, (((id + q) * 11 + i3) * 11 + i2) * 11 + i1);
            // corner (i2, i3): {n(i1), e(i1), n(i1+1), e(i1+1)} are contiguous
            const double *A = P, *B = P + 2 * s3, *Cc = P + 2 * s2, *D = P + 2 * (s2 + s3);
            const double a0n = A[0], a0e = A[1], a1n = A[s1], a1e = A[s1 + 1];
            const double b0n = B[0], b0e = B[1], b1n = B[s1], b1e = B[s1 + 1];
            const double c0n = Cc[0], c0e = Cc[1], c1n = Cc[s1], c1e = Cc[s1 + 1];
            const double d0n = D[0], d0e = D[1], d1n = D[s1], d1e = D[s1 + 1];
            vn[q] = c1 * (w00 * a1n + w01 * b1n + w10 * c1n + w11 * d1n) + (1. - c1) * (w00 * a0n + w01 * b0n + w10 * c0n + w11 * d0n);
            ve[q] = c1 * (w00 * a1e + w01 * b1e + w10 * c1e + w11 * d1e) + (1. - c1) * (w00 * a0e + w01 * b0e + w10 * c0e + w11 * d0e);
        }
    }
    if (dust == 0) { number_rate = exp(vn[0]); heating_rate = exp(ve[0]); return; }
    if (nslab == 1) { vn[1] = vn[0]; ve[1] = ve[0]; } // dust index 10: no slab above, weight 0
    number_rate = exp((1. - cd) * vn[0] + cd * vn[1]);
    heating_rate = exp((1. - cd) * ve[0] + cd * ve[1]);
}

struct Neighbour { int node; double a, b; bool boundary; };

__device__ __forceinline__ int node_child0(const TraceRec &T, int c) { return T.node ? T.node[c].child0 : -1; }
__device__ __forceinline__ int node_level(const TraceRec &T, int c) { return T.node ? T.node[c].level : 0; }
__device__ __forceinline__ int node_parent(const TraceRec &T, int c) { return T.node ? T.node[c].parent : -1; }
__device__ __forceinline__ int node_leaf(const TraceRec &T, int c) { return T.node ? T.node[c].leaf : c; }

// zoom??Neighbour, equiSources.f90:2827-2960: down into the leaf that holds (a,b) on the face the ray crosses.
// axis: 0 the ray crosses an x face (coordinates y,z), 1 a y face (x,z), 2 a z face (x,y).
__device__ __forceinline__ Neighbour zoom(const TraceRec &T, int c, double a, double b, int axis, int side)
{
    for (int first = node_child0(T, c); first >= 0; first = node_child0(T, c)) {
        const int ia = a < 0.5 ? 0 : 1, ib = b < 0.5 ? 0 : 1;
        a = ia ? 2. * a - 1. : 2. * a;
        b = ib ? 2. * b - 1. : 2. * b;
        const int ic = side == 0 ? 1 : 0; // coming down through the face (side 0): the child on the far side
        int i, j, k;
        if (axis == 2) { i = ia; j = ib; k = ic; }
        else if (axis == 0) { i = ic; j = ia; k = ib; }
        else { i = ia; j = ic; k = ib; }
        c = first + 4 * i + 2 * j + k;
    }
    Neighbour N;
    N.node = c; N.a = a; N.b = b; N.boundary = false;
    return N;
}

// find??Neighbour, equiSources.f90:2647-2825
__device__ __forceinline__ Neighbour find_neighbour(const TraceRec &T, int c, double a, double b, int axis, int side)
{
    const int pos = axis == 2 ? 2 : (axis == 0 ? 0 : 1);
    const int pa = axis == 2 ? 0 : (axis == 0 ? 1 : 0), pb = axis == 2 ? 1 : 2;
    int lvl = node_level(T, c);
    while (lvl > 0) {
        const int par = node_parent(T, c);
        const int first = node_child0(T, par);
        const int idx = c - first;
        int ijk[3] = {(idx >> 2) & 1, (idx >> 1) & 1, idx & 1};
        if ((side == 0 && ijk[pos] == 0) || (side == 1 && ijk[pos] == 1)) {
            a = ijk[pa] == 0 ? 0.5 * a : 0.5 * a + 0.5;
            b = ijk[pb] == 0 ? 0.5 * b : 0.5 * b + 0.5;
            c = par;
            --lvl;
        } else {
            ijk[pos] = side == 0 ? 0 : 1;
            return zoom(T, first + 4 * ijk[0] + 2 * ijk[1] + ijk[2], a, b, axis, side);
        }
    }
    const int n = T.n;
    int ijk[3] = {c / (n * n), (c / n) % n, c % n};
    if ((side == 0 && ijk[pos] == 0) || (side == 1 && ijk[pos] == n - 1)) {
        Neighbour N;
        N.node = -1; N.a = a; N.b = b; N.boundary = true;
        return N;
    }
    ijk[pos] += side == 0 ? -1 : 1;
    return zoom(T, (ijk[0] * n + ijk[1]) * n + ijk[2], a, b, axis, side);
}

__global__ void __launch_bounds__(64) point_trace_kernel(const TraceRec T)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= T.nrays) return;
    const int L = T.pixel_level;
    const int n = T.n;
    const double fn = (double)(float)n;
    int pixel, cell, src;
    double pt[3], radius, d1, d2, d3, dd, ndot;
    const int level_off = 4 * ((1 << (2 * (L - 1))) - 1); // 12 (4^(L-1) - 1) / 3

    if (L == 1) {
        const int s = tid / 12;
        src = s;
        pixel = tid % 12;
        cell = T.src_node[s];
        pt[0] = pt[1] = pt[2] = 0.5;
        radius = 0.0; d1 = d2 = d3 = dd = 0.0;
        ndot = T.src_ndot[s] / 12.0;
    } else {
        const SplitRec R = T.in[tid >> 2];
        const int which = tid & 3;
        pixel = 4 * R.pixel + which;
        const double *pd = T.pixdir + 3 * (size_t)(4 * ((1 << (2 * (L - 2))) - 1) + R.pixel);
        double base[3] = {0, 0, 0};
        // the reference walks the four daughters in order and gives up on the rest once one starts outside the box
        // (strategy = boundary is never reset, equiSources.f90:3336-3345): daughter `which` exists only if it and all
        // its elder sisters start inside
        src = R.src;
        bool elder_outside = false;
        for (int sis = 0; sis <= which; ++sis) {
            const double *cd = T.pixdir + 3 * (size_t)(level_off + 4 * R.pixel + sis);
            base[0] = R.pos[0] + R.radius / fn * (cd[0] - pd[0]);
            base[1] = R.pos[1] + R.radius / fn * (cd[1] - pd[1]);
            base[2] = R.pos[2] + R.radius / fn * (cd[2] - pd[2]);
            const bool outside = base[0] < 0. || base[0] > 1. || base[1] < 0. || base[1] > 1. || base[2] < 0. || base[2] > 1.;
            if (outside && sis == which && T.escape) {
                // a daughter that starts outside the box counts as gone through the boundary, whatever her elder sisters did
                // (:3336-3343 is evaluated for every daughter)
                const double tmp = R.radius * T.box / (fn * T.kpc);
                double *E = T.escape + (size_t)src * kEscapeRec;
                for (int ir = 0; ir < kOutputRadii; ++ir)
                    if (T.out_radius_kpc[ir] > tmp) unsafeAtomicAdd(E + kOutputRadii + ir, R.ndot / 4.);
            }
            elder_outside = elder_outside || outside;
        }
        if (elder_outside) return;
        // localizeSplitContinuationCell, :3049-3118
        int ijk[3];
        for (int q = 0; q < 3; ++q) {
            ijk[q] = (int)(base[q] * n);
            if (ijk[q] < 0 || ijk[q] >= n) { atomicMax(T.error, 1); return; }
            pt[q] = base[q] * fn - (double)(float)ijk[q];
        }
        cell = (ijk[0] * n + ijk[1]) * n + ijk[2];
        for (int first = node_child0(T, cell); first >= 0; first = node_child0(T, cell)) {
            int h[3];
            for (int q = 0; q < 3; ++q) { h[q] = pt[q] < 0.5 ? 0 : 1; pt[q] = h[q] ? 2. * pt[q] - 1. : 2. * pt[q]; }
            cell = first + 4 * h[0] + 2 * h[1] + h[2];
        }
        radius = R.radius; d1 = R.depth[0]; d2 = R.depth[1]; d3 = R.depth[2]; dd = R.depth[3];
        ndot = R.ndot / 4.0;
    }
    if (pt[0] < 0. || pt[0] > 1. || pt[1] < 0. || pt[1] > 1. || pt[2] < 0. || pt[2] > 1.) { atomicMax(T.error, 2); return; }

    const double *dir = T.pixdir + 3 * (size_t)(level_off + pixel);
    const double prox = dir[0], proy = dir[1], proz = dir[2];
    const double rm = T.rmax[L];
    bool split = false;
    unsigned crossed = 0;
    for (int step = 0; step < 1000000; ++step) {
        ++crossed;
        // ---- drawSegment, :2412-2595
        const int lvl = node_level(T, cell);
        const double scale = (double)(1 << lvl);
        const double t1 = proz > 0. ? (1. - pt[2]) / proz : -pt[2] / proz;
        const double t2 = prox > 0. ? (1. - pt[0]) / prox : -pt[0] / prox;
        const double t3 = proy > 0. ? (1. - pt[1]) / proy : -pt[1] / proy;
        int axis;
        double len;
        if (t1 < fmin(t2, t3)) { axis = 2; len = t1; }
        else if (t2 < fmin(t1, t3)) { axis = 0; len = t2; }
        else { axis = 1; len = t3; }
        bool stop = false;
        Neighbour N;
        N.node = -1; N.a = N.b = 0.0; N.boundary = false;
        int side = 0;
        const double old_radius = radius;
        if (radius * scale + len < rm || L == kMaxPixelLevel) {
            radius = radius + len / scale;
            const double ex = pt[0] + len * prox, ey = pt[1] + len * proy, ez = pt[2] + len * proz;
            if (axis == 2) { side = proz < 0. ? 0 : 1; N = find_neighbour(T, cell, ex, ey, 2, side); }
            else if (axis == 0) { side = prox < 0. ? 0 : 1; N = find_neighbour(T, cell, ey, ez, 0, side); }
            else { side = proy < 0. ? 0 : 1; N = find_neighbour(T, cell, ex, ez, 1, side); }
            stop = N.boundary;
        } else if (radius * scale >= rm) {
            split = true; len = 0.0;
        } else {
            split = true;
            len = rm - radius * scale;
            radius = radius + len / scale;
            pt[0] = pt[0] + len * prox; pt[1] = pt[1] + len * proy; pt[2] = pt[2] + len * proz;
        }
        // ---- optical depths of the piece and what it absorbs, :3176-3269
        const double cell_size = T.box / ((double)((float)(1 << lvl) * (float)n));
        const double path = cell_size * len;
        const long c = node_leaf(T, cell);
        const double *M = T.medium + c * kCellRec; // HI, HeI, HeII, rho, abun2
        const double hi = M[0];
        const double tau1 = path * hi * (double)6.3e-18f, tau2 = path * M[1] * (double)7.42e-18f, tau3 = path * M[2] * (double)1.58e-18f;
        double taud = 0.0;
        if (T.dust == 1) taud = path * hi * (double)5.4116737e-22f * M[4] / (double)0.2f;
        else if (T.dust == 2) taud = path * (double)0.76f * M[3] / (double)1.6726231e-24f * (double)5.4116737e-22f * M[4] / (double)0.2f;
        if (T.escape) { // :3198-3233: what is left of the ray where it crosses the output radii, what leaves through the box faces
            double *E = T.escape + (size_t)src * kEscapeRec;
            const double tmp1 = old_radius * T.box / fn, tmp2 = radius * T.box / fn;
            for (int ir = 0; ir < kOutputRadii; ++ir) {
                const double tmp = T.out_radius_kpc[ir] * T.kpc;
                if (tmp >= tmp1 && tmp <= tmp2) {
                    const double ratio = (tmp - tmp1) / (tmp2 - tmp1);
                    unsafeAtomicAdd(E + ir, ndot * exp(-(ratio * (tau1 + taud) + d1 + dd)));
                    if (ir == kOutputRadii - 1) {
                        const double o1 = ratio * tau1 + d1, o2 = ratio * tau2 + d2, o3 = ratio * tau3 + d3, od = ratio * taud + dd;
                        unsafeAtomicAdd(E + 2 * kOutputRadii, ndot * exp(-od));
                        if (T.sigma_ratio)
                            for (int ie = 0; ie < kOutputEnergies; ++ie) {
                                const double e1 = T.sigma_ratio[ie] * o1, e2 = T.sigma_ratio[2 * kOutputEnergies + ie] * o2,
                                             e3 = T.sigma_ratio[kOutputEnergies + ie] * o3, e4 = T.sigma_ratio[3 * kOutputEnergies + ie] * od;
                                unsafeAtomicAdd(E + 2 * kOutputRadii + 1 + ie, ndot * exp(-(e1 + e2 + e3 + e4)));
                            }
                    }
                }
            }
            if (stop) { // drawSegment ended on a box face (the depth test below is not a boundary in this sense)
                const double tmp = radius * T.box / (fn * T.kpc);
                for (int ir = 0; ir < kOutputRadii; ++ir)
                    if (T.out_radius_kpc[ir] > tmp) unsafeAtomicAdd(E + kOutputRadii + ir, ndot);
            }
        }
        if (fmin(fmin(d1 + tau1, d2 + tau2), fmin(d3 + tau3, dd + taud)) > 100.) { stop = true; split = false; }
        // a species without opacity in this cell takes nothing from the ray: R(d) - R(d) = 0 (and the reference adds that 0)
        double a, b, ea, eb;
        double *K = T.rates + c * kCellRec; // krate24, krate25, krate26, crate24, crate25, crate26
        if (tau1 != 0.0) {
            lookup_rates(T.logtab, T.dust, 1, d1, d2, d3, dd, a, ea);
            lookup_rates(T.logtab, T.dust, 1, d1 + tau1, d2, d3, dd, b, eb);
            unsafeAtomicAdd(K + 0, ndot * (a - b));
            unsafeAtomicAdd(K + 3, ndot * (ea - eb));
        }
        if (tau2 != 0.0) {
            lookup_rates(T.logtab, T.dust, 2, d1, d2, d3, dd, a, ea);
            lookup_rates(T.logtab, T.dust, 2, d1, d2 + tau2, d3, dd, b, eb);
            unsafeAtomicAdd(K + 2, ndot * (a - b));
            unsafeAtomicAdd(K + 5, ndot * (ea - eb));
        }
        if (tau3 != 0.0) {
            lookup_rates(T.logtab, T.dust, 3, d1, d2, d3, dd, a, ea);
            lookup_rates(T.logtab, T.dust, 3, d1, d2, d3 + tau3, dd, b, eb);
            unsafeAtomicAdd(K + 1, ndot * (a - b));
            unsafeAtomicAdd(K + 4, ndot * (ea - eb));
        }
        d1 = d1 + tau1; d2 = d2 + tau2; d3 = d3 + tau3; dd = dd + taud;
        if (stop || split) break;
        // ---- into the neighbour, :2512-2558
        const double face = side == 0 ? 1. : 0.;
        if (axis == 2) { pt[2] = face; pt[0] = N.a; pt[1] = N.b; }
        else if (axis == 0) { pt[0] = face; pt[1] = N.a; pt[2] = N.b; }
        else { pt[1] = face; pt[0] = N.a; pt[2] = N.b; }
        if (pt[0] < 0. || pt[0] > 1. || pt[1] < 0. || pt[1] > 1. || pt[2] < 0. || pt[2] > 1.) { atomicMax(T.error, 3); return; }
        cell = N.node;
        if (step == 999999) atomicMax(T.error, 4);
    }
    atomicAdd(T.steps, (unsigned long long)crossed);
    if (!split) return;

    // ---- hand the ray over to its four daughters: absoluteCoordinates, :3011-3047
    atomicMax(T.highest_level, L + 1);
    double p[3] = {pt[0], pt[1], pt[2]};
    int c = cell;
    for (int lvl = node_level(T, c); lvl > 0; --lvl) {
        const int par = node_parent(T, c);
        const int idx = c - node_child0(T, par);
        const int h[3] = {(idx >> 2) & 1, (idx >> 1) & 1, idx & 1};
        for (int q = 0; q < 3; ++q) p[q] = h[q] == 0 ? 0.5 * p[q] : 0.5 * p[q] + 0.5;
        c = par;
    }
    const int bi[3] = {c / (n * n), (c / n) % n, c % n};
    const int slot = atomicAdd(T.out_count, 1);
    if (slot >= T.out_capacity) { atomicMax(T.error, 5); return; }
    SplitRec R;
    for (int q = 0; q < 3; ++q) R.pos[q] = ((double)(float)bi[q] + p[q]) / fn;
    R.radius = radius;
    R.depth[0] = d1; R.depth[1] = d2; R.depth[2] = d3; R.depth[3] = dd;
    R.ndot = ndot;
    R.pixel = pixel;
    R.src = src;
    T.out[slot] = R;
}

// getRatesHydrogenHelium for a list of depth tuples: tau[nsample][4] -> out[nsample][3][2] (number rate, heating rate)
__global__ void __launch_bounds__(256) rate_lookup_kernel(const double *__restrict__ logtab, int dust, int nsample,
                                                          const double *__restrict__ tau, double *__restrict__ out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsample) return;
    for (int r = 1; r <= 3; ++r) {
        double a, e;
        lookup_rates(logtab, dust, r, tau[4 * s], tau[4 * s + 1], tau[4 * s + 2], tau[4 * s + 3], a, e);
        out[6 * s + 2 * (r - 1)] = a;
        out[6 * s + 2 * (r - 1) + 1] = e;
    }
}

int launch_rate_lookup(const double *logtab, int dust, int nsample, const double *tau, double *out, hipStream_t stream)
{
    if (nsample <= 0) return 0;
    hipLaunchKernelGGL(rate_lookup_kernel, dim3((nsample + 255) / 256), dim3(256), 0, stream, logtab, dust, nsample, tau, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// packed[c][0..4] = HI, HeI, HeII, rho, abun2 (one 64-byte line per cell)
__global__ void __launch_bounds__(256) pack_medium_kernel(const double *__restrict__ HI, const double *__restrict__ HeI,
                                                          const double *__restrict__ HeII, const double *__restrict__ rho,
                                                          const double *__restrict__ abun2, double *__restrict__ packed, long ncell)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long c = t >> 3;
    const int f = (int)(t & 7);
    if (c >= ncell) return;
    const double *src = f == 0 ? HI : f == 1 ? HeI : f == 2 ? HeII : f == 3 ? rho : f == 4 ? abun2 : nullptr;
    packed[t] = src ? src[c] : 0.0;
}

// to_packed: packed[c][r] = planes[r][c] (r < 6), else planes[r][c] = packed[c][r]; tiles of 32 cells through LDS so that
// both sides move whole lines
__global__ void __launch_bounds__(256) repack_rates_kernel(double *__restrict__ planes, double *__restrict__ packed, long ncell,
                                                           int to_packed)
{
    __shared__ double tile[8][33];
    const long c0 = (long)blockIdx.x * 32;
    const int a = threadIdx.x & 31, b = threadIdx.x >> 5; // 32 x 8
    if (to_packed) {
        tile[b][a] = (b < 6 && c0 + a < ncell) ? planes[(long)b * ncell + c0 + a] : 0.0;
        __syncthreads();
        const int cc = threadIdx.x >> 3, r = threadIdx.x & 7;
        if (c0 + cc < ncell) packed[(c0 + cc) * kCellRec + r] = tile[r][cc];
    } else {
        const int cc = threadIdx.x >> 3, r = threadIdx.x & 7;
        tile[r][cc] = c0 + cc < ncell ? packed[(c0 + cc) * kCellRec + r] : 0.0;
        __syncthreads();
        if (b < 6 && c0 + a < ncell) planes[(long)b * ncell + c0 + a] = tile[b][a];
    }
}

int launch_pack_medium(const double *const field[5], double *packed, long ncell, hipStream_t stream)
{
    const long threads = ncell * kCellRec;
    hipLaunchKernelGGL(pack_medium_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, field[0], field[1], field[2],
                       field[3], field[4], packed, ncell);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_repack_rates(double *planes, double *packed, long ncell, bool to_packed, hipStream_t stream)
{
    hipLaunchKernelGGL(repack_rates_kernel, dim3((unsigned)((ncell + 31) / 32)), dim3(256), 0, stream, planes, packed, ncell,
                       to_packed ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_rate_table(const FreqBin *bins, int nbins, double *tables, double *logtab, hipStream_t stream)
{
    hipLaunchKernelGGL(rate_table_kernel, dim3((kTableSize + 255) / 256), dim3(256), 0, stream, bins, nbins, tables, logtab);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_log_table(const double *tables, double *logtab, hipStream_t stream)
{
    hipLaunchKernelGGL(log_table_kernel, dim3((3 * kTableSize + 255) / 256), dim3(256), 0, stream, tables, logtab);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_point_trace(const TraceRec &T, hipStream_t stream)
{
    if (T.nrays <= 0) return 0;
    hipLaunchKernelGGL(point_trace_kernel, dim3((T.nrays + 63) / 64), dim3(64), 0, stream, T);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Ionisation equilibrium per leaf: solveRateEquations, equiSources.f90:3459-3677.  The reference's arithmetic statement by
// statement (every operation is an IEEE add, multiply or divide; the logarithm of the temperature comes from the host),
// bisection on the electron density until HeI moves by less than 1e-10 of the helium density.
// ------------------------------------------------------------------------------------------------
struct ChemEq { double k1, k2, k3, k4, k5, k6, nh, nhe, kr24, kr25, kr26; };

__device__ __forceinline__ double chem_residual(const ChemEq &q, double de, double &HeI)
{
    // :3592-3596 (repeated at :3600-3604 and :3618-3622)
    // (the divisions through FTTE_DIV: the instruction sequence of an IEEE fp64 division without its range handling -- every
    // denominator here is a positive normal number far from the ends of the range: k de with de >= 1e-30, 1 + ..., -X/Y - 2 --,
    // same bits, a quarter fewer instructions; six divisions per evaluation, some forty evaluations per cell)
    const double X = q.k3 * de + q.kr26, Y = q.k4 * de;
    const double XY = FTTE_DIV(X, Y);
    const double HII = FTTE_DIV(q.nh, 1. + FTTE_DIV(q.k2 * de, q.k1 * de + q.kr24));
    HeI = FTTE_DIV(de - HII - 2. * q.nhe, XY - 2. - FTTE_DIV(2. * X, Y));
    const double HeII = FTTE_DIV(HeI * X, Y);
    return q.k3 * HeI * de + q.k6 * (q.nhe - HeI - HeII) * de + q.kr26 * HeI - HeII * (q.k4 * de + q.k5 * de + q.kr25);
}

__global__ void __launch_bounds__(256) rate_equations_kernel(const ChemRec R)
{
    // (the sweep's three statistics -- the largest change, the bisection steps, the first cell the reference would stop at -- are
    // combined per workgroup before they go to their single words in memory: a quarter of a million atomics on ONE address each
    // cost more than all the arithmetic of the kernel)
    __shared__ unsigned long long blk_change[4], blk_steps[4], blk_bad[4];
    unsigned long long my_change = 0ull, my_steps = 0ull, my_bad = ~0ull;
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < R.ncell; c += (long)gridDim.x * blockDim.x) { // (a few cells per thread)
    const double psi = (double)0.76f, mp = (double)1.6726231e-24f, mn = (double)1.67492728e-24f; // definitionsModule.f90:25-28, 261
    const double mh = mp, mhe = 2. * (mp + mn), pi = (double)3.141592654f;
    ChemEq q;
    const double rho = R.rho[c];
    q.nh = psi * rho / mh;
    q.nhe = (1. - psi) * rho / mhe;
    const double HI0 = R.HI[c], HeI0 = R.HeI[c], HeII0 = R.HeII[c];
    double HI = fmin(HI0, q.nh), HeI = HeI0, HeII = HeII0;
    if (q.nhe - HeI0 - HeII0 < 0. && HeII < 0.) HeII = 0.; // :3504-3513: only this survives of the branch
    // rates per cell -> per absorber, :3520-3542
    const double size = R.box / (double)((float)(1 << R.level[c]) * (float)R.n);
    const double vol = size * size * size;
    q.kr24 = (R.krate && HI > 0.) ? R.krate[c * kCellRec] / (vol * HI) : 0.;
    q.kr25 = (R.krate && HeII > 0.) ? R.krate[c * kCellRec + 1] / (vol * HeII) : 0.;
    q.kr26 = (R.krate && HeI > 0.) ? R.krate[c * kCellRec + 2] / (vol * HeI) : 0.;
    q.kr24 = fmax(q.kr24, 0.); q.kr25 = fmax(q.kr25, 0.); q.kr26 = fmax(q.kr26, 0.);
    if (R.run_uvb) { // :3545-3553
        const double t1 = 4. * pi * R.J[c], t2 = 4. * pi * R.J[R.ncell + c], t3 = 4. * pi * R.J[2 * R.ncell + c];
        q.kr24 = q.kr24 + t1 * R.ksi[0] + t2 * R.ksi[3] + t3 * R.ksi[6];
        q.kr25 = q.kr25 + t3 * R.ksi[7];
        q.kr26 = q.kr26 + t2 * R.ksi[5] + t3 * R.ksi[8];
    } else { // :3554-3562
        const double mfp = 1. / (HI * (double)6.3e-18f + HeI * (double)7.42e-18f + HeII * (double)1.58e-18f);
        if (mfp >= R.threshold) {
            q.kr24 = q.kr24 + 4. * pi * R.uniform[0];
            q.kr25 = q.kr25 + 4. * pi * R.uniform[1];
            q.kr26 = q.kr26 + 4. * pi * R.uniform[2];
        }
    }
    // rate coefficients at this temperature, :3568-3587
    double logtem = R.logtem[c];
    logtem = fmax(logtem, R.logtem0);
    logtem = fmin(logtem, R.logtem9);
    int ix = (int)((logtem - R.logtem0) / R.dlogtem) + 1;
    ix = ix < 1 ? 1 : ix;
    ix = ix > R.nratec - 1 ? R.nratec - 1 : ix;
    const double t1 = R.logtem0 + (double)(ix - 1) * R.dlogtem, t2 = R.logtem0 + (double)ix * R.dlogtem, tdef = t2 - t1;
    double kk[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        const double *ka = R.k + (size_t)r * R.nratec;
        kk[r] = ka[ix - 1] + (logtem - t1) * (ka[ix] - ka[ix - 1]) / tdef;
    }
    q.k1 = kk[0]; q.k2 = kk[1]; q.k3 = kk[2]; q.k4 = kk[3]; q.k5 = kk[4]; q.k6 = kk[5];

    // bisection on the electron density, :3589-3633
    double de1 = (double)1.e-30f, de2 = q.nh + 2. * q.nhe;
    double res1 = chem_residual(q, de1, HeI);
    double de = de2;
    (void)chem_residual(q, de2, HeI);
    double HeIprev = -1.;
    unsigned steps = 0;
    while (fabs(HeI - HeIprev) / q.nhe > 1.e-10 && steps < 4096u) {
        HeIprev = HeI;
        de = 0.5 * (de1 + de2);
        const double res = chem_residual(q, de, HeI);
        const bool opposite = (res > 0. && res1 < 0.) || (res < 0. && res1 > 0.); // :5044-5058
        if (opposite) de2 = de;
        else { de1 = de; res1 = res; }
        ++steps;
    }
    const double X = q.k3 * de + q.kr26, Y = q.k4 * de;
    HeII = HeI * X / Y;
    const double HII = q.nh / (1. + q.k2 * de / (q.k1 * de + q.kr24));
    HI = q.k2 * HII * de / (q.k1 * de + q.kr24);
    // where the reference prints the species and stops, :3637-3654
    const bool ok = (HI / q.nh >= 0. && HI / q.nh <= 1.) && (HeI / q.nhe >= 0. && HeI / q.nhe <= 1.) && steps < 4096u;
    if (!ok) my_bad = (unsigned long long)c < my_bad ? (unsigned long long)c : my_bad;
    else {
        // the reference's (unused) convergence measure, :3671-3674
        const double c1 = fabs(HI - HI0) * mh / (psi * rho), c2 = fabs(HeI - HeI0) * mhe / ((1. - psi) * rho),
                     c3 = fabs(HeII - HeII0) * mhe / ((1. - psi) * rho);
        const double change = fmax(c1, fmax(c2, c3));
        const unsigned long long bits = (unsigned long long)__double_as_longlong(change); // non-negative doubles order like their bits
        my_change = bits > my_change ? bits : my_change;
        my_steps += (unsigned long long)steps;
        R.HI_out[c] = HI; R.HeI_out[c] = HeI; R.HeII_out[c] = HeII;
    }
    }
    // wavefront, then workgroup, then one atomic each
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long oc = __shfl_xor(my_change, off), os = __shfl_xor(my_steps, off), ob = __shfl_xor(my_bad, off);
        my_change = oc > my_change ? oc : my_change;
        my_steps += os;
        my_bad = ob < my_bad ? ob : my_bad;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { blk_change[wave] = my_change; blk_steps[wave] = my_steps; blk_bad[wave] = my_bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            my_change = blk_change[k] > my_change ? blk_change[k] : my_change;
            my_steps += blk_steps[k];
            my_bad = blk_bad[k] < my_bad ? blk_bad[k] : my_bad;
        }
        if (my_bad != ~0ull) atomicMin(R.first_bad, my_bad);
        if (my_change) atomicMax(R.max_change, my_change);
        if (my_steps) atomicAdd(R.steps, my_steps);
    }
}

// assignUvbRadiation, transportRoutinesModule.f90:1056-1093: the optically thin alternative to the sweep.  J_g = uvb_g where the
// Lyman-limit mean free path of the cell is at least the self-shielding threshold, else 0.
__global__ void __launch_bounds__(256) thin_limit_kernel(const double *__restrict__ HI, const double *__restrict__ HeI,
                                                         const double *__restrict__ HeII, const double *__restrict__ rho,
                                                         const double *__restrict__ uvb, double threshold, double *__restrict__ J,
                                                         long ncell, int nnu)
{
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const double psi = (double)0.76f, mh = (double)1.6726231e-24f;
    const double hi = fmin(HI[c], psi * rho[c] / mh);
    const double mfp = 1. / (hi * (double)6.3e-18f + HeI[c] * (double)7.42e-18f + HeII[c] * (double)1.58e-18f);
    const bool lit = mfp >= threshold;
    for (int g = 0; g < nnu; ++g) J[(long)g * ncell + c] = lit ? uvb[g] : 0.0;
}

int launch_thin_limit(const double *HI, const double *HeI, const double *HeII, const double *rho, const double *uvb, double threshold,
                      double *J, long ncell, int nnu, hipStream_t stream)
{
    hipLaunchKernelGGL(thin_limit_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, stream, HI, HeI, HeII, rho, uvb, threshold, J,
                       ncell, nnu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_rate_equations(const ChemRec &R, hipStream_t stream)
{
    if (R.ncell <= 0) return 0;
    // (enough workgroups to fill the GPU several times over, few enough that their atomics are nothing)
    const long blocks = (R.ncell + 255) / 256;
    hipLaunchKernelGGL(rate_equations_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, R);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// kappa[g][c] = HI[c]*beta[0][g] + HeI[c]*beta[1][g] + HeII[c]*beta[2][g]   (equiSources.f90:4977-4980)
__global__ void __launch_bounds__(256) opacity_kernel(const double *__restrict__ HI, const double *__restrict__ HeI,
                                                      const double *__restrict__ HeII, const double *__restrict__ beta,
                                                      double *__restrict__ kappa, long ncell, int nnu)
{
    for (long c = (long)blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += (long)gridDim.x * blockDim.x) {
        const double a = HI[c], b = HeI[c], d = HeII[c];
        for (int g = 0; g < nnu; ++g) kappa[(long)g * ncell + c] = a * beta[g] + b * beta[nnu + g] + d * beta[2 * nnu + g];
    }
}

int launch_opacity(const double *HI, const double *HeI, const double *HeII, const double *beta, double *kappa, long ncell,
                   int nnu, hipStream_t stream)
{
    const int blocks = (int)((ncell + 255) / 256 < 8192 ? (ncell + 255) / 256 : 8192);
    hipLaunchKernelGGL(opacity_kernel, dim3(blocks), dim3(256), 0, stream, HI, HeI, HeII, beta, kappa, ncell, nnu);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- the pieces of J that the devices of one multi-device context swept for different directions, summed (ftte_multi.cpp)
struct SumParts { const double *part[64]; int nparts; };

__global__ void __launch_bounds__(256) sum_parts_kernel(const SumParts P, double *out, long count)
{
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    double sum = P.part[0][q];
    for (int k = 1; k < P.nparts; ++k) sum += P.part[k][q];
    out[q] = sum;
}

int launch_sum_parts(const double *const *parts, int nparts, double *out, long count, hipStream_t stream)
{
    if (nparts < 1 || nparts > 64 || count < 0) return -1;
    if (!count) return 0;
    SumParts P;
    for (int k = 0; k < nparts; ++k) P.part[k] = parts[k];
    P.nparts = nparts;
    hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, P, out, count);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // namespace ftte
