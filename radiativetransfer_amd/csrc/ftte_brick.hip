// ftte_brick.hip -- the cell-fixed brick organisation of the uniform-grid diffuse sweep (CDNA4, gfx950).
//
// Same computation as ftte::sweep_kernel (ftte_kernels.hip): the reference's cell transfer,
//   transportRoutinesModule.f90:587-961  /  equiSources.f90:1580-1788,
// same segment arithmetic (ftte_math.h), organised so that the directions of one izone share every opacity load and
// every J store, and no cell is computed twice.
#include <hip/hip_runtime.h>

#include "ftte_internal.h"
#include "ftte_kernels.h"
#include "ftte_math.h"

namespace ftte {

using gcbyte = const __attribute__((address_space(1))) char;
using gbyte = __attribute__((address_space(1))) char;
using gcdouble = const __attribute__((address_space(1))) double;
using gdouble = __attribute__((address_space(1))) double;

__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// A wave-uniform address the compiler is to keep in scalar registers and use as the base of a load or store whose lane part is a
// 32-bit offset (`global_load v, v_offset, s[base]`): left to itself it folds the lane's offset into a 64-bit vector address that it
// then carries, and advances, in vector registers and vector instructions.
// (the base passes through an empty asm and stays scalar; the caller passes the offset through `here()` once per run of accesses,
// so that its extension to 64 bits stays next to them, where instruction selection can fold it, instead of being hoisted out of
// the loop as a 64-bit vector value)
__device__ __forceinline__ gcbyte *lane_address(gcbyte *base, unsigned offset) { asm volatile("" : "+s"(base)); return base + offset; }
__device__ __forceinline__ gbyte *lane_address(gbyte *base, unsigned offset) { asm volatile("" : "+s"(base)); return base + offset; }
__device__ __forceinline__ unsigned here(unsigned offset) { asm volatile("" : "+v"(offset)); return offset; }
__device__ __forceinline__ long here(long step) { asm volatile("" : "+s"(step)); return step; }

// One value per row of a brick for this lane: rows `step` bytes apart from `row` on, the lane `off` bytes into its row.  A whole
// brick (nrows == ROWS) walks the rows with one scalar addition each; the ragged last brick of a grid that is no multiple of the
// brick repeats its last row's address for the rows it lacks.
template <int ROWS, bool NT>
__device__ __forceinline__ void load_rows(double (&dst)[ROWS], gcbyte *row, unsigned off, long step, int nrows)
{
    if (nrows == ROWS) {
        off = here(off);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            dst[r] = NT ? __builtin_nontemporal_load((gcdouble *)lane_address(row, off)) : *(gcdouble *)lane_address(row, off);
            row += step;
        }
    } else {
        off = here(off);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            dst[r] = NT ? __builtin_nontemporal_load((gcdouble *)lane_address(row, off)) : *(gcdouble *)lane_address(row, off);
            row += r + 1 < nrows ? step : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Cell-fixed bricks (ftte_internal.h: BrickGroup, BrickTask, BrickLaunch).
//
// The tile kernel above follows the rays: a wave's cells drift with its direction, so nothing can be shared between
// directions and every update costs a kappa load and a J read-modify-write of its own (24 B), plus a recomputed halo.
// Here the cells stay put and the rays move through them:
//   * lane l of the wave owns the column u = 64 tu + l + 1, registers hold kBrickRows rows of it;
//   * per layer the opacity of the brick's cells is loaded ONCE and the cells' J contribution stored ONCE for all
//     the directions of the group (directions of one izone: same memory frame, same sweep order);
//   * a ray whose next segment lies in the next column moves one lane up (a DPP shift; lane 0 takes the ray handed
//     over by the brick to the left, lane 63 hands its ray to the brick to the right), one whose next segment lies in
//     the next row moves one register up (row 0 takes it from the brick below, the top row hands it on): every segment
//     is computed by the lane that owns its cell, so a cell's mean needs no exchange and nothing is computed twice;
//   * the ray state of the group's other directions waits in LDS (8 doubles per lane and direction, one slot fewer than
//     there are directions) while one direction crosses the layer: one code path, no barrier, one wavefront per workgroup.
// Faces in memory are rings over two chunks (a brick's consumers run exactly one stage later).
// Arithmetic, segment order and the order of a cell's sum are those of the tile kernel (ftte_math.h): same bits.
// ------------------------------------------------------------------------------------------------

// value held by lane-1; lane 0, which has no lane below, receives in0 (bound_ctrl off: lanes without a source keep `old`)
__device__ __forceinline__ double shift_up_inject(double x, double in0)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    const int ilo = __double2loint(in0), ihi = __double2hiint(in0);
    lo = __builtin_amdgcn_update_dpp(ilo, lo, 0x138, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(ihi, hi, 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// A value another workgroup of THIS launch has stored: past the CU's vector L1, which no other CU's stores refresh (`sc1`: served
// by the L2).
__device__ __forceinline__ double fresh(gcbyte *p) { return __hip_atomic_load((gcdouble *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One direction crosses one layer of the brick.  cur[r]: the ray entering cell (row r, this lane) through its bottom,
// on return the ray leaving through its top.  uin/vin: where the rays handed over by the brick to the left / below wait
// (nullptr: the domain boundary, the inflow enters there); uout/vout: where this brick's leaving rays go (nullptr: they
// leave the domain or nobody is there).
template <int EMIT>
__device__ __forceinline__ double brick_segment(const ftte_consts &K, double lead, double &I, double kap, double x, double dpath)
{
    if (EMIT == 0) return ftte_segment_lead(&K, lead, &I, kap * dpath);
    if (EMIT == 2) return ftte_segment_source(&K, lead, &I, kap * dpath, x); // a source function: the exact path mean
    return ftte_segment_emit(&K, &I, kap * dpath, x, 0.0);              // the reference's emissivity term and its log-mean
}

template <int SHAPE, int EMIT, int RW = kBrickRows>
__device__ __forceinline__ void brick_step(const ftte_consts &K, double lead, double (&cur)[RW], const double (&kap)[RW],
                                           const double (&xs)[EMIT ? RW : 1],
                                           double (&Jacc)[RW], bool third_first, double d0, double d1, double d2,
                                           double w, double uvb, gcbyte *uin, gbyte *uout, gcbyte *vin, gbyte *vout, int lane, bool through = false,
                                           bool same_launch = false, int take_lane = 0, int hand_lane = 63, bool active = true,
                                           const double *carry_in = nullptr, double *carry_out = nullptr)
{
    // take_lane / hand_lane / active: a brick that sweeps only the lanes take_lane .. hand_lane (hybrid sweep of a refined cell
    // array: the others belong to the segment forest).  The rays waiting in `uin` then enter lane take_lane, lane hand_lane hands
    // its rays to `uout`, and only active lanes write to the v-face.  Defaults: the whole brick, nothing added to the code.
    constexpr bool HAS_U = SHAPE == RC_TWO_U || SHAPE == RC_THREE_U || SHAPE == RC_THREE_V;
    constexpr bool HAS_V = SHAPE == RC_TWO_V || SHAPE == RC_THREE_U || SHAPE == RC_THREE_V;
    double ui[RW];
    double carry = uvb; // the ray that moves up from the row below (row 0: from the brick below)
    if (HAS_U) {
#pragma unroll
        for (int r = 0; r < RW; ++r) ui[r] = uvb;
        if (uin) { // one address for the whole wave
            if (same_launch) { // dataflow: written by a brick of this launch, so not through the scalar cache
#pragma unroll
                for (int r = 0; r < RW; ++r) ui[r] = fresh(uin + 8 * r);
            } else { // written by an earlier launch: scalar loads, the values wait in scalar registers
                const __attribute__((address_space(4))) double *us = (const __attribute__((address_space(4))) double *)(unsigned long)uin;
#pragma unroll
                for (int r = 0; r < RW; ++r) ui[r] = us[r];
            }
        }
    }
    if (HAS_V && vin) carry = same_launch ? fresh(vin + 8 * lane) : *(gcdouble *)(vin + 8 * lane);
    if (HAS_V && carry_in) carry = carry_in[lane]; // pair kernel: the wavefront below left it in LDS
    const bool hands_u = HAS_U && uout != nullptr && lane == hand_lane;

#pragma unroll
    for (int r = 0; r < RW; ++r) {
        double I = cur[r];
        const double m0 = brick_segment<EMIT>(K, lead, I, kap[r], xs[EMIT ? r : 0], d0); // xy piece, in the ray's own cell
        double acc = m0;
        if (SHAPE == RC_ONE) {
            cur[r] = I;
            Jacc[r] += ftte_cell_mean(acc, 1, w);
        } else if (SHAPE == RC_TWO_U) {
            if (hands_u) { if (through) __hip_atomic_store((double *)(uout + 8 * r), I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(gdouble *)(uout + 8 * r) = I; }
            I = shift_up_inject(I, ui[r]);
            if (take_lane) I = lane == take_lane ? ui[r] : I;
            acc += brick_segment<EMIT>(K, lead, I, kap[r], xs[EMIT ? r : 0], d1);
            cur[r] = I;
            Jacc[r] += ftte_cell_mean(acc, 2, w);
        } else if (SHAPE == RC_TWO_V) {
            double b = carry;
            carry = I;
            acc += brick_segment<EMIT>(K, lead, b, kap[r], xs[EMIT ? r : 0], d1);
            cur[r] = b;
            Jacc[r] += ftte_cell_mean(acc, 2, w);
        } else if (SHAPE == RC_THREE_U) { // 2nd piece one column on, 3rd one row on
            if (hands_u) { if (through) __hip_atomic_store((double *)(uout + 8 * r), I, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(gdouble *)(uout + 8 * r) = I; }
            I = shift_up_inject(I, ui[r]);
            if (take_lane) I = lane == take_lane ? ui[r] : I;
            const double m1 = brick_segment<EMIT>(K, lead, I, kap[r], xs[EMIT ? r : 0], d1);
            double c = carry;
            carry = I;
            const double m2 = brick_segment<EMIT>(K, lead, c, kap[r], xs[EMIT ? r : 0], d2);
            // reference order: xy + xz + yz, whatever the chain order (transportRoutinesModule.f90:695-941)
            if (third_first) { // one layer record for the wavefront: a scalar branch instead of four selects
                acc += m2;
                acc += m1;
                asm volatile("" : "+v"(acc));
            } else {
                acc += m1;
                acc += m2;
            }
            cur[r] = c;
            Jacc[r] += ftte_cell_mean(acc, 3, w);
        } else { // RC_THREE_V: 2nd piece one row on, 3rd one column on
            double b = carry;
            carry = I;
            const double m1 = brick_segment<EMIT>(K, lead, b, kap[r], xs[EMIT ? r : 0], d1);
            if (hands_u) { if (through) __hip_atomic_store((double *)(uout + 8 * r), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *(gdouble *)(uout + 8 * r) = b; }
            b = shift_up_inject(b, ui[r]);
            if (take_lane) b = lane == take_lane ? ui[r] : b;
            const double m2 = brick_segment<EMIT>(K, lead, b, kap[r], xs[EMIT ? r : 0], d2);
            if (third_first) { // one layer record for the wavefront: a scalar branch instead of four selects
                acc += m2;
                acc += m1;
                asm volatile("" : "+v"(acc));
            } else {
                acc += m1;
                acc += m2;
            }
            cur[r] = b;
            Jacc[r] += ftte_cell_mean(acc, 3, w);
        }
        asm volatile("" : "+v"(Jacc[r]));
        __builtin_amdgcn_sched_barrier(0); // rows in program order: interleaved they multiply the live registers
    }
    if (HAS_V && carry_out) carry_out[lane] = carry; // pair kernel: for the wavefront above
    if (HAS_V && vout && active) { // the top row's ray goes on in the brick above
        if (through) __hip_atomic_store((double *)(vout + 8 * lane), carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *(gdouble *)(vout + 8 * lane) = carry;
    }
}

// grid: ntasks * nnu workgroups of one wavefront; dynamic LDS: 4 KB per direction of the largest group
// MASKED: the tasks sweep a range of lanes only (BrickTask::tu and ::group carry it): the bricks of the hybrid sweep that a box of
// the segment forest cuts through; where the box's u-faces lie inside a brick, rays cross them through two face rings of their own.
// FLOW: 0 a launch per stage; 1 one launch, a workgroup per brick, tickets from one counter (cross-XCD hand-overs: write-through
// stores or an L2 write-back per brick); 2 one launch of persistent workgroups, a queue per XCD (BrickLaunch::queue).
template <int WAVES, int EMIT, int FLOW, bool MASKED = false>
__global__ void __launch_bounds__(64, WAVES) brick_kernel(const BrickLaunch L)
{
    extern __shared__ double state[]; // [slot][row][lane]: the rays of the directions that are not in registers
    constexpr int R = kBrickRows;
    using cgroup = const __attribute__((address_space(4))) BrickGroup;
    using clayer = const __attribute__((address_space(4))) LayerRec;
    const int nnu = L.nnu;
    const int lane = threadIdx.x;
    unsigned work = blockIdx.x;
    int queue = 0;
    unsigned long long waited = 0, began = 0; // polls this workgroup spent waiting; when it started (instrumentation)
    if (FLOW == 2) began = (unsigned long long)wall_clock64();
    if (FLOW == 2) { // which XCD is this?  Its queue holds the bricks whose inputs are written behind this XCD's L2
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        queue = L.xcc_queue[xcc & 15u];
        if (queue < 0) return;
    }
  do {
    if (FLOW == 1) { // dataflow: tasks are taken in list order, whatever order the workgroups start in
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(L.ticket, 1u);
        work = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (work >= (unsigned)L.ntasks * (unsigned)nnu) return;
    }
    if (FLOW == 2) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(L.ticket + 32 * queue, 1u);
        t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= L.qlen[queue]) { // the queue is empty; what this workgroup leaves behind is instrumentation (ftte_counter, FTTE_QUEUE_STATS)
            if (lane == 0) {
                uint32_t *stats = L.ticket + 32 * queue;
                atomicMax((unsigned long long *)(stats + 2), (unsigned long long)wall_clock64());
                atomicAdd((unsigned long long *)(stats + 4), waited);
                atomicAdd(stats + 6, 1u);
                atomicMax((unsigned long long *)(L.error + 2), ~began);
            }
            return;
        }
        work = L.queue[L.qoff[queue] + t];
    }
    // dataflow with write-through stores (sc1: the line goes to memory and leaves this XCD's L2) instead of an L2 write-back
    // before the flag
    const bool through = FLOW == 1 && L.pad_ != 0;
    const int nu = L.nu0 + (int)(work % (unsigned)nnu);
    const unsigned task_index = work / (unsigned)nnu;
    // (the task list, the inflow and the group records are read-only for the launch and wave-uniform: scalar loads, address space 4)
    BrickTask T;
    {
        const unsigned long raw = ((const __attribute__((address_space(4))) unsigned long *)(unsigned long)L.tasks)[task_index];
        T.group = (int16_t)(raw & 0xffffu); T.tu = (int16_t)((raw >> 16) & 0xffffu); T.tv = (int16_t)((raw >> 32) & 0xffffu); T.ti = (int16_t)(raw >> 48);
    }
    const int tu_field = uniform((int)T.tu);
    const int tu = MASKED ? tu_field & kBrickTuMask : tu_field;
    const int group_field = uniform((int)T.group);
    const int lane_lo = MASKED ? (tu_field >> kBrickLaneLoShift) & 63 : 0;
    const int lane_hi = MASKED ? (group_field >> kBrickLaneHiShift) & 63 : 63;
    const int tv_field = uniform((int)T.tv);
    const int tv = MASKED ? tv_field & kBrickTvMask : tv_field, box = MASKED ? (tv_field >> kBrickBoxShift) & kBrickBoxMask : 0;
    const int ti = uniform((int)T.ti) & (kBrickAccumulate - 1);
    const bool accumulate = (uniform((int)T.ti) & kBrickAccumulate) != 0 && !(L.pad2_ & 4);
    const bool atomic_acc = L.atomic_acc != 0 && !through;
    cgroup *G = (cgroup *)(L.groups + (MASKED ? group_field & kBrickGroupMask : group_field));
    const int n = L.n, chunk = L.chunk, up = L.up, vp = L.vp;
    const int ndir = G->ndir;
    const double uvb = ((const __attribute__((address_space(4))) double *)(unsigned long)L.uvb)[nu];
    double lead = L.math.c[9]; // the exponential's leading coefficient, in a vector register for the whole run (ftte_math.h)
    asm volatile("" : "+v"(lead));

    const int sv = G->sv, si = G->si;
    const bool mirror_u = G->su < 0;
    const long org = G->org;
    gcbyte *kbase = (gcbyte *)(G->kappa + (long)nu * L.group_stride + org);
    gbyte *jbase = (gbyte *)(G->J + (long)nu * L.group_stride + org);
    gcbyte *xbase = EMIT ? (gcbyte *)(G->emis + (long)nu * L.group_stride + org) : nullptr;

    // this lane's column and the brick's rows (1-based cell indices; clamped copies for loads of a ragged last brick)
    const int cu = 64 * tu + lane + 1;
    const int cv0 = R * tv + 1;
    const int cuc = cu < n ? cu : n;
    const bool tiled = L.tiled != 0; // brick-ordered kappa and J (BrickLaunch::tiled): the lane's place in its row of the brick
    const unsigned off0 = tiled ? 8u * (unsigned)(mirror_u ? 63 - lane : lane) : 8u * (unsigned)(mirror_u ? n + 1 - cuc : cuc);
    const bool own_lane = cu <= n && (!MASKED || (lane >= lane_lo && lane <= lane_hi));
    const long row_bytes = 8l * sv;
    const int i0 = ti * chunk + 1;
    const int i1 = (i0 + chunk - 1 < n) ? i0 + chunk - 1 : n;

    // (pad2_: diagnostic option "ablate" -- WRONG RESULTS, timing only: bit 0 no rays taken from the left and from below, bit 1 none
    // handed to the right and above, bit 2 no earlier J read, bit 3 no rays taken from or left for the chunks before and after,
    // bit 4 the opacities of a brick's first layer for all its layers, bit 5 no J stored)
    const int ablate = L.pad2_;
    const bool sub = !MASKED && L.sub != 0; // a sub-grid with neighbours all round (BrickLaunch::sub): rings at its faces too
    const bool has_u_in = (tu > 0 || lane_lo > 0 || sub) && !(ablate & 1), has_u_out = (64 * (tu + 1) < n || lane_hi < 63 || sub) && !(ablate & 2);
    const bool has_v_in = (tv > 0 || sub) && !(ablate & 1), has_v_out = (R * (tv + 1) < n || sub) && !(ablate & 2);
    const bool has_i_in = (ti > 0 || sub) && !(ablate & 8), has_i_out = (i1 < n || sub) && !(ablate & 8);
    const long fnu = (long)nu * L.face_stride;
    // element offsets inside a direction's face block (ftte_internal.h)
    const int uw = L.uw, ut = L.ut;
    const int ns = L.nslot, sl = ti % ns;
    // masked bricks: lanes that end inside the brick hand their rays to the near u-face of the box there (ring 2 * box at
    // uqface_off), lanes that start inside it take theirs from that box's far u-face (ring 2 * box + 1)
    const long u_out = lane_hi < 63 ? L.uqface_off + ((long)((2 * box) * ns + sl) * chunk) * uw + ut * tv : ((long)(tu * ns + sl) * chunk) * uw + ut * tv;
    const long u_in = lane_lo > 0 ? L.uqface_off + ((long)((2 * box + 1) * ns + sl) * chunk) * uw + ut * tv
                                  : ((long)((tu > 0 ? tu - 1 : up / 64) * ns + sl) * chunk) * uw + ut * tv; // (tu = 0, sub-grid: ring ntu)
    const long v_out = L.vface_off + ((long)(tv * ns + sl) * chunk) * up + 64 * tu;
    const long v_in = L.vface_off + ((long)((tv > 0 ? tv - 1 : vp / R) * ns + sl) * chunk) * up + 64 * tu;   // (tv = 0, sub-grid: ring ntv)
    const long i_in = L.iface_off + ((long)sl * vp + R * tv) * up + 64 * tu + lane;
    const long i_out = L.iface_off + ((long)((ti + 1) % ns) * vp + R * tv) * up + 64 * tu + lane;

    if (FLOW) {
        // Wait for the bricks this one depends on.  They come earlier in the list, so workgroups that started before this one
        // hold them: no waiting cycle.  Bounded all the same: after about a second without progress the sweep is given up
        // (error flag; every later brick gives up too) rather than left hanging.
        const int32_t *dep = L.deps + (size_t)task_index * kBrickDeps;
        const unsigned slot = (unsigned)(nu - L.nu0);
        bool failed = false;
        // (the bound is on time WITHOUT PROGRESS: every 1024 polls the wavefront looks at the ticket counter it draws from, and
        // starts counting again when that has moved -- somebody has finished a brick meanwhile)
        const uint32_t *progress = FLOW == 2 ? L.ticket + 32 * queue : L.ticket;
        unsigned seen = 0;
        // lane q looks at dependency q: one round trip for all six while they are done, which is the rule
        bool all_done;
        {
            const int32_t dq = lane < kBrickDeps ? dep[lane] : -1;
            const bool ok = dq < 0 || __hip_atomic_load(L.done + (size_t)(dq < 0 ? 0 : dq) * nnu + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == L.epoch;
            all_done = __builtin_amdgcn_ballot_w64(!ok) == 0;
        }
        for (int q = 0; q < kBrickDeps && !failed && !all_done; ++q) {
            const int32_t dq = uniform(dep[q]);
            if (dq < 0) continue;
            const uint32_t *flag = L.done + (size_t)dq * nnu + slot;
            unsigned spins = 0;
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != L.epoch) {
                __builtin_amdgcn_s_sleep(20);
                ++waited;
                if ((++spins & 1023u) == 0) {
                    const unsigned now = __hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (now != seen) { seen = now; spins = 0; }
                    if (spins >= (1u << 20) || __hip_atomic_load(L.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        failed = true;
                        break;
                    }
                }
            }
        }
        if (failed) {
            if (lane == 0) __hip_atomic_store(L.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (FLOW == 1) { // hand-overs from any XCD: this CU's L1 is emptied, the loads that follow are plain
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // (FLOW == 2: every load of handed-over bytes below goes past the L1 -- `fresh`, and the non-temporal loads of J)
    }

    // The rays of one direction are in registers (`cur`), those of the group's other directions wait in LDS: ndir - 1 slots of
    // 4 KB.  Layer i starts with direction p(i) = -(i - 1) mod ndir in registers and direction p + 1 + k in slot k, crosses
    // p, p + 1, ..., each time taking the next direction out of its slot and parking the one just done there, and so ends with
    // p - 1 = p(i + 1) in registers and p + k in slot k: the arrangement layer i + 1 starts from.  (A slot per direction would be
    // simpler and costs a quarter of the wavefronts a CU can hold at three directions.)  The order in which a layer's
    // directions are added into a cell's J follows p(i): fixed by the layer, the same for every chunk length.
    int p0 = (ndir - (i0 - 1) % ndir) % ndir;
    // the opacity of the brick's cells, one layer ahead of the layer being crossed: the first layer's asked for before anything else
    double kap_next[R];
    // (row addresses are walked with scalar additions; the rows a ragged last brick lacks repeat its last row's address for loads
    // and are skipped by stores)
    const int nrows = n - cv0 + 1 < R ? n - cv0 + 1 : R;
    const long row0 = tiled ? 8l * ((long)tu * G->bu + (long)tv * G->bv + (long)ti * G->bi) : cv0 * row_bytes;
    load_rows<R, false>(kap_next, kbase + 8l * i0 * si + row0, off0, row_bytes, nrows);
    double cur[R];
    // rays entering the brick's bottom: the inflow, or what the chunk below left
    {
        gcdouble *f = (gcdouble *)(G->dir[p0].faces + fnu);
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = !has_i_in ? uvb : FLOW == 2 ? fresh((gcbyte *)&f[i_in + (long)r * up]) : f[i_in + (long)r * up];
    }
    for (int k = 0; k + 1 < ndir; ++k) {
        int d = p0 + 1 + k;
        d = d >= ndir ? d - ndir : d;
        gcdouble *f = (gcdouble *)(G->dir[d].faces + fnu);
        // (all eight loads in flight, then the eight LDS stores: left to itself the compiler pairs them, load, load, wait, store,
        // wait, store -- four round trips to memory per direction where one will do, and a brick is a short thing)
        double parked[R];
#pragma unroll
        for (int r = 0; r < R; ++r) parked[r] = !has_i_in ? uvb : FLOW == 2 ? fresh((gcbyte *)&f[i_in + (long)r * up]) : f[i_in + (long)r * up];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; ++r) state[(k * R + r) * 64 + lane] = parked[r];
    }
    for (int i = i0; i <= i1; ++i) {
        const int il = i - i0;
        double kap[R], xs[EMIT ? R : 1], Jacc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { kap[r] = kap_next[r]; Jacc[r] = 0.0; if (EMIT) { const int row = (cv0 + r < n) ? cv0 + r : n; xs[r] = *(gcdouble *)(xbase + 8l * i * si + row * row_bytes + off0); } }
        gbyte *jplane = jbase + 8l * i * si;
        const long rstep = here(row_bytes); // (per layer: the seven steps of a ragged brick are then not kept, and spilled, for the whole run)
        // what the groups before this one left in these cells: read now and added to (the wavefront waits for it at its first sum), or
        // -- BrickLaunch::atomic_acc -- left where it is and added to by the memory system when the layer's sums are stored
        if (accumulate && !atomic_acc) load_rows<R, true>(Jacc, (gcbyte *)jplane + row0, off0, rstep, nrows);
        if (i < i1 && !(ablate & 16)) load_rows<R, false>(kap_next, kbase + 8l * (i + 1) * si + row0, off0, rstep, nrows);
        for (int j = 0; j < ndir; ++j) {
            int d = p0 + j;
            d = d >= ndir ? d - ndir : d;
            if (j) { // direction d leaves slot j - 1, the one just done takes its place
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double parked = state[((j - 1) * R + r) * 64 + lane];
                    state[((j - 1) * R + r) * 64 + lane] = cur[r];
                    cur[r] = parked;
                }
            }
            clayer *rp = (clayer *)(G->dir[d].layers) + (i - 1);
            const double d0 = rp->dpath[0], d1 = rp->dpath[1], d2 = rp->dpath[2];
            const int rc = rp->info & 7;
            const double w = G->dir[d].w;
            gbyte *f = (gbyte *)(G->dir[d].faces + fnu);
            gcbyte *uin = has_u_in ? (gcbyte *)f + 8 * (u_in + (long)il * uw) : nullptr;
            gbyte *uout = has_u_out ? f + 8 * (u_out + (long)il * uw) : nullptr;
            gcbyte *vin = has_v_in ? (gcbyte *)f + 8 * (v_in + (long)il * up) : nullptr;
            gbyte *vout = has_v_out ? f + 8 * (v_out + (long)il * up) : nullptr;
            const bool third_first = rc == RC_THREE_U_SWAP || rc == RC_THREE_V_SWAP;
            switch (rc) {
            case RC_ONE: brick_step<RC_ONE, EMIT>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, through, FLOW != 0, lane_lo, lane_hi, !MASKED || own_lane); break;
            case RC_TWO_U: brick_step<RC_TWO_U, EMIT>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, through, FLOW != 0, lane_lo, lane_hi, !MASKED || own_lane); break;
            case RC_TWO_V: brick_step<RC_TWO_V, EMIT>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, through, FLOW != 0, lane_lo, lane_hi, !MASKED || own_lane); break;
            case RC_THREE_U:
            case RC_THREE_U_SWAP:
                brick_step<RC_THREE_U, EMIT>(L.math, lead, cur, kap, xs, Jacc, third_first, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, through, FLOW != 0, lane_lo, lane_hi, !MASKED || own_lane);
                break;
            default:
                brick_step<RC_THREE_V, EMIT>(L.math, lead, cur, kap, xs, Jacc, third_first, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, through, FLOW != 0, lane_lo, lane_hi, !MASKED || own_lane);
                break;
            }
        }
        p0 = p0 ? p0 - 1 : ndir - 1;
        // the group's contribution to J of this layer's cells: stored once, read only by the merge
        if (own_lane && !(ablate & 32)) {
            gbyte *jrow = jplane + row0;
            const unsigned off = here(off0);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (r < nrows) {
                    if (through) __hip_atomic_store((double *)(jrow + off0), Jacc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else if (accumulate && atomic_acc) (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble *)lane_address(jrow, off), Jacc[r]);
                    else __builtin_nontemporal_store(Jacc[r], (gdouble *)lane_address(jrow, off));
                }
                jrow += rstep;
            }
        }
    }

    if (has_i_out) { // p0 is in registers, p0 + 1 + k in slot k
        for (int j = 0; j < ndir; ++j) {
            int d = p0 + j;
            d = d >= ndir ? d - ndir : d;
            gdouble *f = (gdouble *)(G->dir[d].faces + fnu);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double leaving = j ? state[((j - 1) * R + r) * 64 + lane] : cur[r];
                if (MASKED && !own_lane) continue; // the forest's lanes: its own rays wait there
                if (through) __hip_atomic_store((double *)&f[i_out + (long)r * up], leaving, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else f[i_out + (long)r * up] = leaving;
            }
        }
    }
    if (FLOW) {
        // publish: every store of this wavefront drained (FLOW == 2: they are in this XCD's L2, where every reader of them looks),
        // FLOW == 1 without write-through stores: the XCD's L2 written back; then the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (FLOW == 1 && !through) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) {
            uint32_t *flag = L.done + (size_t)task_index * nnu + (unsigned)(nu - L.nu0);
            if (FLOW == 2) *(volatile uint32_t *)flag = L.epoch; // a plain store: the line stays in this L2 for the pollers
            else __hip_atomic_store(flag, L.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
  } while (FLOW == 2);
}

// The same brick swept by a PAIR of wavefronts: wave 0 the brick's lower four rows, wave 1 the upper four, one layer apart.
// A ray that moves up out of row 3 waits in LDS for the wave above, which takes it one layer later: one barrier per layer
// couples the two, and each wave holds half the rays, opacities and sums (78 VGPRs without emission, 103 with: six and four
// workgroups' waves per SIMD).  The arithmetic and the order of every cell's sum are those of brick_kernel: same bits.
// Twice the wavefronts per stage: ahead where the stages are narrow (up to four frequency groups on the GPU) and with emission
// (brick_form, ftte_context.h).
// grid: ntasks * nnu workgroups of 128 threads; dynamic LDS: 2 x (max_dirs - 1) x 2 KB of parked rays + 2 x max_dirs x 512 B of hand-over
template <int WAVES, int EMIT>
__global__ void __launch_bounds__(128, WAVES) brick_pair_kernel(const BrickLaunch L, int max_dirs)
{
    extern __shared__ double pair_lds[];
    constexpr int R = kBrickRows, H = kBrickRows / 2;
    using cgroup = const __attribute__((address_space(4))) BrickGroup;
    using clayer = const __attribute__((address_space(4))) LayerRec;
    const int lane = threadIdx.x & 63;
    const int wv = uniform((int)(threadIdx.x >> 6));
    double *state = pair_lds + (size_t)wv * (max_dirs - 1) * H * 64; // this wave's parked rays
    double *handover = pair_lds + (size_t)2 * (max_dirs - 1) * H * 64; // [step parity][lane]
    const int nnu = L.nnu;
    const int nu = L.nu0 + blockIdx.x % nnu;
    BrickTask T;
    {
        const unsigned long raw = ((const __attribute__((address_space(4))) unsigned long *)(unsigned long)L.tasks)[blockIdx.x / nnu];
        T.group = (int16_t)(raw & 0xffffu); T.tu = (int16_t)((raw >> 16) & 0xffffu); T.tv = (int16_t)((raw >> 32) & 0xffffu); T.ti = (int16_t)(raw >> 48);
    }
    const int tu = uniform((int)T.tu), tv = uniform((int)T.tv), ti = uniform((int)T.ti) & (kBrickAccumulate - 1);
    const bool accumulate = (uniform((int)T.ti) & kBrickAccumulate) != 0;
    const bool atomic_acc = L.atomic_acc != 0;
    cgroup *G = (cgroup *)(L.groups + uniform((int)T.group));
    const int n = L.n, chunk = L.chunk, up = L.up, vp = L.vp;
    const int ndir = G->ndir;
    const double uvb = ((const __attribute__((address_space(4))) double *)(unsigned long)L.uvb)[nu];
    double lead = L.math.c[9]; // the exponential's leading coefficient, in a vector register for the whole run (ftte_math.h)
    asm volatile("" : "+v"(lead));

    const int sv = G->sv, si = G->si;
    const bool mirror_u = G->su < 0;
    const long org = G->org;
    gcbyte *kbase = (gcbyte *)(G->kappa + (long)nu * L.group_stride + org);
    gbyte *jbase = (gbyte *)(G->J + (long)nu * L.group_stride + org);
    gcbyte *xbase = EMIT ? (gcbyte *)(G->emis + (long)nu * L.group_stride + org) : nullptr;

    const int cu = 64 * tu + lane + 1;
    const int cv0 = R * tv + H * wv + 1; // this wave's first row
    const int cuc = cu < n ? cu : n;
    const bool tiled = L.tiled != 0;
    const unsigned off0 = tiled ? 8u * (unsigned)(mirror_u ? 63 - lane : lane) : 8u * (unsigned)(mirror_u ? n + 1 - cuc : cuc);
    const bool own_lane = cu <= n;
    const long row_bytes = 8l * sv;
    const int i0 = ti * chunk + 1;
    const int i1 = (i0 + chunk - 1 < n) ? i0 + chunk - 1 : n;

    const bool has_u_in = tu > 0, has_u_out = 64 * (tu + 1) < n;
    const bool has_v_in = tv > 0 && wv == 0, has_v_out = R * (tv + 1) < n && wv == 1;
    const bool has_i_in = ti > 0, has_i_out = i1 < n;
    const long fnu = (long)nu * L.face_stride;
    const int uw = L.uw, ut = L.ut;
    const int ns = L.nslot, sl = ti % ns;
    const long u_out = ((long)(tu * ns + sl) * chunk) * uw + ut * tv + H * wv;
    const long u_in = ((long)((tu - 1) * ns + sl) * chunk) * uw + ut * tv + H * wv;
    const long v_out = L.vface_off + ((long)(tv * ns + sl) * chunk) * up + 64 * tu;
    const long v_in = L.vface_off + ((long)((tv - 1) * ns + sl) * chunk) * up + 64 * tu;
    const long i_in = L.iface_off + ((long)sl * vp + R * tv + H * wv) * up + 64 * tu + lane;
    const long i_out = L.iface_off + ((long)((ti + 1) % ns) * vp + R * tv + H * wv) * up + 64 * tu + lane;

    int p0 = (ndir - (i0 - 1) % ndir) % ndir;
    double kap_next[H], kap[H], Jacc[H];
    double xs[EMIT ? H : 1] = {};
    // rows of this wave inside the grid (the upper wave of a ragged last brick may have none: it then reads row n throughout)
    const int nrows = n - cv0 + 1 < H ? (n - cv0 + 1 > 0 ? n - cv0 + 1 : 0) : H;
    const long row0 = tiled ? 8l * ((long)tu * G->bu + (long)tv * G->bv + (long)ti * G->bi) + (long)(H * wv) * row_bytes : (cv0 < n ? cv0 : n) * row_bytes;
#pragma unroll
    for (int r = 0; r < H; ++r) { kap[r] = 0.0; Jacc[r] = 0.0; }
    load_rows<H, false>(kap_next, kbase + 8l * i0 * si + row0, off0, row_bytes, nrows); // (the first layer's opacities: asked for before anything else)
    double cur[H];
    {
        gcdouble *f = (gcdouble *)(G->dir[p0].faces + fnu);
#pragma unroll
        for (int r = 0; r < H; ++r) cur[r] = has_i_in ? f[i_in + (long)r * up] : uvb;
    }
    for (int k = 0; k + 1 < ndir; ++k) {
        int d = p0 + 1 + k;
        d = d >= ndir ? d - ndir : d;
        gcdouble *f = (gcdouble *)(G->dir[d].faces + fnu);
        double parked[H]; // (the loads together, then the LDS stores: brick_kernel)
#pragma unroll
        for (int r = 0; r < H; ++r) parked[r] = has_i_in ? f[i_in + (long)r * up] : uvb;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < H; ++r) state[(k * H + r) * 64 + lane] = parked[r];
    }
    // wave 0 crosses layer i0 + it, wave 1 the layer before: what moves up out of row 3 waits a layer in LDS
    const int nlayers = i1 - i0 + 1;
    for (int it = 0; it <= nlayers; ++it) {
        const int il = it - wv;
        if (il >= 0 && il < nlayers) {
            const int i = i0 + il;
            gbyte *jplane = jbase + 8l * i * si;
#pragma unroll
            for (int r = 0; r < H; ++r) { kap[r] = kap_next[r]; Jacc[r] = 0.0; }
            const long rstep = here(row_bytes);
            if (EMIT) load_rows<EMIT ? H : 1, false>(xs, xbase + 8l * i * si + row0, off0, rstep, nrows); // the layer's emissivities or source functions
            if (accumulate && !atomic_acc) load_rows<H, true>(Jacc, (gcbyte *)jplane + row0, off0, rstep, nrows);
            if (i < i1) load_rows<H, false>(kap_next, kbase + 8l * (i + 1) * si + row0, off0, rstep, nrows);
            double *hand = handover + (size_t)(il & 1) * max_dirs * 64;
            for (int j = 0; j < ndir; ++j) {
                int d = p0 + j;
                d = d >= ndir ? d - ndir : d;
                if (j) {
#pragma unroll
                    for (int r = 0; r < H; ++r) {
                        const double parked = state[((j - 1) * H + r) * 64 + lane];
                        state[((j - 1) * H + r) * 64 + lane] = cur[r];
                        cur[r] = parked;
                    }
                }
                clayer *rp = (clayer *)(G->dir[d].layers) + (i - 1);
                const double d0 = rp->dpath[0], d1 = rp->dpath[1], d2 = rp->dpath[2];
                const int rc = rp->info & 7;
                const double w = G->dir[d].w;
                gbyte *f = (gbyte *)(G->dir[d].faces + fnu);
                gcbyte *uin = has_u_in ? (gcbyte *)f + 8 * (u_in + (long)il * uw) : nullptr;
                gbyte *uout = has_u_out ? f + 8 * (u_out + (long)il * uw) : nullptr;
                gcbyte *vin = has_v_in ? (gcbyte *)f + 8 * (v_in + (long)il * up) : nullptr;
                gbyte *vout = has_v_out ? f + 8 * (v_out + (long)il * up) : nullptr;
                const double *cin = wv ? hand + j * 64 : nullptr;
                double *cout = wv ? nullptr : hand + j * 64;
                const bool third_first = rc == RC_THREE_U_SWAP || rc == RC_THREE_V_SWAP;
                switch (rc) {
                case RC_ONE: brick_step<RC_ONE, EMIT, H>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, false, false, 0, 63, true, cin, cout); break;
                case RC_TWO_U: brick_step<RC_TWO_U, EMIT, H>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, false, false, 0, 63, true, cin, cout); break;
                case RC_TWO_V: brick_step<RC_TWO_V, EMIT, H>(L.math, lead, cur, kap, xs, Jacc, false, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, false, false, 0, 63, true, cin, cout); break;
                case RC_THREE_U:
                case RC_THREE_U_SWAP:
                    brick_step<RC_THREE_U, EMIT, H>(L.math, lead, cur, kap, xs, Jacc, third_first, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, false, false, 0, 63, true, cin, cout);
                    break;
                default:
                    brick_step<RC_THREE_V, EMIT, H>(L.math, lead, cur, kap, xs, Jacc, third_first, d0, d1, d2, w, uvb, uin, uout, vin, vout, lane, false, false, 0, 63, true, cin, cout);
                    break;
                }
            }
            p0 = p0 ? p0 - 1 : ndir - 1;
            if (own_lane) {
                gbyte *jrow = jplane + row0;
                const unsigned off = here(off0);
#pragma unroll
                for (int r = 0; r < H; ++r) {
                    if (r < nrows) {
                        if (accumulate && atomic_acc) (void)__builtin_amdgcn_global_atomic_fadd_f64((gdouble *)lane_address(jrow, off), Jacc[r]);
                        else __builtin_nontemporal_store(Jacc[r], (gdouble *)lane_address(jrow, off));
                    }
                    jrow += rstep;
                }
            }
        }
        __syncthreads(); // the layer's hand-overs are in LDS before the wave above starts that layer
    }

    if (has_i_out) { // p0 is in registers, p0 + 1 + k in slot k
        for (int q = 0; q < ndir; ++q) {
            int d = p0 + q;
            d = d >= ndir ? d - ndir : d;
            gdouble *f = (gdouble *)(G->dir[d].faces + fnu);
#pragma unroll
            for (int r = 0; r < H; ++r) f[i_out + (long)r * up] = q ? state[((q - 1) * H + r) * 64 + lane] : cur[r];
        }
    }
}

// Which XCC ids does this device have?  Every workgroup of a grid large enough to reach every XCD reports the id it reads.
__global__ void xcc_census_kernel(unsigned *mask)
{
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) atomicOr(mask, 1u << (xcc & 15u));
}

int launch_xcc_census(unsigned *mask_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(xcc_census_kernel, dim3(4096), dim3(64), 0, stream, mask_dev);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_brick_pair(const BrickLaunch &L, int max_dirs, int waves, hipStream_t stream)
{
    if (L.ntasks <= 0) return 0;
    if (max_dirs < 1 || max_dirs > kBrickMaxDirs || L.ticket != nullptr) return -1;
    const dim3 grid((unsigned)L.ntasks * (unsigned)L.nnu);
    const size_t lds = ((size_t)(max_dirs - 1) * kBrickRows * 64 + 2 * max_dirs * 64) * sizeof(double) + (size_t)lds_pad();
    // emission: the log-mean's own division and polynomials (ftte_segment_emit) need the registers of three workgroups per SIMD
    if (L.emit == 1) hipLaunchKernelGGL((brick_pair_kernel<3, 1>), grid, dim3(128), lds, stream, L, max_dirs);
    else if (L.emit == 2) hipLaunchKernelGGL((brick_pair_kernel<4, 2>), grid, dim3(128), lds, stream, L, max_dirs);
    else if (L.emit) return -1;
    else switch (waves) {
    case 2: hipLaunchKernelGGL((brick_pair_kernel<2, 0>), grid, dim3(128), lds, stream, L, max_dirs); break;
    case 3: hipLaunchKernelGGL((brick_pair_kernel<3, 0>), grid, dim3(128), lds, stream, L, max_dirs); break;
    case 4: hipLaunchKernelGGL((brick_pair_kernel<4, 0>), grid, dim3(128), lds, stream, L, max_dirs); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

int launch_brick(const BrickLaunch &L, int max_dirs, int waves, hipStream_t stream, bool masked, int persistent)
{
    if (L.ntasks <= 0) return 0;
    if (max_dirs < 1 || max_dirs > kBrickMaxDirs) return -1;
    if (L.tiled && (L.emit || masked)) return -1; // brick-ordered storage: the emission rows and the cut bricks are addressed in frame order only
    const dim3 grid((unsigned)L.ntasks * (unsigned)L.nnu);
    const size_t lds = (size_t)(max_dirs - 1) * kBrickRows * 64 * sizeof(double) + (size_t)lds_pad(); // pad: diagnostic knob "ldspad"
    const bool flow = L.ticket != nullptr;
    if (L.queue) { // persistent form: `persistent` workgroups, what the GPU holds at once (more are harmless: they find the queues empty)
        if (masked || L.emit || !flow || persistent < 1) return -1;
        hipLaunchKernelGGL((brick_kernel<4, 0, 2>), dim3((unsigned)persistent), dim3(64), lds, stream, L);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (masked && !flow && L.emit) { // (the emission forms hold 162 registers with the lane range, as without it: three waves per SIMD)
        if (L.emit == 1) hipLaunchKernelGGL((brick_kernel<2, 1, 0, true>), grid, dim3(64), lds, stream, L);
        else hipLaunchKernelGGL((brick_kernel<3, 2, 0, true>), grid, dim3(64), lds, stream, L);
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    if (masked) {
        if (flow) return -1; // the hybrid sweep issues a launch per stage
        hipLaunchKernelGGL((brick_kernel<3, 0, 0, true>), grid, dim3(64), lds, stream, L); // (126 registers since the trimming: four waves per SIMD in fact)
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    // emission: the log-mean's own division and polynomials need more registers than three waves per SIMD leave
    if (L.emit == 1 && !flow) hipLaunchKernelGGL((brick_kernel<3, 1, 0>), grid, dim3(64), lds, stream, L);
    else if (L.emit == 2 && !flow) hipLaunchKernelGGL((brick_kernel<3, 2, 0>), grid, dim3(64), lds, stream, L); // (the source rows: 145 registers)
    else if (L.emit) return -1; // the dataflow form is built without emission
    else if (flow) hipLaunchKernelGGL((brick_kernel<4, 0, 1>), grid, dim3(64), lds, stream, L);
    else switch (waves) {
    case 2: hipLaunchKernelGGL((brick_kernel<2, 0, 0>), grid, dim3(64), lds, stream, L); break;
    case 3: hipLaunchKernelGGL((brick_kernel<3, 0, 0>), grid, dim3(64), lds, stream, L); break;
    case 4: hipLaunchKernelGGL((brick_kernel<4, 0, 0>), grid, dim3(64), lds, stream, L); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

} // namespace ftte
