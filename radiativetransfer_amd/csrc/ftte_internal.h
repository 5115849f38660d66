// ftte_internal.h -- records shared by the host planner (ftte_plan.cpp) and the kernels
// (ftte_kernels.hip).  Plain structs, no HIP types.
#pragma once
#include <cstdint>

#include "ftte_math.h"

namespace ftte {

// ---- kernel-frame ray classes --------------------------------------------------------------
// In the sweep frame of the reference a layer's ray is one of five chains
// (setPattern, transportRoutinesModule.f90:7-85):  xy | xy->yz | xy->xz | xy->yz->xz | xy->xz->yz,
// a yz segment living one cell further along sweep-k, an xz segment one cell further along
// sweep-j.  The kernel works in a (u, v) frame: u = whichever of sweep-j / sweep-k is contiguous
// in memory for the direction's izone (the wave's lane axis), v = the other one (rows held in
// registers).  In that frame a chain is described by where its 2nd/3rd segments sit:
enum RayClass : int {
    RC_ONE = 0,     // xy only
    RC_TWO_U = 1,   // 2nd segment in cell (v, u+1)
    RC_TWO_V = 2,   // 2nd segment in cell (v+1, u)
    RC_THREE_U = 3, // 2nd in (v, u+1), 3rd in (v+1, u+1)
    RC_THREE_V = 4, // 2nd in (v+1, u), 3rd in (v+1, u+1)
    // same shapes, but the cell's mean adds the chain's 3rd piece before its 2nd: the reference sums
    // xy, xz, yz whatever the chain order (transportRoutinesModule.f90:695-941)
    RC_THREE_U_SWAP = 5,
    RC_THREE_V_SWAP = 6
};

// One layer of one direction, 32 bytes, read with a single scalar load.
struct LayerRec {
    double dpath[3];  // cell size * segment length, chain order (transportRoutinesModule.f90:651)
    int32_t info;     // bits 0-2 RayClass
    int32_t drift;    // cumulative drift of the rays up to this layer: low 16 bits along u, high 16 along v
};
static_assert(sizeof(LayerRec) == 32, "LayerRec must be 32 bytes");

// One direction as a launch sees it.
struct DirRec {
    const LayerRec *layers; // [n], device memory
    const double *kappa;    // opacity in the layout of this direction's march axis, group 0
    double *J;              // accumulator of this direction's slot, same layout, group 0
    const double *emis;     // emissivity or source function in the same layout (LaunchRec::emit), else null
    int64_t org;            // element offset of the virtual cell (i, v, column position) = (0,0,0)
    int32_t si, sv, su;     // element strides along march and v (signed: reflections); su = +1, or -1 when the
                            // u axis is mirrored: the column position of cell u is then n+1-u (stride +1 either way)
    int32_t u_lo, v_lo;     // label of the first owned ray along u / v
    int32_t first;          // 1: J receives a plain store (first direction into this accumulator)
    double w;               // quadrature weight
};

// One wave's work: a tile of 64 x ROWS rays (lane 0 and row 0 are read-only halo) of one
// direction, marched from layer i_first to i_last.
struct WorkItem {
    int16_t slot;    // index into LaunchRec::dir
    int16_t tu, tv;  // tile coordinates
    int16_t i_first, i_last;
    int16_t pad;
};
static_assert(sizeof(WorkItem) == 12, "WorkItem must be 12 bytes");

constexpr int kMaxSlots = 16; // directions of one tile-kernel launch
constexpr int kMaxAcc = 32;   // J accumulators per memory layout (tile kernel: one per slot; bricks: one per group)

struct LaunchRec {
    DirRec dir[kMaxSlots];
    const WorkItem *items;
    const double *uvb;  // [nnu], device memory
    int64_t group_stride; // elements between consecutive frequency groups (= ncell)
    int32_t n;          // grid size
    int32_t nitems;
    int32_t nnu;        // frequency groups; workgroup b handles group b % nnu of work item b / nnu
    int32_t emit;       // 0 none, 1 DirRec::emis is the reference's eta, 2 a source function
    ftte_consts math;   // constants of ftte_math.h, delivered through scalar registers
};

// ---- cell-fixed bricks (brick_kernel) -----------------------------------------------------------------
// The second organisation of the uniform-grid sweep.  A brick is a block of cells fixed in space, 64 cells along u (the
// wave's lanes) x kBrickRows along v (registers) x `chunk` layers along the march axis, swept by one wavefront for ALL
// the directions of one group (directions that share an izone, hence a memory frame and a sweep order), layer by layer:
// the opacity of a layer is loaded once and its J row stored once for the whole group, nothing is recomputed (no halo),
// and the rays that cross a brick face travel through small ring buffers in memory.  A brick needs its three upstream
// neighbours (u-1, v-1, chunk-1) to be done: bricks with equal tu + tv + ti form a stage, one launch per stage.
constexpr int kBrickRows = 8;
constexpr int kBrickDeps = 6;    // the bricks a brick waits for: u-1, v-1, chunk-1, the previous writer of its J tile, the readers of the two ring slots it reuses
constexpr int kBrickQueues = 8;  // task queues of the persistent form: one per XCD of an MI355X
constexpr int kBrickMaxDirs = 8; // directions of one group (their ray state waits in LDS: 4 KB per direction and wave)

struct BrickDir {
    const LayerRec *layers; // [n]
    double *faces;          // this direction's face rings, frequency group 0: [uface | vface | iface] (BrickLaunch)
    double w;
    double pad;
};

struct BrickGroup {
    BrickDir dir[kBrickMaxDirs];
    const double *kappa; // opacity in the layout of this izone's march axis, frequency group 0
    const double *emis;  // emissivity or source function in the same layout (BrickLaunch::emit), else null
    double *J;           // this group's accumulator (same layout); shared with other groups only as BrickTask says
    int64_t org;         // as DirRec
    int32_t si, sv, su;
    int32_t ndir;
    int32_t bu, bv;      // brick-ordered storage (BrickLaunch::tiled): elements from a brick to the next one along u, along v
    int64_t bi;          // tiled == 2 (a whole brick in one piece): what a step to the next brick along the march adds beyond chunk * si
};

// ti: chunk index; bit 14 set (kBrickAccumulate): another group that shares this group's accumulator has been through this
// brick in an earlier launch, so J is read, added to and stored instead of stored
struct BrickTask { int16_t group, tu, tv, ti; };
constexpr int kBrickAccumulate = 0x4000;
// A task that sweeps a range of lanes only (hybrid sweep, brick_kernel<..., MASKED>): tu bits 0-9 the brick, bits 10-15 the first
// lane; group bits 0-7 the group, bits 8-13 the last lane
// lane; tv bits 0-9 the brick, bits 10-14 the box whose face rings the lanes' ends use
constexpr int kBrickTuMask = 0x3ff, kBrickLaneLoShift = 10, kBrickGroupMask = 0xff, kBrickLaneHiShift = 8;
constexpr int kBrickTvMask = 0x3ff, kBrickBoxShift = 10, kBrickBoxMask = 31;
static_assert(sizeof(BrickTask) == 8, "BrickTask must be 8 bytes");

struct BrickLaunch {
    const BrickGroup *groups;
    const BrickTask *tasks;   // of this stage
    const double *uvb;        // [nnu]
    int64_t group_stride;     // elements between frequency groups in kappa / J
    int64_t face_stride;      // elements between frequency groups in a direction's face rings
    int64_t vface_off, iface_off; // where the v-face and i-face rings start inside a direction's block (u-face ring at 0)
    int64_t uqface_off;       // more rings laid out like the u-faces of ONE column of bricks, two per box of the direction: the near and
                              // the far u-face of the box where they lie inside a brick (masked bricks of the hybrid sweep)
    int32_t n, ntasks, nnu, chunk; // nnu: frequency groups THIS launch sweeps, nu0, nu0 + 1, ...
    int32_t nu0;
    int32_t emit;             // 0 none, 1 BrickGroup::emis is the reference's eta, 2 a source function (LaunchRec::emit)
    int32_t up, vp;           // padded extents: 64 * ntu, kBrickRows * ntv
    int32_t nslot;            // face slots along the march: 2 = rings over two chunks (a brick's consumers run one stage later), or the
                              // number of chunks (hybrid sweep of a refined cell array: the forest pass reads faces written many stages earlier)
    int32_t uw, ut;           // u-face ring: doubles per layer (ntv * ut) and per brick (ut = kBrickRows, or 16 = one 128-byte line
                              // of its own per brick and layer when bricks of one launch hand rays to each other)
    // Dataflow form (ticket != nullptr): ONE launch holds every brick of the sweep; a workgroup draws the next task of the
    // (topologically ordered) list from `ticket`, waits until the tasks it depends on have published `epoch` in `done`, and
    // publishes its own when its stores are visible.  deps: [ntasks][kBrickDeps] task indices or -1.
    uint32_t *ticket, *done, *error;
    const int32_t *deps;
    uint32_t epoch;
    int32_t pad_;
    // kappa and the accumulators stored brick by brick: the eight rows of a brick's layer in one piece of 4 KB, the pieces of a
    // layer in brick order (v, then u), layers as in the frame.  Whole bricks only (n a multiple of 64 and of kBrickRows).  The
    // group's org, sv, bu, bv are then those of this order (tiled_index, ftte_kernels.hip); si is the same in both.  tiled == 2: the
    // `chunk` layers of a brick follow each other as well (the whole brick in one piece of chunk x 4 KB; n a multiple of the chunk).
    int32_t tiled, pad2_;
    // Persistent form (queue != nullptr, ftte_brick.hip): as many workgroups as the GPU holds at once, each bound to the XCD it runs
    // on (read from HW_REG_XCC_ID); a workgroup draws (task, frequency group) pairs from ITS XCD's queue only, in queue order, until
    // the queue is empty.  A queue holds whole chains of dependent bricks (everything a brick waits for lies earlier in the same
    // queue), so every ray face and accumulator row a brick reads was written by a CU behind the same L2: plain stores, a drained
    // store counter and a flag suffice, no write-through and no L2 write-back.  Tickets 32 words apart: ticket[32 q].
    const uint32_t *queue;                        // work ids (task index * nnu + frequency slot), queue after queue, each in stage order
    uint32_t qoff[kBrickQueues], qlen[kBrickQueues];
    int8_t xcc_queue[16];                         // XCC id -> queue, from the census of the device's XCC ids (-1: an id the census did not see)
    // A brick that comes second to an accumulator's cells (kBrickAccumulate) adds its sums with fp64 atomic adds instead of reading
    // the earlier ones first: nothing to wait for.  One brick per cell and launch (or, in one launch, in dependency order), so the
    // additions still happen in a fixed order: J stays reproducible bit for bit.
    int32_t atomic_acc;
    // A sweep of a SUB-GRID that has neighbours on every side (the fine cells of a fully refined block of a refined cell array,
    // swept by bricks of their own, ftte_hybrid.cpp): the bricks at the sub-grid's upstream faces take their rays from face rings
    // like everybody else -- ring `ntu` (`ntv`) stands for the missing brick column to the left (row below), chunk slot 0 for the
    // chunk before the first; somebody has filled them (amr_fine_import_kernel) --, and the bricks at its downstream faces leave
    // theirs in their own rings and in chunk slot nti (nslot = nti + 1: no wrap), where the forest behind picks them up.
    int32_t sub;
    ftte_consts math;
};

// ---- refined cell arrays (ftte_amr.h) ------------------------------------------------------------------
constexpr int kAmrBatch = 96; // most directions in flight in the forest path (what fits the device memory decides)

// One active segment of a direction's forest, in processing order (sorted by depth)
struct SegRec {
    int32_t seg;     // 3 * leaf + slot: where its outgoing intensity and mean are stored
    int32_t up, up2; // upstream segment (AmrForest::kInflow: the boundary; kImport: a face buffer), second one of the mean-of-two rule or -1
    int32_t at;      // kImport: element of the direction's face block where the ray waits
    double dpath;    // cell size * segment length
};

struct AmrExport { int32_t at, seg; }; // face element <- outgoing intensity of a segment (a ray leaving the forest's region)
// face element of a fine block's own brick sweep <- what the coarser leaf upstream hands over: the outgoing intensity of segment `up`,
// the mean with `up2` where the coarse-neighbour rule asks for it (transportRoutinesModule.f90:612-634), the inflow for up = -1
struct AmrImport { int32_t at, up, up2; };

struct AmrDirRec {
    const SegRec *rec;       // [active segments], depth after depth
    const uint8_t *active;   // [ncell] bit 0: the leaf has an xz segment, bit 1: a yz segment, bit 2: the leaf lies outside the
                             // region this direction's forest is restricted to (hybrid sweep: a brick computed its J)
    double *faces;           // hybrid sweep: this direction's face block (BrickDir::faces), else nullptr
    const AmrExport *exports;
    int64_t nexports;
    const AmrImport *imports; // hybrid sweep with fine blocks swept by bricks: the rays entering the blocks from the forest
    int64_t nimports;
    double *Iout, *mean;     // [3 ncell][nnu] scratch of this direction's slot
    double w;
};

struct AmrLevelRec {
    const AmrDirRec *dir;         // [ndir], device memory
    const int64_t *count;         // [ndir] segments of this depth per direction, device memory
    const int64_t *begin;         // [ndir] where this depth starts in each direction's `rec`, device memory
    int64_t most;                 // largest count[] of this depth (sizes the launch)
    const double *kappa, *uvb, *emis; // element (group g, cell c) at g * group_stride + c * cell_stride
    int64_t group_stride, cell_stride;
    int64_t ncell;
    const int32_t *cells;    // hybrid sweep: the leaves that lie in the region of at least one direction, else nullptr: every leaf.
    int64_t ncells;          // With a list, segments, opacities, activity bytes and scratch are numbered by POSITION in the list
                             // (segment 3 * position + piece): plan and scratch memory follow the regions, not the tree
    int64_t face_stride;     // elements between frequency groups in a direction's face block
    int32_t ndir, nnu, emit;
    ftte_consts math;
};

// ---- point sources (ftte_point.cpp) --------------------------------------------------------------------------------
constexpr int kTableDepths = 11;                    // ndepth + 1, definitionsModule.f90:72
constexpr int kTableSize = 11 * 11 * 11 * 11;       // one rate table
constexpr int kMaxPixelLevel = 6;                   // localDefinitions, equiSources.f90:9
constexpr int kPixelCount = 12 * (4096 - 1) / 3;    // pixels of levels 1..6: 12 (4^6 - 1)/3 = 16380
constexpr int kOutputRadii = 7;                     // nradius, equiSources.f90:9
constexpr int kOutputEnergies = 300;                // nenergy, definitionsModule.f90:290
// one star's escape bookkeeping (startNewLongRay, equiSources.f90:3198-3233, 3336-3345): ndotRemaining[7], ndotBoundary[7],
// ndotDust, ndotSpectrum[300]
constexpr int kEscapeRec = 2 * kOutputRadii + 1 + kOutputEnergies;

// One frequency bin of stellarBetaTable as the table kernel needs it
struct FreqBin {
    double dtmp;                 // photons/s in the bin
    double r24, r26, r25, rdust; // sigma(nu)/sigma(threshold) per absorber: multiplies the tabulated depth
    double excess[3];            // (nu - nu_threshold) in erg, per reaction; < 0: below threshold, bin does not count
};

// A ray waiting to be split into its four daughter pixels (startNewLongRay, equiSources.f90:3280-3383)
struct SplitRec {
    double pos[3];   // absolute position of the split point, box units (absoluteCoordinates)
    double radius;   // base-cell units
    double depth[4]; // tau1, tau2, tau3, tauDust accumulated so far
    double ndot;     // photons/s carried by the parent ray
    int32_t pixel;   // parent's NESTED index at its level
    int32_t src;     // which star of the batch the ray belongs to (its escape record)
};

// What the tracer touches per cell crossing is packed so that each kind of datum is one 64-byte line (or less):
// the rays of a wavefront are in 64 different places of the grid, and every separate array is another line from HBM.
struct NodeRec { int32_t child0, leaf, parent, level; }; // a tree node: first child or -1, cell-array index or -1, parent or -1
constexpr int kCellRec = 8; // doubles per cell in the packed medium {HI, HeI, HeII, rho, abun2, 0, 0, 0} and in the packed
                            // rates {krate24, krate25, krate26, crate24, crate25, crate26, 0, 0}

struct TraceRec {
    const NodeRec *node;   // nullptr on a uniform grid: node == cell, level 0, no children
    int32_t n;
    int32_t dust;     // dustApproximation: 0 none, 1 ~HI, 2 ~total H
    int64_t ncell;
    double box;
    const double *medium;  // [ncell][kCellRec]
    const double *logtab;  // [3][11^4][2] natural logs of the rate tables, (number, heating) pairs
    const double *pixdir;  // [kPixelCount][3] unit vectors of all pixels of levels 1..6
    double rmax[kMaxPixelLevel + 1];
    double *rates;         // [ncell][kCellRec]
    // this launch: rays of `pixel_level`
    int32_t pixel_level;
    int32_t nrays;         // level 1: 12 * nsources; else 4 * number of split records
    const int32_t *src_node;  // level 1: host leaf node of each source
    const double *src_ndot;
    const SplitRec *in;    // level > 1
    SplitRec *out;         // rays of this level that split
    int32_t *out_count;
    int32_t out_capacity;
    int32_t *highest_level;
    int32_t *error;
    unsigned long long *steps; // cell crossings, all rays (instrumentation)
    // escape bookkeeping
    double *escape;            // [sources of this batch][kEscapeRec], accumulated with atomics
    const double *sigma_ratio; // [4][300]: outputSigma24/6.30e-18, outputSigma25/1.58e-18, outputSigma26/7.42e-18, outputSigmaDust/5.41e-22,
                               // or nullptr (no cross-sections known: ndotSpectrum stays zero)
    double out_radius_kpc[kOutputRadii]; // outputRadius, equiSources.f90:10
    double kpc;
};

// ---- ionisation equilibrium (solveRateEquations, equiSources.f90:3459-3677) -------------------------------------------
struct ChemRec {
    // cell arrays, cell-array order
    const int8_t *level;       // per leaf (not per node)
    const double *rho, *logtem;
    const double *HI, *HeI, *HeII;       // state on entry
    double *HI_out, *HeI_out, *HeII_out; // state on return
    const double *krate;       // [ncell][kCellRec] packed point-source rates or nullptr
    const double *J;           // [3][ncell] or nullptr (uniform background)
    const double *k;           // [6][nratec] rate coefficients k1a..k6a
    int64_t ncell;
    int32_t n, nratec, run_uvb, pad;
    double box, logtem0, logtem9, dlogtem;
    double ksi[9];             // [group][ksi24, ksi25, ksi26]
    double uniform[3];         // uniformQuasar * quasar%ksi + uniformStellar * stellar%ksi, per reaction
    double threshold;          // selfShieldingThreshold
    unsigned long long *first_bad; // lowest cell index at which the reference would stop, or ~0
    unsigned long long *max_change; // bits of the largest change of a species fraction
    unsigned long long *steps;      // bisection steps taken, all cells
};

} // namespace ftte
