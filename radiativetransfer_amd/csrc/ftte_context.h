// ftte_context.h -- what the translation units behind include/ftte.h share: the sweep plans, the context, and the internal
// entry points of the planners (ftte_plan.cpp), the sweeps (ftte_sweeps.cpp: segment forests, cell-fixed bricks), the hybrid sweep
// of refined cell arrays (ftte_hybrid.cpp) and the host-array transfers (ftte_host_arrays.cpp).  ftte_api.cpp is the C ABI itself.
//
// There is no CPU fallback behind any of this: every entry point that computes on the grid needs a HIP device and fails with
// FTTE_ERR_NO_DEVICE otherwise.
#pragma once

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include <thread>

#include "../../include/ftte.h"
#include "ftte_amr.h"
#include "ftte_geometry.h"
#include "ftte_internal.h"
#include "ftte_kernels.h"
#include "ftte_point.h"


namespace ftte {


extern std::string g_create_error; // what ftte_last_error(NULL) returns

// one planned direction
struct DirPlan {
    int izone = 0, layout = 0;
    double phi = 0, theta = 0, w = 0;
    int64_t org = 0;
    int si = 0, sv = 0, su = 0;
    int u_lo = 1, v_lo = 1, ntu = 0, ntv = 0;
    int du_mid = 0, dv_mid = 0; // drift at the middle layer: where a tile's rays are halfway through the grid
    size_t layer_off = 0; // into the layer table
    int slot = 0;
};

struct LaunchPlan {
    int layout = 0;
    bool first = false;
    std::vector<int> dirs; // indices into Plan::dirs, position = slot
    int acc_base = 0;      // slot s of this launch accumulates into acc[layout][acc_base + s]
    size_t item_off = 0;
    int nitems = 0;
    int64_t updates = 0;
};

struct Plan {
    bool valid = false;
    // key
    int n = 0, rows = 0, slots = 0, stack = 0;
    double box = 0;
    std::vector<double> phi, theta, w;
    // content
    std::vector<DirPlan> dirs;
    std::vector<LayerRec> layers;
    std::vector<WorkItem> items;
    std::vector<LaunchPlan> launches;
    bool used[3][kMaxSlots] = {};
};

// The brick organisation of the same sweep (ftte_brick.hip): directions grouped by izone, bricks ordered into stages
struct BrickPlan {
    bool valid = false;
    // key
    int n = 0, chunk = 0, gmax = 0, share = 0, want_glanes = 0, want_dataflow = 0;
    double box = 0;
    std::vector<double> phi, theta, w;
    // content
    std::vector<DirPlan> dirs;
    std::vector<LayerRec> layers;
    struct Group { int izone = 0, layout = 0, acc = 0, offset = 0, lane = 0; std::vector<int> dirs; };
    std::vector<Group> groups;
    std::vector<BrickTask> tasks;      // stage after stage
    bool dataflow = false;             // one launch, bricks wait for each other through flags (needs whole bricks: n % 64 == 0)
    std::vector<int32_t> deps;         // [tasks][kBrickDeps]
    // persistent form (option "dataflow" = 3): one queue of (task, frequency slot) pairs per XCD, whole dependency chains each
    bool persistent = false;
    int qnnu = 0, nq = 0, qmix = 0;    // frequency groups and XCDs the queues were cut for, option "queue_mix"
    std::vector<uint32_t> queue;       // work ids task * nnu + slot, queue after queue
    uint32_t qoff[kBrickQueues] = {}, qlen[kBrickQueues] = {};
    int64_t qload[kBrickQueues] = {};  // cell.direction.frequency updates per queue (balance: instrumentation)
    int ut = kBrickRows, uw = 0;       // u-face ring: doubles per brick and layer, per layer
    int64_t uqface_off = 0;            // BrickLaunch::uqface_off
    int nslot = 2;                     // face slots along the march (BrickLaunch::nslot)
    int glanes = 1, nstages = 0;       // the groups are dealt to `glanes` streams (the groups of one accumulator stay together)
    std::vector<size_t> stage_off;     // [glanes][nstages + 1] into tasks
    int64_t updates = 0;               // cell.direction updates of a sweep (per frequency group)
    int ntu = 0, ntv = 0, nti = 0, up = 0, vp = 0, max_dirs = 0;
    int64_t face_elems = 0, vface_off = 0, iface_off = 0;
    int nacc[3] = {0, 0, 0};
};

struct LaunchTiming {
    hipEvent_t start = nullptr, stop = nullptr;
    int64_t updates = 0;
    // brick sweep with per-lane layouts and merges: the stage launches of lane k lie between first[k] and last[k] (recorded on the
    // lane's stream); the phase is from the earliest first to the latest last, both measured from `start`
    std::vector<hipEvent_t> first, last;
    int lanes = 0;
};


} // namespace ftte

using namespace ftte; // this header is the library's own: every unit that includes it lives in ftte or implements the C ABI


namespace ftte { struct Multi; }

struct ftte_ctx {
    ftte::Multi *multi = nullptr;   // ftte_create with ndev > 1: this context only routes to one single-device context per device (ftte_multi.cpp)
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    bool grid_set = false;
    int n = 0;
    int64_t ncell = 0;
    double box = 0;

    int nnu = 0;
    double *kappa[3] = {nullptr, nullptr, nullptr}; // layouts 0,1,2
    bool kappa_ready[4] = {false, false, false, false}; // [3]: the cell-major copy of the forest path
    // the opacities once more in brick order (BrickLaunch::tiled), per axis order; valid while kappa_tiled_from is the count of
    // opacity changes (n_kappa_sets) they were made at
    double *kappa_tiled[3] = {nullptr, nullptr, nullptr};
    long long kappa_tiled_from[3] = {-1, -1, -1}, n_kappa_sets = 0;
    int kappa_tiled_chunk[3] = {0, 0, 0}; // layers per piece they were made with (tiled == 2), else 0
    int tiled_opt = 0; // option "tiled" (measured: no gain)
    int amr_kappa_form = 0;  // what that copy holds: 0 every leaf in cell-array order, 1 the leaves of the hybrid plan's list
    size_t kappa_cap = 0; // elements per layout buffer

    // emissivity (mode 1: the reference's eta) or source function (mode 2), same three layouts as kappa
    int emit_mode = 0;
    double *emis[3] = {nullptr, nullptr, nullptr};
    bool emis_ready[4] = {false, false, false, false};

    double *acc[3][kMaxAcc] = {};
    size_t acc_cap = 0; // elements per accumulator

    int rows = 8, slots = 8, waves = 4, stack = 1;

    // which organisation sweeps a uniform grid: 0 = the default = 2 = cell-fixed bricks (brick_kernel), 1 = ray-following tiles
    // (sweep_kernel)
    int engine = 0, chunk = 0, group = 0, brick_waves = 4, pair_waves = 4, last_brick_form = -1, share = 2, team = -1, lanes = 2; // chunk, group: 0 = by the parallelism (build_brick_plan)
    std::vector<hipStream_t> lane_stream;   // extra streams of the brick sweep (frequency groups are independent)
    std::vector<hipEvent_t> pipe_up;        // ftte_diffuse_iteration: lane k's opacities have arrived
    bool stage_used[2] = {false, false};    // the pinned staging block has a transfer recorded on stage_ev
    std::vector<hipEvent_t> lane_done;
    hipEvent_t ev_fork = nullptr;
    // option: 0 = a launch per stage; 1, 2 = the bricks of a sweep in ONE launch where the grid allows it, a workgroup per brick,
    // waiting for each other through flags (cross-XCD hand-overs: L2 write-back per brick, or write-through stores); 3 = one launch of
    // persistent workgroups that draw bricks from a queue per XCD (hand-overs stay behind one L2: plain stores)
    int dataflow = 0;
    int atomic_acc = 0;               // option "atomic_acc": later visitors of an accumulator add with fp64 atomics instead of read-add-store
    int ablate = 0;                   // diagnostic option "ablate": parts of the brick kernel's memory traffic left out (wrong J; timing only)
    int queue_mix = 0;                // persistent form: 0 = a frequency group per queue where they divide, else by load; 1 = by load; 2 = (group + accumulator) mod queues
    int32_t *d_bdeps = nullptr; size_t d_bdeps_cap = 0;
    uint32_t *d_bdone = nullptr; size_t d_bdone_cap = 0;
    uint32_t *d_bsync = nullptr;      // [32 q] ticket of queue q (one counter, [0], without queues), [32 kBrickQueues] error
    uint32_t *h_berror = nullptr;     // pinned: [0] the error flag of the last dataflow sweep, [1 + q] its tickets, copied back behind it
    uint32_t bqlen[kBrickQueues] = {}; // what those tickets must have reached (0: no persistent sweep pending)
    uint32_t *d_bqueue = nullptr; size_t d_bqueue_cap = 0; bool bqueue_uploaded = false;
    int xcc_count = -1;               // XCC ids this device reports (census, ftte_brick.hip); -1: not taken yet
    int8_t xcc_queue[16] = {};        // XCC id -> 0 .. xcc_count - 1, or -1
    uint32_t bepoch = 0;
    BrickPlan bplan;
    bool bplan_uploaded = false;
    LayerRec *d_blayers = nullptr; size_t d_blayers_cap = 0;
    BrickGroup *d_bgroups = nullptr; size_t d_bgroups_cap = 0;
    BrickTask *d_btasks = nullptr; size_t d_btasks_cap = 0;
    double *d_faces = nullptr; size_t d_faces_cap = 0;

    Plan plan;
    LayerRec *d_layers = nullptr; size_t d_layers_cap = 0;
    WorkItem *d_items = nullptr;  size_t d_items_cap = 0;
    double *d_uvb = nullptr;      size_t d_uvb_cap = 0;
    std::vector<char> bgroups_sent;   // the bytes d_bgroups holds (brick_sweep), empty: unknown
    std::vector<double> uvb_sent;     // the values d_uvb holds (brick_sweep), empty: unknown
    bool plan_uploaded = false;

    std::vector<LaunchTiming> timing;
    int timing_used = 0;

    // refined cell arrays: the tree, and the per-direction segment forests resident on the device
    AmrTree tree;
    bool use_forest = false;  // refined grid (or option "forest" = 1 on a uniform one, for cross-checks)
    int force_forest = 0;
    struct ForestDev {
        SegRec *rec = nullptr;
        uint8_t *active = nullptr;
        std::vector<int64_t> depth_off;
        double w = 0;
    };
    std::vector<ForestDev> forests;
    std::vector<double> forest_key; // phi, theta, w of the cached forests (+ box)
    AmrDirRec *d_amr_dirs = nullptr; size_t d_amr_dirs_cap = 0;      // per-direction records of the forest batches
    int64_t *d_amr_tables = nullptr; size_t d_amr_tables_cap = 0;    // per batch and depth: count[], begin[]
    double *amr_Iout = nullptr, *amr_mean = nullptr;
    double *amr_kappa = nullptr, *amr_emis = nullptr; // [ncell][nnu] copies
    size_t amr_kappa_cap = 0, amr_emis_cap = 0;
    size_t amr_scratch_cap = 0; // elements per array

    // partial merges run beside the sweeps of the next layout on their own (non-blocking) stream
    hipStream_t merge_stream = nullptr;
    hipEvent_t ev_layout_done = nullptr, ev_merge_done = nullptr, ev_layouts_ready = nullptr;
    // end of the last sweep on whatever stream the caller gave it: the setters and the next sweep wait for it before they
    // overwrite what that sweep reads
    hipEvent_t ev_sweep_done = nullptr;
    bool sweep_pending = false;

    // Hybrid sweep of a refined cell array: bricks outside a box around the refined cells, the segment forest inside it
    int hybrid = 1;                       // option: 0 = the whole tree through the forest path
    int hybrid_slots = 1;                 // option "slots": 0 = a phase of brick stages per pass even with several passes
    int use_graph = 0;                    // option "graph": replay the hybrid sweep's launches from a captured hipGraph (measured slower)
    int forest_batch = 0;                 // option: most directions per forest batch (0: what the path and the memory allow)
    int hybrid_lanes = 1;                 // option "box_lanes": along u the boxes of the hybrid sweep end on multiples of this many lanes
    int halves = 3;                       // option "pipelines": the hybrid sweep as this many independent pipelines on streams of their own (1..kMaxPipes)
    static constexpr int kMaxPipes = 4;
    hipEvent_t ev_combine[kMaxPipes] = {nullptr, nullptr, nullptr, nullptr}; // hybrid sweep: pipeline k's forest means are in J
    struct HybridPlan {
        bool valid = false, worthwhile = false;
        std::vector<double> key;          // box, chunk, group, share, then phi, theta, w
        BrickPlan bricks;                 // groups, tasks of the bricks outside the regions (phase 1, then phase 3)
        size_t phase1_stages = 0;         // stage lists per phase
        hipGraphExec_t graph_exec = nullptr; // the launches of one sweep, captured (hybrid_sweep)
        std::vector<uintptr_t> graph_sig;    // what they name: J, the buffers, the tables
        bool slots = false;               // several passes: launch lists by slot (earliest launch a brick's inputs allow), not by phase
        std::vector<std::vector<int>> pass_at; // [pipeline][pass] the list in front of which the pass's forests are launched
        int most_boxes = 0;               // boxes of the izone that has most
        int npass = 1;                    // passes of the forests (boxes behind other boxes wait for the bricks in between); the
                                          // bricks run in npass + 1 phases: before pass 0, after pass 0, ..., after the last
        size_t nlist = 0;                 // stage lists per half ((npass + 1) x phase1_stages)
        int nhalves = 1;                  // the groups of an accumulator stay in one half; halves share nothing but kappa and J
        std::vector<std::vector<int>> half_dirs; // directions of each half, list order
        std::vector<size_t> stage_off;    // into bricks.tasks: [half][list]
        int64_t brick_updates = 0;        // cell.direction updates the bricks perform (per frequency group)
        struct Dir { SegRec *rec = nullptr; uint8_t *active = nullptr; AmrExport *exports = nullptr; int64_t nexports = 0;
                     AmrImport *imports = nullptr; int64_t nimports = 0; // into the face rings of a fine block's bricks
                     std::vector<int64_t> depth_off; std::vector<int32_t> pass_first; std::vector<int64_t> export_first; };
        std::vector<Dir> dirs;
        int32_t *cells = nullptr; int64_t ncells = 0; // the leaves inside the box of at least one direction
        bool uploaded = false;
        // A fully refined block swept by bricks of its own on the fine level (option "fine_bricks"; one cluster that is a cube of
        // base cells refined exactly once, twice its side a multiple of 64): inside it the fine cells are a uniform grid
        // with a pattern per sub-layer, and the forest keeps only what lies around it (ftte_amr.h: ForestRegion::has_fine)
        struct Fine {
            bool active = false;
            int n = 0;                          // fine cells a side
            int lo[3] = {0, 0, 0};              // the block's first base cell, storage coordinates (1-based)
            BrickPlan plan;                     // the fine grid's groups (those of `bricks`, an accumulator each) and tasks
            std::vector<size_t> stage_off;      // into plan.tasks: list l = pipeline * nstages + stage is [stage_off[l], stage_off[l + 1])
            int nstages = 0;
            int64_t face_base = 0;              // where the fine face block starts inside a direction's face block (= bricks.face_elems)
            int64_t updates = 0;                // cell.direction updates the fine bricks perform (per frequency group)
            int32_t *leaf_of_fine = nullptr;    // device: [n^3], fine cell in storage order -> leaf
            LayerRec *layers = nullptr; BrickTask *tasks = nullptr; BrickGroup *groups = nullptr; // device
        } fine;
    } hplan;
    int forest_fuse = 4096;           // option "forest_fuse": levels of a forest with at most this many (segment, group) pairs in one launch (0: a launch per level)
    int fine_bricks = 1, fine_chunk = 0;  // options "fine_bricks", "fine_chunk" (0: the base bricks' chunk)
    double *fine_kappa[3] = {nullptr, nullptr, nullptr};  // the fine block's opacities, dense, in the three layouts
    size_t fine_kappa_cap = 0;
    double *fine_emis[3] = {nullptr, nullptr, nullptr};   // its emissivity / source function
    size_t fine_emis_cap = 0;
    double *fine_acc[3][kMaxAcc] = {};    // its groups' J accumulators
    size_t fine_acc_cap = 0;
    int32_t *d_leaf_of_base = nullptr;
    double *base_kappa[3] = {nullptr, nullptr, nullptr};
    size_t base_kappa_cap = 0;
    double *base_emis[3] = {nullptr, nullptr, nullptr}; // emissivity / source function of the base cells (hybrid sweep with emission)
    size_t base_emis_cap = 0;

    PointState point; // point sources: rate tables, medium, tracer scratch

    // host-array boundary (ftte_set_opacity / ftte_diffuse_sweep): J lives in a device buffer the context keeps, and
    // pageable host arrays cross PCIe through two pinned staging blocks filled by a few host threads while the other
    // block is in flight; arrays the caller has registered (ftte_host_register) are copied by the DMA engine directly
    double *host_J_dev = nullptr; size_t host_J_cap = 0;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    struct HostRange { const char *base; size_t bytes; };
    std::vector<HostRange> registered;
    std::vector<HostRange> registered_elsewhere; // pinned by another context of the same process (the devices of one multi-device context)

    // instrumentation (ftte_counter): how often the expensive host-side builds ran
    long long n_grid_builds = 0, n_plan_builds = 0, n_forest_builds = 0;

    // ionisation equilibrium (solveRateEquations)
    std::vector<int8_t> leaf_level;  // per leaf, as handed to ftte_set_grid
    int8_t *chem_level = nullptr;
    double *chem_k = nullptr;        // [6][nratec]
    int chem_nratec = 0;
    double chem_logtem0 = 0, chem_logtem9 = 0, chem_dlogtem = 0;
    double *chem_logtem = nullptr;   // [ncell] log of the gas temperature
    bool chem_temperature_set = false;
    double *chem_out = nullptr, *chem_J = nullptr; // [3][ncell] each
    unsigned long long *chem_counters = nullptr;   // first bad cell, bits of the largest change, bisection steps
    long long chem_steps = 0;

    void drop_chem_grid()
    {
        if (chem_level) { (void)hipFree(chem_level); chem_level = nullptr; }
        if (chem_logtem) { (void)hipFree(chem_logtem); chem_logtem = nullptr; }
        if (chem_out) { (void)hipFree(chem_out); chem_out = nullptr; }
        if (chem_J) { (void)hipFree(chem_J); chem_J = nullptr; }
        chem_temperature_set = false;
    }
};

// Which form of the brick kernel sweeps: 0 one wavefront per brick, 2 a pair of wavefronts per brick.
// Option "team" = -1 (the default) leaves it to the parallelism: with four frequency groups or fewer on this GPU (a rank of a
// frequency-sharded run) the stages are narrow, and the pair form's twice as many wavefronts fill them better (5 / 7 / 9 %
// at 4 / 2 / 1 groups); at eight the single wavefront is 1.5 % ahead.  The dataflow launch is built for form 0 only.
inline int brick_form(const ftte_ctx *c, int nnu)
{
    // (with the reference's emissivity term -- its log-mean needs a division and two polynomials per piece -- the pair form is ahead at
    // eight groups as well: 103 instead of 162 VGPRs; a source function costs three instructions per piece and goes as no emission)
    return c->team >= 0 ? c->team : (((nnu <= 4 || c->emit_mode) && !c->dataflow) ? 2 : 0);
}


namespace ftte {

int fail(ftte_ctx *c, int code, const std::string &msg);

#define FTTE_HIP(c, call)                                                                                          \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return fail((c), FTTE_ERR_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

int fold_status(int rc);

template <typename T> int ensure(ftte_ctx *c, T **p, size_t *cap, size_t need)
{
    if (*cap >= need && *p) return FTTE_OK;
    if (*p) FTTE_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    FTTE_HIP(c, hipMalloc((void **)p, std::max<size_t>(need, 1) * sizeof(T)));
    *cap = need;
    return FTTE_OK;
}

// ---- ftte_plan.cpp
// A cubic sub-grid planned like a grid of its own (the fine cells of a fully refined block): side, cell size, and where the layers'
// patterns come from (`patterns` fills n of them for direction d, folded to phi, theta, izone; returns 0 or an ftte_status)
struct SubGridPlan {
    int n = 0;
    double cell = 0;
    std::function<int(int d, double phi, double theta, int izone, ftte_pattern *out)> patterns;
};
int plan_direction(ftte_ctx *c, int d, double phi_d, double theta_d, double w_d, int tile_rows, std::vector<ftte_pattern> &pat,
                   std::vector<int> &du_cum, std::vector<int> &dv_cum, DirPlan &D, LayerRec *layers, size_t layer_off,
                   const SubGridPlan *sub = nullptr);
int build_plan(ftte_ctx *c, int rows, int stack, int ndir, const double *phi, const double *theta, const double *w);
int plan_brick_groups(ftte_ctx *c, BrickPlan &P, int ndir, const double *phi, const double *theta, const double *w, int chunk, int gmax,
                      int want_dataflow, bool whole_faces, const SubGridPlan *sub = nullptr);
int build_brick_plan(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w);
int xcc_census(ftte_ctx *c); // fills ftte_ctx::xcc_count, xcc_queue (once per context)

// ---- ftte_sweeps.cpp
int ensure_kappa(ftte_ctx *c, int nnu);
int check_ready(ftte_ctx *c, bool need_kappa);
int wait_sweep(ftte_ctx *c);
int mark_sweep(ftte_ctx *c, hipStream_t stream);
void free_forests(ftte_ctx *c);

// One direction of a forest pass as the host knows it
struct ForestDirHost {
    const SegRec *rec; const uint8_t *active; double w;
    double *faces; const AmrExport *exports; int64_t nexports; // hybrid sweep only, else null / 0
    const std::vector<int64_t> *depth_off;
    const std::vector<int32_t> *pass_first;    // the passes of depth_off (AmrForest::pass_first), or null: one pass
    const std::vector<int64_t> *export_first;  // the passes of exports, or null: all in the first
    const AmrImport *imports = nullptr; int64_t nimports = 0;  // hybrid sweep with a fine block swept by bricks
};

// A forest pass made ready: the per-direction records and the per-depth tables are in device memory (a batch of 96 would not
// fit the kernel arguments), what is left is a list of launches.
struct ForestRun {
    struct Pass { size_t table_at = 0, most_at = 0, maxdepth = 0, export_at = 0; int64_t most_exports = 0; };
    struct Batch { int d0 = 0, nb = 0; std::vector<Pass> passes; };
    std::vector<Batch> batches;
    std::vector<int64_t> most_of;
    size_t dir_at = 0;
};

int prepare_forests(ftte_ctx *c, hipStream_t stream, const std::vector<std::vector<ForestDirHost>> &sets, const std::vector<int> &slot0,
                    int batch, size_t per_dir, std::vector<ForestRun> *runs);
int launch_forest_pass(ftte_ctx *c, hipStream_t stream, const ForestRun &R, size_t b, size_t p, AmrLevelRec A);
int launch_forest_combine(ftte_ctx *c, hipStream_t stream, const ForestRun &R, size_t b, AmrLevelRec A, double *J_dev, bool zero_first);
int launch_forests(ftte_ctx *c, hipStream_t stream, const ForestRun &R, AmrLevelRec A, double *J_dev, bool zero_first, bool time_batches,
                   hipEvent_t before_combine, hipEvent_t after_combine);
int run_forests(ftte_ctx *c, hipStream_t stream, const std::vector<ForestDirHost> &dirs, int batch, size_t per_dir, AmrLevelRec A,
                double *J_dev, bool zero_first, bool time_batches);
int forest_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb,
                 double *J_dev, hipStream_t stream);

// The sweep of a uniform grid by cell-fixed bricks (ftte_brick.hip): one launch per stage, then one merge of the groups'
// accumulators (layout after layout, group after group: a fixed order) into J.
// Host arrays handed over with the sweep (ftte_diffuse_iteration): the opacities go up and J comes back one lane of frequency
// groups at a time, on the lane's own stream, so that the first lane is swept while the second one's opacities are still on the
// PCIe link and its J travels back while the second is swept.
struct HostPipe { const double *kappa; double *J; };
int brick_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
                hipStream_t stream, const HostPipe *pipe = nullptr);

int tile_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
               hipStream_t stream);

// ---- ftte_hybrid.cpp
void free_hybrid(ftte_ctx *c);
int hybrid_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
                 hipStream_t stream, bool *done);

// ---- ftte_multi.cpp: several devices behind one context
int multi_create(ftte_ctx **out, int ndev, const int *dev_ids);
int multi_destroy(ftte_ctx *c);
int multi_set_grid(ftte_ctx *c, int nx, int ny, int nz, int64_t ncell, const int32_t *level, double box_cm);
int multi_set_opacity(ftte_ctx *c, int nnu, const double *kappa);
int multi_set_emission(ftte_ctx *c, int mode, const double *values);
int multi_set_option(ftte_ctx *c, const char *key, int value);
int multi_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J);
int multi_iteration(ftte_ctx *c, int nnu, const double *kappa, int ndir, const double *phi, const double *theta, const double *w,
                    const double *uvb, double *J);
long long multi_counter(const ftte_ctx *c, const char *name);
const char *multi_how(const ftte_ctx *c);
ftte_ctx *multi_first(const ftte_ctx *c);
int multi_host_register(ftte_ctx *c, void *ptr, size_t bytes, bool on);

// ---- ftte_host_arrays.cpp
bool is_registered(const ftte_ctx *c, const void *p, size_t bytes);
int upload(ftte_ctx *c, void *dst_dev, const void *src_host, size_t bytes);
int download(ftte_ctx *c, void *dst_host, const void *src_dev, size_t bytes);
int upload_on(ftte_ctx *c, hipStream_t q, void *dst_dev, const void *src_host, size_t bytes);
int download_on(ftte_ctx *c, hipStream_t q, void *dst_host, const void *src_dev, size_t bytes);

} // namespace ftte
