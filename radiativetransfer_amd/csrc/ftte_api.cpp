// ftte_api.cpp -- the C ABI of include/ftte.h: context, device buffers, the per-sweep host
// planner (directions -> layer tables, tiles, launches) and the launch sequence.
//
// There is no CPU fallback in this file: every entry point that computes on the grid needs a
// HIP device and fails with FTTE_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <thread>

#include "../../include/ftte.h"
#include "ftte_amr.h"
#include "ftte_geometry.h"
#include "ftte_internal.h"
#include "ftte_kernels.h"
#include "ftte_point.h"

using namespace ftte;

namespace {

std::string g_create_error;

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
    int ut = kBrickRows, uw = 0;       // u-face ring: doubles per brick and layer, per layer
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

} // namespace

struct ftte_ctx {
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
    int engine = 0, chunk = 0, group = 0, brick_waves = 4, share = 2, team = 0, lanes = 2; // chunk, group: 0 = by the parallelism (build_brick_plan)
    std::vector<hipStream_t> lane_stream;   // extra streams of the brick sweep (frequency groups are independent)
    std::vector<hipEvent_t> pipe_up;        // ftte_diffuse_iteration: lane k's opacities have arrived
    bool stage_used[2] = {false, false};    // the pinned staging block has a transfer recorded on stage_ev
    std::vector<hipEvent_t> lane_done;
    hipEvent_t ev_fork = nullptr;
    // option: 0 = a launch per stage (default); 1, 2 = the bricks of a sweep in ONE launch where the grid allows it, waiting for each
    // other through flags (measured: no faster -- the stage boundaries are not what limits the sweep, DESIGN.md -- so not the default)
    int dataflow = 0;
    int32_t *d_bdeps = nullptr; size_t d_bdeps_cap = 0;
    uint32_t *d_bdone = nullptr; size_t d_bdone_cap = 0;
    uint32_t *d_bsync = nullptr;      // [0] ticket, [1] error
    uint32_t *h_berror = nullptr;     // pinned: the error flag of the last dataflow sweep, copied back behind it
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
    int halves = 3;                       // option "pipelines": the hybrid sweep as this many independent pipelines on streams of their own (1..kMaxPipes)
    static constexpr int kMaxPipes = 4;
    hipEvent_t ev_combine[kMaxPipes] = {nullptr, nullptr, nullptr, nullptr}; // hybrid sweep: pipeline k's forest means are in J
    struct HybridPlan {
        bool valid = false, worthwhile = false;
        std::vector<double> key;          // box, chunk, group, share, then phi, theta, w
        BrickPlan bricks;                 // groups, tasks of the bricks outside the regions (phase 1, then phase 3)
        size_t phase1_stages = 0;         // per half: stage lists [0, phase1_stages) come before the forest pass, the rest after it
        size_t nlist = 0;                 // stage lists per half (2 x phase1_stages)
        int nhalves = 1;                  // the groups of an accumulator stay in one half; halves share nothing but kappa and J
        std::vector<std::vector<int>> half_dirs; // directions of each half, list order
        std::vector<size_t> stage_off;    // into bricks.tasks: [half][list]
        int64_t brick_updates = 0;        // cell.direction updates the bricks perform (per frequency group)
        struct Dir { SegRec *rec = nullptr; uint8_t *active = nullptr; AmrExport *exports = nullptr; int64_t nexports = 0;
                     std::vector<int64_t> depth_off; };
        std::vector<Dir> dirs;
        int32_t *cells = nullptr; int64_t ncells = 0; // the leaves inside the box of at least one direction
        bool uploaded = false;
    } hplan;
    int32_t *d_leaf_of_base = nullptr;
    double *base_kappa[3] = {nullptr, nullptr, nullptr};
    size_t base_kappa_cap = 0;

    PointState point; // point sources: rate tables, medium, tracer scratch

    // host-array boundary (ftte_set_opacity / ftte_diffuse_sweep): J lives in a device buffer the context keeps, and
    // pageable host arrays cross PCIe through two pinned staging blocks filled by a few host threads while the other
    // block is in flight; arrays the caller has registered (ftte_host_register) are copied by the DMA engine directly
    double *host_J_dev = nullptr; size_t host_J_cap = 0;
    void *stage[2] = {nullptr, nullptr};
    hipEvent_t stage_ev[2] = {nullptr, nullptr};
    struct HostRange { const char *base; size_t bytes; };
    std::vector<HostRange> registered;

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

namespace {

int fail(ftte_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define FTTE_HIP(c, call)                                                                                          \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return fail((c), FTTE_ERR_NO_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

int fold_status(int rc)
{
    return rc == 1 ? FTTE_ERR_PHI : rc == 2 ? FTTE_ERR_THETA : FTTE_ERR_DOMINANT_AXIS;
}

// ---- planner ------------------------------------------------------------------------------------
// One direction: fold it (equiSources.f90:1395-1454), build its per-layer patterns (:1495-1534, setPattern) and turn them
// into what the kernels read: the memory frame of its izone and one LayerRec per layer.
int plan_direction(ftte_ctx *c, int d, double phi_d, double theta_d, double w_d, int tile_rows, std::vector<ftte_pattern> &pat,
                   std::vector<int> &du_cum, std::vector<int> &dv_cum, DirPlan &D, LayerRec *layers_of_d, size_t layer_off)
{
    const int n = c->n;
    const double cell = c->box / (double)n; // cellSizeAbsoluteUnits, equiSources.f90:1570
    const long nn = (long)n * n;
    D.w = w_d;

    int rc = fold_direction(phi_d, theta_d, &D.phi, &D.theta, &D.izone);
    if (rc) {
        char buf[160];
        std::snprintf(buf, sizeof buf, "direction %d (phi=%.17g, theta=%.17g) cannot be folded: %s", d, phi_d, theta_d,
                      rc == 1 ? "phi on a quadrant boundary" : rc == 2 ? "theta outside (-pi/2,0)u(0,pi/2)"
                                                                       : "tie between dominant axes");
        return fail(c, fold_status(rc), buf);
    }
    if (layer_patterns(n, D.phi, D.theta, pat.data())) {
        char buf[128];
        std::snprintf(buf, sizeof buf, "direction %d: ray pattern left the unit cell (setPattern consistency check)", d);
        return fail(c, FTTE_ERR_PATTERN, buf);
    }

    // memory frame of this izone: which storage axis the march runs along decides the layout;
    // within it u = the sweep axis that lands on the contiguous storage axis
    ZoneMap zm;
    zone_map(D.izone, &zm);
    int march_c = 0;
    for (int a = 0; a < 3; ++a) if (zm.src[a] == 0) march_c = a;
    D.layout = march_c;
    const int fast_c = (march_c == 2) ? 1 : 2;
    const int mid_c = (march_c == 0) ? 1 : 0;
    const bool u_is_k = zm.src[fast_c] == 2;
    D.su = zm.mirror[fast_c] ? -1 : 1;
    D.sv = zm.mirror[mid_c] ? -n : n;
    D.si = (int)(zm.mirror[march_c] ? -nn : nn);
    // the column enters as a position p = u (or n+1-u when mirrored) with stride +1: offset p - 1
    D.org = -1 + (zm.mirror[mid_c] ? (long)n * n : -(long)n) +
            (zm.mirror[march_c] ? (long)n * nn : -nn);

    // layers: reference chain -> kernel-frame class, lengths in chain order, cumulative drift
    D.layer_off = layer_off;
    int du = 0, dv = 0;
    for (int i = 0; i < n; ++i) {
        const ftte_pattern &p = pat[i];
        LayerRec &R = layers_of_d[i];
        R.dpath[0] = cell * p.xy_len;
        R.dpath[1] = R.dpath[2] = 0.0;
        int rc_class = RC_ONE, step_k = 0, step_j = 0;
        if (p.xz_active && p.yz_active) {
            step_k = step_j = 1;
            if (p.xy_top == 3) { // xy -> yz -> xz (the xz piece reaches the top)
                R.dpath[1] = cell * p.yz_len; R.dpath[2] = cell * p.xz_len;
                rc_class = u_is_k ? RC_THREE_U_SWAP : RC_THREE_V_SWAP; // mean adds xy, xz, yz: 3rd piece before 2nd
            } else {             // xy -> xz -> yz
                R.dpath[1] = cell * p.xz_len; R.dpath[2] = cell * p.yz_len;
                rc_class = u_is_k ? RC_THREE_V : RC_THREE_U;
            }
        } else if (p.yz_active) { // xy -> yz: one cell further along sweep-k
            step_k = 1;
            R.dpath[1] = cell * p.yz_len;
            rc_class = u_is_k ? RC_TWO_U : RC_TWO_V;
        } else if (p.xz_active) { // xy -> xz: one cell further along sweep-j
            step_j = 1;
            R.dpath[1] = cell * p.xz_len;
            rc_class = u_is_k ? RC_TWO_V : RC_TWO_U;
        }
        R.info = rc_class;
        R.drift = (du & 0xffff) | (dv << 16);
        du_cum[i] = du; dv_cum[i] = dv;
        du += u_is_k ? step_k : step_j;
        dv += u_is_k ? step_j : step_k;
    }
    // rays present at the last layer start at label -drift (base cell 0, second piece in cell 1)
    D.u_lo = 1 - du_cum[n - 1];
    D.v_lo = 1 - dv_cum[n - 1];
    D.du_mid = du_cum[n / 2];
    D.dv_mid = dv_cum[n / 2];
    D.ntu = (n - D.u_lo + 1 + 62) / 63;
    D.ntv = (n - D.v_lo + 1 + tile_rows - 1) / tile_rows;

    return FTTE_OK;
}

// Turns the direction list into what the kernel consumes.  O(ndir * (n + tiles)) host work,
// cached in the context for as long as the directions, the grid and the tuning stay the same.
int build_plan(ftte_ctx *c, int rows, int stack, int ndir, const double *phi, const double *theta, const double *w)
{
    Plan &P = c->plan;
    const int n = c->n, slots = c->slots;
    const int tile_rows = stack * rows - 1; // owned rows of one work item
    if (P.valid && P.n == n && P.rows == rows && P.slots == slots && P.stack == stack && P.box == c->box && (int)P.phi.size() == ndir &&
        (ndir == 0 || (!std::memcmp(P.phi.data(), phi, sizeof(double) * ndir) &&
                       !std::memcmp(P.theta.data(), theta, sizeof(double) * ndir) &&
                       !std::memcmp(P.w.data(), w, sizeof(double) * ndir))))
        return FTTE_OK;

    ++c->n_plan_builds;
    P = Plan();
    P.n = n; P.rows = rows; P.slots = slots; P.stack = stack; P.box = c->box;
    P.phi.assign(phi, phi + ndir); P.theta.assign(theta, theta + ndir); P.w.assign(w, w + ndir);
    P.dirs.resize(ndir);
    P.layers.resize((size_t)ndir * n);
    c->plan_uploaded = false;

    std::vector<ftte_pattern> pat(n);
    std::vector<int> du_cum(n + 1), dv_cum(n + 1);
    int in_layout[3] = {0, 0, 0};

    for (int d = 0; d < ndir; ++d) {
        DirPlan &D = P.dirs[d];
        const int rc = plan_direction(c, d, phi[d], theta[d], w[d], tile_rows, pat, du_cum, dv_cum, D, &P.layers[(size_t)d * n], (size_t)d * n);
        if (rc) return rc;
        D.slot = in_layout[D.layout]++ % slots;
    }

    // launches: per layout, batches of `slots` directions in input order
    for (int layout = 0; layout < 3; ++layout) {
        std::vector<int> members;
        for (int d = 0; d < ndir; ++d) if (P.dirs[d].layout == layout) members.push_back(d);
        for (size_t b = 0; b < members.size(); b += slots) {
            LaunchPlan LP;
            LP.layout = layout;
            LP.first = (b == 0);
            // a short last batch takes the highest accumulators: the lower ones are final one launch earlier and can be
            // merged while it runs, without changing the order in which the accumulators are added up
            const int in_batch = (int)(std::min(members.size(), b + (size_t)slots) - b);
            LP.acc_base = (b > 0 && in_batch < slots) ? slots - in_batch : 0;
            LP.item_off = P.items.size();
            std::vector<uint32_t> where; // per item of this launch: the tile's place in the plane halfway through the march
            for (size_t s = b; s < std::min(members.size(), b + (size_t)slots); ++s) {
                const int d = members[s];
                const DirPlan &D = P.dirs[d];
                const int slot = (int)(s - b);
                LP.dirs.push_back(d);
                P.used[layout][LP.acc_base + slot] = true;
                const LayerRec *Ls = &P.layers[D.layer_off];
                for (int tv = 0; tv < D.ntv; ++tv) {
                    for (int tu = 0; tu < D.ntu; ++tu) {
                        // owned labels of this tile; a layer is active when any owned ray, or the cell one
                        // step beyond it, is inside the domain
                        const int ul_min = D.u_lo + 63 * tu, ul_max = ul_min + 62;
                        const int vl_min = D.v_lo + tile_rows * tv, vl_max = vl_min + tile_rows - 1;
                        int i_first = 0, i_last = -1;
                        for (int i = 1; i <= n; ++i) {
                            const int cu_d = (int)(short)(Ls[i - 1].drift & 0xffff), cv_d = Ls[i - 1].drift >> 16;
                            const bool act = ul_min + cu_d <= n && ul_max + cu_d + 1 >= 1 && vl_min + cv_d <= n &&
                                             vl_max + cv_d + 1 >= 1;
                            if (act) { if (!i_first) i_first = i; i_last = i; }
                        }
                        if (!i_first) continue;
                        WorkItem it;
                        it.slot = (int16_t)slot; it.tu = (int16_t)tu; it.tv = (int16_t)tv;
                        it.i_first = (int16_t)i_first; it.i_last = (int16_t)i_last; it.pad = 0;
                        P.items.push_back(it);
                        const int pu = std::max(0, ul_min + D.du_mid + 64) / 64, pv = std::max(0, vl_min + D.dv_mid + 64) / std::max(tile_rows, 1);
                        where.push_back(((uint32_t)pv << 16) | (uint32_t)(pu & 0xffff));
                    }
                }
                LP.updates += (int64_t)n * n * n;
            }
            LP.nitems = (int)(P.items.size() - LP.item_off);
            // longest marches first, so that the short corner tiles fill the tail of the launch.  (Grouping the tiles
            // of one direction together instead -- hoping for L2 hits on shared halo rows -- was measured: no drop in
            // FETCH_SIZE, 6 % slower through worse load balance.)
            {
                // longest marches first, so that the short corner tiles fill the tail of the launch; among equally long
                // ones, tiles of the directions in flight that cross the same part of the grid side by side, so that they
                // read the same part of a kappa plane at about the same time (+2 %; the place is taken halfway through the march.
                // Grouping by direction instead: -6 %)
                std::vector<uint32_t> idx(where.size());
                for (size_t q = 0; q < idx.size(); ++q) idx[q] = (uint32_t)q;
                const WorkItem *base = P.items.data() + LP.item_off;
                std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
                    const int lx = base[x].i_last - base[x].i_first, ly = base[y].i_last - base[y].i_first;
                    if (lx != ly) return lx > ly;
                    if (where[x] != where[y]) return where[x] < where[y];
                    return base[x].slot < base[y].slot;
                });
                std::vector<WorkItem> sorted(idx.size());
                for (size_t q = 0; q < idx.size(); ++q) sorted[q] = base[idx[q]];
                std::copy(sorted.begin(), sorted.end(), P.items.begin() + LP.item_off);
            }
            P.launches.push_back(LP);
        }
    }
    P.valid = true;
    return FTTE_OK;
}

// The part of a brick plan that does not depend on which bricks are swept: the directions, the brick geometry and the face
// block layout, the groups and their accumulators.
int plan_brick_groups(ftte_ctx *c, BrickPlan &P, int ndir, const double *phi, const double *theta, const double *w, int chunk, int gmax,
                      int want_dataflow, bool whole_faces)
{
    const int n = c->n;
    ++c->n_plan_builds;
    P = BrickPlan();
    P.n = n; P.chunk = chunk; P.gmax = gmax; P.share = c->share; P.want_dataflow = want_dataflow; P.box = c->box;
    P.phi.assign(phi, phi + ndir); P.theta.assign(theta, theta + ndir); P.w.assign(w, w + ndir);
    P.dirs.resize(ndir);
    P.layers.resize((size_t)ndir * n);

    std::vector<ftte_pattern> pat(n);
    std::vector<int> du_cum(n + 1), dv_cum(n + 1);
    for (int d = 0; d < ndir; ++d) {
        const int rc = plan_direction(c, d, phi[d], theta[d], w[d], 7, pat, du_cum, dv_cum, P.dirs[d], &P.layers[(size_t)d * n], (size_t)d * n);
        if (rc) return rc;
    }
    P.ntu = (n + 63) / 64; P.ntv = (n + kBrickRows - 1) / kBrickRows; P.nti = (n + chunk - 1) / chunk;
    P.up = 64 * P.ntu; P.vp = kBrickRows * P.ntv;
    P.dataflow = want_dataflow != 0;
    P.ut = P.dataflow ? 16 : kBrickRows; // a 128-byte line of its own per brick and layer when bricks of one launch exchange rays
    P.uw = P.ntv * P.ut;
    P.nslot = whole_faces ? P.nti : 2; // rings over two chunks, or every chunk's faces kept (hybrid sweep)
    P.vface_off = (int64_t)P.ntu * P.nslot * chunk * P.uw;
    P.iface_off = P.vface_off + (int64_t)P.ntv * P.nslot * chunk * P.up;
    P.face_elems = P.iface_off + (int64_t)P.nslot * P.vp * P.up;

    if (P.nti >= kBrickAccumulate) return fail(c, FTTE_ERR_UNSUPPORTED, "brick engine: more than 16383 chunks along the march axis: raise option \"chunk\"");

    // Groups: layout after layout (the order in which the merge adds the accumulators), izone after izone, at most gmax
    // directions each.  Accumulators: a group stores its J contribution once per cell, and every accumulator costs the merge
    // one more read of the grid, so groups share an accumulator where they provably never meet in a brick in the same launch
    // (the later one then reads, adds and stores, BrickTask):
    //   * the passes of one izone sweep the bricks in the same order: started in different launches they never meet;
    //   * two izones of one layout differ by reflections of the brick order along some axes; with t -> N-1-t along an axis
    //     of even brick count N the difference of their stage numbers in a brick changes by an odd amount, so if an odd number
    //     of such axes is reflected the difference is odd in every brick, and start launches that differ by an even number
    //     never bring them together.  Needs bricks that coincide under reflection: n a multiple of 64, 8 and the chunk.
    const bool aligned = n % 64 == 0 && n % kBrickRows == 0 && n % chunk == 0;
    const int nbricks[3] = {P.ntu, P.ntv, P.nti};
    for (int layout = 0; layout < 3; ++layout) {
        struct Zone { int izone, parity; std::vector<std::vector<int>> passes; };
        std::vector<Zone> zones;
        for (int izone = 1; izone <= 24; ++izone) {
            std::vector<int> members;
            for (int d = 0; d < ndir; ++d)
                if (P.dirs[d].izone == izone && P.dirs[d].layout == layout) members.push_back(d);
            if (members.empty()) continue;
            Zone Z;
            Z.izone = izone;
            const DirPlan &D0 = P.dirs[members[0]];
            const bool mirror[3] = {D0.su < 0, D0.sv < 0, D0.si < 0};
            Z.parity = 0;
            for (int a = 0; a < 3; ++a) if (mirror[a] && nbricks[a] % 2 == 0) Z.parity ^= 1;
            // as few passes as gmax allows, of equal size where possible (5 directions, gmax 4: 3 + 2, not 4 + 1)
            const size_t npass = (members.size() + (size_t)gmax - 1) / (size_t)gmax;
            for (size_t b = 0, q = 0; q < npass; ++q) {
                const size_t len = members.size() / npass + (q < members.size() % npass ? 1 : 0);
                Z.passes.emplace_back(members.begin() + (long)b, members.begin() + (long)(b + len));
                b += len;
            }
            zones.push_back(Z);
        }
        // pair the izones of opposite parity (share = 2); share = 1: only the passes of one izone share; 0: nobody shares
        std::vector<int> partner(zones.size(), -1);
        if (aligned && c->share >= 2)
            for (size_t x = 0; x < zones.size(); ++x) {
                if (partner[x] >= 0) continue;
                for (size_t y = x + 1; y < zones.size(); ++y)
                    if (partner[y] < 0 && zones[y].parity != zones[x].parity) { partner[x] = (int)y; partner[y] = (int)x; break; }
            }
        std::vector<int> acc_of(zones.size(), -1);
        for (size_t x = 0; x < zones.size(); ++x) {
            const bool paired = partner[x] >= 0;
            if (c->share >= 1) {
                if (acc_of[x] < 0) {
                    acc_of[x] = P.nacc[layout]++;
                    if (paired) acc_of[(size_t)partner[x]] = acc_of[x];
                }
            }
            for (size_t p = 0; p < zones[x].passes.size(); ++p) {
                BrickPlan::Group G;
                G.izone = zones[x].izone; G.layout = layout;
                G.acc = c->share >= 1 ? acc_of[x] : P.nacc[layout]++;
                G.offset = c->share >= 1 ? (int)p * (paired ? 2 : 1) : 0;
                G.dirs = zones[x].passes[p];
                P.max_dirs = std::max(P.max_dirs, (int)G.dirs.size());
                P.groups.push_back(G);
            }
        }
    }
    for (int layout = 0; layout < 3; ++layout)
        if (P.nacc[layout] > kMaxAcc) return fail(c, FTTE_ERR_UNSUPPORTED, "too many direction groups for one memory layout: raise option \"group\"");

    return FTTE_OK;
}

// Bricks: group the directions by izone (input order within an izone, at most `group` per group), cut the grid into
// bricks of 64 x kBrickRows x chunk cells, and order the bricks of every group into stages tu + tv + ti: a brick's three
// upstream neighbours lie one stage earlier, its consumers exactly one stage later (which is what lets the face buffers be
// rings over two chunks).  Pure host work, cached like the tile plan.
int build_brick_plan(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w)
{
    BrickPlan &P = c->bplan;
    const int n = c->n, nnu = c->nnu;
    int rc;
    // Unset options (0) follow the parallelism there is: a stage offers (bricks of a plane) x groups x frequency groups
    // tasks, and with few frequency groups on this GPU (a rank of a frequency-sharded run) shorter bricks and smaller
    // groups keep the stages wide enough; the groups are then dealt to the streams instead of the frequency groups.
    const int chunk = std::min(c->chunk > 0 ? c->chunk : (nnu >= 4 ? 16 : nnu >= 2 ? 8 : 4), n);
    const int gmax = c->group > 0 ? c->group : (nnu >= 2 ? 3 : 2);
    const int want_dataflow = (c->dataflow && n % 64 == 0 && n % kBrickRows == 0 && n % chunk == 0 && !c->team && !c->emit_mode) ? 1 : 0;
    const int want_glanes = want_dataflow ? 1 : (nnu >= c->lanes ? 1 : c->lanes);
    if (P.valid && P.n == n && P.chunk == chunk && P.gmax == gmax && P.share == c->share && P.want_glanes == want_glanes &&
        P.want_dataflow == want_dataflow && P.box == c->box &&
        (int)P.phi.size() == ndir &&
        (ndir == 0 || (!std::memcmp(P.phi.data(), phi, sizeof(double) * ndir) &&
                       !std::memcmp(P.theta.data(), theta, sizeof(double) * ndir) &&
                       !std::memcmp(P.w.data(), w, sizeof(double) * ndir))))
        return FTTE_OK;

    if ((rc = plan_brick_groups(c, P, ndir, phi, theta, w, chunk, gmax, want_dataflow, false))) return rc;
    P.want_glanes = want_glanes;
    c->bplan_uploaded = false;

    // streams: the groups of one accumulator stay on one stream (their launches are ordered against each other)
    P.glanes = std::max(1, std::min(want_glanes, P.nacc[0] + P.nacc[1] + P.nacc[2]));
    {
        int next = 0;
        std::vector<int> lane_of(3 * (size_t)kMaxAcc, -1);
        for (auto &G : P.groups) {
            int &l = lane_of[(size_t)G.layout * kMaxAcc + G.acc];
            if (l < 0) l = next++ % P.glanes;
            G.lane = l;
        }
    }
    int max_offset = 0;
    for (const auto &G : P.groups) max_offset = std::max(max_offset, G.offset);
    const int nstages = P.groups.empty() ? 0 : P.ntu + P.ntv + P.nti - 2 + max_offset;
    P.nstages = nstages;
    const size_t per_lane = (size_t)nstages + 1;
    P.stage_off.assign((size_t)P.glanes * per_lane, 0);
    P.updates = 0;
    if (!P.groups.empty()) {
        // launch in which each accumulator's cells are first written, per physical brick: whoever comes later accumulates
        const size_t nb = (size_t)P.ntu * P.ntv * P.nti;
        std::vector<std::vector<int>> first(3 * (size_t)kMaxAcc);
        auto brick_of = [&](const BrickPlan::Group &G, int tu, int tv, int ti) {
            const DirPlan &D0 = P.dirs[G.dirs[0]];
            const int bu = D0.su < 0 ? P.ntu - 1 - tu : tu, bv = D0.sv < 0 ? P.ntv - 1 - tv : tv, bi = D0.si < 0 ? P.nti - 1 - ti : ti;
            return ((size_t)bi * P.ntv + bv) * P.ntu + bu;
        };
        for (const auto &G : P.groups) {
            std::vector<int> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
            if (F.empty()) F.assign(nb, 1 << 30);
            for (int ti = 0; ti < P.nti; ++ti)
                for (int tv = 0; tv < P.ntv; ++tv)
                    for (int tu = 0; tu < P.ntu; ++tu) {
                        int &f = F[brick_of(G, tu, tv, ti)];
                        f = std::min(f, tu + tv + ti + G.offset);
                    }
        }
        // count per (lane, stage) in slot [lane][stage + 1], turn into offsets (lanes one after the other), then fill
        for (const auto &G : P.groups)
            for (int ti = 0; ti < P.nti; ++ti)
                for (int tv = 0; tv < P.ntv; ++tv)
                    for (int tu = 0; tu < P.ntu; ++tu) ++P.stage_off[(size_t)G.lane * per_lane + (size_t)(tu + tv + ti + G.offset) + 1];
        size_t run = 0;
        for (int l = 0; l < P.glanes; ++l) {
            P.stage_off[(size_t)l * per_lane] = run;
            for (int st = 0; st < nstages; ++st) {
                const size_t cnt = P.stage_off[(size_t)l * per_lane + (size_t)st + 1];
                P.stage_off[(size_t)l * per_lane + (size_t)st + 1] = P.stage_off[(size_t)l * per_lane + (size_t)st] + cnt;
            }
            run = P.stage_off[(size_t)l * per_lane + (size_t)nstages];
        }
        P.tasks.resize(run);
        std::vector<size_t> fill(P.stage_off);
        // within a stage the groups with the most directions first: their bricks take longest, the short ones fill the tail
        std::vector<size_t> by_size(P.groups.size());
        for (size_t g = 0; g < by_size.size(); ++g) by_size[g] = g;
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return P.groups[x].dirs.size() > P.groups[y].dirs.size(); });
        for (size_t g : by_size) {
            const BrickPlan::Group &G = P.groups[g];
            const std::vector<int> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
            for (int ti = 0; ti < P.nti; ++ti)
                for (int tv = 0; tv < P.ntv; ++tv)
                    for (int tu = 0; tu < P.ntu; ++tu) {
                        const int st = tu + tv + ti + G.offset;
                        BrickTask T;
                        T.group = (int16_t)g; T.tu = (int16_t)tu; T.tv = (int16_t)tv;
                        T.ti = (int16_t)(ti | (st > F[brick_of(G, tu, tv, ti)] ? kBrickAccumulate : 0));
                        P.tasks[fill[(size_t)G.lane * per_lane + (size_t)st]++] = T;
                        const int64_t cu = std::min(64, n - 64 * tu), cv = std::min(kBrickRows, n - kBrickRows * tv),
                                      ci = std::min(chunk, n - chunk * ti);
                        P.updates += cu * cv * ci * (int64_t)G.dirs.size();
                    }
        }
    }
    if (P.dataflow && !P.tasks.empty()) {
        // what each brick waits for.  All of them lie earlier in the (stage-ordered) list.
        const size_t nt = P.tasks.size(), nb = (size_t)P.ntu * P.ntv * P.nti;
        std::vector<int32_t> index(P.groups.size() * nb, -1);
        auto at = [&](size_t g, int tu, int tv, int ti) -> int32_t & { return index[g * nb + ((size_t)ti * P.ntv + tv) * P.ntu + tu]; };
        for (size_t q = 0; q < nt; ++q) at((size_t)P.tasks[q].group, P.tasks[q].tu, P.tasks[q].tv, P.tasks[q].ti & (kBrickAccumulate - 1)) = (int32_t)q;
        P.deps.assign(nt * kBrickDeps, -1);
        // the visitors of every J tile, per accumulator, in launch order
        struct Visit { int launch; int32_t task; };
        std::vector<std::vector<std::vector<Visit>>> visits(3 * (size_t)kMaxAcc);
        for (size_t q = 0; q < nt; ++q) {
            const BrickTask &T = P.tasks[q];
            const BrickPlan::Group &G = P.groups[(size_t)T.group];
            const int ti = T.ti & (kBrickAccumulate - 1);
            int32_t *D = &P.deps[q * kBrickDeps];
            if (T.tu > 0) D[0] = at((size_t)T.group, T.tu - 1, T.tv, ti);
            if (T.tv > 0) D[1] = at((size_t)T.group, T.tu, T.tv - 1, ti);
            if (ti > 0) D[2] = at((size_t)T.group, T.tu, T.tv, ti - 1);
            if (ti >= 2 && T.tu + 1 < P.ntu) D[4] = at((size_t)T.group, T.tu + 1, T.tv, ti - 2); // read the u-face slot this brick rewrites
            if (ti >= 2 && T.tv + 1 < P.ntv) D[5] = at((size_t)T.group, T.tu, T.tv + 1, ti - 2); // the v-face slot
            auto &V = visits[(size_t)G.layout * kMaxAcc + G.acc];
            if (V.empty()) V.resize(nb);
            const DirPlan &D0 = P.dirs[G.dirs[0]];
            const int bu = D0.su < 0 ? P.ntu - 1 - T.tu : T.tu, bv = D0.sv < 0 ? P.ntv - 1 - T.tv : T.tv, bi = D0.si < 0 ? P.nti - 1 - ti : ti;
            V[((size_t)bi * P.ntv + bv) * P.ntu + bu].push_back({T.tu + T.tv + ti + G.offset, (int32_t)q});
        }
        for (auto &V : visits)
            for (auto &list : V) {
                std::sort(list.begin(), list.end(), [](const Visit &x, const Visit &y) { return x.launch < y.launch; });
                for (size_t k = 1; k < list.size(); ++k) P.deps[(size_t)list[k].task * kBrickDeps + 3] = list[k - 1].task;
            }
        for (size_t q = 0; q < nt; ++q)
            for (int k = 0; k < kBrickDeps; ++k)
                if (P.deps[q * kBrickDeps + k] >= (int32_t)q) return fail(c, FTTE_ERR_STATE, "brick plan: a dependency does not precede its brick");
    }
    P.valid = true;
    return FTTE_OK;
}

template <typename T> int ensure(ftte_ctx *c, T **p, size_t *cap, size_t need)
{
    if (*cap >= need && *p) return FTTE_OK;
    if (*p) FTTE_HIP(c, hipFree(*p));
    *p = nullptr; *cap = 0;
    FTTE_HIP(c, hipMalloc((void **)p, std::max<size_t>(need, 1) * sizeof(T)));
    *cap = need;
    return FTTE_OK;
}

int ensure_kappa(ftte_ctx *c, int nnu)
{
    const size_t need = (size_t)nnu * c->ncell;
    if (c->kappa[0] && c->kappa_cap >= need) return FTTE_OK;
    for (int l = 0; l < 3; ++l) {
        if (c->kappa[l]) { FTTE_HIP(c, hipFree(c->kappa[l])); c->kappa[l] = nullptr; }
        if (c->emis[l]) { FTTE_HIP(c, hipFree(c->emis[l])); c->emis[l] = nullptr; }
    }
    c->emit_mode = 0; // sized by the old number of groups: has to be set again
    FTTE_HIP(c, hipMalloc((void **)&c->kappa[0], need * sizeof(double)));
    c->kappa_cap = need;
    return FTTE_OK;
}

int check_ready(ftte_ctx *c, bool need_kappa)
{
    if (!c) return FTTE_ERR_ARG;
    if (!c->grid_set) return fail(c, FTTE_ERR_STATE, "ftte_set_grid has not been called");
    if (need_kappa && (!c->nnu || !c->kappa[0])) return fail(c, FTTE_ERR_STATE, "no opacities: call ftte_set_opacity / ftte_set_species first");
    return FTTE_OK;
}

// the previous sweep may have been issued on a stream of the caller's: wait for its end before its inputs are rewritten
int wait_sweep(ftte_ctx *c)
{
    if (c->sweep_pending) {
        FTTE_HIP(c, hipEventSynchronize(c->ev_sweep_done));
        c->sweep_pending = false;
        if (c->h_berror && *c->h_berror) {
            *c->h_berror = 0;
            return fail(c, FTTE_ERR_NO_DEVICE, "the previous sweep gave up: a brick waited too long for the bricks it depends on (its J is not valid)");
        }
    }
    return FTTE_OK;
}

int mark_sweep(ftte_ctx *c, hipStream_t stream)
{
    if (!c->ev_sweep_done) FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_sweep_done, hipEventDisableTiming));
    FTTE_HIP(c, hipEventRecord(c->ev_sweep_done, stream));
    c->sweep_pending = true;
    return FTTE_OK;
}

void free_forests(ftte_ctx *c)
{
    for (auto &f : c->forests) {
        if (f.rec) (void)hipFree(f.rec);
        if (f.active) (void)hipFree(f.active);
    }
    c->forests.clear();
    c->forest_key.clear();
}

// One direction of a forest pass as the host knows it
struct ForestDirHost {
    const SegRec *rec; const uint8_t *active; double w;
    double *faces; const AmrExport *exports; int64_t nexports; // hybrid sweep only, else null / 0
    const std::vector<int64_t> *depth_off;
};

// A forest pass made ready: the per-direction records and the per-depth tables are in device memory (a batch of 96 would not
// fit the kernel arguments), what is left is a list of launches.
struct ForestRun {
    struct Batch { int d0, nb; size_t table_at, most_at, maxdepth; int64_t most_exports; };
    std::vector<Batch> batches;
    std::vector<int64_t> most_of;
    size_t dir_at = 0;
};

// Tables of several independent passes (`sets`: direction lists that may run side by side on different streams, set q using the
// scratch slots from slot0[q] on), `batch` directions at a time each; built and uploaded in one go on `stream`.
int prepare_forests(ftte_ctx *c, hipStream_t stream, const std::vector<std::vector<ForestDirHost>> &sets, const std::vector<int> &slot0,
                    int batch, size_t per_dir, std::vector<ForestRun> *runs)
{
    int rc;
    std::vector<AmrDirRec> recs;
    std::vector<int64_t> tables;
    runs->assign(sets.size(), ForestRun());
    for (size_t q = 0; q < sets.size(); ++q) {
        const std::vector<ForestDirHost> &dirs = sets[q];
        ForestRun &R = (*runs)[q];
        const int ndir = (int)dirs.size();
        R.dir_at = recs.size();
        for (int d0 = 0; d0 < ndir; d0 += batch) {
            const int nb = std::min(batch, ndir - d0);
            ForestRun::Batch B{d0, nb, 0, 0, 0, 0};
            for (int t = 0; t < nb; ++t) {
                const ForestDirHost &D = dirs[(size_t)(d0 + t)];
                AmrDirRec rec;
                std::memset(&rec, 0, sizeof rec);
                rec.rec = D.rec; rec.active = D.active; rec.w = D.w;
                rec.Iout = c->amr_Iout + per_dir * (size_t)(slot0[q] + t);
                rec.mean = c->amr_mean + per_dir * (size_t)(slot0[q] + t);
                rec.faces = D.faces; rec.exports = D.exports; rec.nexports = D.nexports;
                recs.push_back(rec);
                B.maxdepth = std::max(B.maxdepth, D.depth_off->size() - 1);
                B.most_exports = std::max(B.most_exports, D.nexports);
            }
            B.table_at = tables.size();
            B.most_at = R.most_of.size();
            for (size_t depth = 0; depth < B.maxdepth; ++depth) {
                int64_t most = 0;
                const size_t at = tables.size();
                tables.resize(at + 2 * (size_t)nb, 0);
                for (int t = 0; t < nb; ++t) {
                    const std::vector<int64_t> &off = *dirs[(size_t)(d0 + t)].depth_off;
                    if (depth + 1 < off.size()) {
                        tables[at + (size_t)t] = off[depth + 1] - off[depth];
                        tables[at + (size_t)nb + (size_t)t] = off[depth];
                        most = std::max(most, off[depth + 1] - off[depth]);
                    }
                }
                R.most_of.push_back(most);
            }
            R.batches.push_back(B);
        }
    }
    if ((rc = ensure(c, &c->d_amr_dirs, &c->d_amr_dirs_cap, recs.size()))) return rc;
    if ((rc = ensure(c, &c->d_amr_tables, &c->d_amr_tables_cap, tables.size()))) return rc;
    if (!recs.empty()) FTTE_HIP(c, hipMemcpyAsync(c->d_amr_dirs, recs.data(), sizeof(AmrDirRec) * recs.size(), hipMemcpyHostToDevice, stream));
    if (!tables.empty()) FTTE_HIP(c, hipMemcpyAsync(c->d_amr_tables, tables.data(), sizeof(int64_t) * tables.size(), hipMemcpyHostToDevice, stream));
    FTTE_HIP(c, hipStreamSynchronize(stream)); // the host vectors leave scope; pageable copies are staged anyway
    return FTTE_OK;
}

// One prepared pass on `stream`: depth after depth (one launch per depth for the whole batch), then the rays that leave the region
// (hybrid), then the per-leaf means into J in list order.  The combine launches read-modify-write J: `before_combine` (if any) is
// waited for in front of the first one, `after_combine` (if any) recorded behind the last, which is how two passes on two streams
// keep a fixed order of additions.
int launch_forests(ftte_ctx *c, hipStream_t stream, const ForestRun &R, AmrLevelRec A, double *J_dev, bool zero_first, bool time_batches,
                   hipEvent_t before_combine, hipEvent_t after_combine)
{
    const int nnu = c->nnu;
    for (size_t b = 0; b < R.batches.size(); ++b) {
        const ForestRun::Batch &B = R.batches[b];
        A.dir = c->d_amr_dirs + R.dir_at + (size_t)B.d0;
        A.ndir = B.nb;
        if (time_batches) {
            c->timing[b].updates = (int64_t)B.nb * c->ncell * nnu; c->timing[b].lanes = 0;
            FTTE_HIP(c, hipEventRecord(c->timing[b].start, stream));
        }
        for (size_t depth = 0; depth < B.maxdepth; ++depth) {
            A.count = c->d_amr_tables + B.table_at + depth * 2 * (size_t)B.nb;
            A.begin = A.count + B.nb;
            A.most = R.most_of[B.most_at + depth];
            if (launch_amr_level(A, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest level kernel launch failed");
        }
        if (launch_amr_export(A, B.most_exports, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest export kernel launch failed");
        if (b == 0 && before_combine) FTTE_HIP(c, hipStreamWaitEvent(stream, before_combine, 0));
        if (launch_amr_combine(A, J_dev, zero_first && b == 0, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest combine kernel launch failed");
        if (time_batches) {
            FTTE_HIP(c, hipEventRecord(c->timing[b].stop, stream));
            c->timing_used = (int)b + 1;
        }
    }
    if (after_combine) FTTE_HIP(c, hipEventRecord(after_combine, stream));
    return FTTE_OK;
}

// The forests of `dirs`, `batch` directions at a time (A.dir / A.count / A.begin are filled here).
int run_forests(ftte_ctx *c, hipStream_t stream, const std::vector<ForestDirHost> &dirs, int batch, size_t per_dir, AmrLevelRec A,
                double *J_dev, bool zero_first, bool time_batches)
{
    std::vector<ForestRun> runs;
    int rc;
    if ((rc = prepare_forests(c, stream, {dirs}, {0}, batch, per_dir, &runs))) return rc;
    return launch_forests(c, stream, runs[0], A, J_dev, zero_first, time_batches, nullptr, nullptr);
}

// The sweep on a refined cell array: per-direction segment forests (ftte_amr.h), processed depth by depth.
int forest_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb,
                 double *J_dev, hipStream_t stream)
{
    const int nnu = c->nnu;
    const int64_t ncell = c->ncell, nseg = 3 * ncell;
    int rc;
    if ((rc = wait_sweep(c))) return rc;

    // ---- plan: fold, link, order; cached while the direction list, the tree and the box stay the same
    std::vector<double> key;
    key.reserve(3 * (size_t)ndir + 1);
    key.push_back(c->box);
    key.insert(key.end(), phi, phi + ndir);
    key.insert(key.end(), theta, theta + ndir);
    key.insert(key.end(), w, w + ndir);
    if (key != c->forest_key || (int)c->forests.size() != ndir) {
        FTTE_HIP(c, hipStreamSynchronize(stream));
        free_forests(c);
        ++c->n_forest_builds;
        std::vector<double> fphi(ndir), ftheta(ndir);
        std::vector<int> fzone(ndir);
        for (int d = 0; d < ndir; ++d) {
            const int frc = fold_direction(phi[d], theta[d], &fphi[d], &ftheta[d], &fzone[d]);
            if (frc) {
                char buf[160];
                std::snprintf(buf, sizeof buf, "direction %d (phi=%.17g, theta=%.17g) cannot be folded", d, phi[d], theta[d]);
                return fail(c, fold_status(frc), buf);
            }
        }
        c->forests.resize(ndir);
        // link on the host, a few directions at a time on separate threads, upload, drop the host copy
        const int nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        for (int d0 = 0; d0 < ndir; d0 += nthreads) {
            const int nb = std::min(nthreads, ndir - d0);
            std::vector<AmrForest> F(nb);
            std::vector<std::vector<SegRec>> rec(nb);
            std::vector<std::vector<uint8_t>> active(nb);
            std::vector<int> st(nb, 0);
            std::vector<std::string> msg(nb);
            std::vector<std::thread> pool;
            for (int t = 0; t < nb; ++t)
                pool.emplace_back([&, t] {
                    st[t] = build_forest(c->tree, fphi[d0 + t], ftheta[d0 + t], fzone[d0 + t], c->box, &F[t], &msg[t]);
                    if (st[t]) return;
                    // pack what the device reads per segment into one record, in processing order
                    const AmrForest &f = F[t];
                    const size_t nact = f.order.size();
                    rec[t].resize(std::max<size_t>(nact, 1));
                    for (size_t q = 0; q < nact; ++q) {
                        const int32_t sg = f.order[q];
                        rec[t][q].seg = sg; rec[t][q].up = f.up[sg]; rec[t][q].up2 = f.up2[sg]; rec[t][q].at = 0;
                        rec[t][q].dpath = f.dpath[sg];
                    }
                    active[t].resize((size_t)ncell);
                    for (int64_t q = 0; q < ncell; ++q)
                        active[t][q] = (uint8_t)((f.up[3 * q + 1] != AmrForest::kInactive ? 1 : 0) | (f.up[3 * q + 2] != AmrForest::kInactive ? 2 : 0));
                });
            for (auto &th : pool) th.join();
            for (int t = 0; t < nb; ++t) {
                if (st[t]) { free_forests(c); return fail(c, st[t], "direction " + std::to_string(d0 + t) + ": " + msg[t]); }
                ftte_ctx::ForestDev &D = c->forests[d0 + t];
                D.w = w[d0 + t];
                D.depth_off = F[t].depth_off;
                FTTE_HIP(c, hipMalloc((void **)&D.rec, sizeof(SegRec) * rec[t].size()));
                FTTE_HIP(c, hipMalloc((void **)&D.active, (size_t)ncell));
                FTTE_HIP(c, hipMemcpy(D.rec, rec[t].data(), sizeof(SegRec) * rec[t].size(), hipMemcpyHostToDevice));
                FTTE_HIP(c, hipMemcpy(D.active, active[t].data(), (size_t)ncell, hipMemcpyHostToDevice));
            }
        }
        c->forest_key = key;
    }

    // Scratch: outgoing intensity and mean of every segment of every direction of a batch.  The batch is as large as the
    // direction list, kAmrBatch and the free memory allow (two arrays of 3 ncell nnu doubles per direction: 38 GB for 48
    // directions of a 128^3 x 8 tree), and shrinks once more if the allocation still fails.
    const size_t per_dir = (size_t)nseg * nnu;
    int batch = std::max(1, std::min(ndir, kAmrBatch));
    if (c->amr_scratch_cap < per_dir * (size_t)batch) {
        FTTE_HIP(c, hipStreamSynchronize(stream));
        if (c->amr_Iout) { FTTE_HIP(c, hipFree(c->amr_Iout)); c->amr_Iout = nullptr; }
        if (c->amr_mean) { FTTE_HIP(c, hipFree(c->amr_mean)); c->amr_mean = nullptr; }
        c->amr_scratch_cap = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t fits = (size_t)(0.9 * (double)free_b) / (2 * sizeof(double) * per_dir);
            batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, fits));
        }
        for (;;) {
            hipError_t e1 = hipMalloc((void **)&c->amr_Iout, sizeof(double) * per_dir * (size_t)batch);
            hipError_t e2 = e1 == hipSuccess ? hipMalloc((void **)&c->amr_mean, sizeof(double) * per_dir * (size_t)batch) : e1;
            if (e1 == hipSuccess && e2 == hipSuccess) break;
            if (c->amr_Iout) { (void)hipFree(c->amr_Iout); c->amr_Iout = nullptr; }
            c->amr_mean = nullptr;
            (void)hipGetLastError();
            if (batch == 1) return fail(c, FTTE_ERR_MEMORY, "refined-grid sweep: not enough device memory for the segment scratch of one direction");
            batch = (batch + 1) / 2;
        }
        c->amr_scratch_cap = per_dir * (size_t)batch;
    } else batch = (int)std::min<size_t>((size_t)kAmrBatch, c->amr_scratch_cap / per_dir);
    FTTE_HIP(c, hipStreamSynchronize(stream)); // d_uvb below may still be read by the previous sweep
    if ((rc = ensure(c, &c->d_uvb, &c->d_uvb_cap, (size_t)nnu))) return rc;
    FTTE_HIP(c, hipMemcpy(c->d_uvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice)); c->uvb_sent.clear();

    // the forest path gathers by cell: all groups of a cell side by side (beyond 96 groups the transposing kernel's
    // tile no longer fits the LDS of a workgroup; the strided layout is read as it is)
    const bool cell_major = nnu <= 96;
    if (cell_major) {
    if ((rc = ensure(c, &c->amr_kappa, &c->amr_kappa_cap, (size_t)nnu * ncell))) return rc;
    if (!c->kappa_ready[3] || c->amr_kappa_form != 0) {
        if (launch_cell_major(c->kappa[0], c->amr_kappa, ncell, nnu, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        c->kappa_ready[3] = true; c->amr_kappa_form = 0;
    }
    if (c->emit_mode) {
        if ((rc = ensure(c, &c->amr_emis, &c->amr_emis_cap, (size_t)nnu * ncell))) return rc;
        if (!c->emis_ready[3]) {
            if (launch_cell_major(c->emis[0], c->amr_emis, ncell, nnu, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            c->emis_ready[3] = true;
        }
    }
    }

    const int nbatch = (ndir + batch - 1) / batch;
    while ((int)c->timing.size() < nbatch) {
        LaunchTiming t;
        FTTE_HIP(c, hipEventCreate(&t.start));
        FTTE_HIP(c, hipEventCreate(&t.stop));
        c->timing.push_back(t);
    }
    c->timing_used = 0;
    if (ndir == 0) FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * ncell, stream));

    static const ftte_consts kMath = FTTE_CONSTS_INIT;
    {
        AmrLevelRec A;
        std::memset(&A, 0, sizeof A);
        A.kappa = cell_major ? c->amr_kappa : c->kappa[0];
        A.emis = !c->emit_mode ? nullptr : cell_major ? c->amr_emis : c->emis[0];
        A.group_stride = cell_major ? 1 : ncell;
        A.cell_stride = cell_major ? nnu : 1;
        A.emit = c->emit_mode;
        A.uvb = c->d_uvb;
        A.ncell = ncell;
        A.nnu = nnu;
        A.math = kMath;
        std::vector<ForestDirHost> dirs((size_t)ndir);
        for (int d = 0; d < ndir; ++d) {
            const ftte_ctx::ForestDev &D = c->forests[(size_t)d];
            dirs[(size_t)d] = ForestDirHost{D.rec, D.active, D.w, nullptr, nullptr, 0, &D.depth_off};
        }
        if ((rc = run_forests(c, stream, dirs, batch, per_dir, A, J_dev, true, true))) return rc;
    }
    return mark_sweep(c, stream);
}


// The sweep of a uniform grid by cell-fixed bricks (ftte_brick.hip): one launch per stage, then one merge of the groups'
// accumulators (layout after layout, group after group: a fixed order) into J.
// Host arrays handed over with the sweep (ftte_diffuse_iteration): the opacities go up and J comes back one lane of frequency
// groups at a time, on the lane's own stream, so that the first lane is swept while the second one's opacities are still on the
// PCIe link and its J travels back while the second is swept.
struct HostPipe { const double *kappa; double *J; };
int upload_on(ftte_ctx *c, hipStream_t q, void *dst_dev, const void *src_host, size_t bytes);
int download_on(ftte_ctx *c, hipStream_t q, void *dst_host, const void *src_dev, size_t bytes);
bool is_registered(const ftte_ctx *c, const void *p, size_t bytes);

int brick_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
                hipStream_t stream, const HostPipe *pipe = nullptr)
{
    int rc;
    if ((rc = build_brick_plan(c, ndir, phi, theta, w))) return rc;
    BrickPlan &P = c->bplan;
    const int n = c->n, nnu = c->nnu;
    const size_t per_acc = (size_t)nnu * c->ncell;

    // everything below overwrites device tables the previous sweep may still be reading
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(stream));
    if (stream != c->stream) FTTE_HIP(c, hipStreamSynchronize(c->stream));

    if (c->acc_cap < per_acc) {
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < kMaxAcc; ++s)
                if (c->acc[l][s]) { FTTE_HIP(c, hipFree(c->acc[l][s])); c->acc[l][s] = nullptr; }
        c->acc_cap = per_acc;
    }
    const size_t face_need = (size_t)ndir * nnu * (size_t)P.face_elems;
    if ((rc = ensure(c, &c->d_faces, &c->d_faces_cap, face_need))) return rc;
    if (!c->merge_stream) {
        FTTE_HIP(c, hipStreamCreateWithFlags(&c->merge_stream, hipStreamNonBlocking));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_layout_done, hipEventDisableTiming));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_merge_done, hipEventDisableTiming));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_layouts_ready, hipEventDisableTiming));
    }
    // Host arrays (ftte_diffuse_iteration): lanes of frequency groups (below) do their own layouts before their first stage and their
    // own merge after their last.
    // (With device-resident opacities, layouts up front and one merge at the end are faster: 37.4-38.0 against 38.6 ms per
    // 256^3 x 8 x 96 step.  With host arrays in flight the lanes are staggered by the transfers and their ends fall into each
    // other's sweeps anyway.)
    const bool lane_ends = pipe != nullptr;
    bool lane_layout[3] = {false, false, false};
    // accumulators and the opacity in the layouts the groups march through
    bool transposed = false;
    for (int l = 0; l < 3; ++l) {
        for (int s = 0; s < P.nacc[l]; ++s)
            if (!c->acc[l][s]) FTTE_HIP(c, hipMalloc((void **)&c->acc[l][s], sizeof(double) * c->acc_cap));
        if (P.nacc[l] && !c->kappa_ready[l]) {
            if (!c->kappa[l]) FTTE_HIP(c, hipMalloc((void **)&c->kappa[l], sizeof(double) * c->kappa_cap));
            if (!lane_ends) {
                if (launch_to_layout(l, c->kappa[0], c->kappa[l], n, nnu, (long)c->ncell, stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
                c->kappa_ready[l] = true;
            } else lane_layout[l] = true;
            transposed = true;
        }
        if (P.nacc[l] && c->emit_mode && !c->emis_ready[l]) {
            if (!c->emis[l]) FTTE_HIP(c, hipMalloc((void **)&c->emis[l], sizeof(double) * c->kappa_cap));
            if (launch_to_layout(l, c->emis[0], c->emis[l], n, nnu, (long)c->ncell, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            c->emis_ready[l] = true;
        }
    }
    (void)transposed;

    if (!c->bplan_uploaded) {
        c->bgroups_sent.clear();
        if ((rc = ensure(c, &c->d_blayers, &c->d_blayers_cap, P.layers.size()))) return rc;
        if ((rc = ensure(c, &c->d_btasks, &c->d_btasks_cap, P.tasks.size()))) return rc;
        if ((rc = ensure(c, &c->d_bgroups, &c->d_bgroups_cap, P.groups.size()))) return rc;
        if (!P.layers.empty())
            FTTE_HIP(c, hipMemcpy(c->d_blayers, P.layers.data(), sizeof(LayerRec) * P.layers.size(), hipMemcpyHostToDevice));
        if (!P.tasks.empty())
            FTTE_HIP(c, hipMemcpy(c->d_btasks, P.tasks.data(), sizeof(BrickTask) * P.tasks.size(), hipMemcpyHostToDevice));
        if (P.dataflow && !P.deps.empty()) {
            if ((rc = ensure(c, &c->d_bdeps, &c->d_bdeps_cap, P.deps.size()))) return rc;
            FTTE_HIP(c, hipMemcpy(c->d_bdeps, P.deps.data(), sizeof(int32_t) * P.deps.size(), hipMemcpyHostToDevice));
        }
        c->bplan_uploaded = true;
    }
    // the group records carry pointers that depend on nnu (face blocks) and on the buffers: rebuilt per sweep (a few KB)
    {
        std::vector<BrickGroup> G(P.groups.size());
        std::memset(G.data(), 0, sizeof(BrickGroup) * G.size());
        for (size_t g = 0; g < P.groups.size(); ++g) {
            const BrickPlan::Group &H = P.groups[g];
            const DirPlan &D0 = P.dirs[H.dirs[0]];
            G[g].kappa = c->kappa[H.layout];
            G[g].emis = c->emit_mode ? c->emis[H.layout] : nullptr;
            G[g].J = c->acc[H.layout][H.acc];
            G[g].org = D0.org; G[g].si = D0.si; G[g].sv = D0.sv; G[g].su = D0.su;
            G[g].ndir = (int)H.dirs.size();
            for (size_t q = 0; q < H.dirs.size(); ++q) {
                const int d = H.dirs[q];
                G[g].dir[q].layers = c->d_blayers + P.dirs[d].layer_off;
                G[g].dir[q].faces = c->d_faces + (size_t)d * nnu * (size_t)P.face_elems;
                G[g].dir[q].w = P.dirs[d].w;
            }
        }
        // (a blocking copy each: skipped when the device already holds exactly these bytes, which is every iteration after the first)
        const size_t bytes = sizeof(BrickGroup) * G.size();
        if (bytes && (c->bgroups_sent.size() != bytes || std::memcmp(c->bgroups_sent.data(), G.data(), bytes) != 0)) {
            FTTE_HIP(c, hipMemcpy(c->d_bgroups, G.data(), bytes, hipMemcpyHostToDevice));
            c->bgroups_sent.assign((const char *)G.data(), (const char *)G.data() + bytes);
        }
    }
    if ((rc = ensure(c, &c->d_uvb, &c->d_uvb_cap, (size_t)nnu))) return rc;
    if (c->uvb_sent.size() != (size_t)nnu || std::memcmp(c->uvb_sent.data(), uvb, sizeof(double) * nnu) != 0) {
        FTTE_HIP(c, hipMemcpy(c->d_uvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice));
        c->uvb_sent.assign(uvb, uvb + nnu);
    }

    // The frequency groups never touch each other's data (own slices of the accumulators and of the face rings), and a
    // stage is a launch that drains before the next one starts: the stage sequence is therefore issued once per "lane"
    // (a subset of the frequency groups) on streams of their own, so that the tail of one lane's stage overlaps the next
    // stage of another.  Lane 0 is the caller's stream.  One pair of events brackets the whole phase: with kernels of
    // several streams in flight together the time of a single launch says little.
    const size_t nstages = (size_t)P.nstages, per_lane = nstages + 1;
    const int nulanes = P.glanes > 1 ? 1 : std::max(1, std::min(c->lanes, nnu)); // streams over frequency groups ...
    const int nlanes = nulanes * P.glanes;                                        // ... or over the groups of directions
    while ((int)c->lane_stream.size() < nlanes - 1) {
        hipStream_t q; hipEvent_t e;
        FTTE_HIP(c, hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
        FTTE_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->lane_stream.push_back(q); c->lane_done.push_back(e);
    }
    if (!c->ev_fork) FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    while (c->timing.size() < 1) {
        LaunchTiming t;
        FTTE_HIP(c, hipEventCreate(&t.start));
        FTTE_HIP(c, hipEventCreate(&t.stop));
        c->timing.push_back(t);
    }
    c->timing_used = 0;

    static const ftte_consts kMath = FTTE_CONSTS_INIT;
    if (!P.groups.empty()) {
        LaunchTiming &T = c->timing[0];
        T.updates = P.updates * nnu;
        T.lanes = 0;
        if (lane_ends) {
            while ((int)T.first.size() < nlanes) {
                hipEvent_t a, b;
                FTTE_HIP(c, hipEventCreate(&a));
                FTTE_HIP(c, hipEventCreate(&b));
                T.first.push_back(a); T.last.push_back(b);
            }
        }
        FTTE_HIP(c, hipEventRecord(T.start, stream));
        if (P.dataflow) {
            // every brick of the sweep in one launch; flags of `epoch` mark the finished ones (the array is zeroed when it is
            // (re)allocated and when the epoch wraps, never in between)
            const size_t nflags = P.tasks.size() * (size_t)nnu;
            if (c->d_bdone_cap < nflags || c->bepoch == 0xffffffffu) {
                if ((rc = ensure(c, &c->d_bdone, &c->d_bdone_cap, nflags))) return rc;
                FTTE_HIP(c, hipMemsetAsync(c->d_bdone, 0, sizeof(uint32_t) * c->d_bdone_cap, stream));
                c->bepoch = 0;
            }
            if (!c->d_bsync) {
                FTTE_HIP(c, hipMalloc((void **)&c->d_bsync, sizeof(uint32_t) * 2));
                FTTE_HIP(c, hipHostMalloc((void **)&c->h_berror, sizeof(uint32_t), hipHostMallocDefault));
                *c->h_berror = 0;
            }
            FTTE_HIP(c, hipMemsetAsync(c->d_bsync, 0, sizeof(uint32_t) * 2, stream));
            BrickLaunch L;
            std::memset(&L, 0, sizeof L);
            L.groups = c->d_bgroups;
            L.tasks = c->d_btasks;
            L.uvb = c->d_uvb;
            L.group_stride = c->ncell;
            L.face_stride = P.face_elems;
            L.vface_off = P.vface_off; L.iface_off = P.iface_off;
            L.n = n; L.ntasks = (int)P.tasks.size(); L.nnu = nnu; L.nu0 = 0; L.chunk = P.chunk;
            L.up = P.up; L.vp = P.vp; L.uw = P.uw; L.ut = P.ut; L.nslot = P.nslot;
            L.emit = c->emit_mode;
            L.ticket = c->d_bsync; L.error = c->d_bsync + 1; L.done = c->d_bdone; L.deps = c->d_bdeps; L.epoch = ++c->bepoch; L.pad_ = c->dataflow == 2 ? 1 : 0;
            L.math = kMath;
            const int lrc = launch_brick(L, P.max_dirs, c->brick_waves, stream);
            if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
            FTTE_HIP(c, hipMemcpyAsync(c->h_berror, c->d_bsync + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        }
        FTTE_HIP(c, hipEventRecord(c->ev_fork, stream));
        for (int lane = 0; lane < nlanes && !P.dataflow; ++lane) {
            hipStream_t q = lane == 0 ? stream : c->lane_stream[(size_t)lane - 1];
            if (lane) FTTE_HIP(c, hipStreamWaitEvent(q, c->ev_fork, 0));
            const int gl = P.glanes > 1 ? lane : 0, nl = P.glanes > 1 ? 0 : lane;
            const int nu0 = (int)((int64_t)nnu * nl / nulanes), nu1 = (int)((int64_t)nnu * (nl + 1) / nulanes);
            const size_t *off = &P.stage_off[(size_t)gl * per_lane];
            const size_t slice0 = (size_t)nu0 * c->ncell, slice_bytes = sizeof(double) * (size_t)(nu1 - nu0) * c->ncell;
            if (pipe) {
                // this lane's opacities: after the lane before (one transfer at a time has the link to itself), then its layouts
                while (c->pipe_up.size() < (size_t)nlanes) {
                    hipEvent_t e;
                    FTTE_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    c->pipe_up.push_back(e);
                }
                if (lane) FTTE_HIP(c, hipStreamWaitEvent(q, c->pipe_up[(size_t)lane - 1], 0));
                if ((rc = upload_on(c, q, c->kappa[0] + slice0, pipe->kappa + slice0, slice_bytes))) return rc;
                FTTE_HIP(c, hipEventRecord(c->pipe_up[(size_t)lane], q));
            }
            if (lane_ends) {
                for (int l = 1; l < 3; ++l)
                    if (lane_layout[l] && launch_to_layout(l, c->kappa[0] + slice0, c->kappa[l] + slice0, n, nu1 - nu0, (long)c->ncell, q))
                        return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
                FTTE_HIP(c, hipEventRecord(T.first[(size_t)lane], q));
            }
            for (size_t st = 0; st < nstages; ++st) {
                if (off[st + 1] == off[st]) continue;
                BrickLaunch L;
                std::memset(&L, 0, sizeof L);
                L.groups = c->d_bgroups;
                L.tasks = c->d_btasks + off[st];
                L.uvb = c->d_uvb;
                L.group_stride = c->ncell;
                L.face_stride = P.face_elems;
                L.vface_off = P.vface_off; L.iface_off = P.iface_off;
                L.n = n; L.ntasks = (int)(off[st + 1] - off[st]); L.nnu = nu1 - nu0; L.nu0 = nu0; L.chunk = P.chunk;
                L.up = P.up; L.vp = P.vp; L.uw = P.uw; L.ut = P.ut; L.nslot = P.nslot;
                L.emit = c->emit_mode;
                L.math = kMath;
                const int lrc = (c->team && !c->emit_mode) ? launch_brick_team(L, P.max_dirs, c->brick_waves, q) : launch_brick(L, P.max_dirs, c->brick_waves, q);
                if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
            }
            if (lane_ends) { // this lane's J: merged as soon as its stages are done, and on its way back (pinned arrays) behind that
                FTTE_HIP(c, hipEventRecord(T.last[(size_t)lane], q));
                const double *accs[3 * kMaxAcc];
                int layouts[3 * kMaxAcc], count = 0;
                for (int l = 0; l < 3; ++l)
                    for (int s2 = 0; s2 < P.nacc[l]; ++s2) { accs[count] = c->acc[l][s2] + slice0; layouts[count++] = l; }
                if (launch_merge(accs, layouts, count, J_dev + slice0, n, nu1 - nu0, (long)c->ncell, false, q))
                    return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
                if (pipe && is_registered(c, pipe->J + slice0, slice_bytes))
                    FTTE_HIP(c, hipMemcpyAsync(pipe->J + slice0, J_dev + slice0, slice_bytes, hipMemcpyDeviceToHost, q));
            }
            if (lane) {
                FTTE_HIP(c, hipEventRecord(c->lane_done[(size_t)lane - 1], q));
                FTTE_HIP(c, hipStreamWaitEvent(stream, c->lane_done[(size_t)lane - 1], 0));
            }
        }
        if (pipe) { // pageable J: through the staging blocks, lane after lane (the later lanes are still being swept)
            for (int lane = 0; lane < nlanes; ++lane) {
                const int nu0 = (int)((int64_t)nnu * lane / nulanes), nu1 = (int)((int64_t)nnu * (lane + 1) / nulanes);
                const size_t slice0 = (size_t)nu0 * c->ncell, slice_bytes = sizeof(double) * (size_t)(nu1 - nu0) * c->ncell;
                if (is_registered(c, pipe->J + slice0, slice_bytes)) continue;
                hipStream_t q = lane == 0 ? stream : c->lane_stream[(size_t)lane - 1];
                if ((rc = download_on(c, q, pipe->J + slice0, J_dev + slice0, slice_bytes))) return rc;
            }
            c->kappa_ready[0] = true; // every lane has brought its groups
        }
        if (lane_ends) {
            for (int l = 1; l < 3; ++l) if (lane_layout[l]) c->kappa_ready[l] = true; // ... and transposed them
            T.lanes = nlanes;
        }
        FTTE_HIP(c, hipEventRecord(T.stop, stream));
        c->timing_used = 1;
    }
    // J = the groups' accumulators, layout after layout
    if (!lane_ends || P.groups.empty()) {
        const double *accs[3 * kMaxAcc];
        int layouts[3 * kMaxAcc], count = 0;
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < P.nacc[l]; ++s) { accs[count] = c->acc[l][s]; layouts[count++] = l; }
        if (count) {
            if (launch_merge(accs, layouts, count, J_dev, n, nnu, (long)c->ncell, false, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
        } else FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * c->ncell, stream)); // no directions
    }
    return mark_sweep(c, stream);
}

// ---- hybrid sweep of a refined cell array -----------------------------------------------------------------------------
// The reference recurses into refined cells wherever they are (transport, transportRoutinesModule.f90:577-586) and walks the
// tree for every upstream link of every cell of every direction.  Most of a cell array is plain base cells; here those are
// swept by the brick kernel, and only a box around the refined cells -- widened by one brick, so that its surface separates
// unrefined base cells, across which a ray is handed over exactly as between two bricks -- by the segment forest.  Per group of
// directions: the bricks that do not lie behind the box, then the forest (rays entering it read from the bricks' face
// buffers, rays leaving it written there), then the bricks behind it.  J of a cell = what the bricks stored for the
// directions in whose box it does not lie + what the forest adds for the others.

void free_hybrid(ftte_ctx *c)
{
    for (auto &d : c->hplan.dirs) {
        if (d.rec) (void)hipFree(d.rec);
        if (d.active) (void)hipFree(d.active);
        if (d.exports) (void)hipFree(d.exports);
    }
    if (c->hplan.cells) (void)hipFree(c->hplan.cells);
    c->hplan = ftte_ctx::HybridPlan();
}

// the box of izone `izone`, sweep frame, tile-aligned and widened by a brick; false if the tree has no refined cell
bool hybrid_region(const ftte_ctx *c, const BrickPlan &P, int izone, ForestRegion *R, int tile_lo[3], int tile_hi[3])
{
    const AmrTree &T = c->tree;
    const int n = T.n;
    int clo[3] = {n + 1, n + 1, n + 1}, chi[3] = {0, 0, 0}; // storage coordinates of the refined base cells
    for (int64_t b = 0; b < (int64_t)n * n * n; ++b)
        if (T.child0[(size_t)b] >= 0) {
            const int cc[3] = {(int)(b / ((int64_t)n * n)) + 1, (int)((b / n) % n) + 1, (int)(b % n) + 1};
            for (int a = 0; a < 3; ++a) { clo[a] = std::min(clo[a], cc[a]); chi[a] = std::max(chi[a], cc[a]); }
        }
    if (chi[0] == 0) return false;
    ZoneMap zm;
    zone_map(izone, &zm);
    int slo[3], shi[3]; // sweep frame: i, j, k
    int march_c = 0;
    for (int a = 0; a < 3; ++a) {
        const int sa = zm.src[a];
        slo[sa] = zm.mirror[a] ? n + 1 - chi[a] : clo[a];
        shi[sa] = zm.mirror[a] ? n + 1 - clo[a] : chi[a];
        if (sa == 0) march_c = a;
    }
    const int fast_c = (march_c == 2) ? 1 : 2;
    const bool u_is_k = zm.src[fast_c] == 2;
    const int ju = u_is_k ? 2 : 1, jv = u_is_k ? 1 : 2; // sweep axes of u and v
    const int size[3] = {P.chunk, 0, 0};
    (void)size;
    const int tsize_i = P.chunk, tsize_u = 64, tsize_v = kBrickRows;
    tile_lo[0] = std::max(0, (slo[ju] - 1) / tsize_u - 1); tile_hi[0] = std::min(P.ntu - 1, (shi[ju] - 1) / tsize_u + 1);
    tile_lo[1] = std::max(0, (slo[jv] - 1) / tsize_v - 1); tile_hi[1] = std::min(P.ntv - 1, (shi[jv] - 1) / tsize_v + 1);
    tile_lo[2] = std::max(0, (slo[0] - 1) / tsize_i - 1);  tile_hi[2] = std::min(P.nti - 1, (shi[0] - 1) / tsize_i + 1);
    R->u_is_k = u_is_k;
    R->lo[0] = tile_lo[2] * tsize_i + 1; R->hi[0] = std::min(n, (tile_hi[2] + 1) * tsize_i);
    R->lo[ju] = tile_lo[0] * tsize_u + 1; R->hi[ju] = std::min(n, (tile_hi[0] + 1) * tsize_u);
    R->lo[jv] = tile_lo[1] * tsize_v + 1; R->hi[jv] = std::min(n, (tile_hi[1] + 1) * tsize_v);
    R->chunk = P.chunk; R->ut = P.ut; R->nslot = P.nslot; R->ntv = P.ntv; R->up = P.up; R->vp = P.vp;
    R->vface_off = P.vface_off; R->iface_off = P.iface_off;
    return true;
}

int build_hybrid_plan(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w)
{
    ftte_ctx::HybridPlan &H = c->hplan;
    const int n = c->n, nnu = c->nnu;
    // short bricks: the box is widened by one brick on every side, and what lies inside it costs several times a brick's bytes
    const int chunk = std::min(c->chunk > 0 ? c->chunk : 4, n);
    const int gmax = c->group > 0 ? c->group : (nnu >= 2 ? 3 : 2);
    std::vector<double> key = {c->box, (double)chunk, (double)gmax, (double)c->share, (double)c->halves};
    key.insert(key.end(), phi, phi + ndir);
    key.insert(key.end(), theta, theta + ndir);
    key.insert(key.end(), w, w + ndir);
    if (H.valid && H.key == key) return FTTE_OK;
    free_hybrid(c);
    int rc;
    BrickPlan &P = H.bricks;
    if ((rc = plan_brick_groups(c, P, ndir, phi, theta, w, chunk, gmax, 0, true))) return rc;
    P.glanes = 1;

    // the box of every group; is the part outside the boxes worth a brick sweep?
    struct Box { ForestRegion R; int lo[3], hi[3]; bool any; };
    std::vector<Box> box(P.groups.size());
    int64_t inside_bricks = 0, all_bricks = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        box[g].any = hybrid_region(c, P, P.groups[g].izone, &box[g].R, box[g].lo, box[g].hi);
        all_bricks += (int64_t)P.ntu * P.ntv * P.nti;
        if (box[g].any) inside_bricks += (int64_t)(box[g].hi[0] - box[g].lo[0] + 1) * (box[g].hi[1] - box[g].lo[1] + 1) * (box[g].hi[2] - box[g].lo[2] + 1);
    }
    H.key = key;
    H.valid = true;
    H.worthwhile = !P.groups.empty() && inside_bricks * 2 <= all_bricks; // else: the forest path for the whole tree
    if (!H.worthwhile) return FTTE_OK;

    // Halves: the forests stream records at the memory system's rate while the brick stages of a 128^3 grid are short launches
    // that leave most of it idle, so the sweep runs as two pipelines (bricks - forests - bricks each) on two streams.  What the
    // groups of one accumulator write is ordered by their launches, so an accumulator's groups stay together; halves are
    // balanced by direction count.
    std::vector<int> half_of_group(P.groups.size(), 0);
    H.nhalves = 1;
    if (c->halves > 1 && P.nacc[0] + P.nacc[1] + P.nacc[2] >= 2) {
        H.nhalves = std::min(c->halves, P.nacc[0] + P.nacc[1] + P.nacc[2]);
        std::vector<int> weight(3 * (size_t)kMaxAcc, 0), order;
        for (const auto &G : P.groups) weight[(size_t)G.layout * kMaxAcc + G.acc] += (int)G.dirs.size();
        for (int a = 0; a < 3 * kMaxAcc; ++a) if (weight[(size_t)a]) order.push_back(a);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return weight[(size_t)x] > weight[(size_t)y]; });
        std::vector<int> half_of_acc(3 * (size_t)kMaxAcc, 0);
        int load[ftte_ctx::kMaxPipes] = {0, 0, 0, 0};
        for (int a : order) {
            int h = 0;
            for (int q = 1; q < H.nhalves; ++q) if (load[q] < load[h]) h = q;
            half_of_acc[(size_t)a] = h; load[h] += weight[(size_t)a];
        }
        for (size_t g = 0; g < P.groups.size(); ++g) half_of_group[g] = half_of_acc[(size_t)P.groups[g].layout * kMaxAcc + P.groups[g].acc];
    }
    H.half_dirs.assign((size_t)H.nhalves, std::vector<int>());
    {
        std::vector<int> half_of_dir((size_t)ndir, 0);
        for (size_t g = 0; g < P.groups.size(); ++g) for (int d : P.groups[g].dirs) half_of_dir[(size_t)d] = half_of_group[g];
        for (int d = 0; d < ndir; ++d) H.half_dirs[(size_t)half_of_dir[(size_t)d]].push_back(d);
    }

    // tasks: the bricks outside the boxes.  Phase 1: those that do not lie behind their group's box (no tile index at or beyond
    // the box's first one in all three directions); phase 3: the others.  Within a phase stage by stage as in a plain sweep.
    int max_offset = 0;
    for (const auto &G : P.groups) max_offset = std::max(max_offset, G.offset);
    const int per_phase = P.ntu + P.ntv + P.nti - 2 + max_offset;
    H.nlist = 2 * (size_t)per_phase;
    const size_t nlist = (size_t)H.nhalves * H.nlist;
    H.phase1_stages = (size_t)per_phase;
    auto list_of = [&](size_t g, const Box &B, int tu, int tv, int ti, int offset) {
        const bool behind = B.any && tu >= B.lo[0] && tv >= B.lo[1] && ti >= B.lo[2];
        return (size_t)half_of_group[g] * H.nlist + (size_t)(behind ? per_phase : 0) + (size_t)(tu + tv + ti + offset);
    };
    auto in_box = [&](const Box &B, int tu, int tv, int ti) {
        return B.any && tu >= B.lo[0] && tu <= B.hi[0] && tv >= B.lo[1] && tv <= B.hi[1] && ti >= B.lo[2] && ti <= B.hi[2];
    };
    const size_t nb = (size_t)P.ntu * P.ntv * P.nti;
    std::vector<std::vector<size_t>> first(3 * (size_t)kMaxAcc);
    auto brick_of = [&](const BrickPlan::Group &G, int tu, int tv, int ti) {
        const DirPlan &D0 = P.dirs[G.dirs[0]];
        const int bu = D0.su < 0 ? P.ntu - 1 - tu : tu, bv = D0.sv < 0 ? P.ntv - 1 - tv : tv, bi = D0.si < 0 ? P.nti - 1 - ti : ti;
        return ((size_t)bi * P.ntv + bv) * P.ntu + bu;
    };
    H.stage_off.assign(nlist + 1, 0);
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const BrickPlan::Group &G = P.groups[g];
        std::vector<size_t> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
        if (F.empty()) F.assign(nb, ~(size_t)0);
        for (int ti = 0; ti < P.nti; ++ti)
            for (int tv = 0; tv < P.ntv; ++tv)
                for (int tu = 0; tu < P.ntu; ++tu) {
                    if (in_box(box[g], tu, tv, ti)) continue;
                    const size_t l = list_of(g, box[g], tu, tv, ti, G.offset);
                    ++H.stage_off[l + 1];
                    size_t &f = F[brick_of(G, tu, tv, ti)];
                    f = std::min(f, l);
                }
    }
    for (size_t l = 0; l < nlist; ++l) H.stage_off[l + 1] += H.stage_off[l];
    P.tasks.resize(H.stage_off[nlist]);
    std::vector<size_t> fill(H.stage_off.begin(), H.stage_off.end() - 1);
    H.brick_updates = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const BrickPlan::Group &G = P.groups[g];
        const std::vector<size_t> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
        for (int ti = 0; ti < P.nti; ++ti)
            for (int tv = 0; tv < P.ntv; ++tv)
                for (int tu = 0; tu < P.ntu; ++tu) {
                    if (in_box(box[g], tu, tv, ti)) continue;
                    const size_t l = list_of(g, box[g], tu, tv, ti, G.offset);
                    BrickTask T;
                    T.group = (int16_t)g; T.tu = (int16_t)tu; T.tv = (int16_t)tv;
                    T.ti = (int16_t)(ti | (l > F[brick_of(G, tu, tv, ti)] ? kBrickAccumulate : 0));
                    P.tasks[fill[l]++] = T;
                    const int64_t cu = std::min(64, n - 64 * tu), cv = std::min(kBrickRows, n - kBrickRows * tv), ci = std::min(chunk, n - chunk * ti);
                    H.brick_updates += cu * cv * ci * (int64_t)G.dirs.size();
                }
    }

    // The forests, restricted to the boxes: linked on the host a few directions at a time.  Once the leaves that lie in any box are
    // known they are numbered by their place in that list, and segments (3 * place + piece), activity bytes, opacities and scratch
    // use those numbers: what the forests need of memory follows the boxes, not the tree.
    std::vector<int> group_of((size_t)ndir, -1);
    for (size_t g = 0; g < P.groups.size(); ++g) for (int d : P.groups[g].dirs) group_of[(size_t)d] = (int)g;
    H.dirs.resize((size_t)ndir);
    const int64_t ncell = c->ncell;
    const int nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<uint8_t> in_any((size_t)ncell, 0);
    std::vector<std::vector<SegRec>> rec((size_t)ndir);
    std::vector<std::vector<uint8_t>> active((size_t)ndir);  // per leaf, until the list is known
    std::vector<std::vector<AmrExport>> exports((size_t)ndir);
    ++c->n_forest_builds;
    for (int d0 = 0; d0 < ndir; d0 += nthreads) {
        const int nbt = std::min(nthreads, ndir - d0);
        std::vector<AmrForest> F(nbt);
        std::vector<int> st(nbt, 0);
        std::vector<std::string> msg(nbt);
        std::vector<std::thread> pool;
        for (int t = 0; t < nbt; ++t)
            pool.emplace_back([&, t] {
                const int d = d0 + t;
                const DirPlan &D = P.dirs[(size_t)d];
                const Box &B = box[(size_t)group_of[(size_t)d]];
                st[t] = build_forest(c->tree, D.phi, D.theta, D.izone, c->box, &F[t], &msg[t], &B.R);
                if (st[t]) return;
                const AmrForest &f = F[t];
                const size_t nact = f.order.size();
                rec[(size_t)d].resize(std::max<size_t>(nact, 1));
                for (size_t q = 0; q < nact; ++q) {
                    const int32_t sg = f.order[q];
                    SegRec &R = rec[(size_t)d][q];
                    R.seg = sg; R.up = f.up[sg]; R.up2 = f.up2[sg];
                    R.at = f.up[sg] == AmrForest::kImport ? f.import_at[sg] : 0;
                    R.dpath = f.dpath[sg];
                }
                active[(size_t)d].resize((size_t)ncell);
                for (int64_t q = 0; q < ncell; ++q)
                    active[(size_t)d][(size_t)q] = (uint8_t)((f.up[3 * q + 1] != AmrForest::kInactive ? 1 : 0) | (f.up[3 * q + 2] != AmrForest::kInactive ? 2 : 0) |
                                                             (f.inside[(size_t)q] ? 0 : 4));
            });
        for (auto &th : pool) th.join();
        for (int t = 0; t < nbt; ++t) {
            if (st[t]) { const std::string m = msg[t]; const int code = st[t]; free_hybrid(c); return fail(c, code, "direction " + std::to_string(d0 + t) + ": " + m); }
            ftte_ctx::HybridPlan::Dir &D = H.dirs[(size_t)(d0 + t)];
            D.depth_off = F[t].depth_off;
            D.nexports = (int64_t)F[t].exports.size();
            static_assert(sizeof(AmrForest::Export) == sizeof(AmrExport), "export records: host and device forms must agree");
            exports[(size_t)(d0 + t)].resize(F[t].exports.size());
            if (!F[t].exports.empty()) std::memcpy(exports[(size_t)(d0 + t)].data(), F[t].exports.data(), sizeof(AmrExport) * F[t].exports.size());
            for (int64_t q = 0; q < ncell; ++q) in_any[(size_t)q] |= F[t].inside[(size_t)q];
        }
    }
    std::vector<int32_t> cells, place((size_t)ncell, -1);
    for (int64_t q = 0; q < ncell; ++q)
        if (in_any[(size_t)q]) { place[(size_t)q] = (int32_t)cells.size(); cells.push_back((int32_t)q); }
    H.ncells = (int64_t)cells.size();
    FTTE_HIP(c, hipMalloc((void **)&H.cells, sizeof(int32_t) * std::max<size_t>(cells.size(), 1)));
    if (!cells.empty()) FTTE_HIP(c, hipMemcpy(H.cells, cells.data(), sizeof(int32_t) * cells.size(), hipMemcpyHostToDevice));
    {
        auto renumber = [&](int32_t sg) { return sg < 0 ? sg : 3 * place[(size_t)(sg / 3)] + sg % 3; }; // negative: inflow / import marks
        std::vector<int> bad((size_t)ndir, 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&, t] {
                std::vector<uint8_t> compact(cells.size());
                for (int d = t; d < ndir; d += nthreads) {
                    for (SegRec &R : rec[(size_t)d]) {
                        if (place[(size_t)(R.seg / 3)] < 0 || (R.up >= 0 && place[(size_t)(R.up / 3)] < 0) || (R.up2 >= 0 && place[(size_t)(R.up2 / 3)] < 0)) { bad[(size_t)d] = 1; break; }
                        R.seg = renumber(R.seg); R.up = renumber(R.up); R.up2 = renumber(R.up2);
                    }
                    for (AmrExport &X : exports[(size_t)d]) {
                        if (place[(size_t)(X.seg / 3)] < 0) { bad[(size_t)d] = 1; break; }
                        X.seg = renumber(X.seg);
                    }
                    for (size_t q = 0; q < cells.size(); ++q) compact[q] = active[(size_t)d][(size_t)cells[q]];
                    active[(size_t)d].assign(compact.begin(), compact.end());
                }
            });
        for (auto &th : pool) th.join();
        for (int d = 0; d < ndir; ++d)
            if (bad[(size_t)d]) { free_hybrid(c); return fail(c, FTTE_ERR_STATE, "hybrid plan: a forest segment lies outside every box"); }
    }
    for (int d = 0; d < ndir; ++d) {
        ftte_ctx::HybridPlan::Dir &D = H.dirs[(size_t)d];
        FTTE_HIP(c, hipMalloc((void **)&D.rec, sizeof(SegRec) * rec[(size_t)d].size()));
        FTTE_HIP(c, hipMalloc((void **)&D.active, std::max<size_t>(active[(size_t)d].size(), 1)));
        FTTE_HIP(c, hipMalloc((void **)&D.exports, sizeof(AmrExport) * std::max<size_t>(exports[(size_t)d].size(), 1)));
        FTTE_HIP(c, hipMemcpy(D.rec, rec[(size_t)d].data(), sizeof(SegRec) * rec[(size_t)d].size(), hipMemcpyHostToDevice));
        if (!active[(size_t)d].empty()) FTTE_HIP(c, hipMemcpy(D.active, active[(size_t)d].data(), active[(size_t)d].size(), hipMemcpyHostToDevice));
        if (!exports[(size_t)d].empty())
            FTTE_HIP(c, hipMemcpy(D.exports, exports[(size_t)d].data(), sizeof(AmrExport) * exports[(size_t)d].size(), hipMemcpyHostToDevice));
        std::vector<SegRec>().swap(rec[(size_t)d]);
        std::vector<uint8_t>().swap(active[(size_t)d]);
    }
    c->kappa_ready[3] = false; // the forests' copy of the opacities follows the list
    H.uploaded = false;
    return FTTE_OK;
}

int hybrid_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
                 hipStream_t stream, bool *done)
{
    *done = false;
    int rc;
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(stream));
    if (stream != c->stream) FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = build_hybrid_plan(c, ndir, phi, theta, w))) return rc;
    ftte_ctx::HybridPlan &H = c->hplan;
    if (!H.worthwhile) return FTTE_OK; // the caller takes the forest path for the whole tree
    BrickPlan &P = H.bricks;
    const int n = c->n, nnu = c->nnu;
    const int64_t ncell = c->ncell, nbase = (int64_t)n * n * n;

    // ---- device state that depends on the tree only
    if (!c->d_leaf_of_base) {
        std::vector<int32_t> map((size_t)nbase);
        for (int64_t b = 0; b < nbase; ++b) map[(size_t)b] = c->tree.leaf[(size_t)b];
        FTTE_HIP(c, hipMalloc((void **)&c->d_leaf_of_base, sizeof(int32_t) * (size_t)nbase));
        FTTE_HIP(c, hipMemcpy(c->d_leaf_of_base, map.data(), sizeof(int32_t) * (size_t)nbase, hipMemcpyHostToDevice));
    }
    const size_t per_base = (size_t)nnu * (size_t)nbase;
    if (c->base_kappa_cap < per_base) {
        for (int l = 0; l < 3; ++l) if (c->base_kappa[l]) { FTTE_HIP(c, hipFree(c->base_kappa[l])); c->base_kappa[l] = nullptr; }
        for (int l = 0; l < 3; ++l) FTTE_HIP(c, hipMalloc((void **)&c->base_kappa[l], sizeof(double) * per_base));
        c->base_kappa_cap = per_base;
    }
    if (c->acc_cap < (size_t)nnu * (size_t)ncell) {
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < kMaxAcc; ++s)
                if (c->acc[l][s]) { FTTE_HIP(c, hipFree(c->acc[l][s])); c->acc[l][s] = nullptr; }
        c->acc_cap = (size_t)nnu * (size_t)ncell;
    }
    for (int l = 0; l < 3; ++l)
        for (int s = 0; s < P.nacc[l]; ++s)
            if (!c->acc[l][s]) FTTE_HIP(c, hipMalloc((void **)&c->acc[l][s], sizeof(double) * c->acc_cap));
    const size_t face_need = (size_t)ndir * nnu * (size_t)P.face_elems;
    if ((rc = ensure(c, &c->d_faces, &c->d_faces_cap, face_need))) return rc;
    if (!H.uploaded) {
        if ((rc = ensure(c, &c->d_blayers, &c->d_blayers_cap, P.layers.size()))) return rc;
        if ((rc = ensure(c, &c->d_btasks, &c->d_btasks_cap, P.tasks.size()))) return rc;
        if ((rc = ensure(c, &c->d_bgroups, &c->d_bgroups_cap, P.groups.size()))) return rc;
        FTTE_HIP(c, hipMemcpy(c->d_blayers, P.layers.data(), sizeof(LayerRec) * P.layers.size(), hipMemcpyHostToDevice));
        if (!P.tasks.empty()) FTTE_HIP(c, hipMemcpy(c->d_btasks, P.tasks.data(), sizeof(BrickTask) * P.tasks.size(), hipMemcpyHostToDevice));
        H.uploaded = true;
        c->bplan_uploaded = false; c->bplan.valid = false; // the uniform-grid plan shared these buffers
    }
    {
        std::vector<BrickGroup> G(P.groups.size());
        std::memset(G.data(), 0, sizeof(BrickGroup) * G.size());
        for (size_t g = 0; g < P.groups.size(); ++g) {
            const BrickPlan::Group &Hg = P.groups[g];
            const DirPlan &D0 = P.dirs[Hg.dirs[0]];
            G[g].kappa = c->base_kappa[Hg.layout];
            G[g].J = c->acc[Hg.layout][Hg.acc];
            G[g].org = D0.org; G[g].si = D0.si; G[g].sv = D0.sv; G[g].su = D0.su;
            G[g].ndir = (int)Hg.dirs.size();
            for (size_t q = 0; q < Hg.dirs.size(); ++q) {
                const int d = Hg.dirs[q];
                G[g].dir[q].layers = c->d_blayers + P.dirs[d].layer_off;
                G[g].dir[q].faces = c->d_faces + (size_t)d * nnu * (size_t)P.face_elems;
                G[g].dir[q].w = P.dirs[d].w;
            }
        }
        FTTE_HIP(c, hipMemcpy(c->d_bgroups, G.data(), sizeof(BrickGroup) * G.size(), hipMemcpyHostToDevice)); c->bgroups_sent.clear();
    }
    if ((rc = ensure(c, &c->d_uvb, &c->d_uvb_cap, (size_t)nnu))) return rc;
    FTTE_HIP(c, hipMemcpy(c->d_uvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice)); c->uvb_sent.clear();

    // forest scratch: as forest_sweep, for the leaves of the plan's list only
    const size_t per_dir = (size_t)3 * (size_t)std::max<int64_t>(H.ncells, 1) * nnu;
    int batch = std::max(1, std::min(ndir, kAmrBatch));
    if (c->amr_scratch_cap < per_dir * (size_t)batch) {
        if (c->amr_Iout) { FTTE_HIP(c, hipFree(c->amr_Iout)); c->amr_Iout = nullptr; }
        if (c->amr_mean) { FTTE_HIP(c, hipFree(c->amr_mean)); c->amr_mean = nullptr; }
        c->amr_scratch_cap = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, (size_t)(0.9 * (double)free_b) / (2 * sizeof(double) * per_dir)));
        FTTE_HIP(c, hipMalloc((void **)&c->amr_Iout, sizeof(double) * per_dir * (size_t)batch));
        FTTE_HIP(c, hipMalloc((void **)&c->amr_mean, sizeof(double) * per_dir * (size_t)batch));
        c->amr_scratch_cap = per_dir * (size_t)batch;
    } else batch = (int)std::min<size_t>((size_t)kAmrBatch, c->amr_scratch_cap / per_dir);
    if (nnu > 96) return FTTE_OK; // the cell-major copy of kappa is what the level kernel reads here: leave it to the forest path
    if ((rc = ensure(c, &c->amr_kappa, &c->amr_kappa_cap, (size_t)nnu * (size_t)std::max<int64_t>(H.ncells, 1)))) return rc;
    if (!c->kappa_ready[3] || c->amr_kappa_form != 1) {
        if (launch_cell_major(c->kappa[0], c->amr_kappa, ncell, nnu, stream, H.cells, (long)H.ncells)) return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        c->kappa_ready[3] = true; c->amr_kappa_form = 1;
    }

    while (c->timing.size() < 1) {
        LaunchTiming t;
        FTTE_HIP(c, hipEventCreate(&t.start));
        FTTE_HIP(c, hipEventCreate(&t.stop));
        c->timing.push_back(t);
    }
    LaunchTiming &Tm = c->timing[0];
    Tm.updates = (int64_t)ndir * ncell * nnu; Tm.lanes = 0;
    c->timing_used = 0;
    FTTE_HIP(c, hipEventRecord(Tm.start, stream));

    // ---- opacity of the base cells in the three layouts; accumulators and J start from zero
    if (launch_base_cells(c->kappa[0], c->d_leaf_of_base, c->base_kappa[0], (long)nbase, (long)ncell, nnu, stream))
        return fail(c, FTTE_ERR_NO_DEVICE, "base-cell kernel launch failed");
    for (int l = 1; l < 3; ++l)
        if (P.nacc[l] && launch_to_layout(l, c->base_kappa[0], c->base_kappa[l], n, nnu, (long)nbase, stream))
            return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
    for (int l = 0; l < 3; ++l)
        for (int s = 0; s < P.nacc[l]; ++s) FTTE_HIP(c, hipMemsetAsync(c->acc[l][s], 0, sizeof(double) * per_base, stream));
    FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * ncell, stream));

    static const ftte_consts kMath = FTTE_CONSTS_INIT;
    auto brick_stages = [&](int half, size_t from, size_t to, hipStream_t q) -> int {
        const size_t *off = &H.stage_off[(size_t)half * H.nlist];
        for (size_t l = from; l < to; ++l) {
            if (off[l + 1] == off[l]) continue;
            BrickLaunch L;
            std::memset(&L, 0, sizeof L);
            L.groups = c->d_bgroups;
            L.tasks = c->d_btasks + off[l];
            L.uvb = c->d_uvb;
            L.group_stride = nbase;
            L.face_stride = P.face_elems;
            L.vface_off = P.vface_off; L.iface_off = P.iface_off;
            L.n = n; L.ntasks = (int)(off[l + 1] - off[l]); L.nnu = nnu; L.nu0 = 0; L.chunk = P.chunk;
            L.up = P.up; L.vp = P.vp; L.uw = P.uw; L.ut = P.ut; L.nslot = P.nslot;
            L.math = kMath;
            const int lrc = launch_brick(L, P.max_dirs, c->brick_waves, q);
            if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
        }
        return FTTE_OK;
    };

    // ---- per half: bricks not behind the boxes, the forests of the boxes (all directions of the half per depth launch), the
    // bricks behind them.  The halves run side by side on two streams and meet only in J: the second half's means are added
    // after the first half's (an event), the bricks' accumulators after both.
    const int nh = (H.nhalves > 1 && batch >= ndir) ? H.nhalves : 1; // scratch for every direction at once, or one pipeline
    hipStream_t qs[ftte_ctx::kMaxPipes] = {stream, stream, stream, stream};
    if (nh > 1) {
        while ((int)c->lane_stream.size() < nh - 1) {
            hipStream_t q; hipEvent_t e;
            FTTE_HIP(c, hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
            FTTE_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            c->lane_stream.push_back(q); c->lane_done.push_back(e);
        }
        if (!c->ev_fork) FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        for (int r = 0; r < nh; ++r) {
            if (!c->ev_combine[r]) FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_combine[r], hipEventDisableTiming));
            if (r) qs[r] = c->lane_stream[(size_t)r - 1];
        }
    }
    AmrLevelRec A;
    std::memset(&A, 0, sizeof A);
    A.kappa = c->amr_kappa; A.emis = nullptr;
    A.group_stride = 1; A.cell_stride = nnu;
    A.emit = 0;
    A.uvb = c->d_uvb;
    A.ncell = ncell; A.nnu = nnu;
    A.cells = H.cells; A.ncells = H.ncells;
    A.face_stride = P.face_elems;
    A.math = kMath;
    std::vector<ForestRun> runs;
    {
        std::vector<std::vector<ForestDirHost>> sets((size_t)nh);
        std::vector<int> slot0((size_t)nh, 0);
        for (int h = 0; h < H.nhalves; ++h) {
            const int to = nh > 1 ? h : 0;
            for (int d : H.half_dirs[(size_t)h]) {
                const ftte_ctx::HybridPlan::Dir &D = H.dirs[(size_t)d];
                sets[(size_t)to].push_back(ForestDirHost{D.rec, D.active, P.dirs[(size_t)d].w, c->d_faces + (size_t)d * nnu * (size_t)P.face_elems,
                                                         D.exports, D.nexports, &D.depth_off});
            }
        }
        for (int r = 1; r < nh; ++r) slot0[(size_t)r] = slot0[(size_t)r - 1] + (int)sets[(size_t)r - 1].size();
        // one batch per pipeline when they run side by side (their scratch must not overlap), else `batch` directions at a time
        if ((rc = prepare_forests(c, stream, sets, slot0, nh > 1 ? ndir : batch, per_dir, &runs))) return rc;
    }
    if (nh > 1) {
        FTTE_HIP(c, hipEventRecord(c->ev_fork, stream));
        for (int r = 1; r < nh; ++r) FTTE_HIP(c, hipStreamWaitEvent(qs[r], c->ev_fork, 0));
    }
    // issued phase by phase, alternating between the streams, so that none waits for the host to finish with the others
    for (int h = 0; h < H.nhalves; ++h)
        if ((rc = brick_stages(h, 0, H.phase1_stages, qs[nh > 1 ? h : 0]))) return rc;
    for (int r = 0; r < nh; ++r)
        if ((rc = launch_forests(c, qs[r], runs[(size_t)r], A, J_dev, false, false, (nh > 1 && r > 0) ? c->ev_combine[r - 1] : nullptr,
                                 (nh > 1 && r + 1 < nh) ? c->ev_combine[r] : nullptr))) return rc;
    for (int h = 0; h < H.nhalves; ++h)
        if ((rc = brick_stages(h, H.phase1_stages, H.nlist, qs[nh > 1 ? h : 0]))) return rc;
    for (int r = 1; r < nh; ++r) {
        FTTE_HIP(c, hipEventRecord(c->lane_done[(size_t)r - 1], qs[r]));
        FTTE_HIP(c, hipStreamWaitEvent(stream, c->lane_done[(size_t)r - 1], 0));
    }

    // ---- J of the unrefined base cells += what the bricks stored (layout after layout, accumulator after accumulator)
    {
        const double *accs[3 * kMaxAcc];
        int layouts[3 * kMaxAcc], count = 0;
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < P.nacc[l]; ++s) { accs[count] = c->acc[l][s]; layouts[count++] = l; }
        if (count && launch_merge(accs, layouts, count, J_dev, n, nnu, (long)nbase, true, stream, c->d_leaf_of_base, (long)ncell))
            return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
    }
    FTTE_HIP(c, hipEventRecord(Tm.stop, stream));
    c->timing_used = 1;
    *done = true;
    return mark_sweep(c, stream);
}

// ---- host arrays across PCIe ------------------------------------------------------------------------------------
constexpr size_t kStageBytes = (size_t)64 << 20;

bool is_registered(const ftte_ctx *c, const void *p, size_t bytes)
{
    const char *b = (const char *)p;
    for (const auto &r : c->registered)
        if (b >= r.base && b + bytes <= r.base + r.bytes) return true;
    return false;
}

void parallel_copy(void *dst, const void *src, size_t bytes)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthreads = std::min<size_t>(std::min(8u, hw), std::max<size_t>(1, bytes >> 22));
    if (nthreads <= 1) { std::memcpy(dst, src, bytes); return; }
    const size_t chunk = ((bytes + nthreads - 1) / nthreads + 63) & ~(size_t)63;
    std::vector<std::thread> pool;
    for (size_t t = 0; t < nthreads; ++t) {
        const size_t lo = t * chunk;
        if (lo >= bytes) break;
        const size_t len = std::min(chunk, bytes - lo);
        pool.emplace_back([=] { std::memcpy((char *)dst + lo, (const char *)src + lo, len); });
    }
    for (auto &th : pool) th.join();
}

int ensure_stage(ftte_ctx *c)
{
    for (int q = 0; q < 2; ++q) {
        if (!c->stage[q]) FTTE_HIP(c, hipHostMalloc(&c->stage[q], kStageBytes, hipHostMallocDefault));
        if (!c->stage_ev[q]) FTTE_HIP(c, hipEventCreateWithFlags(&c->stage_ev[q], hipEventDisableTiming));
    }
    return FTTE_OK;
}

// host -> device on c->stream; returns with the copy complete
int upload(ftte_ctx *c, void *dst_dev, const void *src_host, size_t bytes)
{
    if (is_registered(c, src_host, bytes) || bytes < ((size_t)1 << 20)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
        FTTE_HIP(c, hipStreamSynchronize(c->stream));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    int q = 0;
    bool busy[2] = {false, false};
    for (size_t off = 0; off < bytes; off += kStageBytes, q ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        if (busy[q]) FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q]));
        parallel_copy(c->stage[q], (const char *)src_host + off, len);
        FTTE_HIP(c, hipMemcpyAsync((char *)dst_dev + off, c->stage[q], len, hipMemcpyHostToDevice, c->stream));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[q], c->stream));
        busy[q] = true;
    }
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

// device -> host on c->stream (after whatever is queued there); returns with the copy complete
int download(ftte_ctx *c, void *dst_host, const void *src_dev, size_t bytes)
{
    if (is_registered(c, dst_host, bytes) || bytes < ((size_t)1 << 20)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
        FTTE_HIP(c, hipStreamSynchronize(c->stream));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    // block q is filled by the DMA engine while the host threads empty block q^1
    size_t off_prev = 0, len_prev = 0;
    bool have_prev = false;
    int q = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, q ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        FTTE_HIP(c, hipMemcpyAsync(c->stage[q], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, c->stream));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[q], c->stream));
        if (have_prev) {
            FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q ^ 1]));
            parallel_copy((char *)dst_host + off_prev, c->stage[q ^ 1], len_prev);
        }
        off_prev = off; len_prev = len; have_prev = true;
    }
    if (have_prev) {
        FTTE_HIP(c, hipEventSynchronize(c->stage_ev[q ^ 1]));
        parallel_copy((char *)dst_host + off_prev, c->stage[q ^ 1], len_prev);
    }
    return FTTE_OK;
}

// host -> device on stream q; returns when the last piece has been handed to the DMA engine (not when it has arrived)
int upload_on(ftte_ctx *c, hipStream_t q, void *dst_dev, const void *src_host, size_t bytes)
{
    if (is_registered(c, src_host, bytes)) {
        FTTE_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, q));
        return FTTE_OK;
    }
    int rc = ensure_stage(c);
    if (rc) return rc;
    int b = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, b ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        if (c->stage_used[b]) FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b]));
        parallel_copy(c->stage[b], (const char *)src_host + off, len);
        FTTE_HIP(c, hipMemcpyAsync((char *)dst_dev + off, c->stage[b], len, hipMemcpyHostToDevice, q));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[b], q));
        c->stage_used[b] = true;
    }
    return FTTE_OK;
}

// device -> pageable host memory behind whatever is queued on stream q; returns with the copy complete
int download_on(ftte_ctx *c, hipStream_t q, void *dst_host, const void *src_dev, size_t bytes)
{
    int rc = ensure_stage(c);
    if (rc) return rc;
    for (int b = 0; b < 2; ++b)
        if (c->stage_used[b]) { FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b])); c->stage_used[b] = false; }
    size_t off_prev = 0, len_prev = 0;
    bool have_prev = false;
    int b = 0;
    for (size_t off = 0; off < bytes; off += kStageBytes, b ^= 1) {
        const size_t len = std::min(kStageBytes, bytes - off);
        FTTE_HIP(c, hipMemcpyAsync(c->stage[b], (const char *)src_dev + off, len, hipMemcpyDeviceToHost, q));
        FTTE_HIP(c, hipEventRecord(c->stage_ev[b], q));
        if (have_prev) {
            FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b ^ 1]));
            parallel_copy((char *)dst_host + off_prev, c->stage[b ^ 1], len_prev);
        }
        off_prev = off; len_prev = len; have_prev = true;
    }
    if (have_prev) {
        FTTE_HIP(c, hipEventSynchronize(c->stage_ev[b ^ 1]));
        parallel_copy((char *)dst_host + off_prev, c->stage[b ^ 1], len_prev);
    }
    return FTTE_OK;
}

} // namespace

// =================================================================================================
extern "C" {

int ftte_create(ftte_ctx **out, int ndev, const int *dev_ids)
{
    if (!out) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: ctx is NULL");
    *out = nullptr;
    if (ndev != 1)
        return fail(nullptr, FTTE_ERR_UNSUPPORTED,
                    "ftte_create: one context drives one device (ndev must be 1); run one process per GPU and reduce J "
                    "with RCCL in the host driver");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, FTTE_ERR_NO_DEVICE, std::string("ftte_create: no HIP device (") +
                                                     (e != hipSuccess ? hipGetErrorString(e) : "device count is 0") + ")");
    int dev = 0;
    if (dev_ids) dev = dev_ids[0];
    else if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= count) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(nullptr, FTTE_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, FTTE_ERR_NO_DEVICE, std::string("ftte_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    ftte_ctx *c = new ftte_ctx;
    c->device = dev;
    if (hipSetDevice(dev) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) {
        delete c;
        return fail(nullptr, FTTE_ERR_NO_DEVICE, "ftte_create: cannot create a stream on the device");
    }
    *out = c;
    return FTTE_OK;
}

int ftte_destroy(ftte_ctx *c)
{
    if (!c) return FTTE_ERR_ARG;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < 3; ++l) {
        if (c->kappa[l]) (void)hipFree(c->kappa[l]);
        if (c->emis[l]) (void)hipFree(c->emis[l]);
        for (int s = 0; s < kMaxAcc; ++s) if (c->acc[l][s]) (void)hipFree(c->acc[l][s]);
    }
    if (c->d_bdeps) (void)hipFree(c->d_bdeps);
    if (c->d_bdone) (void)hipFree(c->d_bdone);
    if (c->d_bsync) (void)hipFree(c->d_bsync);
    if (c->h_berror) (void)hipHostFree(c->h_berror);
    if (c->d_blayers) (void)hipFree(c->d_blayers);
    if (c->d_bgroups) (void)hipFree(c->d_bgroups);
    if (c->d_btasks) (void)hipFree(c->d_btasks);
    if (c->d_faces) (void)hipFree(c->d_faces);
    if (c->d_layers) (void)hipFree(c->d_layers);
    if (c->d_items) (void)hipFree(c->d_items);
    if (c->d_uvb) (void)hipFree(c->d_uvb);
    free_forests(c);
    free_hybrid(c);
    if (c->d_leaf_of_base) (void)hipFree(c->d_leaf_of_base);
    for (int l = 0; l < 3; ++l) if (c->base_kappa[l]) (void)hipFree(c->base_kappa[l]);
    if (c->amr_Iout) (void)hipFree(c->amr_Iout);
    if (c->amr_mean) (void)hipFree(c->amr_mean);
    if (c->d_amr_dirs) (void)hipFree(c->d_amr_dirs);
    if (c->d_amr_tables) (void)hipFree(c->d_amr_tables);
    if (c->amr_kappa) (void)hipFree(c->amr_kappa);
    if (c->amr_emis) (void)hipFree(c->amr_emis);
    if (c->merge_stream) (void)hipStreamDestroy(c->merge_stream);
    if (c->ev_layout_done) (void)hipEventDestroy(c->ev_layout_done);
    if (c->ev_merge_done) (void)hipEventDestroy(c->ev_merge_done);
    if (c->ev_layouts_ready) (void)hipEventDestroy(c->ev_layouts_ready);
    if (c->ev_sweep_done) (void)hipEventDestroy(c->ev_sweep_done);
    for (auto &q : c->lane_stream) (void)hipStreamDestroy(q);
    for (auto &e : c->pipe_up) (void)hipEventDestroy(e);
    for (auto &e : c->lane_done) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (auto &e : c->ev_combine) if (e) (void)hipEventDestroy(e);
    if (c->host_J_dev) (void)hipFree(c->host_J_dev);
    for (int q = 0; q < 2; ++q) {
        if (c->stage[q]) (void)hipHostFree(c->stage[q]);
        if (c->stage_ev[q]) (void)hipEventDestroy(c->stage_ev[q]);
    }
    for (auto &r : c->registered) (void)hipHostUnregister((void *)r.base);
    c->point.release();
    c->drop_chem_grid();
    if (c->chem_k) (void)hipFree(c->chem_k);
    if (c->chem_counters) (void)hipFree(c->chem_counters);
    for (auto &t : c->timing) {
        (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop);
        for (auto &e : t.first) (void)hipEventDestroy(e);
        for (auto &e : t.last) (void)hipEventDestroy(e);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FTTE_OK;
}

const char *ftte_last_error(const ftte_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ftte_set_grid(ftte_ctx *c, int nx, int ny, int nz, int64_t ncell, const int32_t *level, double box_cm)
{
    if (!c) return FTTE_ERR_ARG;
    if (nx < 1 || !level || ncell < 1 || !(box_cm > 0.0)) return fail(c, FTTE_ERR_ARG, "ftte_set_grid: bad argument");
    if (nx != ny || nx != nz) return fail(c, FTTE_ERR_NOT_CUBIC, "base grid needs to be of size n^3");
    if (nx > 32000) return fail(c, FTTE_ERR_UNSUPPORTED, "ftte_set_grid: n > 32000");
    // The reference's tree is static over a run while its driver would hand the same list over on every outer iteration
    // (the drop-ins do): an unchanged list keeps the tree, the sweep plan, the segment forests and the resident medium.
    // Only the box may differ (the plans are keyed on it themselves).
    if (c->grid_set && c->n == nx && c->ncell == ncell && (int64_t)c->leaf_level.size() == ncell) {
        bool same = true;
        const int8_t *have = c->leaf_level.data();
        for (int64_t q = 0; q < ncell; ++q)
            if ((int32_t)have[q] != level[q]) { same = false; break; }
        if (same) { c->box = box_cm; return FTTE_OK; }
    }
    // rebuild the tree exactly as createFullyThreadedStructure does (readCellArray.f90:154-187); this also
    // validates the list
    ++c->n_grid_builds;
    if (c->sweep_pending) { (void)hipSetDevice(c->device); (void)hipEventSynchronize(c->ev_sweep_done); c->sweep_pending = false; }
    AmrTree tree;
    const std::string terr = tree.build(nx, ncell, level);
    if (!terr.empty()) return fail(c, FTTE_ERR_LEVELS, terr);
    if (tree.refined() && 3 * ncell >= (int64_t)1 << 31) return fail(c, FTTE_ERR_UNSUPPORTED, "refined cell array with more than 7.1e8 leaves");
    if (c->grid_set && (c->n != nx || c->ncell != ncell)) {
        // a different grid: drop everything sized by the old one
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        for (int l = 0; l < 3; ++l) {
            if (c->kappa[l]) { (void)hipFree(c->kappa[l]); c->kappa[l] = nullptr; }
            if (c->emis[l]) { (void)hipFree(c->emis[l]); c->emis[l] = nullptr; }
            for (int s = 0; s < kMaxAcc; ++s) if (c->acc[l][s]) { (void)hipFree(c->acc[l][s]); c->acc[l][s] = nullptr; }
        }
        c->emit_mode = 0;
        if (c->amr_Iout) { (void)hipFree(c->amr_Iout); c->amr_Iout = nullptr; }
        if (c->amr_mean) { (void)hipFree(c->amr_mean); c->amr_mean = nullptr; }
        if (c->amr_kappa) { (void)hipFree(c->amr_kappa); c->amr_kappa = nullptr; }
        if (c->amr_emis) { (void)hipFree(c->amr_emis); c->amr_emis = nullptr; }
        c->kappa_cap = c->acc_cap = c->amr_scratch_cap = c->amr_kappa_cap = c->amr_emis_cap = 0;
        c->nnu = 0;
    }
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_forests(c);
    free_hybrid(c);
    if (c->d_leaf_of_base) { (void)hipFree(c->d_leaf_of_base); c->d_leaf_of_base = nullptr; }
    c->point.drop_grid();
    c->drop_chem_grid();
    c->leaf_level.assign(level, level + ncell);
    c->n = nx; c->ncell = ncell; c->box = box_cm; c->grid_set = true;
    c->kappa_ready[0] = c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    c->plan.valid = false;
    c->bplan.valid = false;
    c->tree = std::move(tree);
    c->use_forest = c->tree.refined() || c->force_forest;
    return FTTE_OK;
}

int ftte_set_opacity(ftte_ctx *c, int nnu, const double *kappa)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !kappa) return fail(c, FTTE_ERR_ARG, "ftte_set_opacity: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure_kappa(c, nnu))) return rc;
    if ((rc = upload(c, c->kappa[0], kappa, sizeof(double) * nnu * c->ncell))) return rc;
    c->nnu = nnu;
    c->kappa_ready[0] = true; c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    return FTTE_OK;
}

int ftte_set_opacity_device(ftte_ctx *c, int nnu, const double *kappa_dev)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !kappa_dev) return fail(c, FTTE_ERR_ARG, "ftte_set_opacity_device: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    if ((rc = ensure_kappa(c, nnu))) return rc;
    FTTE_HIP(c, hipMemcpyAsync(c->kappa[0], kappa_dev, sizeof(double) * nnu * c->ncell, hipMemcpyDeviceToDevice, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream)); // the sweep may run on another stream: the copy must have landed
    c->nnu = nnu;
    c->kappa_ready[0] = true; c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    return FTTE_OK;
}

int ftte_set_species(ftte_ctx *c, int nnu, const double *HI, const double *HeI, const double *HeII, const double *beta)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !HI || !HeI || !HeII || !beta) return fail(c, FTTE_ERR_ARG, "ftte_set_species: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure_kappa(c, nnu))) return rc;
    double *tmp = nullptr;
    const size_t nc = (size_t)c->ncell;
    FTTE_HIP(c, hipMalloc((void **)&tmp, sizeof(double) * (3 * nc + 3 * (size_t)nnu)));
    hipError_t e = hipMemcpyAsync(tmp, HI, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + nc, HeI, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + 2 * nc, HeII, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp + 3 * nc, beta, sizeof(double) * 3 * nnu, hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess) lrc = launch_opacity(tmp, tmp + nc, tmp + 2 * nc, tmp + 3 * nc, c->kappa[0], (long)nc, nnu, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(c, FTTE_ERR_NO_DEVICE, std::string("ftte_set_species: ") + hipGetErrorString(e));
    if (lrc) return fail(c, FTTE_ERR_NO_DEVICE, "ftte_set_species: kernel launch failed");
    c->nnu = nnu;
    c->kappa_ready[0] = true; c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    return FTTE_OK;
}

static int set_emission(ftte_ctx *c, int mode, const double *values, bool on_device, const char *who)
{
    if (!c) return FTTE_ERR_ARG;
    if (!values) { c->emit_mode = 0; return FTTE_OK; }
    int rc = check_ready(c, true);
    if (rc) return rc;
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    if (!on_device) FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if (!c->emis[0]) FTTE_HIP(c, hipMalloc((void **)&c->emis[0], sizeof(double) * c->kappa_cap));
    const size_t bytes = sizeof(double) * (size_t)c->nnu * c->ncell;
    FTTE_HIP(c, hipMemcpyAsync(c->emis[0], values, bytes, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream)); // the sweep may run on another stream: the copy must have landed
    c->emit_mode = mode;
    c->emis_ready[0] = true; c->emis_ready[1] = c->emis_ready[2] = c->emis_ready[3] = false;
    (void)who;
    return FTTE_OK;
}

int ftte_set_emissivity(ftte_ctx *c, const double *eta) { return set_emission(c, 1, eta, false, "ftte_set_emissivity"); }
int ftte_set_emissivity_device(ftte_ctx *c, const double *eta_dev) { return set_emission(c, 1, eta_dev, true, "ftte_set_emissivity_device"); }
int ftte_set_source_function(ftte_ctx *c, const double *S) { return set_emission(c, 2, S, false, "ftte_set_source_function"); }
int ftte_set_source_function_device(ftte_ctx *c, const double *S_dev) { return set_emission(c, 2, S_dev, true, "ftte_set_source_function_device"); }

int ftte_set_option(ftte_ctx *c, const char *key, int value)
{
    if (!c || !key) return FTTE_ERR_ARG;
    if (!std::strcmp(key, "rows")) {
        if (value != 4 && value != 8 && value != 16) return fail(c, FTTE_ERR_ARG, "rows must be 4, 8 or 16");
        c->rows = value;
    } else if (!std::strcmp(key, "slots")) {
        if (value < 1 || value > kMaxSlots) return fail(c, FTTE_ERR_ARG, "slots must be 1..16");
        c->slots = value;
    } else if (!std::strcmp(key, "waves")) {
        if (value < 2 || value > 6) return fail(c, FTTE_ERR_ARG, "waves must be 2..6");
        c->waves = value;
    } else if (!std::strcmp(key, "forest")) {
        if (value != 0 && value != 1) return fail(c, FTTE_ERR_ARG, "forest must be 0 or 1");
        c->force_forest = value;
        c->use_forest = (c->grid_set && c->tree.refined()) || value;
    } else if (!std::strcmp(key, "ldspad")) {
        if (value < 0 || value > 160 * 1024) return fail(c, FTTE_ERR_ARG, "ldspad must be 0..163840 bytes");
        set_lds_pad(value);
    } else if (!std::strcmp(key, "engine")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "engine must be 0 (automatic), 1 (ray-following tiles) or 2 (cell-fixed bricks)");
        c->engine = value;
    } else if (!std::strcmp(key, "chunk")) {
        if (value < 0 || value > 4096) return fail(c, FTTE_ERR_ARG, "chunk (layers per brick) must be 1..4096, or 0 for the default");
        c->chunk = value;
    } else if (!std::strcmp(key, "group")) {
        if (value < 0 || value > kBrickMaxDirs) return fail(c, FTTE_ERR_ARG, "group (directions sharing a brick pass) must be 1..8, or 0 for the default");
        c->group = value;
    } else if (!std::strcmp(key, "hybrid")) {
        if (value != 0 && value != 1) return fail(c, FTTE_ERR_ARG, "hybrid must be 0 (a refined cell array goes through the forest path as a whole) or 1 (bricks outside a box around the refined cells)");
        c->hybrid = value;
        c->hplan.valid = false;
    } else if (!std::strcmp(key, "pipelines")) {
        if (value < 1 || value > ftte_ctx::kMaxPipes) return fail(c, FTTE_ERR_ARG, "pipelines (independent bricks-forests-bricks sequences of the hybrid sweep, each on a stream of its own) must be 1..4");
        c->halves = value;
        c->hplan.valid = false;
    } else if (!std::strcmp(key, "dataflow")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "dataflow must be 0 (a launch per stage), 1 (one launch, bricks wait for each other) or 2 (the same with write-through stores)");
        c->dataflow = value;
    } else if (!std::strcmp(key, "lanes")) {
        if (value < 1 || value > 16) return fail(c, FTTE_ERR_ARG, "lanes (streams the brick sweep spreads its frequency groups over) must be 1..16");
        c->lanes = value;
    } else if (!std::strcmp(key, "team")) {
        if (value != 0 && value != 1) return fail(c, FTTE_ERR_ARG, "team must be 0 (one wavefront sweeps a group's directions in turn) or 1 (one wavefront per direction)");
        c->team = value;
    } else if (!std::strcmp(key, "share")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "share (groups sharing a J accumulator) must be 0 (none), 1 (passes of one izone) or 2 (and izone pairs)");
        c->share = value;
    } else if (!std::strcmp(key, "brick_waves")) {
        if (value < 2 || value > 4) return fail(c, FTTE_ERR_ARG, "brick_waves must be 2..4");
        c->brick_waves = value;
    } else if (!std::strcmp(key, "stack")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return fail(c, FTTE_ERR_ARG, "stack must be 1, 2, 4 or 8");
        c->stack = value;
    } else return fail(c, FTTE_ERR_ARG, std::string("unknown option: ") + key);
    c->plan.valid = false;
    c->bplan.valid = false;
    c->hplan.valid = false;
    return FTTE_OK;
}

int ftte_diffuse_sweep_device(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w,
                              const double *uvb, double *J_dev, void *stream_v)
{
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (ndir < 0 || (ndir > 0 && (!phi || !theta || !w)) || !uvb || !J_dev)
        return fail(c, FTTE_ERR_ARG, "ftte_diffuse_sweep: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : c->stream;
    const int n = c->n, nnu = c->nnu;
    const size_t per_acc = (size_t)nnu * c->ncell;

    if (c->use_forest) {
        if (c->hybrid && c->tree.refined() && !c->force_forest && !c->emit_mode && ndir > 0) {
            bool done = false;
            if ((rc = hybrid_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream, &done)) || done) return rc;
        }
        return forest_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream);
    }
    if (c->engine != 1) return brick_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream);
    // the emission variants of the tiled kernel are built for one shape
    const int rows = c->emit_mode ? 8 : c->rows, stack = c->emit_mode ? 1 : c->stack;
    if ((rc = build_plan(c, rows, stack, ndir, phi, theta, w))) return rc;
    Plan &P = c->plan;

    // everything below overwrites device tables the previous sweep may still be reading
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(stream));
    if (stream != c->stream) FTTE_HIP(c, hipStreamSynchronize(c->stream));

    if (!c->plan_uploaded) {
        if ((rc = ensure(c, &c->d_layers, &c->d_layers_cap, P.layers.size()))) return rc;
        if ((rc = ensure(c, &c->d_items, &c->d_items_cap, P.items.size()))) return rc;
        if (!P.layers.empty())
            FTTE_HIP(c, hipMemcpy(c->d_layers, P.layers.data(), sizeof(LayerRec) * P.layers.size(), hipMemcpyHostToDevice));
        if (!P.items.empty())
            FTTE_HIP(c, hipMemcpy(c->d_items, P.items.data(), sizeof(WorkItem) * P.items.size(), hipMemcpyHostToDevice));
        c->plan_uploaded = true;
    }
    if ((rc = ensure(c, &c->d_uvb, &c->d_uvb_cap, (size_t)nnu))) return rc;
    FTTE_HIP(c, hipMemcpy(c->d_uvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice)); c->uvb_sent.clear();

    // accumulators sized for this nnu
    if (c->acc_cap < per_acc) {
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < kMaxAcc; ++s)
                if (c->acc[l][s]) { FTTE_HIP(c, hipFree(c->acc[l][s])); c->acc[l][s] = nullptr; }
        c->acc_cap = per_acc;
    }
    // a second (non-blocking) stream: the transposed copies of the opacity are made there while the directions that march
    // along storage-i (layout 0, the array as it was handed over) are already being swept, and later the merges run there
    if (!c->merge_stream) {
        FTTE_HIP(c, hipStreamCreateWithFlags(&c->merge_stream, hipStreamNonBlocking));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_layout_done, hipEventDisableTiming));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_merge_done, hipEventDisableTiming));
        FTTE_HIP(c, hipEventCreateWithFlags(&c->ev_layouts_ready, hipEventDisableTiming));
    }
    // everything queued on `stream` so far (and the previous sweep's merges) comes first
    FTTE_HIP(c, hipEventRecord(c->ev_layout_done, stream));
    FTTE_HIP(c, hipStreamWaitEvent(c->merge_stream, c->ev_layout_done, 0));
    for (int l = 0; l < 3; ++l) {
        bool any = false;
        for (int s = 0; s < kMaxSlots; ++s) {
            if (!P.used[l][s]) continue;
            any = true;
            if (!c->acc[l][s]) FTTE_HIP(c, hipMalloc((void **)&c->acc[l][s], sizeof(double) * c->acc_cap));
        }
        // opacity in the layout this march axis needs
        if (any && !c->kappa_ready[l]) {
            if (!c->kappa[l]) FTTE_HIP(c, hipMalloc((void **)&c->kappa[l], sizeof(double) * c->kappa_cap));
            if (launch_to_layout(l, c->kappa[0], c->kappa[l], n, nnu, (long)c->ncell, c->merge_stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            c->kappa_ready[l] = true;
        }
        if (any && c->emit_mode && !c->emis_ready[l]) {
            if (!c->emis[l]) FTTE_HIP(c, hipMalloc((void **)&c->emis[l], sizeof(double) * c->kappa_cap));
            if (launch_to_layout(l, c->emis[0], c->emis[l], n, nnu, (long)c->ncell, c->merge_stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            c->emis_ready[l] = true;
        }
    }
    FTTE_HIP(c, hipEventRecord(c->ev_layouts_ready, c->merge_stream));
    bool layouts_awaited = false;

    // events for the launch records
    while (c->timing.size() < P.launches.size()) {
        LaunchTiming t;
        FTTE_HIP(c, hipEventCreate(&t.start));
        FTTE_HIP(c, hipEventCreate(&t.stop));
        c->timing.push_back(t);
    }
    c->timing_used = 0;

    bool merged_any = false;
    for (size_t li = 0; li < P.launches.size(); ++li) {
        const LaunchPlan &LP = P.launches[li];
        LaunchRec L;
        std::memset(&L, 0, sizeof L);
        for (size_t s = 0; s < LP.dirs.size(); ++s) {
            const DirPlan &D = P.dirs[LP.dirs[s]];
            DirRec &R = L.dir[s];
            R.layers = c->d_layers + D.layer_off;
            R.kappa = c->kappa[LP.layout];
            R.J = c->acc[LP.layout][LP.acc_base + s];
            R.emis = c->emit_mode ? c->emis[LP.layout] : nullptr;
            R.org = D.org;
            R.si = D.si; R.sv = D.sv; R.su = D.su;
            R.u_lo = D.u_lo; R.v_lo = D.v_lo;
            R.first = LP.first ? 1 : 0;
            R.w = D.w;
        }
        L.items = c->d_items + LP.item_off;
        L.uvb = c->d_uvb;
        L.group_stride = c->ncell;
        L.n = n;
        L.nitems = LP.nitems;
        L.nnu = nnu;
        L.emit = c->emit_mode;
        static const ftte_consts kMath = FTTE_CONSTS_INIT;
        L.math = kMath;
        LaunchTiming &T = c->timing[li];
        T.updates = LP.updates * nnu; T.lanes = 0;
        if (LP.layout != 0 && !layouts_awaited) { // the first launch that reads a transposed copy
            FTTE_HIP(c, hipStreamWaitEvent(stream, c->ev_layouts_ready, 0));
            layouts_awaited = true;
        }
        FTTE_HIP(c, hipEventRecord(T.start, stream));
        const int lrc = launch_sweep(L, rows, c->waves, stack, nnu, stream);
        if (lrc == -1)
            return fail(c, FTTE_ERR_ARG, "no sweep kernel variant for this rows/stack/waves combination (rows x stack: 4x{1,4,8}, "
                                         "8x{1,2,4}, 16x1; waves 2, 3, 4, 6)");
        if (lrc) return fail(c, FTTE_ERR_NO_DEVICE, "sweep kernel launch failed");
        FTTE_HIP(c, hipEventRecord(T.stop, stream));
        c->timing_used = (int)li + 1;

        // J (+)= the accumulators of this layout, slots in order, layout 0 first -- the same sequence of additions as one
        // merge over all of them -- on the second stream, beside the sweeps that follow: the accumulators that the
        // layout's (short) last launch does not touch as soon as the launch before it is done, the rest after the last one.
        // Only the tail of the last layout's merge has nothing to hide behind.
        const bool last_of_layout = li + 1 == P.launches.size() || P.launches[li + 1].layout != LP.layout;
        const bool before_last = !last_of_layout && (li + 2 == P.launches.size() || P.launches[li + 2].layout != LP.layout);
        int lo = -1, hi = -1; // accumulator range [lo, hi) to merge now
        if (before_last && P.launches[li + 1].acc_base > 0) { lo = 0; hi = P.launches[li + 1].acc_base; }
        if (last_of_layout) { lo = LP.acc_base; hi = kMaxSlots; }
        if (lo >= 0) {
            const double *accs[kMaxSlots];
            int layouts[kMaxSlots], count = 0;
            for (int s = lo; s < hi; ++s)
                if (P.used[LP.layout][s]) { accs[count] = c->acc[LP.layout][s]; layouts[count++] = LP.layout; }
            if (count) {
                FTTE_HIP(c, hipEventRecord(c->ev_layout_done, stream));
                FTTE_HIP(c, hipStreamWaitEvent(c->merge_stream, c->ev_layout_done, 0));
                if (launch_merge(accs, layouts, count, J_dev, n, nnu, (long)c->ncell, merged_any, c->merge_stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
                merged_any = true;
            }
        }
    }
    if (!merged_any) FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * c->ncell, stream)); // no directions
    FTTE_HIP(c, hipEventRecord(c->ev_merge_done, c->merge_stream));
    FTTE_HIP(c, hipStreamWaitEvent(stream, c->ev_merge_done, 0));
    return mark_sweep(c, stream);
}

int ftte_diffuse_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb,
                       double *J)
{
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!J) return fail(c, FTTE_ERR_ARG, "ftte_diffuse_sweep: J is NULL");
    FTTE_HIP(c, hipSetDevice(c->device));
    const size_t elems = (size_t)c->nnu * c->ncell;
    if ((rc = ensure(c, &c->host_J_dev, &c->host_J_cap, elems))) return rc; // kept from call to call
    if ((rc = ftte_diffuse_sweep_device(c, ndir, phi, theta, w, uvb, c->host_J_dev, nullptr))) return rc;
    if ((rc = download(c, J, c->host_J_dev, sizeof(double) * elems))) return rc;
    return wait_sweep(c); // the sweep has drained: report a dataflow sweep that gave up now rather than at the next call
}

/* ftte_set_opacity + ftte_diffuse_sweep in one call, and faster than the two: on a uniform grid swept by the brick engine the
 * frequency groups travel in lanes (option "lanes") -- the first lane is swept while the second one's opacities are still
 * crossing PCIe, and its J goes back while the second is swept.  Same results; elsewhere the two calls one after the other. */
int ftte_diffuse_iteration(ftte_ctx *c, int nnu, const double *kappa, int ndir, const double *phi, const double *theta, const double *w,
                           const double *uvb, double *J)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !kappa || !J || ndir < 0 || (ndir > 0 && (!phi || !theta || !w)) || !uvb)
        return fail(c, FTTE_ERR_ARG, "ftte_diffuse_iteration: bad argument");
    const bool lanes_apply = !c->use_forest && c->engine != 1 && !c->emit_mode && !c->team && !c->dataflow && ndir > 0 && c->lanes >= 2 &&
                             nnu >= c->lanes;
    if (!lanes_apply) {
        if ((rc = ftte_set_opacity(c, nnu, kappa))) return rc;
        return ftte_diffuse_sweep(c, ndir, phi, theta, w, uvb, J);
    }
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure_kappa(c, nnu))) return rc;
    c->nnu = nnu;
    c->kappa_ready[0] = c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    const size_t elems = (size_t)nnu * c->ncell;
    if ((rc = ensure(c, &c->host_J_dev, &c->host_J_cap, elems))) return rc;
    const HostPipe pipe{kappa, J};
    if ((rc = brick_sweep(c, ndir, phi, theta, w, uvb, c->host_J_dev, c->stream, &pipe))) {
        c->kappa_ready[0] = c->kappa_ready[1] = c->kappa_ready[2] = false;
        return rc;
    }
    return wait_sweep(c); // J is in the caller's array on return
}

/* Pins a caller-owned host array for as long as it stays registered: ftte_set_opacity / ftte_diffuse_sweep then move
 * it by DMA directly (no staging copy).  The caller unregisters it before freeing it. */
int ftte_host_register(ftte_ctx *c, void *ptr, size_t bytes)
{
    if (!c) return FTTE_ERR_ARG;
    if (!ptr || !bytes) return fail(c, FTTE_ERR_ARG, "ftte_host_register: bad argument");
    if (is_registered(c, ptr, bytes)) return FTTE_OK;
    FTTE_HIP(c, hipSetDevice(c->device));
    FTTE_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    c->registered.push_back({(const char *)ptr, bytes});
    return FTTE_OK;
}

int ftte_host_unregister(ftte_ctx *c, void *ptr)
{
    if (!c) return FTTE_ERR_ARG;
    for (size_t q = 0; q < c->registered.size(); ++q)
        if (c->registered[q].base == (const char *)ptr) {
            FTTE_HIP(c, hipSetDevice(c->device));
            FTTE_HIP(c, hipStreamSynchronize(c->stream));
            FTTE_HIP(c, hipHostUnregister(ptr));
            c->registered.erase(c->registered.begin() + (long)q);
            return FTTE_OK;
        }
    return fail(c, FTTE_ERR_ARG, "ftte_host_unregister: not a registered array");
}

long long ftte_counter(const ftte_ctx *c, const char *name)
{
    if (!c || !name) return -1;
    if (!std::strcmp(name, "grid_builds")) return c->n_grid_builds;
    if (!std::strcmp(name, "plan_builds")) return c->n_plan_builds;
    if (!std::strcmp(name, "forest_builds")) return c->n_forest_builds;
    return -1;
}

int ftte_launch_count(const ftte_ctx *c) { return c ? c->timing_used : 0; }

int ftte_launch_info(ftte_ctx *c, int idx, double *ms, int64_t *updates)
{
    if (!c || idx < 0 || idx >= c->timing_used) return FTTE_ERR_ARG;
    float t = 0.f;
    const LaunchTiming &T = c->timing[idx];
    if (T.lanes > 0) {
        float begin = 0.f, end = 0.f;
        for (int l = 0; l < T.lanes; ++l) {
            float b = 0.f, e = 0.f;
            FTTE_HIP(c, hipEventElapsedTime(&b, T.start, T.first[(size_t)l]));
            FTTE_HIP(c, hipEventElapsedTime(&e, T.start, T.last[(size_t)l]));
            begin = l ? std::min(begin, b) : b;
            end = l ? std::max(end, e) : e;
        }
        t = end - begin;
    } else FTTE_HIP(c, hipEventElapsedTime(&t, T.start, T.stop));
    if (ms) *ms = t;
    if (updates) *updates = c->timing[idx].updates;
    return FTTE_OK;
}

// ---- host geometry ----------------------------------------------------------------------------------
int ftte_rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *ic, int *jc, int *kc)
{
    if (!ic || !jc || !kc) return FTTE_ERR_ARG;
    return rotate_indices(i, j, k, nx, ny, nz, izone, ic, jc, kc) ? FTTE_ERR_IZONE : FTTE_OK;
}

int ftte_pix2ang_nest(int nside, int64_t ipix, double *phi, double *theta)
{
    if (!phi || !theta) return FTTE_ERR_ARG;
    return pix2ang_nest(nside, ipix, phi, theta) ? FTTE_ERR_PIXEL : FTTE_OK;
}

int ftte_fold_direction(double phi_large, double theta_large, double *phi, double *theta, int *izone)
{
    if (!phi || !theta || !izone) return FTTE_ERR_ARG;
    const int rc = fold_direction(phi_large, theta_large, phi, theta, izone);
    return rc ? fold_status(rc) : FTTE_OK;
}

int ftte_set_pattern(ftte_pattern *pattern, double phi, double theta)
{
    if (!pattern) return FTTE_ERR_ARG;
    return set_pattern(pattern, phi, theta) ? FTTE_ERR_PATTERN : FTTE_OK;
}

int ftte_layer_patterns(int n, double phi, double theta, ftte_pattern *layers)
{
    if (n < 1 || !layers) return FTTE_ERR_ARG;
    return layer_patterns(n, phi, theta, layers) ? FTTE_ERR_PATTERN : FTTE_OK;
}

void ftte_compute_cell_intensity(double *Jmean, double Iin, double Iout)
{
    // transportRoutinesModule.f90:1044-1048, the reference's own formula (host helper; the device
    // evaluates the same mean through ftte_math.h)
    if (Iout < Iin) *Jmean += (Iin - Iout) / std::log(Iin / Iout);
    else *Jmean += 0.5 * (Iin + Iout);
}

// ---- point sources --------------------------------------------------------------------------------------------

int ftte_stellar_beta_table(ftte_ctx *c, const double *a_smc, int nwave, const double *wavelength_cm, int nspectrum, int nmetal,
                            const double *specific_luminosity, int iSpectrum, double coefSpectrum, int iMetal, double coefMetal,
                            double *total_integral)
{
    if (!c) return FTTE_ERR_ARG;
    if (!a_smc || !wavelength_cm || !specific_luminosity || nwave < 2 || nspectrum < 2 || nmetal < 2)
        return fail(c, FTTE_ERR_ARG, "ftte_stellar_beta_table: bad argument");
    if (iSpectrum < 1 || iSpectrum + 1 > nspectrum || iMetal < 1 || iMetal + 1 > nmetal)
        return fail(c, FTTE_ERR_ARG, "ftte_stellar_beta_table: iSpectrum / iMetal outside the library");
    FTTE_HIP(c, hipSetDevice(c->device));
    const int rc = point_stellar_beta_table(c->point, c->stream, a_smc, nwave, wavelength_cm, nspectrum, nmetal, specific_luminosity,
                                            iSpectrum, coefSpectrum, iMetal, coefMetal, total_integral, &c->err);
    return rc;
}

int ftte_set_rate_tables(ftte_ctx *c, const double *tables)
{
    if (!c) return FTTE_ERR_ARG;
    if (!tables) return fail(c, FTTE_ERR_ARG, "ftte_set_rate_tables: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_set_tables(c->point, c->stream, tables, &c->err);
}

int ftte_get_rate_tables(ftte_ctx *c, double *tables)
{
    if (!c) return FTTE_ERR_ARG;
    if (!tables) return fail(c, FTTE_ERR_ARG, "ftte_get_rate_tables: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_get_tables(c->point, c->stream, tables, &c->err);
}

int ftte_get_rates_hydrogen_helium(ftte_ctx *c, int dust_approximation, int nsample, const double *tau, double *rates)
{
    if (!c) return FTTE_ERR_ARG;
    if (nsample < 0 || (nsample && (!tau || !rates)) || dust_approximation < 0 || dust_approximation > 2)
        return fail(c, FTTE_ERR_ARG, "ftte_get_rates_hydrogen_helium: bad argument");
    if (!nsample) return FTTE_OK;
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_lookup(c->point, c->stream, dust_approximation, nsample, tau, rates, &c->err);
}

static int set_medium(ftte_ctx *c, const double *HI, const double *HeI, const double *HeII, const double *rho, const double *abun2,
                      int dust, bool on_device, const char *who)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!HI || !HeI || !HeII || dust < 0 || dust > 2) return fail(c, FTTE_ERR_ARG, std::string(who) + ": bad argument");
    if ((dust >= 1 && !abun2) || (dust == 2 && !rho)) return fail(c, FTTE_ERR_ARG, std::string(who) + ": this dust approximation needs abun2 (and rho)");
    FTTE_HIP(c, hipSetDevice(c->device));
    const double *const field[5] = {HI, HeI, HeII, rho, abun2};
    return point_set_medium(c->point, c->stream, c->ncell, field, on_device, dust, &c->err);
}

int ftte_set_medium(ftte_ctx *c, const double *HI, const double *HeI, const double *HeII, const double *rho, const double *abun2,
                    int dust_approximation)
{
    return set_medium(c, HI, HeI, HeII, rho, abun2, dust_approximation, false, "ftte_set_medium");
}

int ftte_set_medium_device(ftte_ctx *c, const double *HI, const double *HeI, const double *HeII, const double *rho,
                           const double *abun2, int dust_approximation)
{
    return set_medium(c, HI, HeI, HeII, rho, abun2, dust_approximation, true, "ftte_set_medium_device");
}

int ftte_set_zero_rates(ftte_ctx *c)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = point_zero_rates(c->point, c->stream, c->ncell, &c->err))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

int ftte_locate_cell(ftte_ctx *c, int level, const int32_t *position, int64_t *cell)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (level < 0 || !position || !cell) return fail(c, FTTE_ERR_ARG, "ftte_locate_cell: bad argument");
    const int n = c->n;
    for (int q = 0; q < 3; ++q)
        if (position[q] < 1 || position[q] > n) return fail(c, FTTE_ERR_ARG, "ftte_locate_cell: base index outside 1..n");
    int32_t node = ((position[0] - 1) * n + (position[1] - 1)) * n + (position[2] - 1);
    for (int l = 0; l < level; ++l) {
        // localizeCellFromStar, equiSources.f90:2597-2620
        if (c->tree.child0[node] < 0) return fail(c, FTTE_ERR_LEVELS, "error in star particle position: cell not refined");
        const int32_t *p = position + 3 * l + 3;
        for (int q = 0; q < 3; ++q)
            if (p[q] < 1 || p[q] > 2) return fail(c, FTTE_ERR_ARG, "ftte_locate_cell: child index outside 1..2");
        node = c->tree.child0[node] + 4 * (p[0] - 1) + 2 * (p[1] - 1) + (p[2] - 1);
    }
    if (c->tree.leaf[node] < 0) return fail(c, FTTE_ERR_LEVELS, "ftte_locate_cell: the call sequence ends on a refined cell");
    *cell = c->tree.leaf[node];
    return FTTE_OK;
}

int ftte_point_sources(ftte_ctx *c, int nsrc, const int64_t *src_cell, const double *src_ndot, int *highest_pixel_level)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nsrc < 0 || (nsrc && (!src_cell || !src_ndot))) return fail(c, FTTE_ERR_ARG, "ftte_point_sources: bad argument");
    if (highest_pixel_level) *highest_pixel_level = 0;
    if (!nsrc) return FTTE_OK;
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_trace(c->point, c->stream, c->tree, c->box, nsrc, src_cell, src_ndot, highest_pixel_level, &c->err);
}

int ftte_point_escape(ftte_ctx *c, int nsrc, double *remaining, double *boundary, double *dust, double *spectrum, double *fraction)
{
    if (!c) return FTTE_ERR_ARG;
    const PointState &P = c->point;
    if (nsrc < 0 || (size_t)nsrc * kEscapeRec != P.escape_host.size())
        return fail(c, FTTE_ERR_ARG, "ftte_point_escape: nsrc is not the number of stars of the last ftte_point_sources");
    for (int s = 0; s < nsrc; ++s) {
        const double *E = P.escape_host.data() + (size_t)s * kEscapeRec;
        for (int ir = 0; ir < kOutputRadii; ++ir) {
            if (remaining) remaining[s * kOutputRadii + ir] = E[ir];
            if (boundary) boundary[s * kOutputRadii + ir] = E[kOutputRadii + ir];
            // equiSources.f90:1342-1348
            if (fraction) fraction[s * kOutputRadii + ir] = E[kOutputRadii + ir] < 1. ? E[ir] / (P.escape_ndot[(size_t)s] - E[kOutputRadii + ir]) : 0.;
        }
        if (dust) dust[s] = E[2 * kOutputRadii];
        if (spectrum) std::memcpy(spectrum + (size_t)s * kOutputEnergies, E + 2 * kOutputRadii + 1, sizeof(double) * kOutputEnergies);
    }
    return FTTE_OK;
}

int ftte_set_output_sigma(ftte_ctx *c, const double *sigma)
{
    if (!c) return FTTE_ERR_ARG;
    if (!sigma) return fail(c, FTTE_ERR_ARG, "ftte_set_output_sigma: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_set_output_sigma(c->point, c->stream, sigma, &c->err);
}

int ftte_get_point_rates(ftte_ctx *c, double *rates)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!rates) return fail(c, FTTE_ERR_ARG, "ftte_get_point_rates: bad argument");
    if (!c->point.rates || c->point.rates_cells != c->ncell) return fail(c, FTTE_ERR_STATE, "no rates: call ftte_set_zero_rates / ftte_point_sources first");
    FTTE_HIP(c, hipSetDevice(c->device));
    double *planes = nullptr;
    if ((rc = point_rate_planes(c->point, c->stream, &planes, &c->err))) return rc;
    FTTE_HIP(c, hipMemcpyAsync(rates, planes, sizeof(double) * 6 * c->ncell, hipMemcpyDeviceToHost, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

int ftte_set_point_rates(ftte_ctx *c, const double *rates)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!rates) return fail(c, FTTE_ERR_ARG, "ftte_set_point_rates: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_set_rates(c->point, c->stream, c->ncell, rates, &c->err);
}

int ftte_point_rates_device(ftte_ctx *c, double **rates_dev)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!rates_dev) return fail(c, FTTE_ERR_ARG, "ftte_point_rates_device: bad argument");
    if (!c->point.rates || c->point.rates_cells != c->ncell) return fail(c, FTTE_ERR_STATE, "no rates: call ftte_set_zero_rates / ftte_point_sources first");
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = point_rate_planes(c->point, c->stream, rates_dev, &c->err))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

// ---- ionisation equilibrium ---------------------------------------------------------------------------------------

int ftte_set_rate_coefficients(ftte_ctx *c, int nratec, double logtem0, double logtem9, double dlogtem, const double *k1a,
                               const double *k2a, const double *k3a, const double *k4a, const double *k5a, const double *k6a)
{
    if (!c) return FTTE_ERR_ARG;
    if (nratec < 2 || !(dlogtem > 0.0) || !(logtem9 > logtem0) || !k1a || !k2a || !k3a || !k4a || !k5a || !k6a)
        return fail(c, FTTE_ERR_ARG, "ftte_set_rate_coefficients: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if (c->chem_k && c->chem_nratec != nratec) { FTTE_HIP(c, hipFree(c->chem_k)); c->chem_k = nullptr; }
    if (!c->chem_k) FTTE_HIP(c, hipMalloc((void **)&c->chem_k, sizeof(double) * 6 * (size_t)nratec));
    const double *src[6] = {k1a, k2a, k3a, k4a, k5a, k6a};
    for (int r = 0; r < 6; ++r)
        FTTE_HIP(c, hipMemcpyAsync(c->chem_k + (size_t)r * nratec, src[r], sizeof(double) * nratec, hipMemcpyHostToDevice, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    c->chem_nratec = nratec;
    c->chem_logtem0 = logtem0; c->chem_logtem9 = logtem9; c->chem_dlogtem = dlogtem;
    return FTTE_OK;
}

int ftte_set_temperature(ftte_ctx *c, const double *tgas)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!tgas) return fail(c, FTTE_ERR_ARG, "ftte_set_temperature: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    // the logarithm is taken here, on the host, so that the device update consists of IEEE-exact operations only
    std::vector<double> logtem((size_t)c->ncell);
    {
        const int nthreads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const int64_t chunk = (c->ncell + nthreads - 1) / nthreads;
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&, t] {
                const int64_t lo = t * chunk, hi = std::min<int64_t>(c->ncell, lo + chunk);
                for (int64_t q = lo; q < hi; ++q) logtem[(size_t)q] = std::log(tgas[q]);
            });
        for (auto &th : pool) th.join();
    }
    if (!c->chem_logtem) FTTE_HIP(c, hipMalloc((void **)&c->chem_logtem, sizeof(double) * (size_t)c->ncell));
    FTTE_HIP(c, hipMemcpyAsync(c->chem_logtem, logtem.data(), sizeof(double) * (size_t)c->ncell, hipMemcpyHostToDevice, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    c->chem_temperature_set = true;
    return FTTE_OK;
}

static int solve_rates(ftte_ctx *c, int run_uvb, const double *J, bool J_on_device, const double *ksi, const double *uniform,
                       double threshold, int use_point_rates, double *max_change, const char *who)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    PointState &P = c->point;
    if (!c->chem_k) return fail(c, FTTE_ERR_STATE, std::string(who) + ": no rate coefficients (ftte_set_rate_coefficients)");
    if (!c->chem_temperature_set) return fail(c, FTTE_ERR_STATE, std::string(who) + ": no temperature (ftte_set_temperature)");
    if (!P.medium_ready || P.medium_cells != c->ncell || !P.rho_given)
        return fail(c, FTTE_ERR_STATE, std::string(who) + ": no medium with density (ftte_set_medium with rho)");
    if (run_uvb && (!J || !ksi)) return fail(c, FTTE_ERR_ARG, std::string(who) + ": the transfer-driven update needs J and ksi");
    if (!run_uvb && !uniform) return fail(c, FTTE_ERR_ARG, std::string(who) + ": the uniform-background update needs the background rates");
    if (use_point_rates && (!P.rates || P.rates_cells != c->ncell))
        return fail(c, FTTE_ERR_STATE, std::string(who) + ": no point-source rates (ftte_set_zero_rates / ftte_point_sources)");
    FTTE_HIP(c, hipSetDevice(c->device));
    const size_t nc = (size_t)c->ncell;
    if (!c->chem_level) {
        FTTE_HIP(c, hipMalloc((void **)&c->chem_level, nc));
        FTTE_HIP(c, hipMemcpyAsync(c->chem_level, c->leaf_level.data(), nc, hipMemcpyHostToDevice, c->stream));
    }
    if (!c->chem_out) FTTE_HIP(c, hipMalloc((void **)&c->chem_out, sizeof(double) * 3 * nc));
    if (!c->chem_counters) FTTE_HIP(c, hipMalloc((void **)&c->chem_counters, sizeof(unsigned long long) * 4));
    const double *J_dev = nullptr;
    if (run_uvb) {
        if (J_on_device) J_dev = J;
        else {
            if (!c->chem_J) FTTE_HIP(c, hipMalloc((void **)&c->chem_J, sizeof(double) * 3 * nc));
            FTTE_HIP(c, hipMemcpyAsync(c->chem_J, J, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, c->stream));
            J_dev = c->chem_J;
        }
    }
    const unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull};
    FTTE_HIP(c, hipMemcpyAsync(c->chem_counters, init, sizeof init, hipMemcpyHostToDevice, c->stream));

    ChemRec R;
    std::memset(&R, 0, sizeof R);
    R.level = c->chem_level;
    R.rho = P.medium[3]; R.logtem = c->chem_logtem;
    R.HI = P.medium[0]; R.HeI = P.medium[1]; R.HeII = P.medium[2];
    R.HI_out = c->chem_out; R.HeI_out = c->chem_out + nc; R.HeII_out = c->chem_out + 2 * nc;
    R.krate = use_point_rates ? P.rates : nullptr;
    R.J = J_dev;
    R.k = c->chem_k;
    R.ncell = c->ncell; R.n = c->n; R.nratec = c->chem_nratec; R.run_uvb = run_uvb ? 1 : 0;
    R.box = c->box; R.logtem0 = c->chem_logtem0; R.logtem9 = c->chem_logtem9; R.dlogtem = c->chem_dlogtem;
    if (ksi) std::memcpy(R.ksi, ksi, sizeof R.ksi);
    if (uniform) std::memcpy(R.uniform, uniform, sizeof R.uniform);
    R.threshold = threshold;
    R.first_bad = c->chem_counters; R.max_change = c->chem_counters + 1; R.steps = c->chem_counters + 2;
    if (launch_rate_equations(R, c->stream)) return fail(c, FTTE_ERR_NO_DEVICE, std::string(who) + ": kernel launch failed");
    unsigned long long out[4];
    FTTE_HIP(c, hipMemcpyAsync(out, c->chem_counters, sizeof out, hipMemcpyDeviceToHost, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if (out[0] != ~0ull) {
        // the reference prints the species of the cell and stops (equiSources.f90:3637-3654); the state is left as it was
        return fail(c, FTTE_ERR_RATES, std::string(who) + ": species fraction outside [0, 1] in cell " + std::to_string(out[0]) +
                                           " (0-based cell-array index)");
    }
    for (int f = 0; f < 3; ++f)
        FTTE_HIP(c, hipMemcpyAsync(P.medium[f], c->chem_out + f * nc, sizeof(double) * nc, hipMemcpyDeviceToDevice, c->stream));
    P.packed_ready = false; // the tracer's packed copy of the medium is stale now
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    double change;
    std::memcpy(&change, &out[1], sizeof change);
    if (max_change) *max_change = change;
    c->chem_steps = (long long)out[2];
    return FTTE_OK;
}

int ftte_solve_rate_equations(ftte_ctx *c, int run_uvb_transfer, const double *J, const double *ksi, const double *uniform,
                              double self_shielding_threshold, int use_point_rates, double *max_change)
{
    return solve_rates(c, run_uvb_transfer, J, false, ksi, uniform, self_shielding_threshold, use_point_rates, max_change,
                       "ftte_solve_rate_equations");
}

int ftte_solve_rate_equations_device(ftte_ctx *c, int run_uvb_transfer, const double *J_dev, const double *ksi, const double *uniform,
                                     double self_shielding_threshold, int use_point_rates, double *max_change)
{
    return solve_rates(c, run_uvb_transfer, J_dev, true, ksi, uniform, self_shielding_threshold, use_point_rates, max_change,
                       "ftte_solve_rate_equations_device");
}

int ftte_get_medium(ftte_ctx *c, double *HI, double *HeI, double *HeII)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!HI || !HeI || !HeII) return fail(c, FTTE_ERR_ARG, "ftte_get_medium: bad argument");
    if (!c->point.medium_ready || c->point.medium_cells != c->ncell) return fail(c, FTTE_ERR_STATE, "no medium: call ftte_set_medium first");
    FTTE_HIP(c, hipSetDevice(c->device));
    double *dst[3] = {HI, HeI, HeII};
    for (int f = 0; f < 3; ++f)
        FTTE_HIP(c, hipMemcpyAsync(dst[f], c->point.medium[f], sizeof(double) * (size_t)c->ncell, hipMemcpyDeviceToHost, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    return FTTE_OK;
}

int ftte_compute_opacities(ftte_ctx *c, int nnu, const double *beta)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !beta) return fail(c, FTTE_ERR_ARG, "ftte_compute_opacities: bad argument");
    if (!c->point.medium_ready || c->point.medium_cells != c->ncell) return fail(c, FTTE_ERR_STATE, "no medium: call ftte_set_medium first");
    FTTE_HIP(c, hipSetDevice(c->device));
    if ((rc = wait_sweep(c))) return rc;
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure_kappa(c, nnu))) return rc;
    double *dbeta = nullptr;
    FTTE_HIP(c, hipMalloc((void **)&dbeta, sizeof(double) * 3 * (size_t)nnu));
    hipError_t e = hipMemcpyAsync(dbeta, beta, sizeof(double) * 3 * nnu, hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess)
        lrc = launch_opacity(c->point.medium[0], c->point.medium[1], c->point.medium[2], dbeta, c->kappa[0], (long)c->ncell, nnu, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dbeta);
    if (e != hipSuccess) return fail(c, FTTE_ERR_NO_DEVICE, std::string("ftte_compute_opacities: ") + hipGetErrorString(e));
    if (lrc) return fail(c, FTTE_ERR_NO_DEVICE, "ftte_compute_opacities: kernel launch failed");
    c->nnu = nnu;
    c->kappa_ready[0] = true; c->kappa_ready[1] = c->kappa_ready[2] = c->kappa_ready[3] = false;
    return FTTE_OK;
}

static int assign_uvb(ftte_ctx *c, int nnu, const double *uvb, double threshold, double *J, bool J_on_device, const char *who)
{
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !uvb || !J) return fail(c, FTTE_ERR_ARG, std::string(who) + ": bad argument");
    PointState &P = c->point;
    if (!P.medium_ready || P.medium_cells != c->ncell || !P.rho_given)
        return fail(c, FTTE_ERR_STATE, std::string(who) + ": no medium with density (ftte_set_medium with rho)");
    FTTE_HIP(c, hipSetDevice(c->device));
    const size_t nc = (size_t)c->ncell;
    double *duvb = nullptr, *dJ = J_on_device ? J : nullptr;
    FTTE_HIP(c, hipMalloc((void **)&duvb, sizeof(double) * nnu));
    hipError_t e = hipSuccess;
    if (!J_on_device) e = hipMalloc((void **)&dJ, sizeof(double) * nc * nnu);
    if (e == hipSuccess) e = hipMemcpyAsync(duvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess) lrc = launch_thin_limit(P.medium[0], P.medium[1], P.medium[2], P.medium[3], duvb, threshold, dJ, (long)nc, nnu, c->stream);
    if (e == hipSuccess && !J_on_device) e = hipMemcpyAsync(J, dJ, sizeof(double) * nc * nnu, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(duvb);
    if (!J_on_device && dJ) (void)hipFree(dJ);
    if (e != hipSuccess) return fail(c, FTTE_ERR_NO_DEVICE, std::string(who) + ": " + hipGetErrorString(e));
    if (lrc) return fail(c, FTTE_ERR_NO_DEVICE, std::string(who) + ": kernel launch failed");
    return FTTE_OK;
}

int ftte_assign_uvb_radiation(ftte_ctx *c, int nnu, const double *uvb, double self_shielding_threshold, double *J)
{
    return assign_uvb(c, nnu, uvb, self_shielding_threshold, J, false, "ftte_assign_uvb_radiation");
}

int ftte_assign_uvb_radiation_device(ftte_ctx *c, int nnu, const double *uvb, double self_shielding_threshold, double *J_dev)
{
    return assign_uvb(c, nnu, uvb, self_shielding_threshold, J_dev, true, "ftte_assign_uvb_radiation_device");
}

long long ftte_rate_equation_steps(const ftte_ctx *c) { return c ? c->chem_steps : 0; }

long long ftte_point_ray_steps(const ftte_ctx *c) { return c ? c->point.ray_steps : 0; }

int ftte_rmax(double *rmax30)
{
    if (!rmax30) return FTTE_ERR_ARG;
    rmax_table(rmax30);
    return FTTE_OK;
}

int ftte_uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma)
{
    if (nfreq < 2 || !(freqdel > 0.0) || !alpha || !beta || !ksi || !gamma) return FTTE_ERR_ARG;
    uvb_beta_table(nfreq, freqdel, alpha, beta, ksi, gamma);
    return FTTE_OK;
}

int ftte_coll_rates(double T, int recombination_type, double *k)
{
    if (!(T > 0.0) || (recombination_type != 1 && recombination_type != 2) || !k) return FTTE_ERR_ARG;
    coll_rates(T, recombination_type, k);
    return FTTE_OK;
}

int ftte_rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                                 double *logtem9, double *dlogtem)
{
    if (nratec < 2 || !(temstart > 0.0) || !(temend > temstart) || (recombination_type != 1 && recombination_type != 2) || !k || !logtem0 ||
        !logtem9 || !dlogtem)
        return FTTE_ERR_ARG;
    rate_coefficient_tables(nratec, temstart, temend, recombination_type, k, logtem0, logtem9, dlogtem);
    return FTTE_OK;
}

int ftte_uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma)
{
    if (nfreq < 2 || !(freqdel > 0.0) || !ksi || !gamma) return FTTE_ERR_ARG;
    uniform_table(nfreq, freqdel, alpha_quasar, alpha_stellar, ksi, gamma);
    return FTTE_OK;
}

double ftte_dust_cross_section(double lambda_micron, const double *a_smc)
{
    return dust_cross_section(lambda_micron, a_smc);
}

} // extern "C"
