// ftte_api.cpp -- the C ABI of include/ftte.h: context life cycle, grid and field setters, options, the sweep entry points, point
// sources, equilibrium, host helpers.  The work behind them: ftte_plan.cpp (planners), ftte_sweeps.cpp (launch sequences),
// ftte_hybrid.cpp (refined cell arrays), ftte_host_arrays.cpp (PCIe), ftte_point.cpp, ftte_amr.cpp, ftte_ingest.cpp.
//
// There is no CPU fallback: every entry point that computes on the grid needs a HIP device and fails with FTTE_ERR_NO_DEVICE
// otherwise.
#include "ftte_context.h"

using namespace ftte;

namespace ftte {

std::string g_create_error;

int fail(ftte_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

int fold_status(int rc)
{
    return rc == 1 ? FTTE_ERR_PHI : rc == 2 ? FTTE_ERR_THETA : FTTE_ERR_DOMINANT_AXIS;
}

} // namespace ftte


// =================================================================================================
extern "C" {

int ftte_create(ftte_ctx **out, int ndev, const int *dev_ids)
{
    if (!out) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: ctx is NULL");
    *out = nullptr;
    if (ndev > 1) return multi_create(out, ndev, dev_ids); // one single-device context per device behind this one (ftte_multi.cpp)
    if (ndev != 1) return fail(nullptr, FTTE_ERR_ARG, "ftte_create: ndev must be at least 1");
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
    if (c->multi) return multi_destroy(c);
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < 3; ++l) {
        if (c->kappa[l]) (void)hipFree(c->kappa[l]);
        if (c->kappa_tiled[l]) (void)hipFree(c->kappa_tiled[l]);
        if (c->emis[l]) (void)hipFree(c->emis[l]);
        for (int s = 0; s < kMaxAcc; ++s) if (c->acc[l][s]) (void)hipFree(c->acc[l][s]);
    }
    if (c->d_bdeps) (void)hipFree(c->d_bdeps);
    if (c->d_bdone) (void)hipFree(c->d_bdone);
    if (c->d_bqueue) (void)hipFree(c->d_bqueue);
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
    for (int l = 0; l < 3; ++l) if (c->base_emis[l]) (void)hipFree(c->base_emis[l]);
    for (int l = 0; l < 3; ++l) {
        if (c->fine_kappa[l]) (void)hipFree(c->fine_kappa[l]);
        if (c->fine_emis[l]) (void)hipFree(c->fine_emis[l]);
        for (int a = 0; a < kMaxAcc; ++a) if (c->fine_acc[l][a]) (void)hipFree(c->fine_acc[l][a]);
    }
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

const char *ftte_multi_info(const ftte_ctx *c) { return (c && c->multi) ? multi_how(c) : ""; }

int ftte_set_grid(ftte_ctx *c, int nx, int ny, int nz, int64_t ncell, const int32_t *level, double box_cm)
{
    if (!c) return FTTE_ERR_ARG;
    if (nx < 1 || !level || ncell < 1 || !(box_cm > 0.0)) return fail(c, FTTE_ERR_ARG, "ftte_set_grid: bad argument");
    if (nx != ny || nx != nz) return fail(c, FTTE_ERR_NOT_CUBIC, "base grid needs to be of size n^3");
    if (nx > 32000) return fail(c, FTTE_ERR_UNSUPPORTED, "ftte_set_grid: n > 32000");
    if (c->multi) return multi_set_grid(c, nx, ny, nz, ncell, level, box_cm);
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
            if (c->kappa_tiled[l]) { (void)hipFree(c->kappa_tiled[l]); c->kappa_tiled[l] = nullptr; }
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
    ++c->n_kappa_sets;
    c->plan.valid = false;
    c->bplan.valid = false;
    c->tree = std::move(tree);
    c->use_forest = c->tree.refined() || c->force_forest;
    return FTTE_OK;
}

int ftte_set_opacity(ftte_ctx *c, int nnu, const double *kappa)
{
    if (c && c->multi) return multi_set_opacity(c, nnu, kappa);
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
    ++c->n_kappa_sets;
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
    // A uniform grid whose last sweep used all three layouts (the copies are there): the copy and the two transposes in one pass
    // over the caller's array.  Else the copy alone; the sweep makes what it needs.
    const bool all_three = !c->use_forest && c->kappa[1] && c->kappa[2] && c->kappa_ready[1] && c->kappa_ready[2] && c->nnu == nnu && !c->tiled_opt;
    if (all_three) {
        if (launch_set_layouts(kappa_dev, c->kappa[0], c->kappa[1], c->kappa[2], c->n, nnu, (long)c->ncell, c->stream))
            return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
    } else FTTE_HIP(c, hipMemcpyAsync(c->kappa[0], kappa_dev, sizeof(double) * nnu * c->ncell, hipMemcpyDeviceToDevice, c->stream));
    FTTE_HIP(c, hipStreamSynchronize(c->stream)); // the sweep may run on another stream: the copies must have landed
    c->nnu = nnu;
    c->kappa_ready[0] = true; c->kappa_ready[1] = c->kappa_ready[2] = all_three; c->kappa_ready[3] = false;
    ++c->n_kappa_sets;
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
    ++c->n_kappa_sets;
    return FTTE_OK;
}

static int set_emission(ftte_ctx *c, int mode, const double *values, bool on_device, const char *who)
{
    if (!c) return FTTE_ERR_ARG;
    if (c->multi) return on_device ? fail(c, FTTE_ERR_UNSUPPORTED, std::string(who) + ": a multi-device context takes host arrays") : multi_set_emission(c, mode, values);
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
    if (c->multi) return multi_set_option(c, key, value);
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
    } else if (!std::strcmp(key, "hybrid_slots")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "hybrid_slots must be 0 (phases), 1 (slots where there are several passes) or 2 (slots always)");
        c->hybrid_slots = value;
    } else if (!std::strcmp(key, "graph")) {
        if (value != 0 && value != 1) return fail(c, FTTE_ERR_ARG, "graph must be 0 (every launch of the hybrid sweep issued every time) or 1 (captured once, replayed)");
        c->use_graph = value;
    } else if (!std::strcmp(key, "box_lanes")) {
        if (value < 1 || value > 64 || 64 % value) return fail(c, FTTE_ERR_ARG, "box_lanes (the boxes of the hybrid sweep end on multiples of it along a brick's 64 lanes) must divide 64");
        c->hybrid_lanes = value;
    } else if (!std::strcmp(key, "forest_batch")) {
        if (value < 0 || value > 65535) return fail(c, FTTE_ERR_ARG, "forest_batch (directions per launch of the segment forests) must be 1..65535, or 0 for the default");
        c->forest_batch = value;
    } else if (!std::strcmp(key, "pipelines")) {
        if (value < 1 || value > ftte_ctx::kMaxPipes) return fail(c, FTTE_ERR_ARG, "pipelines (independent bricks-forests-bricks sequences of the hybrid sweep, each on a stream of its own) must be 1..4");
        c->halves = value;
        c->hplan.valid = false;
    } else if (!std::strcmp(key, "fine_bricks")) {
        if (value < 0 || value > 1) return fail(c, FTTE_ERR_ARG, "fine_bricks must be 0 (a fully refined block stays in the segment forest) or 1 (bricks of its own on the fine level where the block allows it)");
        c->fine_bricks = value;
    } else if (!std::strcmp(key, "fine_chunk")) {
        if (value < 0 || value > 4096) return fail(c, FTTE_ERR_ARG, "fine_chunk (layers per brick on the fine level of a refined block) must be 1..4096, or 0 for the base bricks' chunk");
        c->fine_chunk = value;
    } else if (!std::strcmp(key, "atomic_acc")) {
        if (value < 0 || value > 1) return fail(c, FTTE_ERR_ARG, "atomic_acc must be 0 or 1");
        c->atomic_acc = value;
    } else if (!std::strcmp(key, "ablate")) {
        if (value < 0 || value > 63) return fail(c, FTTE_ERR_ARG, "ablate is a mask of 6 bits");
        c->ablate = value;
    } else if (!std::strcmp(key, "forest_fuse")) {
        if (value < 0 || value > (1 << 24)) return fail(c, FTTE_ERR_ARG, "forest_fuse must be 0 (a launch per level) .. 16777216");
        c->forest_fuse = value;
    } else if (!std::strcmp(key, "queue_mix")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "queue_mix must be 0, 1 or 2");
        c->queue_mix = value;
    } else if (!std::strcmp(key, "dataflow")) {
        if (value < 0 || value > 3) return fail(c, FTTE_ERR_ARG, "dataflow must be 0 (a launch per stage), 1 (one launch, bricks wait for each other), 2 (the same with write-through stores) or 3 (persistent workgroups, a task queue per XCD)");
        c->dataflow = value;
    } else if (!std::strcmp(key, "lanes")) {
        if (value < 1 || value > 16) return fail(c, FTTE_ERR_ARG, "lanes (streams the brick sweep spreads its frequency groups over) must be 1..16");
        c->lanes = value;
    } else if (!std::strcmp(key, "team")) {
        if (value < -1 || value > 2 || value == 1) return fail(c, FTTE_ERR_ARG, "team must be -1 (by the number of frequency groups: 2 up to four, else 0), 0 (one wavefront sweeps a group's directions in turn) or 2 (two wavefronts per brick, four rows each); 1 (a wavefront per direction) lost everywhere and is gone");
        c->team = value;
    } else if (!std::strcmp(key, "tiled")) {
        if (value < 0 || value > 2) return fail(c, FTTE_ERR_ARG, "tiled must be 0 (default), 1 (bricks: opacities and accumulators stored brick by brick where the grid is made of whole bricks: a brick's layer in one piece) or 2 (the whole brick in one piece)");
        c->tiled_opt = value;
    } else if (!std::strcmp(key, "pair_waves")) {
        if (value < 2 || value > 4) return fail(c, FTTE_ERR_ARG, "pair_waves (workgroups of two wavefronts per SIMD the pair kernel is built for) must be 2..4");
        c->pair_waves = value;
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

    if (c->use_forest) {
        if (c->hybrid && c->tree.refined() && !c->force_forest && ndir > 0) {
            bool done = false;
            if ((rc = hybrid_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream, &done)) || done) return rc;
        }
        return forest_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream);
    }
    if (c->engine != 1) return brick_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream);
    return tile_sweep(c, ndir, phi, theta, w, uvb, J_dev, stream);
}

int ftte_diffuse_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb,
                       double *J)
{
    if (c && c->multi) return multi_sweep(c, ndir, phi, theta, w, uvb, J);
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
    if (c && c->multi) return multi_iteration(c, nnu, kappa, ndir, phi, theta, w, uvb, J);
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (nnu < 1 || !kappa || !J || ndir < 0 || (ndir > 0 && (!phi || !theta || !w)) || !uvb)
        return fail(c, FTTE_ERR_ARG, "ftte_diffuse_iteration: bad argument");
    const bool lanes_apply = !c->use_forest && c->engine != 1 && !c->emit_mode && brick_form(c, nnu) != 1 && !c->dataflow && ndir > 0 && c->lanes >= 2 &&
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
    ++c->n_kappa_sets;
    const size_t elems = (size_t)nnu * c->ncell;
    if ((rc = ensure(c, &c->host_J_dev, &c->host_J_cap, elems))) return rc;
    const HostPipe pipe{kappa, J};
    if ((rc = brick_sweep(c, ndir, phi, theta, w, uvb, c->host_J_dev, c->stream, &pipe))) {
        c->kappa_ready[0] = c->kappa_ready[1] = c->kappa_ready[2] = false;
        ++c->n_kappa_sets;
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
    if (c->multi) return multi_host_register(c, ptr, bytes, true);
    if (is_registered(c, ptr, bytes)) return FTTE_OK;
    FTTE_HIP(c, hipSetDevice(c->device));
    FTTE_HIP(c, hipHostRegister(ptr, bytes, hipHostRegisterPortable)); // (portable: pinned for every device of the process)
    c->registered.push_back({(const char *)ptr, bytes});
    return FTTE_OK;
}

int ftte_host_unregister(ftte_ctx *c, void *ptr)
{
    if (!c) return FTTE_ERR_ARG;
    if (c->multi) return multi_host_register(c, ptr, 0, false);
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
    if (c->multi) return multi_counter(c, name);
    if (!std::strcmp(name, "devices")) return 1;
    if (!std::strcmp(name, "grid_builds")) return c->n_grid_builds;
    if (!std::strcmp(name, "plan_builds")) return c->n_plan_builds;
    if (!std::strcmp(name, "forest_builds")) return c->n_forest_builds;
    if (!std::strcmp(name, "hybrid_boxes")) return (c->hplan.valid && c->hplan.worthwhile) ? c->hplan.most_boxes : 0;
    if (!std::strcmp(name, "hybrid_passes")) return (c->hplan.valid && c->hplan.worthwhile) ? c->hplan.npass : 0;
    if (!std::strcmp(name, "fine_block")) return (c->hplan.valid && c->hplan.worthwhile && c->hplan.fine.active) ? c->hplan.fine.n : 0;
    if (!std::strcmp(name, "brick_form")) return c->last_brick_form;
    if (!std::strcmp(name, "brick_groups")) return c->bplan.valid ? (long long)c->bplan.groups.size() : 0;
    if (!std::strcmp(name, "brick_accumulators")) return c->bplan.valid ? c->bplan.nacc[0] + c->bplan.nacc[1] + c->bplan.nacc[2] : 0;
    if (!std::strcmp(name, "brick_accumulators_0")) return c->bplan.valid ? c->bplan.nacc[0] : 0;
    if (!std::strcmp(name, "brick_accumulators_1")) return c->bplan.valid ? c->bplan.nacc[1] : 0;
    if (!std::strcmp(name, "brick_accumulators_2")) return c->bplan.valid ? c->bplan.nacc[2] : 0;
    return -1;
}

int ftte_launch_count(const ftte_ctx *c) { return !c ? 0 : c->multi ? multi_first(c)->timing_used : c->timing_used; }

int ftte_launch_info(ftte_ctx *c, int idx, double *ms, int64_t *updates)
{
    if (c && c->multi) return ftte_launch_info(multi_first(c), idx, ms, updates); // (the first device's share of the sweep)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
    if (!c) return FTTE_ERR_ARG;
    if (!tables) return fail(c, FTTE_ERR_ARG, "ftte_set_rate_tables: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_set_tables(c->point, c->stream, tables, &c->err);
}

int ftte_get_rate_tables(ftte_ctx *c, double *tables)
{
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
    if (!c) return FTTE_ERR_ARG;
    if (!tables) return fail(c, FTTE_ERR_ARG, "ftte_get_rate_tables: bad argument");
    FTTE_HIP(c, hipSetDevice(c->device));
    return point_get_tables(c->point, c->stream, tables, &c->err);
}

int ftte_get_rates_hydrogen_helium(ftte_ctx *c, int dust_approximation, int nsample, const double *tau, double *rates)
{
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
    ++c->n_kappa_sets;
    return FTTE_OK;
}

static int assign_uvb(ftte_ctx *c, int nnu, const double *uvb, double threshold, double *J, bool J_on_device, const char *who)
{
    if (c && c->multi) return check_ready(c, false); // (refuses: a multi-device context routes the diffuse sweep only)
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
