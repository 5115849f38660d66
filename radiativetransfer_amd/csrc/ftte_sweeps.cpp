// ftte_sweeps.cpp -- the launch sequences of the diffuse sweep: per-direction segment forests on refined cell arrays
// (forest_sweep and its pieces, also used by the hybrid sweep) and cell-fixed bricks on uniform grids (brick_sweep).
#include <cstdlib>

#include "ftte_context.h"

namespace ftte {

int ensure_kappa(ftte_ctx *c, int nnu)
{
    const size_t need = (size_t)nnu * c->ncell;
    if (c->kappa[0] && c->kappa_cap >= need) return FTTE_OK;
    for (int l = 0; l < 3; ++l) {
        if (c->kappa[l]) { FTTE_HIP(c, hipFree(c->kappa[l])); c->kappa[l] = nullptr; }
        if (c->kappa_tiled[l]) { FTTE_HIP(c, hipFree(c->kappa_tiled[l])); c->kappa_tiled[l] = nullptr; }
        if (c->emis[l]) { FTTE_HIP(c, hipFree(c->emis[l])); c->emis[l] = nullptr; }
    }
    c->emit_mode = 0; // sized by the old number of groups: has to be set again
    FTTE_HIP(c, hipMalloc((void **)&c->kappa[0], need * sizeof(double)));
    c->kappa_cap = need;
    return FTTE_OK;
}

int check_ready(ftte_ctx *c, bool need_kappa)
{
    if (c && c->multi)
        return fail(c, FTTE_ERR_UNSUPPORTED, "a multi-device context (ftte_create with ndev > 1) takes host arrays through ftte_set_grid, ftte_set_opacity, "
                                             "ftte_set_emissivity / ftte_set_source_function, ftte_diffuse_sweep and ftte_diffuse_iteration; for "
                                             "everything else create a context per device");
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
        if (c->h_berror && c->h_berror[32 * kBrickQueues]) {
            c->h_berror[32 * kBrickQueues] = 0;
            std::memset(c->bqlen, 0, sizeof c->bqlen);
            return fail(c, FTTE_ERR_STALLED, "the previous sweep gave up: a brick waited for the bricks it depends on while nothing moved (its J is not valid)");
        }
        if (c->h_berror && c->bqlen[0] && std::getenv("FTTE_QUEUE_STATS")) { // instrumentation of the persistent form, per queue
            unsigned long long began = 0;
            std::memcpy(&began, c->h_berror + 32 * kBrickQueues + 2, 8);
            began = ~began;
            for (int q = 0; q < kBrickQueues; ++q) {
                unsigned long long fin = 0, waited = 0;
                std::memcpy(&fin, c->h_berror + 32 * q + 2, 8);
                std::memcpy(&waited, c->h_berror + 32 * q + 4, 8);
                std::fprintf(stderr, "[ftte] queue %d: %u tasks, %u workgroups, drained after %.3f ms, %.1f polls per task, load %lld updates\n", q, c->bqlen[q],
                             c->h_berror[32 * q + 6], (double)(fin - began) * 1e-5, c->bqlen[q] ? (double)waited / c->bqlen[q] : 0.0, (long long)c->bplan.qload[q]);
            }
        }
        for (int q = 0; q < kBrickQueues; ++q) {
            const uint32_t want = c->bqlen[q];
            c->bqlen[q] = 0;
            if (want && c->h_berror && c->h_berror[32 * q] < want)
                return fail(c, FTTE_ERR_STALLED, "the previous sweep left a task queue undrained: no workgroup ran on that queue's XCD (its J is not valid)");
        }
    }
    return FTTE_OK;
}

// The XCC ids this device's workgroups report (HW_REG_XCC_ID), numbered 0 .. count - 1 in ascending order: the persistent form of
// the brick sweep keeps a task queue per XCD.  Once per context.
int xcc_census(ftte_ctx *c)
{
    if (c->xcc_count >= 0) return FTTE_OK;
    unsigned *mask_dev = nullptr, mask = 0;
    FTTE_HIP(c, hipMalloc((void **)&mask_dev, sizeof(unsigned)));
    FTTE_HIP(c, hipMemset(mask_dev, 0, sizeof(unsigned)));
    if (launch_xcc_census(mask_dev, c->stream)) { (void)hipFree(mask_dev); return fail(c, FTTE_ERR_NO_DEVICE, "census kernel launch failed"); }
    FTTE_HIP(c, hipStreamSynchronize(c->stream));
    FTTE_HIP(c, hipMemcpy(&mask, mask_dev, sizeof(unsigned), hipMemcpyDeviceToHost));
    FTTE_HIP(c, hipFree(mask_dev));
    c->xcc_count = 0;
    for (int x = 0; x < 16; ++x) c->xcc_queue[x] = (mask >> x) & 1u ? (int8_t)c->xcc_count++ : (int8_t)-1;
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

// Tables of several independent runs (`sets`: direction lists that may run side by side on different streams, set q using the
// scratch slots from slot0[q] on), `batch` directions at a time each; built and uploaded in one go on `stream`.  A direction's
// forest may come in passes (hybrid sweep with several boxes): every batch gets per-depth tables pass by pass, and per pass the
// range of the rays that leave its boxes.
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
            ForestRun::Batch B;
            B.d0 = d0; B.nb = nb;
            size_t npass = 1;
            for (int t = 0; t < nb; ++t) {
                const ForestDirHost &D = dirs[(size_t)(d0 + t)];
                AmrDirRec rec;
                std::memset(&rec, 0, sizeof rec);
                rec.rec = D.rec; rec.active = D.active; rec.w = D.w;
                rec.Iout = c->amr_Iout + per_dir * (size_t)(slot0[q] + t);
                rec.mean = c->amr_mean + per_dir * (size_t)(slot0[q] + t);
                rec.faces = D.faces; rec.exports = D.exports; rec.nexports = D.nexports;
                rec.imports = D.imports; rec.nimports = D.nimports;
                recs.push_back(rec);
                if (D.pass_first) npass = std::max(npass, D.pass_first->size() - 1);
            }
            for (size_t p = 0; p < npass; ++p) {
                ForestRun::Pass P;
                // depth d of pass p is entry first(p) + d of the direction's depth_off, while that lies inside the pass
                auto range = [&](const ForestDirHost &D, size_t *first, size_t *count) {
                    const size_t all = D.depth_off->size() - 1;
                    if (!D.pass_first) { *first = 0; *count = p == 0 ? all : 0; return; }
                    if (p + 1 >= D.pass_first->size()) { *first = all; *count = 0; return; }
                    *first = (size_t)(*D.pass_first)[p]; *count = (size_t)((*D.pass_first)[p + 1] - (*D.pass_first)[p]);
                };
                for (int t = 0; t < nb; ++t) {
                    size_t first, count;
                    range(dirs[(size_t)(d0 + t)], &first, &count);
                    P.maxdepth = std::max(P.maxdepth, count);
                }
                P.table_at = tables.size();
                P.most_at = R.most_of.size();
                for (size_t depth = 0; depth < P.maxdepth; ++depth) {
                    int64_t most = 0;
                    const size_t at = tables.size();
                    tables.resize(at + 2 * (size_t)nb, 0);
                    for (int t = 0; t < nb; ++t) {
                        const ForestDirHost &D = dirs[(size_t)(d0 + t)];
                        size_t first, count;
                        range(D, &first, &count);
                        if (depth < count) {
                            const std::vector<int64_t> &off = *D.depth_off;
                            tables[at + (size_t)t] = off[first + depth + 1] - off[first + depth];
                            tables[at + (size_t)nb + (size_t)t] = off[first + depth];
                            most = std::max(most, off[first + depth + 1] - off[first + depth]);
                        }
                    }
                    R.most_of.push_back(most);
                }
                // the rays that leave this pass's boxes: count[], first[] per direction
                P.export_at = tables.size();
                tables.resize(P.export_at + 2 * (size_t)nb, 0);
                for (int t = 0; t < nb; ++t) {
                    const ForestDirHost &D = dirs[(size_t)(d0 + t)];
                    int64_t first = 0, count = p == 0 ? D.nexports : 0;
                    if (D.export_first) {
                        first = p + 1 < D.export_first->size() ? (*D.export_first)[p] : D.nexports;
                        count = p + 1 < D.export_first->size() ? (*D.export_first)[p + 1] - first : 0;
                    }
                    tables[P.export_at + (size_t)t] = count;
                    tables[P.export_at + (size_t)nb + (size_t)t] = first;
                    P.most_exports = std::max(P.most_exports, count);
                }
                B.passes.push_back(P);
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

// One pass of one batch of a prepared run on `stream`: depth after depth (one launch per depth for the whole batch), then the rays
// that leave the pass's boxes (hybrid sweep) into the bricks' face buffers.
int launch_forest_pass(ftte_ctx *c, hipStream_t stream, const ForestRun &R, size_t b, size_t p, AmrLevelRec A)
{
    const ForestRun::Batch &B = R.batches[b];
    if (p >= B.passes.size()) return FTTE_OK;
    const ForestRun::Pass &P = B.passes[p];
    A.dir = c->d_amr_dirs + R.dir_at + (size_t)B.d0;
    A.ndir = B.nb;
    // Runs of thin levels (option "forest_fuse": at most that many (segment, frequency group) pairs in the fullest direction) go in
    // one launch, a workgroup per direction and a barrier per level; a thick level gets a launch of its own, a thread per pair.
    const int64_t thin = (int64_t)c->forest_fuse;
    for (size_t depth = 0; depth < P.maxdepth;) {
        size_t end = depth;
        while (end < P.maxdepth && R.most_of[P.most_at + end] * (int64_t)c->nnu <= thin) ++end;
        if (end > depth + 1) {
            A.count = A.begin = nullptr; A.most = 0;
            if (launch_amr_levels(A, c->d_amr_tables + P.table_at + depth * 2 * (size_t)B.nb, (int)(end - depth), stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "forest level kernel launch failed");
            depth = end;
            continue;
        }
        A.count = c->d_amr_tables + P.table_at + depth * 2 * (size_t)B.nb;
        A.begin = A.count + B.nb;
        A.most = R.most_of[P.most_at + depth];
        if (launch_amr_level(A, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest level kernel launch failed");
        ++depth;
    }
    A.count = c->d_amr_tables + P.export_at;
    A.begin = A.count + B.nb;
    if (launch_amr_export(A, P.most_exports, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest export kernel launch failed");
    return FTTE_OK;
}

// The per-leaf means of one batch into J, directions in list order.
int launch_forest_combine(ftte_ctx *c, hipStream_t stream, const ForestRun &R, size_t b, AmrLevelRec A, double *J_dev, bool zero_first)
{
    const ForestRun::Batch &B = R.batches[b];
    A.dir = c->d_amr_dirs + R.dir_at + (size_t)B.d0;
    A.ndir = B.nb;
    if (launch_amr_combine(A, J_dev, zero_first, stream)) return fail(c, FTTE_ERR_NO_DEVICE, "forest combine kernel launch failed");
    return FTTE_OK;
}

// A prepared run on `stream`, batch after batch: its passes, then the means into J.  The combine launches read-modify-write J:
// `before_combine` (if any) is waited for in front of the first one, `after_combine` (if any) recorded behind the last, which is
// how two runs on two streams keep a fixed order of additions.
int launch_forests(ftte_ctx *c, hipStream_t stream, const ForestRun &R, AmrLevelRec A, double *J_dev, bool zero_first, bool time_batches,
                   hipEvent_t before_combine, hipEvent_t after_combine)
{
    const int nnu = c->nnu;
    int rc;
    for (size_t b = 0; b < R.batches.size(); ++b) {
        const ForestRun::Batch &B = R.batches[b];
        if (time_batches) {
            c->timing[b].updates = (int64_t)B.nb * c->ncell * nnu; c->timing[b].lanes = 0;
            FTTE_HIP(c, hipEventRecord(c->timing[b].start, stream));
        }
        for (size_t p = 0; p < B.passes.size(); ++p)
            if ((rc = launch_forest_pass(c, stream, R, b, p, A))) return rc;
        if (b == 0 && before_combine) FTTE_HIP(c, hipStreamWaitEvent(stream, before_combine, 0));
        if ((rc = launch_forest_combine(c, stream, R, b, A, J_dev, zero_first && b == 0))) return rc;
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
    const int most = c->forest_batch > 0 ? c->forest_batch : kAmrBatch;
    int batch = std::max(1, std::min(ndir, most));
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
    } else batch = (int)std::min<size_t>((size_t)most, c->amr_scratch_cap / per_dir);
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
            dirs[(size_t)d] = ForestDirHost{D.rec, D.active, D.w, nullptr, nullptr, 0, &D.depth_off, nullptr, nullptr};
        }
        if ((rc = run_forests(c, stream, dirs, batch, per_dir, A, J_dev, true, true))) return rc;
    }
    return mark_sweep(c, stream);
}


int brick_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
                hipStream_t stream, const HostPipe *pipe)
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
    // Brick order for the opacities and the accumulators (BrickLaunch::tiled, option "tiled"): a brick's layer is then one piece of
    // 4 KB instead of eight rows of 512 B a row of the frame apart.  A kernel that does nothing but these loads and stores gets a
    // third more out of the memory system that way (tools/membench.hip); the sweep gets nothing (31.66 against 31.67 ms), so the
    // option is off by default.  For grids made of whole bricks, device-resident opacities, the forms of the kernel that know it;
    // costs one more copy of the opacities per axis order.
    const bool tiled = c->tiled_opt && !pipe && !c->emit_mode && n % 64 == 0 && n % kBrickRows == 0 &&
                       (c->tiled_opt == 1 || n % P.chunk == 0);
    const int tchunk = tiled && c->tiled_opt == 2 ? P.chunk : 0; // 2: a whole brick in one piece
    // accumulators and the opacity in the layouts the groups march through
    for (int l = 0; l < 3; ++l) {
        for (int s = 0; s < P.nacc[l]; ++s)
            if (!c->acc[l][s]) FTTE_HIP(c, hipMalloc((void **)&c->acc[l][s], sizeof(double) * c->acc_cap));
        if (P.nacc[l] && tiled) {
            if (c->kappa_tiled_from[l] != c->n_kappa_sets || c->kappa_tiled_chunk[l] != tchunk) {
                if (!c->kappa_tiled[l]) FTTE_HIP(c, hipMalloc((void **)&c->kappa_tiled[l], sizeof(double) * c->kappa_cap));
                if (launch_to_layout(l, c->kappa[0], c->kappa_tiled[l], n, nnu, (long)c->ncell, stream, true, tchunk))
                    return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
                c->kappa_tiled_from[l] = c->n_kappa_sets;
                c->kappa_tiled_chunk[l] = tchunk;
            }
        } else if (P.nacc[l] && !c->kappa_ready[l]) {
            if (!c->kappa[l]) FTTE_HIP(c, hipMalloc((void **)&c->kappa[l], sizeof(double) * c->kappa_cap));
            if (!lane_ends) {
                if (launch_to_layout(l, c->kappa[0], c->kappa[l], n, nnu, (long)c->ncell, stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
                c->kappa_ready[l] = true;
            } else lane_layout[l] = true;
        }
        if (P.nacc[l] && c->emit_mode && !c->emis_ready[l]) {
            if (!c->emis[l]) FTTE_HIP(c, hipMalloc((void **)&c->emis[l], sizeof(double) * c->kappa_cap));
            if (launch_to_layout(l, c->emis[0], c->emis[l], n, nnu, (long)c->ncell, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            c->emis_ready[l] = true;
        }
    }

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
        c->bqueue_uploaded = false;
    }
    // the group records carry pointers that depend on nnu (face blocks) and on the buffers: rebuilt per sweep (a few KB)
    {
        std::vector<BrickGroup> G(P.groups.size());
        std::memset(G.data(), 0, sizeof(BrickGroup) * G.size());
        for (size_t g = 0; g < P.groups.size(); ++g) {
            const BrickPlan::Group &H = P.groups[g];
            const DirPlan &D0 = P.dirs[H.dirs[0]];
            G[g].kappa = tiled ? c->kappa_tiled[H.layout] : c->kappa[H.layout];
            G[g].emis = c->emit_mode ? c->emis[H.layout] : nullptr;
            G[g].J = c->acc[H.layout][H.acc];
            G[g].org = D0.org; G[g].si = D0.si; G[g].sv = D0.sv; G[g].su = D0.su;
            if (tiled) { // cell (brick tu, tv; row r; lane; layer i) at org + i * si + tu * bu + tv * bv + r * sv + lane (63 - lane mirrored)
                const int64_t ntu = n / 64, ntv = n / kBrickRows, piece = 64 * kBrickRows;
                G[g].sv = D0.sv > 0 ? 64 : -64;
                G[g].bu = (int32_t)(D0.su > 0 ? piece : -piece);
                G[g].bv = (int32_t)(D0.sv > 0 ? ntu * piece : -ntu * piece);
                G[g].org = (int64_t)(D0.si > 0 ? -1 : n) * n * n + (D0.sv > 0 ? 0 : (ntv - 1) * ntu * piece + (kBrickRows - 1) * 64) +
                           (D0.su > 0 ? 0 : (ntu - 1) * piece);
                if (tchunk) { // layer i = chunk * ti + il + 1 at org + i * si + ti * bi: si = +-piece inside the brick
                    const int64_t nti = n / tchunk, brick = piece * tchunk;
                    G[g].si = (int32_t)(D0.si > 0 ? piece : -piece);
                    G[g].bu = (int32_t)(D0.su > 0 ? brick : -brick);
                    G[g].bv = (int32_t)(D0.sv > 0 ? ntu * brick : -ntu * brick);
                    const int64_t step_i = D0.si > 0 ? ntv * ntu * brick : -ntv * ntu * brick;
                    G[g].bi = step_i - (int64_t)tchunk * G[g].si;
                    // i = 1 (ti = 0, il = 0): the first layer of brick 0 (si > 0) or the last layer of the last brick (si < 0)
                    G[g].org = (D0.si > 0 ? -piece : (nti - 1) * ntv * ntu * brick + (int64_t)tchunk * piece) +
                               (D0.sv > 0 ? 0 : (ntv - 1) * ntu * brick + (kBrickRows - 1) * 64) + (D0.su > 0 ? 0 : (ntu - 1) * brick);
                }
            }
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
            constexpr size_t kSyncWords = 32 * (kBrickQueues + 1);
            if (!c->d_bsync) {
                FTTE_HIP(c, hipMalloc((void **)&c->d_bsync, sizeof(uint32_t) * kSyncWords));
                FTTE_HIP(c, hipHostMalloc((void **)&c->h_berror, sizeof(uint32_t) * kSyncWords, hipHostMallocDefault));
                std::memset(c->h_berror, 0, sizeof(uint32_t) * kSyncWords);
            }
            FTTE_HIP(c, hipMemsetAsync(c->d_bsync, 0, sizeof(uint32_t) * kSyncWords, stream));
            if (P.persistent && (c->d_bqueue_cap < P.queue.size() || !c->bqueue_uploaded)) {
                if ((rc = ensure(c, &c->d_bqueue, &c->d_bqueue_cap, P.queue.size()))) return rc;
                FTTE_HIP(c, hipMemcpy(c->d_bqueue, P.queue.data(), sizeof(uint32_t) * P.queue.size(), hipMemcpyHostToDevice));
                c->bqueue_uploaded = true;
            }
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
            L.ticket = c->d_bsync; L.error = c->d_bsync + 32 * kBrickQueues; L.done = c->d_bdone; L.deps = c->d_bdeps; L.epoch = ++c->bepoch; L.pad_ = c->dataflow == 2 ? 1 : 0;
            L.math = kMath;
            L.tiled = tiled ? 1 : 0;
            L.pad2_ = c->ablate;
            L.atomic_acc = c->atomic_acc;
            int persistent = 0;
            if (P.persistent) {
                L.queue = c->d_bqueue;
                std::memcpy(L.qoff, P.qoff, sizeof L.qoff);
                std::memcpy(L.qlen, P.qlen, sizeof L.qlen);
                std::memcpy(L.xcc_queue, c->xcc_queue, sizeof L.xcc_queue);
                // as many workgroups as the GPU holds (four waves per SIMD; fewer fit when LDS is padded: the rest start late and
                // find the queues empty)
                hipDeviceProp_t prop;
                FTTE_HIP(c, hipGetDeviceProperties(&prop, c->device));
                persistent = (int)std::min<size_t>((size_t)prop.multiProcessorCount * 16, P.queue.size());
            }
            const int lrc = launch_brick(L, P.max_dirs, c->brick_waves, stream, false, persistent);
            if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
            FTTE_HIP(c, hipMemcpyAsync(c->h_berror, c->d_bsync, sizeof(uint32_t) * kSyncWords, hipMemcpyDeviceToHost, stream));
            if (P.persistent) std::memcpy(c->bqlen, P.qlen, sizeof c->bqlen);
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
                L.tiled = tiled ? 1 : 0;
                L.pad2_ = c->ablate;
                L.atomic_acc = c->atomic_acc;
                const int form = brick_form(c, nnu);
                c->last_brick_form = form;
                const int lrc = form == 2 ? launch_brick_pair(L, P.max_dirs, c->pair_waves, q) : launch_brick(L, P.max_dirs, c->brick_waves, q);
                if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
            }
            if (lane_ends) { // this lane's J: merged as soon as its stages are done, and on its way back (pinned arrays) behind that
                FTTE_HIP(c, hipEventRecord(T.last[(size_t)lane], q));
                const double *accs[3 * kMaxAcc];
                int layouts[3 * kMaxAcc], count = 0;
                for (int l = 0; l < 3; ++l)
                    for (int s2 = 0; s2 < P.nacc[l]; ++s2) { accs[count] = c->acc[l][s2] + slice0; layouts[count++] = l; }
                if (launch_merge(accs, layouts, count, J_dev + slice0, n, nu1 - nu0, (long)c->ncell, false, q, nullptr, 0, tiled, tchunk))
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
            ++c->n_kappa_sets;
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
            if (launch_merge(accs, layouts, count, J_dev, n, nnu, (long)c->ncell, false, stream, nullptr, 0, tiled, tchunk))
                return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
        } else FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * c->ncell, stream)); // no directions
    }
    return mark_sweep(c, stream);
}


// The sweep of a uniform grid by ray-following tiles (ftte::sweep_kernel, option "engine" = 1): launches of up to `slots`
// directions of one layout, each layout's accumulators merged into J on a second stream while the next layout is swept.
int tile_sweep(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w, const double *uvb, double *J_dev,
               hipStream_t stream)
{
    int rc;
    const int n = c->n, nnu = c->nnu;
    const size_t per_acc = (size_t)nnu * c->ncell;
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

} // namespace ftte
