// ftte_plan.cpp -- the host planners of the uniform-grid sweeps: directions -> per-layer tables, then either ray-following
// tiles and launches (build_plan, ftte::sweep_kernel) or cell-fixed bricks, groups, accumulators and stages (plan_brick_groups,
// build_brick_plan, ftte::brick_kernel).  Pure host work, cached in the context.
#include "ftte_context.h"

namespace ftte {

// ---- planner ------------------------------------------------------------------------------------
// One direction: fold it (equiSources.f90:1395-1454), build its per-layer patterns (:1495-1534, setPattern) and turn them
// into what the kernels read: the memory frame of its izone and one LayerRec per layer.
int plan_direction(ftte_ctx *c, int d, double phi_d, double theta_d, double w_d, int tile_rows, std::vector<ftte_pattern> &pat,
                   std::vector<int> &du_cum, std::vector<int> &dv_cum, DirPlan &D, LayerRec *layers_of_d, size_t layer_off,
                   const SubGridPlan *sub)
{
    // sub: the planes of a cubic sub-grid of `sub->n` cells a side and cell size sub->cell, whose layers carry the patterns
    // sub->patterns(d) instead of a ray's own march from (0.5, 0.5): the fine cells of a fully refined block (ftte_hybrid.cpp)
    const int n = sub ? sub->n : c->n;
    const double cell = sub ? sub->cell : c->box / (double)n; // cellSizeAbsoluteUnits, equiSources.f90:1570
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
    if (sub) {
        const int src = sub->patterns(d, D.phi, D.theta, D.izone, pat.data());
        if (src) return fail(c, src, "direction " + std::to_string(d) + ": ray pattern left the unit cell (sub-layer patterns of a refined block)");
    } else if (layer_patterns(n, D.phi, D.theta, pat.data())) {
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
                      int want_dataflow, bool whole_faces, const SubGridPlan *sub)
{
    const int n = sub ? sub->n : c->n;
    ++c->n_plan_builds;
    P = BrickPlan();
    P.n = n; P.chunk = chunk; P.gmax = gmax; P.share = c->share; P.want_dataflow = want_dataflow; P.box = c->box;
    P.phi.assign(phi, phi + ndir); P.theta.assign(theta, theta + ndir); P.w.assign(w, w + ndir);
    P.dirs.resize(ndir);
    P.layers.resize((size_t)ndir * n);

    std::vector<ftte_pattern> pat(n);
    std::vector<int> du_cum(n + 1), dv_cum(n + 1);
    for (int d = 0; d < ndir; ++d) {
        const int rc = plan_direction(c, d, phi[d], theta[d], w[d], 7, pat, du_cum, dv_cum, P.dirs[d], &P.layers[(size_t)d * n], (size_t)d * n, sub);
        if (rc) return rc;
    }
    P.ntu = (n + 63) / 64; P.ntv = (n + kBrickRows - 1) / kBrickRows; P.nti = (n + chunk - 1) / chunk;
    P.up = 64 * P.ntu; P.vp = kBrickRows * P.ntv;
    P.dataflow = want_dataflow != 0;
    P.ut = P.dataflow ? 16 : kBrickRows; // a 128-byte line of its own per brick and layer when bricks of one launch exchange rays
    P.uw = P.ntv * P.ut;
    // rings over two chunks, or every chunk's faces kept (hybrid sweep); a sub-grid keeps one slot more (the chunk before its first)
    // and one ring more along u and v (the brick columns / rows before its first): BrickLaunch::sub
    P.nslot = sub ? P.nti + 1 : whole_faces ? P.nti : 2;
    const int edge = sub ? 1 : 0;
    P.vface_off = (int64_t)(P.ntu + edge) * P.nslot * chunk * P.uw;
    P.iface_off = P.vface_off + (int64_t)(P.ntv + edge) * P.nslot * chunk * P.up;
    P.uqface_off = P.iface_off + (int64_t)P.nslot * P.vp * P.up;
    // (the faces inside a brick are used by the hybrid sweep only, which keeps every chunk's faces)
    P.face_elems = P.uqface_off + (whole_faces ? 2 * (int64_t)P.nslot * chunk * P.uw : 0);

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
    const int form = brick_form(c, nnu);
    const int gmax = c->group > 0 ? c->group : (nnu >= 2 || form == 2 ? 3 : 2);
    int want_dataflow = (c->dataflow && n % 64 == 0 && n % kBrickRows == 0 && n % chunk == 0 && form == 0 && !c->emit_mode) ? 1 : 0;
    if (want_dataflow && c->dataflow == 3) { // persistent workgroups, a queue per XCD: needs to know the XCDs
        if ((rc = xcc_census(c))) return rc;
        want_dataflow = (c->xcc_count >= 1 && c->xcc_count <= kBrickQueues) ? 3 : 1;
    }
    const int want_glanes = want_dataflow ? 1 : (nnu >= c->lanes ? 1 : c->lanes);
    if (P.valid && P.n == n && P.chunk == chunk && P.gmax == gmax && P.share == c->share && P.want_glanes == want_glanes &&
        P.want_dataflow == want_dataflow && P.box == c->box && (want_dataflow != 3 || (P.qnnu == nnu && P.nq == c->xcc_count && P.qmix == c->queue_mix)) &&
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
    P.persistent = false;
    if (P.dataflow && want_dataflow == 3 && !P.tasks.empty()) {
        // Queues.  What a brick waits for belongs to its own frequency group and to the groups of directions that share its
        // accumulator, so (frequency group, accumulator) pairs are the units that can be dealt out.  With a multiple of the queue
        // count in frequency groups, queue = group mod queues (all direction groups of a frequency group read the same opacities:
        // one L2 for them); else the units go, largest first, to the queue with the least work so far.
        const int nq = c->xcc_count;
        P.persistent = true; P.qnnu = nnu; P.nq = nq; P.qmix = c->queue_mix;
        std::vector<int64_t> acc_dirs(3 * (size_t)kMaxAcc, 0);
        for (const auto &G : P.groups) acc_dirs[(size_t)G.layout * kMaxAcc + G.acc] += (int64_t)G.dirs.size();
        std::vector<int> queue_of((size_t)nnu * 3 * kMaxAcc, -1);
        int64_t load[kBrickQueues] = {};
        if (nnu % nq == 0 && c->queue_mix == 0) {
            for (int nu = 0; nu < nnu; ++nu)
                for (size_t a = 0; a < acc_dirs.size(); ++a)
                    if (acc_dirs[a]) { queue_of[(size_t)nu * acc_dirs.size() + a] = nu % nq; load[nu % nq] += acc_dirs[a]; }
        } else if (c->queue_mix == 2) { // every queue a share of every frequency group: accumulator a of group nu to queue (nu + a) mod queues
            for (int nu = 0; nu < nnu; ++nu) {
                int k = 0;
                for (size_t a = 0; a < acc_dirs.size(); ++a)
                    if (acc_dirs[a]) { const int q = (nu + k++) % nq; queue_of[(size_t)nu * acc_dirs.size() + a] = q; load[q] += acc_dirs[a]; }
            }
        } else {
            std::vector<std::pair<int64_t, size_t>> units;
            for (int nu = 0; nu < nnu; ++nu)
                for (size_t a = 0; a < acc_dirs.size(); ++a)
                    if (acc_dirs[a]) units.push_back({acc_dirs[a], (size_t)nu * acc_dirs.size() + a});
            std::stable_sort(units.begin(), units.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
            for (const auto &u : units) {
                int q = 0;
                for (int k = 1; k < nq; ++k) if (load[k] < load[q]) q = k;
                queue_of[u.second] = q;
                load[q] += u.first;
            }
        }
        std::vector<std::vector<uint32_t>> lists((size_t)nq);
        for (size_t t = 0; t < P.tasks.size(); ++t) {
            const BrickPlan::Group &G = P.groups[(size_t)P.tasks[t].group];
            const size_t a = (size_t)G.layout * kMaxAcc + G.acc;
            for (int nu = 0; nu < nnu; ++nu)
                lists[(size_t)queue_of[(size_t)nu * acc_dirs.size() + a]].push_back((uint32_t)(t * (size_t)nnu + (size_t)nu));
        }
        P.queue.clear();
        for (int q = 0; q < kBrickQueues; ++q) {
            P.qoff[q] = (uint32_t)P.queue.size();
            P.qlen[q] = q < nq ? (uint32_t)lists[(size_t)q].size() : 0;
            P.qload[q] = q < nq ? load[q] * (int64_t)n * n * n : 0;
            if (q < nq) P.queue.insert(P.queue.end(), lists[(size_t)q].begin(), lists[(size_t)q].end());
        }
    }
    P.valid = true;
    return FTTE_OK;
}


} // namespace ftte
