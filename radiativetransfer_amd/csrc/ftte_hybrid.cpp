// ftte_hybrid.cpp -- the hybrid sweep of a refined cell array: bricks outside a box around the refined cells, segment forests inside.
#include "ftte_context.h"

namespace ftte {

// ---- hybrid sweep of a refined cell array -----------------------------------------------------------------------------
// The reference recurses into refined cells wherever they are (transport, transportRoutinesModule.f90:577-586) and walks the
// tree for every upstream link of every cell of every direction.  Most of a cell array is plain base cells; here those are
// swept by the brick kernel, and only a box around the refined cells -- widened by one brick, so that its surface separates
// unrefined base cells, across which a ray is handed over exactly as between two bricks -- by the segment forest.  Per group of
// directions: the bricks that do not lie behind the box, then the forest (rays entering it read from the bricks' face
// buffers, rays leaving it written there), then the bricks behind it.  J of a cell = what the bricks stored for the
// directions in whose box it does not lie + what the forest adds for the others.

static void drop_graph(ftte_ctx::HybridPlan &H)
{
    if (H.graph_exec) (void)hipGraphExecDestroy(H.graph_exec);
    H.graph_exec = nullptr;
    H.graph_sig.clear();
}

void free_hybrid(ftte_ctx *c)
{
    drop_graph(c->hplan);
    for (auto &d : c->hplan.dirs) {
        if (d.rec) (void)hipFree(d.rec);
        if (d.active) (void)hipFree(d.active);
        if (d.exports) (void)hipFree(d.exports);
    }
    if (c->hplan.cells) (void)hipFree(c->hplan.cells);
    for (auto &D : c->hplan.dirs) if (D.imports) (void)hipFree(D.imports);
    if (c->hplan.fine.leaf_of_fine) (void)hipFree(c->hplan.fine.leaf_of_fine);
    if (c->hplan.fine.layers) (void)hipFree(c->hplan.fine.layers);
    if (c->hplan.fine.tasks) (void)hipFree(c->hplan.fine.tasks);
    if (c->hplan.fine.groups) (void)hipFree(c->hplan.fine.groups);
    c->hplan = ftte_ctx::HybridPlan();
}

// Where the refined base cells are, in storage coordinates (1-based, inclusive): one bounding box per cluster.  Two refined cells
// belong to one cluster when they lie within two 8-cell blocks of each other; what the per-izone alignment below still brings
// into contact is merged there.
struct Extent { int lo[3], hi[3]; };

std::vector<Extent> refined_clusters(const AmrTree &T)
{
    const int n = T.n, M = 8, nm = (n + M - 1) / M;
    std::vector<int32_t> label((size_t)nm * nm * nm, -2); // -2: no refined cell, -1: not yet labelled
    for (int64_t b = 0; b < (int64_t)n * n * n; ++b)
        if (T.child0[(size_t)b] >= 0)
            label[(((size_t)(b / ((int64_t)n * n)) / M) * nm + (size_t)((b / n) % n) / M) * nm + (size_t)(b % n) / M] = -1;
    int32_t count = 0;
    std::vector<int32_t> stack;
    for (int32_t m0 = 0; m0 < (int32_t)label.size(); ++m0) {
        if (label[(size_t)m0] != -1) continue;
        label[(size_t)m0] = count;
        stack.assign(1, m0);
        while (!stack.empty()) {
            const int32_t m = stack.back();
            stack.pop_back();
            const int a = m / (nm * nm), b = (m / nm) % nm, c2 = m % nm;
            for (int da = -2; da <= 2; ++da)
                for (int db = -2; db <= 2; ++db)
                    for (int dc = -2; dc <= 2; ++dc) {
                        const int x = a + da, y = b + db, z = c2 + dc;
                        if (x < 0 || y < 0 || z < 0 || x >= nm || y >= nm || z >= nm) continue;
                        int32_t &l = label[((size_t)x * nm + y) * nm + z];
                        if (l == -1) { l = count; stack.push_back((int32_t)(((size_t)x * nm + y) * nm + z)); }
                    }
        }
        ++count;
    }
    std::vector<Extent> out((size_t)count);
    for (auto &e : out) for (int a = 0; a < 3; ++a) { e.lo[a] = n + 1; e.hi[a] = 0; }
    for (int64_t b = 0; b < (int64_t)n * n * n; ++b)
        if (T.child0[(size_t)b] >= 0) {
            const int cc[3] = {(int)(b / ((int64_t)n * n)) + 1, (int)((b / n) % n) + 1, (int)(b % n) + 1};
            Extent &e = out[(size_t)label[(((size_t)(cc[0] - 1) / M) * nm + (size_t)(cc[1] - 1) / M) * nm + (size_t)(cc[2] - 1) / M]];
            for (int a = 0; a < 3; ++a) { e.lo[a] = std::min(e.lo[a], cc[a]); e.hi[a] = std::max(e.hi[a], cc[a]); }
        }
    return out;
}

// A box of one izone: the refined cells of a cluster (`fine`: their extent in the sweep frame i, j, k) and a rim of unrefined ones,
// on brick boundaries along v and the march axis (one brick of rim) and, along u, one cell of rim (option "box_lanes": then outwards
// to the next multiple of it): a brick is 64 lanes wide, and whole bricks of rim would put every column of a 128^3 grid into the box
// of a 32^3 patch.  Bricks that a box cuts through sweep the lanes outside it (brick_kernel<..., MASKED>).
struct HybridBox {
    Extent fine;        // sweep frame
    ForestRegion R;
    int lo[3], hi[3];   // the bricks the box touches: u, v, march axis
    int ulo, uhi;       // its cells along u
    int level = 0;      // pass of its forest: 1 + the highest level among the boxes it lies behind
};

void align_box(const ftte_ctx *c, const BrickPlan &P, bool u_is_k, HybridBox *B)
{
    const int n = c->n;
    const int ju = u_is_k ? 2 : 1, jv = u_is_k ? 1 : 2; // sweep axes of u and v
    const int *slo = B->fine.lo, *shi = B->fine.hi;
    const int tsize_i = P.chunk, tsize_u = 64, tsize_v = kBrickRows;
    const int lanes = c->hybrid_lanes; // 1 (the rim and no more), a multiple such as 16, or 64: whole bricks along u as along the other axes
    B->ulo = lanes == 64 ? std::max(0, (slo[ju] - 1) / 64 - 1) * 64 + 1 : std::max(0, (slo[ju] - 2) / lanes) * lanes + 1;
    B->uhi = lanes == 64 ? std::min(n, (std::min(P.ntu - 1, (shi[ju] - 1) / 64 + 1) + 1) * 64) : std::min(n, (shi[ju] + lanes) / lanes * lanes);
    B->lo[0] = (B->ulo - 1) / tsize_u; B->hi[0] = (B->uhi - 1) / tsize_u;
    B->lo[1] = std::max(0, (slo[jv] - 1) / tsize_v - 1); B->hi[1] = std::min(P.ntv - 1, (shi[jv] - 1) / tsize_v + 1);
    B->lo[2] = std::max(0, (slo[0] - 1) / tsize_i - 1);  B->hi[2] = std::min(P.nti - 1, (shi[0] - 1) / tsize_i + 1);
    ForestRegion *R = &B->R;
    R->u_is_k = u_is_k;
    R->lo[0] = B->lo[2] * tsize_i + 1; R->hi[0] = std::min(n, (B->hi[2] + 1) * tsize_i);
    R->lo[ju] = B->ulo; R->hi[ju] = B->uhi;
    R->lo[jv] = B->lo[1] * tsize_v + 1; R->hi[jv] = std::min(n, (B->hi[1] + 1) * tsize_v);
    R->chunk = P.chunk; R->ut = P.ut; R->nslot = P.nslot; R->ntv = P.ntv; R->up = P.up; R->vp = P.vp;
    R->vface_off = P.vface_off; R->iface_off = P.iface_off; R->uqface_off = P.uqface_off;
}

// The boxes of one izone: every cluster's, merged where two would touch or cut through the same brick, with their levels: box B lies
// behind box A when some ray can pass A first and B later (B's last brick >= A's first one on every axis).  Then B's forest needs the
// bricks in between, which need A's: A is swept in an earlier pass.
std::vector<HybridBox> izone_boxes(const ftte_ctx *c, const BrickPlan &P, int izone, const std::vector<Extent> &clusters)
{
    const int n = c->n;
    ZoneMap zm;
    zone_map(izone, &zm);
    int march_c = 0;
    for (int a = 0; a < 3; ++a) if (zm.src[a] == 0) march_c = a;
    const int fast_c = (march_c == 2) ? 1 : 2;
    const bool u_is_k = zm.src[fast_c] == 2;
    std::vector<HybridBox> boxes;
    for (const Extent &e : clusters) {
        HybridBox B;
        for (int a = 0; a < 3; ++a) {
            const int sa = zm.src[a];
            B.fine.lo[sa] = zm.mirror[a] ? n + 1 - e.hi[a] : e.lo[a];
            B.fine.hi[sa] = zm.mirror[a] ? n + 1 - e.lo[a] : e.hi[a];
        }
        align_box(c, P, u_is_k, &B);
        boxes.push_back(B);
    }
    auto join = [&](size_t x, size_t y) {
        for (int a = 0; a < 3; ++a) {
            boxes[x].fine.lo[a] = std::min(boxes[x].fine.lo[a], boxes[y].fine.lo[a]);
            boxes[x].fine.hi[a] = std::max(boxes[x].fine.hi[a], boxes[y].fine.hi[a]);
        }
        align_box(c, P, u_is_k, &boxes[x]);
        boxes.erase(boxes.begin() + (long)y);
    };
    for (bool changed = true; changed;) {
        changed = false;
        for (size_t x = 0; x < boxes.size() && !changed; ++x)
            for (size_t y = x + 1; y < boxes.size() && !changed; ++y) {
                const HybridBox &A = boxes[x], &B = boxes[y];
                // too close: next to each other brick-wise along v and the march axis and, along u, in one brick or within two cells
                const bool near_v = A.lo[1] - 1 <= B.hi[1] && B.lo[1] - 1 <= A.hi[1], near_i = A.lo[2] - 1 <= B.hi[2] && B.lo[2] - 1 <= A.hi[2];
                const bool near_u = (A.lo[0] <= B.hi[0] && B.lo[0] <= A.hi[0]) || (A.ulo - 2 <= B.uhi && B.ulo - 2 <= A.uhi);
                // each behind the other: cannot be ordered
                bool a_then_b = true, b_then_a = true;
                for (int k = 0; k < 3; ++k) { a_then_b = a_then_b && B.hi[k] >= A.lo[k]; b_then_a = b_then_a && A.hi[k] >= B.lo[k]; }
                if ((near_u && near_v && near_i) || (a_then_b && b_then_a)) { join(x, y); changed = true; }
            }
    }
    // levels: longest chain of boxes in front.  The relation has no cycles among boxes that do not intersect; should the relaxation
    // not settle all the same, one box takes everything.
    const size_t K = boxes.size();
    bool settled = false;
    for (size_t round = 0; round <= K && !settled; ++round) {
        settled = true;
        for (size_t y = 0; y < K; ++y)
            for (size_t x = 0; x < K; ++x) {
                if (x == y) continue;
                bool x_then_y = true;
                for (int k = 0; k < 3; ++k) x_then_y = x_then_y && boxes[y].hi[k] >= boxes[x].lo[k];
                if (x_then_y && boxes[y].level < boxes[x].level + 1) { boxes[y].level = boxes[x].level + 1; settled = false; }
            }
    }
    if (!settled) {
        while (boxes.size() > 1) join(0, 1);
        boxes[0].level = 0;
    }
    for (size_t x = 0; x < boxes.size(); ++x) { boxes[x].R.id = (int)x; boxes[x].R.pass = boxes[x].level; }
    return boxes;
}

int build_hybrid_plan(ftte_ctx *c, int ndir, const double *phi, const double *theta, const double *w)
{
    ftte_ctx::HybridPlan &H = c->hplan;
    const int n = c->n, nnu = c->nnu;
    // short bricks: the box is widened by one brick on every side, and what lies inside it costs several times a brick's bytes
    const int chunk = std::min(c->chunk > 0 ? c->chunk : 4, n);
    const int gmax = c->group > 0 ? c->group : (nnu >= 2 ? 3 : 2);
    std::vector<double> key = {c->box, (double)chunk, (double)gmax, (double)c->share, (double)c->halves, (double)c->hybrid_lanes, (double)c->hybrid_slots,
                               (double)c->forest_batch, (double)c->fine_bricks, (double)c->fine_chunk};
    key.insert(key.end(), phi, phi + ndir);
    key.insert(key.end(), theta, theta + ndir);
    key.insert(key.end(), w, w + ndir);
    if (H.valid && H.key == key) return FTTE_OK;
    free_hybrid(c);
    int rc;
    BrickPlan &P = H.bricks;
    if ((rc = plan_brick_groups(c, P, ndir, phi, theta, w, chunk, gmax, 0, true))) return rc;
    P.glanes = 1;

    // the boxes of every group; is the part outside them worth a brick sweep?
    const std::vector<Extent> clusters = refined_clusters(c->tree);
    std::vector<std::vector<HybridBox>> boxes(P.groups.size());
    int64_t inside_bricks = 0, all_bricks = 0;
    int most_boxes = 0, top_level = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        if (g > 0 && P.groups[g].izone == P.groups[g - 1].izone) boxes[g] = boxes[g - 1];
        else boxes[g] = izone_boxes(c, P, P.groups[g].izone, clusters);
        all_bricks += (int64_t)P.ntu * P.ntv * P.nti;
        most_boxes = std::max(most_boxes, (int)boxes[g].size());
        for (const HybridBox &B : boxes[g]) {
            inside_bricks += (int64_t)(B.hi[0] - B.lo[0] + 1) * (B.hi[1] - B.lo[1] + 1) * (B.hi[2] - B.lo[2] + 1);
            top_level = std::max(top_level, B.level);
        }
    }
    // every box has its own pair of face rings for rays that cross its u-faces inside a brick
    P.face_elems = P.uqface_off + 2 * (int64_t)std::max(most_boxes, 1) * P.nslot * P.chunk * P.uw;
    H.key = key;
    H.valid = true;
    H.worthwhile = !P.groups.empty() && most_boxes > 0 && most_boxes <= kBrickBoxMask && inside_bricks * 2 <= all_bricks; // else: the forest path for the whole tree
    H.npass = top_level + 1;
    H.most_boxes = most_boxes;
    if (!H.worthwhile) return FTTE_OK;

    // Several passes: every brick gets the earliest launch its own inputs allow ("slots", below) instead of a phase per pass,
    // which needs accumulators that are not shared between groups (the proof that two groups of one accumulator never meet in a
    // launch rests on launch = stage + offset).  36 accumulators of a 128^3 base grid are 5 GB and 1 ms of merge.
    H.slots = (H.npass > 1 && c->hybrid_slots) || c->hybrid_slots == 2;
    // Forest batches smaller than the direction list (option "forest_batch") put all pipelines' forests into one run on one stream,
    // which has ONE place in the launch sequence: only the phase form gives every pipeline the same place for a pass.  (Where it is
    // the device memory that makes the batch small, hybrid_sweep finds out later and leaves such a sweep to the forest path.)
    if (c->forest_batch > 0 && c->forest_batch < ndir) H.slots = false;
    if (H.slots) {
        int per_layout[3] = {0, 0, 0};
        for (const auto &G : P.groups) ++per_layout[G.layout];
        if (per_layout[0] > kMaxAcc || per_layout[1] > kMaxAcc || per_layout[2] > kMaxAcc) H.slots = false;
    }
    if (H.slots) {
        P.nacc[0] = P.nacc[1] = P.nacc[2] = 0;
        for (auto &G : P.groups) { G.acc = P.nacc[G.layout]++; G.offset = 0; }
    }

    // Pipelines ("halves" in the names below): the forests stream records at the memory system's rate while the brick stages of a
    // 128^3 grid are short launches that leave most of it idle, so the sweep runs as up to four independent sequences (bricks -
    // forests - bricks ...) on streams of their own.  What the groups of one accumulator write is ordered by their launches, so
    // an accumulator's groups stay together; the pipelines are balanced by direction count.
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

    // ---- a fully refined block swept by bricks of its own on the fine level.  One cluster, a cube of q base cells a side refined
    // exactly once, 2 q a multiple of the bricks' 64 lanes; one pass, launch lists by phase.  Inside the block the fine
    // cells are a uniform grid of 2 q cells a side whose sub-layers carry the patterns of setRaysRefined
    // (transportRoutinesModule.f90:150-187): the brick kernel sweeps it like a grid of its own (plan_brick_groups with a SubGridPlan),
    // rays cross its faces through rings of its own face block, and the forest keeps what lies around it.
    ftte_ctx::HybridPlan::Fine &FN = H.fine;
    FN = ftte_ctx::HybridPlan::Fine();
    const int fine_chunk = c->fine_chunk > 0 ? c->fine_chunk : chunk; // layers per brick on the fine level (option "fine_chunk")
    if (c->fine_bricks && clusters.size() == 1 && H.npass == 1 && !H.slots) {
        const Extent &e = clusters[0];
        const int q = e.hi[0] - e.lo[0] + 1;
        bool cube = q == e.hi[1] - e.lo[1] + 1 && q == e.hi[2] - e.lo[2] + 1 && (2 * q) % 64 == 0 && (2 * q) % fine_chunk == 0 && 2 * q <= 32000;
        for (int a = e.lo[0]; a <= e.hi[0] && cube; ++a)
            for (int b = e.lo[1]; b <= e.hi[1] && cube; ++b)
                for (int d = e.lo[2]; d <= e.hi[2] && cube; ++d) {
                    const int32_t node = (int32_t)(((int64_t)(a - 1) * n + (b - 1)) * n + (d - 1));
                    const int32_t c0 = c->tree.child0[(size_t)node];
                    if (c0 < 0) { cube = false; break; }
                    for (int k = 0; k < 8; ++k) if (c->tree.child0[(size_t)(c0 + k)] >= 0) cube = false; // refined once, no deeper
                }
        if (cube) {
            FN.active = true;
            FN.n = 2 * q;
            for (int a = 0; a < 3; ++a) FN.lo[a] = e.lo[a];
        }
    }
    if (FN.active) {
        SubGridPlan sg;
        sg.n = FN.n;
        sg.cell = c->box / (double)n / 2.0; // the size of a cell halves per level (transportRoutinesModule.f90:583)
        const Extent cluster = clusters[0];
        sg.patterns = [&, cluster](int, double phi_f, double theta_f, int izone, ftte_pattern *out) -> int {
            // the base layers the block spans along this izone's march axis, each with its two sub-layers
            ZoneMap zm;
            zone_map(izone, &zm);
            int lo0 = 1;
            for (int a = 0; a < 3; ++a)
                if (zm.src[a] == 0) lo0 = zm.mirror[a] ? n + 1 - cluster.hi[a] : cluster.lo[a];
            std::vector<ftte_pattern> base((size_t)n);
            if (layer_patterns(n, phi_f, theta_f, base.data())) return FTTE_ERR_PATTERN;
            for (int i = 0; i < FN.n / 2; ++i)
                if (sub_layer_patterns(base[(size_t)(lo0 - 1 + i)], phi_f, theta_f, &out[2 * i], &out[2 * i + 1])) return FTTE_ERR_PATTERN;
            return 0;
        };
        const int share = c->share;
        c->share = 0; // an accumulator per group: the fine grid is small and every launch a plain store
        rc = plan_brick_groups(c, FN.plan, ndir, phi, theta, w, fine_chunk, gmax, 0, true, &sg);
        c->share = share;
        if (rc) { free_hybrid(c); return rc; }
        BrickPlan &Q = FN.plan;
        if (Q.groups.size() != P.groups.size()) { free_hybrid(c); return fail(c, FTTE_ERR_STATE, "hybrid plan: the fine block's groups differ from the base grid's"); }
        Q.face_elems = Q.uqface_off; // (no boxes inside the fine grid)
        FN.face_base = P.face_elems;
        // stage lists per pipeline: stage = tu + tv + ti, the groups with the most directions first
        FN.nstages = Q.ntu + Q.ntv + Q.nti - 2;
        const size_t nst = (size_t)FN.nstages, nl = (size_t)H.nhalves * nst;
        FN.stage_off.assign(nl + 1, 0); // list l = pipeline * nstages + stage: tasks [stage_off[l], stage_off[l + 1])
        for (size_t g = 0; g < Q.groups.size(); ++g)
            for (int ti = 0; ti < Q.nti; ++ti)
                for (int tv = 0; tv < Q.ntv; ++tv)
                    for (int tu = 0; tu < Q.ntu; ++tu) ++FN.stage_off[(size_t)half_of_group[g] * nst + (size_t)(tu + tv + ti) + 1];
        for (size_t l = 0; l < nl; ++l) FN.stage_off[l + 1] += FN.stage_off[l];
        Q.tasks.resize(FN.stage_off[nl]);
        std::vector<size_t> at(FN.stage_off.begin(), FN.stage_off.end() - 1);
        std::vector<size_t> by_size(Q.groups.size());
        for (size_t g = 0; g < by_size.size(); ++g) by_size[g] = g;
        std::stable_sort(by_size.begin(), by_size.end(), [&](size_t x, size_t y) { return Q.groups[x].dirs.size() > Q.groups[y].dirs.size(); });
        for (size_t g : by_size)
            for (int ti = 0; ti < Q.nti; ++ti)
                for (int tv = 0; tv < Q.ntv; ++tv)
                    for (int tu = 0; tu < Q.ntu; ++tu) {
                        BrickTask T;
                        T.group = (int16_t)g; T.tu = (int16_t)tu; T.tv = (int16_t)tv; T.ti = (int16_t)ti;
                        Q.tasks[at[(size_t)half_of_group[g] * nst + (size_t)(tu + tv + ti)]++] = T;
                    }
        FN.updates = (int64_t)FN.n * FN.n * FN.n * ndir;
        // the boxes learn about the block: its extent in their sweep frame (the refined cells' own) and the fine face block
        for (auto &BX : boxes)
            for (HybridBox &B : BX) {
                ForestRegion &R = B.R;
                R.has_fine = true;
                for (int a = 0; a < 3; ++a) { R.flo[a] = B.fine.lo[a]; R.fhi[a] = B.fine.hi[a]; }
                R.fine.chunk = Q.chunk; R.fine.ut = Q.ut; R.fine.nslot = Q.nslot; R.fine.ntu = Q.ntu; R.fine.ntv = Q.ntv; R.fine.up = Q.up; R.fine.vp = Q.vp;
                R.fine.vface_off = Q.vface_off; R.fine.iface_off = Q.iface_off; R.fine.base = FN.face_base;
            }
    }

    // tasks: the bricks outside the boxes, and what the boxes leave of the bricks they cut through.  Phase 0: what lies behind no
    // box (no brick index at or beyond a box's first one on all three axes); phase k: what needs the forests up to pass k - 1.
    // Within a phase stage by stage as in a plain sweep.
    int max_offset = 0;
    for (const auto &G : P.groups) max_offset = std::max(max_offset, G.offset);
    const int per_phase = P.ntu + P.ntv + P.nti - 2 + max_offset;
    H.phase1_stages = (size_t)per_phase;
    // What a group sweeps of brick (tu, tv, ti): nothing (a box holds it), all of it, or -- a box cuts through it along u -- the
    // lanes on the near side of that box and / or those on the far side.  Every piece runs in the phase after the last pass it
    // depends on: 1 + the highest level among the boxes it lies behind (phase 0: behind none).
    struct Piece { int lane_lo, lane_hi, phase, box; bool masked; };
    auto pieces_of = [&](const std::vector<HybridBox> &BX, int tu, int tv, int ti, Piece out[2]) -> int {
        const int last = std::min(63, n - 64 * tu - 1); // last lane with a cell
        int behind_all = 0, cut = -1;
        for (size_t x = 0; x < BX.size(); ++x) {
            const HybridBox &B = BX[x];
            if (tu >= B.lo[0] && tv >= B.lo[1] && ti >= B.lo[2]) behind_all = std::max(behind_all, B.level + 1);
            if (tv >= B.lo[1] && tv <= B.hi[1] && ti >= B.lo[2] && ti <= B.hi[2] && B.ulo <= 64 * tu + 64 && B.uhi >= 64 * tu + 1) cut = (int)x;
        }
        if (cut < 0) { out[0] = Piece{0, 63, behind_all, 0, false}; return 1; } // no box reaches into this brick
        const HybridBox &X = BX[(size_t)cut];
        const int first_in = std::max(X.ulo, 64 * tu + 1) - (64 * tu + 1), last_in = std::min(X.uhi, 64 * tu + 64) - (64 * tu + 1);
        int count = 0;
        if (first_in > 0) { // the near side: not behind the box it belongs to
            int behind_others = 0;
            for (size_t x = 0; x < BX.size(); ++x)
                if ((int)x != cut && tu >= BX[x].lo[0] && tv >= BX[x].lo[1] && ti >= BX[x].lo[2]) behind_others = std::max(behind_others, BX[x].level + 1);
            out[count++] = Piece{0, first_in - 1, behind_others, cut, true};
        }
        if (last_in < last) out[count++] = Piece{last_in + 1, 63, behind_all, cut, true};
        return count;
    };
    const size_t nb = (size_t)P.ntu * P.ntv * P.nti;
    auto brick_index = [&](int tu, int tv, int ti) { return ((size_t)ti * P.ntv + tv) * P.ntu + tu; };

    // ---- slots (several passes).  A piece's slot is the first launch after everything it takes rays from: the pieces of the three
    // bricks upstream that share lanes with it, and the forests of the boxes directly upstream of it (pass k of a pipeline is
    // issued in front of the launches of slot pass_at[k], which lies behind every piece that feeds a box of level k in that
    // pipeline).  Level after level, because a pass's place needs the slots of its feeders and its consumers' slots need its place.
    std::vector<std::vector<int32_t>> slot(P.groups.size());
    H.pass_at.assign((size_t)H.nhalves, std::vector<int>((size_t)H.npass, 0));
    int nslots = 0;
    if (H.slots) {
        auto box_over = [&](const std::vector<HybridBox> &BX, int tu, int tv, int ti, int lane_lo, int lane_hi) -> int {
            for (size_t x = 0; x < BX.size(); ++x) {
                const HybridBox &B = BX[x];
                if (tv >= B.lo[1] && tv <= B.hi[1] && ti >= B.lo[2] && ti <= B.hi[2] && B.ulo <= 64 * tu + 1 + lane_hi && B.uhi >= 64 * tu + 1 + lane_lo) return (int)x;
            }
            return -1;
        };
        for (size_t g = 0; g < P.groups.size(); ++g) slot[g].assign(2 * nb, -1);
        for (int k = 0; k <= H.npass; ++k) {
            for (size_t g = 0; g < P.groups.size(); ++g) {
                const std::vector<HybridBox> &BX = boxes[g];
                const std::vector<int> &at = H.pass_at[(size_t)half_of_group[g]];
                std::vector<int32_t> &S = slot[g];
                for (int ti = 0; ti < P.nti; ++ti)
                    for (int tv = 0; tv < P.ntv; ++tv)
                        for (int tu = 0; tu < P.ntu; ++tu) {
                            Piece pc[2];
                            const int np = pieces_of(BX, tu, tv, ti, pc);
                            for (int q = 0; q < np; ++q) {
                                int32_t &mine = S[2 * brick_index(tu, tv, ti) + (size_t)q];
                                if (mine >= 0) continue;
                                int s2 = 0;
                                bool known = true;
                                auto after_pass = [&](int x) { if (BX[(size_t)x].level >= k) known = false; else s2 = std::max(s2, at[(size_t)BX[(size_t)x].level]); };
                                // the brick on the near side along u, or (lanes that start inside the brick) the box there
                                if (pc[q].lane_lo > 0) after_pass(pc[q].box);
                                else if (tu > 0) {
                                    Piece up[2];
                                    const int nu2 = pieces_of(BX, tu - 1, tv, ti, up);
                                    if (nu2 > 0 && up[nu2 - 1].lane_hi == 63) {
                                        const int32_t v = S[2 * brick_index(tu - 1, tv, ti) + (size_t)(nu2 - 1)];
                                        if (v < 0) known = false; else s2 = std::max(s2, v + 1);
                                    } else { const int x = box_over(BX, tu - 1, tv, ti, 63, 63); if (x >= 0) after_pass(x); }
                                }
                                // the bricks below along v and the march axis: their pieces that share lanes, and the box between them
                                for (int axis = 1; axis <= 2; ++axis) {
                                    const int nv = axis == 1 ? tv - 1 : tv, ni = axis == 2 ? ti - 1 : ti;
                                    if (nv < 0 || ni < 0) continue;
                                    Piece up[2];
                                    const int nu2 = pieces_of(BX, tu, nv, ni, up);
                                    for (int r = 0; r < nu2; ++r)
                                        if (up[r].lane_lo <= pc[q].lane_hi && up[r].lane_hi >= pc[q].lane_lo) {
                                            const int32_t v = S[2 * brick_index(tu, nv, ni) + (size_t)r];
                                            if (v < 0) known = false; else s2 = std::max(s2, v + 1);
                                        }
                                    const int x = box_over(BX, tu, nv, ni, pc[q].lane_lo, pc[q].lane_hi);
                                    if (x >= 0) after_pass(x);
                                }
                                if (known) mine = s2;
                            }
                        }
            }
            if (k == H.npass) break;
            // where pass k goes: behind every piece that hands rays to a box of level k
            for (size_t g = 0; g < P.groups.size(); ++g) {
                const std::vector<HybridBox> &BX = boxes[g];
                int &at = H.pass_at[(size_t)half_of_group[g]][(size_t)k];
                if (k > 0) at = std::max(at, H.pass_at[(size_t)half_of_group[g]][(size_t)k - 1]);
                for (int ti = 0; ti < P.nti; ++ti)
                    for (int tv = 0; tv < P.ntv; ++tv)
                        for (int tu = 0; tu < P.ntu; ++tu) {
                            Piece pc[2];
                            const int np = pieces_of(BX, tu, tv, ti, pc);
                            for (int q = 0; q < np; ++q) {
                                bool feeds = false;
                                if (pc[q].lane_hi < 63) feeds = BX[(size_t)pc[q].box].level == k; // lanes that end inside the brick: at a box
                                else if (tu + 1 < P.ntu) { const int x = box_over(BX, tu + 1, tv, ti, 0, 0); feeds = x >= 0 && BX[(size_t)x].level == k; }
                                if (!feeds && tv + 1 < P.ntv) { const int x = box_over(BX, tu, tv + 1, ti, pc[q].lane_lo, pc[q].lane_hi); feeds = x >= 0 && BX[(size_t)x].level == k; }
                                if (!feeds && ti + 1 < P.nti) { const int x = box_over(BX, tu, tv, ti + 1, pc[q].lane_lo, pc[q].lane_hi); feeds = x >= 0 && BX[(size_t)x].level == k; }
                                if (!feeds) continue;
                                const int32_t v = slot[g][2 * brick_index(tu, tv, ti) + (size_t)q];
                                if (v < 0) { free_hybrid(c); return fail(c, FTTE_ERR_STATE, "hybrid plan: a brick that feeds a box waits for a later pass"); }
                                at = std::max(at, v + 1);
                            }
                        }
            }
        }
        for (size_t g = 0; g < P.groups.size(); ++g)
            for (int32_t v : slot[g]) nslots = std::max(nslots, v + 1);
        for (auto &at : H.pass_at) for (int v : at) nslots = std::max(nslots, v);
    }
    // the launch lists of a pipeline: a piece's slot, or (one pass, or slots switched off) a phase per pass, in it the stages of a
    // plain sweep: phase 0 before the first pass of the forests, phase k after pass k - 1
    H.nlist = H.slots ? (size_t)std::max(nslots, 1) : (size_t)(H.npass + 1) * (size_t)per_phase;
    if (!H.slots) for (auto &at : H.pass_at) for (int k = 0; k < H.npass; ++k) at[(size_t)k] = (k + 1) * per_phase;
    const size_t nlist = (size_t)H.nhalves * H.nlist;
    auto list_of = [&](size_t g, int phase, int tu, int tv, int ti, int offset, int q) {
        if (H.slots) return (size_t)half_of_group[g] * H.nlist + (size_t)slot[g][2 * brick_index(tu, tv, ti) + (size_t)q];
        return (size_t)half_of_group[g] * H.nlist + (size_t)phase * (size_t)per_phase + (size_t)(tu + tv + ti + offset);
    };
    std::vector<std::vector<size_t>> first(3 * (size_t)kMaxAcc);
    auto brick_of = [&](const BrickPlan::Group &G, int tu, int tv, int ti) {
        const DirPlan &D0 = P.dirs[G.dirs[0]];
        const int bu = D0.su < 0 ? P.ntu - 1 - tu : tu, bv = D0.sv < 0 ? P.ntv - 1 - tv : tv, bi = D0.si < 0 ? P.nti - 1 - ti : ti;
        return ((size_t)bi * P.ntv + bv) * P.ntu + bu;
    };
    // Two sets of lists in one task array, [whole lists][masked lists]: the masked kernel (a lane range per task)
    // takes every brick of a stage in which some brick is cut by a box -- two launches per stage would run one after the other,
    // and a launch of a few bricks lasts as long as one of many --, the plain kernel the stages without.
    std::vector<uint8_t> cut(nlist, 0);
    for (size_t g = 0; g < P.groups.size(); ++g)
        for (int ti = 0; ti < P.nti; ++ti)
            for (int tv = 0; tv < P.ntv; ++tv)
                for (int tu = 0; tu < P.ntu; ++tu) {
                    Piece pc[2];
                    const int np = pieces_of(boxes[g], tu, tv, ti, pc);
                    for (int q = 0; q < np; ++q)
                        if (pc[q].masked) cut[list_of(g, pc[q].phase, tu, tv, ti, P.groups[g].offset, q)] = 1;
                }
    H.stage_off.assign(2 * nlist + 1, 0);
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const BrickPlan::Group &G = P.groups[g];
        std::vector<size_t> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
        if (F.empty()) F.assign(nb, ~(size_t)0);
        for (int ti = 0; ti < P.nti; ++ti)
            for (int tv = 0; tv < P.ntv; ++tv)
                for (int tu = 0; tu < P.ntu; ++tu) {
                    Piece pc[2];
                    const int np = pieces_of(boxes[g], tu, tv, ti, pc);
                    for (int q = 0; q < np; ++q) {
                        const size_t l = list_of(g, pc[q].phase, tu, tv, ti, G.offset, q);
                        ++H.stage_off[(cut[l] ? nlist : 0) + l + 1];
                        size_t &f = F[brick_of(G, tu, tv, ti)];
                        f = std::min(f, l);
                    }
                }
    }
    for (size_t l = 0; l < 2 * nlist; ++l) H.stage_off[l + 1] += H.stage_off[l];
    P.tasks.resize(H.stage_off[2 * nlist]);
    std::vector<size_t> fill(H.stage_off.begin(), H.stage_off.end() - 1);
    H.brick_updates = 0;
    for (size_t g = 0; g < P.groups.size(); ++g) {
        const BrickPlan::Group &G = P.groups[g];
        const std::vector<size_t> &F = first[(size_t)G.layout * kMaxAcc + G.acc];
        for (int ti = 0; ti < P.nti; ++ti)
            for (int tv = 0; tv < P.ntv; ++tv)
                for (int tu = 0; tu < P.ntu; ++tu) {
                    Piece pc[2];
                    const int np = pieces_of(boxes[g], tu, tv, ti, pc);
                    for (int q = 0; q < np; ++q) {
                        const size_t l = list_of(g, pc[q].phase, tu, tv, ti, G.offset, q);
                        BrickTask T;
                        T.tv = (int16_t)(cut[l] ? tv | (pc[q].box << kBrickBoxShift) : tv);
                        T.group = (int16_t)(cut[l] ? (int)g | (pc[q].lane_hi << kBrickLaneHiShift) : (int)g);
                        T.tu = (int16_t)(uint16_t)(cut[l] ? tu | (pc[q].lane_lo << kBrickLaneLoShift) : tu);
                        // (two pieces of one brick write different lanes of rows that start from zero: either may come first)
                        T.ti = (int16_t)(ti | (l > F[brick_of(G, tu, tv, ti)] ? kBrickAccumulate : 0));
                        P.tasks[fill[(cut[l] ? nlist : 0) + l]++] = T;
                        const int64_t cu = std::max(0, std::min(pc[q].lane_hi, n - 64 * tu - 1) - pc[q].lane_lo + 1), cv = std::min(kBrickRows, n - kBrickRows * tv),
                                      ci = std::min(chunk, n - chunk * ti);
                        H.brick_updates += cu * cv * ci * (int64_t)G.dirs.size();
                    }
                }
    }
    if (P.ntu > kBrickTuMask || P.ntv > kBrickTvMask || (int)P.groups.size() > kBrickGroupMask) { free_hybrid(c); return fail(c, FTTE_ERR_UNSUPPORTED, "hybrid sweep: more than 1023 bricks along a row, or more than 255 groups of directions"); }

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
    // per direction: which pieces the leaves it walked have, and whether they are the forest's (every other leaf: not the forest's)
    std::vector<std::vector<std::pair<int32_t, uint8_t>>> active((size_t)ndir);
    std::vector<std::vector<AmrExport>> exports((size_t)ndir);
    std::vector<std::vector<AmrImport>> imports((size_t)ndir);
    ++c->n_forest_builds;
    static_assert(sizeof(AmrForest::Export) == sizeof(AmrExport), "export records: host and device forms must agree");
    static_assert(sizeof(AmrForest::FineImport) == sizeof(AmrImport), "import records: host and device forms must agree");
    {
        // a thread keeps its forest from direction to direction: after the first, a build touches only what the boxes hold (ftte_amr.h)
        const int nt = std::min(nthreads, ndir);
        std::vector<int> st((size_t)ndir, 0);
        std::vector<std::string> msg((size_t)ndir);
        std::vector<std::vector<int32_t>> seen((size_t)nt); // leaves inside a box of at least one of the thread's directions
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t] {
                AmrForest f;
                for (int d = t; d < ndir; d += nt) {
                    const DirPlan &D = P.dirs[(size_t)d];
                    std::vector<ForestRegion> regions;
                    for (const HybridBox &B : boxes[(size_t)group_of[(size_t)d]]) regions.push_back(B.R);
                    st[(size_t)d] = build_forest_regions(c->tree, D.phi, D.theta, D.izone, c->box, &f, &msg[(size_t)d], regions);
                    if (st[(size_t)d]) return;
                    const size_t nact = f.order.size();
                    rec[(size_t)d].resize(std::max<size_t>(nact, 1));
                    for (size_t q = 0; q < nact; ++q) {
                        const int32_t sg = f.order[q];
                        SegRec &R = rec[(size_t)d][q];
                        R.seg = sg; R.up = f.up[sg]; R.up2 = f.up2[sg];
                        R.at = f.up[sg] == AmrForest::kImport ? f.import_at[sg] : 0;
                        R.dpath = f.dpath[sg];
                    }
                    active[(size_t)d].reserve(f.visited.size());
                    for (int32_t q : f.visited) {
                        active[(size_t)d].push_back({q, (uint8_t)((f.up[3 * (size_t)q + 1] != AmrForest::kInactive ? 1 : 0) | (f.up[3 * (size_t)q + 2] != AmrForest::kInactive ? 2 : 0) |
                                                                  (f.inside[(size_t)q] ? 0 : 4))});
                        if (f.inside[(size_t)q]) seen[(size_t)t].push_back(q);
                    }
                    ftte_ctx::HybridPlan::Dir &HD = H.dirs[(size_t)d];
                    HD.depth_off = f.depth_off;
                    HD.pass_first = f.pass_first;
                    HD.export_first = f.export_first;
                    HD.nexports = (int64_t)f.exports.size();
                    exports[(size_t)d].resize(f.exports.size());
                    if (!f.exports.empty()) std::memcpy(exports[(size_t)d].data(), f.exports.data(), sizeof(AmrExport) * f.exports.size());
                    imports[(size_t)d].resize(f.fine_imports.size());
                    if (!f.fine_imports.empty()) std::memcpy(imports[(size_t)d].data(), f.fine_imports.data(), sizeof(AmrImport) * f.fine_imports.size());
                }
            });
        for (auto &th : pool) th.join();
        for (int d = 0; d < ndir; ++d)
            if (st[(size_t)d]) { const std::string m = msg[(size_t)d]; const int code = st[(size_t)d]; free_hybrid(c); return fail(c, code, "direction " + std::to_string(d) + ": " + m); }
        for (const auto &list : seen) for (int32_t q : list) in_any[(size_t)q] = 1;
    }
    std::vector<int32_t> cells, place((size_t)ncell, -1);
    std::vector<std::vector<uint8_t>> bytes((size_t)ndir); // the activity bytes of the leaves in `cells`, per direction
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
                    for (AmrImport &X : imports[(size_t)d]) {
                        if ((X.up >= 0 && place[(size_t)(X.up / 3)] < 0) || (X.up2 >= 0 && place[(size_t)(X.up2 / 3)] < 0)) { bad[(size_t)d] = 1; break; }
                        X.up = renumber(X.up); X.up2 = renumber(X.up2);
                    }
                    std::fill(compact.begin(), compact.end(), (uint8_t)4);
                    for (const auto &a : active[(size_t)d]) if (place[(size_t)a.first] >= 0) compact[(size_t)place[(size_t)a.first]] = a.second;
                    bytes[(size_t)d] = compact;
                }
            });
        for (auto &th : pool) th.join();
        for (int d = 0; d < ndir; ++d)
            if (bad[(size_t)d]) { free_hybrid(c); return fail(c, FTTE_ERR_STATE, "hybrid plan: a forest segment lies outside every box"); }
    }
    for (int d = 0; d < ndir; ++d) {
        ftte_ctx::HybridPlan::Dir &D = H.dirs[(size_t)d];
        FTTE_HIP(c, hipMalloc((void **)&D.rec, sizeof(SegRec) * rec[(size_t)d].size()));
        FTTE_HIP(c, hipMalloc((void **)&D.active, std::max<size_t>(bytes[(size_t)d].size(), 1)));
        FTTE_HIP(c, hipMalloc((void **)&D.exports, sizeof(AmrExport) * std::max<size_t>(exports[(size_t)d].size(), 1)));
        FTTE_HIP(c, hipMemcpy(D.rec, rec[(size_t)d].data(), sizeof(SegRec) * rec[(size_t)d].size(), hipMemcpyHostToDevice));
        if (!bytes[(size_t)d].empty()) FTTE_HIP(c, hipMemcpy(D.active, bytes[(size_t)d].data(), bytes[(size_t)d].size(), hipMemcpyHostToDevice));
        if (!exports[(size_t)d].empty())
            FTTE_HIP(c, hipMemcpy(D.exports, exports[(size_t)d].data(), sizeof(AmrExport) * exports[(size_t)d].size(), hipMemcpyHostToDevice));
        D.nimports = (int64_t)imports[(size_t)d].size();
        if (D.nimports) {
            FTTE_HIP(c, hipMalloc((void **)&D.imports, sizeof(AmrImport) * imports[(size_t)d].size()));
            FTTE_HIP(c, hipMemcpy(D.imports, imports[(size_t)d].data(), sizeof(AmrImport) * imports[(size_t)d].size(), hipMemcpyHostToDevice));
        }
        std::vector<SegRec>().swap(rec[(size_t)d]);
        std::vector<uint8_t>().swap(bytes[(size_t)d]);
    }
    if (FN.active) {
        // fine cell (storage order inside the block) -> leaf: the children of a refined base cell follow each other in the cell array
        // in storage order 4 (a - 1) + 2 (b - 1) + (c - 1) (equiSources.f90:4044-4079)
        const int nf = FN.n;
        std::vector<int32_t> map((size_t)nf * nf * nf);
        for (int a = 0; a < nf; ++a)
            for (int b = 0; b < nf; ++b)
                for (int d = 0; d < nf; ++d) {
                    const int32_t node = (int32_t)(((int64_t)(FN.lo[0] - 1 + a / 2) * n + (FN.lo[1] - 1 + b / 2)) * n + (FN.lo[2] - 1 + d / 2));
                    map[((size_t)a * nf + b) * nf + d] = c->tree.leaf[(size_t)(c->tree.child0[(size_t)node] + 4 * (a % 2) + 2 * (b % 2) + (d % 2))];
                }
        const BrickPlan &Q = FN.plan;
        FTTE_HIP(c, hipMalloc((void **)&FN.leaf_of_fine, sizeof(int32_t) * map.size()));
        FTTE_HIP(c, hipMemcpy(FN.leaf_of_fine, map.data(), sizeof(int32_t) * map.size(), hipMemcpyHostToDevice));
        FTTE_HIP(c, hipMalloc((void **)&FN.layers, sizeof(LayerRec) * Q.layers.size()));
        FTTE_HIP(c, hipMemcpy(FN.layers, Q.layers.data(), sizeof(LayerRec) * Q.layers.size(), hipMemcpyHostToDevice));
        FTTE_HIP(c, hipMalloc((void **)&FN.tasks, sizeof(BrickTask) * Q.tasks.size()));
        FTTE_HIP(c, hipMemcpy(FN.tasks, Q.tasks.data(), sizeof(BrickTask) * Q.tasks.size(), hipMemcpyHostToDevice));
        FTTE_HIP(c, hipMalloc((void **)&FN.groups, sizeof(BrickGroup) * Q.groups.size()));
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
    // (decided before any device state is touched: the forest path then builds its own plan and scratch, not both)
    if (c->nnu > 96) return FTTE_OK; // the cell-major copy of kappa is what the level kernel reads here: leave it to the forest path
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
    const int emit = c->emit_mode;
    if (emit && c->base_emis_cap < per_base) {
        for (int l = 0; l < 3; ++l) if (c->base_emis[l]) { FTTE_HIP(c, hipFree(c->base_emis[l])); c->base_emis[l] = nullptr; }
        for (int l = 0; l < 3; ++l) FTTE_HIP(c, hipMalloc((void **)&c->base_emis[l], sizeof(double) * per_base));
        c->base_emis_cap = per_base;
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
    ftte_ctx::HybridPlan::Fine &FN = H.fine;
    // a direction's face block: the base bricks' rings, then (a fine block swept by bricks) the fine bricks' own
    const int64_t face_elems = P.face_elems + (FN.active ? FN.plan.face_elems : 0);
    const size_t face_need = (size_t)ndir * nnu * (size_t)face_elems;
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
            G[g].emis = emit ? c->base_emis[Hg.layout] : nullptr;
            G[g].J = c->acc[Hg.layout][Hg.acc];
            G[g].org = D0.org; G[g].si = D0.si; G[g].sv = D0.sv; G[g].su = D0.su;
            G[g].ndir = (int)Hg.dirs.size();
            for (size_t q = 0; q < Hg.dirs.size(); ++q) {
                const int d = Hg.dirs[q];
                G[g].dir[q].layers = c->d_blayers + P.dirs[d].layer_off;
                G[g].dir[q].faces = c->d_faces + (size_t)d * nnu * (size_t)face_elems;
                G[g].dir[q].w = P.dirs[d].w;
            }
        }
        FTTE_HIP(c, hipMemcpy(c->d_bgroups, G.data(), sizeof(BrickGroup) * G.size(), hipMemcpyHostToDevice)); c->bgroups_sent.clear();
    }
    if (FN.active) {
        // the fine block's own arrays -- opacities in the three layouts, an accumulator per group -- and its group records
        const BrickPlan &Q = FN.plan;
        const size_t per_fine = (size_t)nnu * (size_t)FN.n * FN.n * FN.n;
        if (c->fine_kappa_cap < per_fine) {
            for (int l = 0; l < 3; ++l) if (c->fine_kappa[l]) { FTTE_HIP(c, hipFree(c->fine_kappa[l])); c->fine_kappa[l] = nullptr; }
            for (int l = 0; l < 3; ++l) FTTE_HIP(c, hipMalloc((void **)&c->fine_kappa[l], sizeof(double) * per_fine));
            c->fine_kappa_cap = per_fine;
        }
        if (c->fine_acc_cap < per_fine) {
            for (int l = 0; l < 3; ++l)
                for (int a = 0; a < kMaxAcc; ++a) if (c->fine_acc[l][a]) { FTTE_HIP(c, hipFree(c->fine_acc[l][a])); c->fine_acc[l][a] = nullptr; }
            c->fine_acc_cap = per_fine;
        }
        for (int l = 0; l < 3; ++l)
            for (int a = 0; a < Q.nacc[l]; ++a)
                if (!c->fine_acc[l][a]) FTTE_HIP(c, hipMalloc((void **)&c->fine_acc[l][a], sizeof(double) * c->fine_acc_cap));
        if (emit && c->fine_emis_cap < per_fine) {
            for (int l = 0; l < 3; ++l) if (c->fine_emis[l]) { FTTE_HIP(c, hipFree(c->fine_emis[l])); c->fine_emis[l] = nullptr; }
            for (int l = 0; l < 3; ++l) FTTE_HIP(c, hipMalloc((void **)&c->fine_emis[l], sizeof(double) * per_fine));
            c->fine_emis_cap = per_fine;
        }
        std::vector<BrickGroup> G(Q.groups.size());
        std::memset(G.data(), 0, sizeof(BrickGroup) * G.size());
        for (size_t g = 0; g < Q.groups.size(); ++g) {
            const BrickPlan::Group &Hg = Q.groups[g];
            const DirPlan &D0 = Q.dirs[Hg.dirs[0]];
            G[g].kappa = c->fine_kappa[Hg.layout];
            G[g].emis = emit ? c->fine_emis[Hg.layout] : nullptr;
            G[g].J = c->fine_acc[Hg.layout][Hg.acc];
            G[g].org = D0.org; G[g].si = D0.si; G[g].sv = D0.sv; G[g].su = D0.su;
            G[g].ndir = (int)Hg.dirs.size();
            for (size_t q = 0; q < Hg.dirs.size(); ++q) {
                const int d = Hg.dirs[q];
                G[g].dir[q].layers = FN.layers + Q.dirs[d].layer_off;
                G[g].dir[q].faces = c->d_faces + (size_t)d * nnu * (size_t)face_elems + (size_t)FN.face_base; // behind the base bricks' rings
                G[g].dir[q].w = Q.dirs[d].w;
            }
        }
        FTTE_HIP(c, hipMemcpy(FN.groups, G.data(), sizeof(BrickGroup) * G.size(), hipMemcpyHostToDevice));
    }
    if ((rc = ensure(c, &c->d_uvb, &c->d_uvb_cap, (size_t)nnu))) return rc;
    FTTE_HIP(c, hipMemcpy(c->d_uvb, uvb, sizeof(double) * nnu, hipMemcpyHostToDevice)); c->uvb_sent.clear();

    // forest scratch: as forest_sweep, for the leaves of the plan's list only
    const size_t per_dir = (size_t)3 * (size_t)std::max<int64_t>(H.ncells, 1) * nnu;
    // (numbered by the list, a direction's scratch is small: every direction at once, where the memory is there)
    const int most = c->forest_batch > 0 ? c->forest_batch : 1024;
    int batch = std::max(1, std::min(ndir, most));
    if (c->amr_scratch_cap < per_dir * (size_t)batch) {
        if (c->amr_Iout) { FTTE_HIP(c, hipFree(c->amr_Iout)); c->amr_Iout = nullptr; }
        if (c->amr_mean) { FTTE_HIP(c, hipFree(c->amr_mean)); c->amr_mean = nullptr; }
        c->amr_scratch_cap = 0;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)batch, (size_t)(0.6 * (double)free_b) / (2 * sizeof(double) * per_dir)));
        FTTE_HIP(c, hipMalloc((void **)&c->amr_Iout, sizeof(double) * per_dir * (size_t)batch));
        FTTE_HIP(c, hipMalloc((void **)&c->amr_mean, sizeof(double) * per_dir * (size_t)batch));
        c->amr_scratch_cap = per_dir * (size_t)batch;
    } else batch = (int)std::min<size_t>((size_t)most, c->amr_scratch_cap / per_dir);
    // Several passes keep every direction's scratch from pass to pass, and pipelines whose launch lists go by slot have each their
    // own place for a pass (pass_at[pipeline][pass]): with fewer directions resident than the sweep has, all pipelines' forests
    // would have to go in one run at ONE place, in front of bricks of the other pipelines that feed them or behind bricks that
    // read what they export.  Both are left to the forest path for the whole tree.
    if (batch < ndir && (H.npass > 1 || (H.slots && H.nhalves > 1) || FN.active)) return FTTE_OK; // (a fine block's forests come in two passes)
    if ((rc = ensure(c, &c->amr_kappa, &c->amr_kappa_cap, (size_t)nnu * (size_t)std::max<int64_t>(H.ncells, 1)))) return rc;
    if (!c->kappa_ready[3] || c->amr_kappa_form != 1) {
        if (launch_cell_major(c->kappa[0], c->amr_kappa, ncell, nnu, stream, H.cells, (long)H.ncells)) return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        c->kappa_ready[3] = true; c->amr_kappa_form = 1;
    }
    if (emit) { // the emissivity / source function of the boxes' leaves, cell-major like their opacities; new every iteration
        if ((rc = ensure(c, &c->amr_emis, &c->amr_emis_cap, (size_t)nnu * (size_t)std::max<int64_t>(H.ncells, 1)))) return rc;
        if (launch_cell_major(c->emis[0], c->amr_emis, ncell, nnu, stream, H.cells, (long)H.ncells)) return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        c->emis_ready[3] = false; // (the forest path for the whole tree keeps every leaf there)
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
    static const ftte_consts kMath = FTTE_CONSTS_INIT;
    const size_t masked_lists = (size_t)H.nhalves * H.nlist;
    auto brick_stages = [&](int half, size_t from, size_t to, hipStream_t q) -> int {
      for (size_t l = from; l < to; ++l)
        for (int masked = 0; masked < 2; ++masked) { // the stage's whole bricks, then those a box cuts through
            const size_t *off = &H.stage_off[(masked ? masked_lists : 0) + (size_t)half * H.nlist];
            if (off[l + 1] == off[l]) continue;
            BrickLaunch L;
            std::memset(&L, 0, sizeof L);
            L.groups = c->d_bgroups;
            L.tasks = c->d_btasks + off[l];
            L.uvb = c->d_uvb;
            L.group_stride = nbase;
            L.face_stride = face_elems;
            L.vface_off = P.vface_off; L.iface_off = P.iface_off; L.uqface_off = P.uqface_off;
            L.n = n; L.ntasks = (int)(off[l + 1] - off[l]); L.nnu = nnu; L.nu0 = 0; L.chunk = P.chunk;
            L.up = P.up; L.vp = P.vp; L.uw = P.uw; L.ut = P.ut; L.nslot = P.nslot;
            L.emit = emit;
            L.math = kMath;
            const int lrc = launch_brick(L, P.max_dirs, c->brick_waves, q, masked != 0);
            if (lrc) return fail(c, lrc == -1 ? FTTE_ERR_ARG : FTTE_ERR_NO_DEVICE, "brick kernel launch failed");
        }
      return FTTE_OK;
    };

    // the fine block's own sweep, pipeline `half`: its rays in from the forest, then its bricks stage by stage
    int64_t most_imports = 0;
    for (const auto &D : H.dirs) most_imports = std::max(most_imports, D.nimports);
    auto fine_sweep = [&](int half, hipStream_t q, const ForestRun &R, AmrLevelRec A) -> int {
        const ForestRun::Batch &B = R.batches[0];
        A.dir = c->d_amr_dirs + R.dir_at + (size_t)B.d0;
        A.ndir = B.nb;
        if (launch_amr_fine_import(A, most_imports, q)) return fail(c, FTTE_ERR_NO_DEVICE, "fine import kernel launch failed");
        const BrickPlan &Q = FN.plan;
        for (int st = 0; st < FN.nstages; ++st) {
            const size_t l = (size_t)half * (size_t)FN.nstages + (size_t)st;
            if (FN.stage_off[l + 1] == FN.stage_off[l]) continue;
            BrickLaunch L;
            std::memset(&L, 0, sizeof L);
            L.groups = FN.groups;
            L.tasks = FN.tasks + FN.stage_off[l];
            L.uvb = c->d_uvb;
            L.group_stride = (int64_t)FN.n * FN.n * FN.n;
            L.face_stride = face_elems;
            L.vface_off = Q.vface_off; L.iface_off = Q.iface_off; L.uqface_off = Q.uqface_off;
            L.n = FN.n; L.ntasks = (int)(FN.stage_off[l + 1] - FN.stage_off[l]); L.nnu = nnu; L.nu0 = 0; L.chunk = Q.chunk;
            L.up = Q.up; L.vp = Q.vp; L.uw = Q.uw; L.ut = Q.ut; L.nslot = Q.nslot;
            L.sub = 1;
            L.emit = emit;
            L.math = kMath;
            const int lrc = launch_brick(L, Q.max_dirs, c->brick_waves, q, false);
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
    A.kappa = c->amr_kappa; A.emis = emit ? c->amr_emis : nullptr;
    A.group_stride = 1; A.cell_stride = nnu;
    A.emit = emit;
    A.uvb = c->d_uvb;
    A.ncell = ncell; A.nnu = nnu;
    A.cells = H.cells; A.ncells = H.ncells;
    A.face_stride = face_elems;
    A.math = kMath;
    std::vector<ForestRun> runs;
    {
        std::vector<std::vector<ForestDirHost>> sets((size_t)nh);
        std::vector<int> slot0((size_t)nh, 0);
        for (int h = 0; h < H.nhalves; ++h) {
            const int to = nh > 1 ? h : 0;
            for (int d : H.half_dirs[(size_t)h]) {
                const ftte_ctx::HybridPlan::Dir &D = H.dirs[(size_t)d];
                sets[(size_t)to].push_back(ForestDirHost{D.rec, D.active, P.dirs[(size_t)d].w, c->d_faces + (size_t)d * nnu * (size_t)face_elems,
                                                         D.exports, D.nexports, &D.depth_off, &D.pass_first, &D.export_first, D.imports, D.nimports});
            }
        }
        for (int r = 1; r < nh; ++r) slot0[(size_t)r] = slot0[(size_t)r - 1] + (int)sets[(size_t)r - 1].size();
        // one batch per pipeline when they run side by side (their scratch must not overlap), else `batch` directions at a time
        if ((rc = prepare_forests(c, stream, sets, slot0, nh > 1 ? ndir : batch, per_dir, &runs))) return rc;
    }
    // (FTTE_HYBRID_TIMELINE: events at the phase boundaries of every pipeline, printed when the sweep is over -- a timeline without
    // a tracer, whose own cost per launch changes what overlaps what)
    static const bool timeline = std::getenv("FTTE_HYBRID_TIMELINE") != nullptr;
    struct Mark { hipEvent_t e; int pipe; const char *what; };
    std::vector<Mark> marks;
    auto mark = [&](hipStream_t q, int pipe, const char *what) {
        if (!timeline) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, q);
        marks.push_back({e, pipe, what});
    };
    // ---- the launches of one sweep: the same sequence every iteration while plan and buffers stay what they are
    auto issue = [&]() -> int {
        // ---- opacity of the base cells in the three layouts; accumulators and J start from zero
        if (launch_base_cells(c->kappa[0], c->d_leaf_of_base, c->base_kappa[0], (long)nbase, (long)ncell, nnu, stream))
            return fail(c, FTTE_ERR_NO_DEVICE, "base-cell kernel launch failed");
        for (int l = 1; l < 3; ++l)
            if (P.nacc[l] && launch_to_layout(l, c->base_kappa[0], c->base_kappa[l], n, nnu, (long)nbase, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        if (emit) { // and their emissivity / source function
            if (launch_base_cells(c->emis[0], c->d_leaf_of_base, c->base_emis[0], (long)nbase, (long)ncell, nnu, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "base-cell kernel launch failed");
            for (int l = 1; l < 3; ++l)
                if (P.nacc[l] && launch_to_layout(l, c->base_emis[0], c->base_emis[l], n, nnu, (long)nbase, stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
        }
        if (FN.active) { // the fine block's opacities, dense in its own storage order, then in the layouts its groups march through
            const long nfine = (long)FN.n * FN.n * FN.n;
            if (launch_base_cells(c->kappa[0], FN.leaf_of_fine, c->fine_kappa[0], nfine, (long)ncell, nnu, stream))
                return fail(c, FTTE_ERR_NO_DEVICE, "base-cell kernel launch failed");
            for (int l = 1; l < 3; ++l)
                if (FN.plan.nacc[l] && launch_to_layout(l, c->fine_kappa[0], c->fine_kappa[l], FN.n, nnu, nfine, stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            if (emit) { // and its emissivity / source function
                if (launch_base_cells(c->emis[0], FN.leaf_of_fine, c->fine_emis[0], nfine, (long)ncell, nnu, stream))
                    return fail(c, FTTE_ERR_NO_DEVICE, "base-cell kernel launch failed");
                for (int l = 1; l < 3; ++l)
                    if (FN.plan.nacc[l] && launch_to_layout(l, c->fine_emis[0], c->fine_emis[l], FN.n, nnu, nfine, stream))
                        return fail(c, FTTE_ERR_NO_DEVICE, "layout kernel launch failed");
            }
        }
        for (int l = 0; l < 3; ++l)
            for (int s = 0; s < P.nacc[l]; ++s) FTTE_HIP(c, hipMemsetAsync(c->acc[l][s], 0, sizeof(double) * per_base, stream));
        FTTE_HIP(c, hipMemsetAsync(J_dev, 0, sizeof(double) * (size_t)nnu * ncell, stream));

        mark(stream, -1, "layouts and zeroing done");
        if (nh > 1) {
            FTTE_HIP(c, hipEventRecord(c->ev_fork, stream));
            for (int r = 1; r < nh; ++r) FTTE_HIP(c, hipStreamWaitEvent(qs[r], c->ev_fork, 0));
        }
        // Issued list by list, alternating between the streams, so that none waits for the host to finish with the others.  In
        // front of list pass_at[k] of a pipeline its forests of pass k; behind the last pass the means into J.
        for (size_t l = 0; l <= H.nlist; ++l) {
            for (int h = 0; h < H.nhalves; ++h) {
                const int r = nh > 1 ? h : 0;
                for (int pass = 0; pass < H.npass; ++pass) {
                    if ((size_t)H.pass_at[(size_t)h][(size_t)pass] != l || (nh == 1 && h > 0)) continue;
                    hipEvent_t before = (nh > 1 && r > 0) ? c->ev_combine[r - 1] : nullptr, after = (nh > 1 && r + 1 < nh) ? c->ev_combine[r] : nullptr;
                    if (FN.active) { // the forest before the fine block's bricks, those, the forest behind them; the means below
                        mark(qs[r], r, "bricks before the box done");
                        if ((rc = launch_forest_pass(c, qs[r], runs[(size_t)r], 0, 0, A))) return rc;
                        mark(qs[r], r, "forest before the fine block done");
                        if ((rc = fine_sweep(h, qs[r], runs[(size_t)r], A))) return rc;
                        mark(qs[r], r, "fine bricks done");
                        if ((rc = launch_forest_pass(c, qs[r], runs[(size_t)r], 0, 1, A))) return rc;
                        mark(qs[r], r, "forest behind the fine block done");
                        continue;
                    }
                    mark(qs[r], r, "bricks before a forest pass done");
                    if (H.npass == 1) { // one pass: batch by batch, each with its means
                        if ((rc = launch_forests(c, qs[r], runs[(size_t)r], A, J_dev, false, false, before, after))) return rc;
                        continue;
                    }
                    // (several passes: every direction of the run is resident, one batch)
                    if ((rc = launch_forest_pass(c, qs[r], runs[(size_t)r], 0, (size_t)pass, A))) return rc;
                }
                if (l < H.nlist && (rc = brick_stages(h, l, l + 1, qs[r]))) return rc;
            }
        }
        // several passes: the means into J when everything is issued, pipeline after pipeline (a pipeline's last pass may come
        // earlier or later than another's, and the events that order the additions must be recorded before they are waited for)
        for (int r = 0; r < nh; ++r) mark(qs[r], r, "last bricks done");
        if (H.npass > 1 || FN.active)
            for (int r = 0; r < nh; ++r) {
                if (nh > 1 && r > 0) FTTE_HIP(c, hipStreamWaitEvent(qs[r], c->ev_combine[r - 1], 0));
                if ((rc = launch_forest_combine(c, qs[r], runs[(size_t)r], 0, A, J_dev, false))) return rc;
                if (nh > 1 && r + 1 < nh) FTTE_HIP(c, hipEventRecord(c->ev_combine[r], qs[r]));
            }
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
        if (FN.active) { // ... and J of the fine block's cells += what its bricks stored
            const double *accs[3 * kMaxAcc];
            int layouts[3 * kMaxAcc], count = 0;
            for (int l = 0; l < 3; ++l)
                for (int s = 0; s < FN.plan.nacc[l]; ++s) { accs[count] = c->fine_acc[l][s]; layouts[count++] = l; }
            if (count && launch_merge(accs, layouts, count, J_dev, FN.n, nnu, (long)FN.n * FN.n * FN.n, true, stream, FN.leaf_of_fine, (long)ncell))
                return fail(c, FTTE_ERR_NO_DEVICE, "merge kernel launch failed");
        }
        return FTTE_OK;
    };

    // The sequence is hundreds of short launches on up to three streams (more with several passes).  Option "graph" = 1 captures it
    // once into a hipGraph -- the streams' forks and joins become dependencies of the graph -- and replays it while the plan, J and
    // every buffer and table the launches name stay the same.  Measured on ROCm 7.2 / MI355X the replay is SLOWER than issuing the
    // launches (configs[3]: 15.8 against 12.2 ms; 8 clusters in 5 passes: 23.9 against 15.1 ms), so it is off by default.
    std::vector<uintptr_t> sig = {(uintptr_t)J_dev, (uintptr_t)stream, (uintptr_t)nnu, (uintptr_t)nh, (uintptr_t)c->kappa[0], (uintptr_t)c->amr_kappa,
                                  (uintptr_t)c->d_faces, (uintptr_t)c->amr_Iout, (uintptr_t)c->amr_mean, (uintptr_t)c->d_bgroups, (uintptr_t)c->d_btasks,
                                  (uintptr_t)c->d_amr_dirs, (uintptr_t)c->d_amr_tables, (uintptr_t)c->d_uvb, (uintptr_t)c->d_leaf_of_base, (uintptr_t)H.cells,
                                  (uintptr_t)c->brick_waves, (uintptr_t)emit, (uintptr_t)c->emis[0], (uintptr_t)c->amr_emis, (uintptr_t)c->forest_fuse,
                                  (uintptr_t)c->base_emis[0], (uintptr_t)c->base_emis[1], (uintptr_t)c->base_emis[2]};
    for (int l = 0; l < 3; ++l) { sig.push_back((uintptr_t)c->base_kappa[l]); for (int s2 = 0; s2 < P.nacc[l]; ++s2) sig.push_back((uintptr_t)c->acc[l][s2]); }
    if (FN.active) for (int l = 0; l < 3; ++l) { sig.push_back((uintptr_t)c->fine_kappa[l]); sig.push_back((uintptr_t)c->fine_emis[l]); for (int s2 = 0; s2 < FN.plan.nacc[l]; ++s2) sig.push_back((uintptr_t)c->fine_acc[l][s2]); }
    FTTE_HIP(c, hipEventRecord(Tm.start, stream));
    bool replayed = false;
    if (c->use_graph && H.graph_exec && H.graph_sig == sig) {
        if (hipGraphLaunch(H.graph_exec, stream) == hipSuccess) replayed = true;
        else { (void)hipGetLastError(); drop_graph(H); }
    }
    if (!replayed && c->use_graph) {
        drop_graph(H);
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
            const int irc = issue();
            const hipError_t e = hipStreamEndCapture(stream, &graph);
            if (irc == FTTE_OK && e == hipSuccess && graph && hipGraphInstantiate(&H.graph_exec, graph, nullptr, nullptr, 0) == hipSuccess &&
                hipGraphLaunch(H.graph_exec, stream) == hipSuccess) {
                H.graph_sig = sig;
                replayed = true;
            } else {
                (void)hipGetLastError();
                drop_graph(H);
                c->use_graph = 0; // this runtime or this sequence does not capture: launches one by one from now on
            }
            if (graph) (void)hipGraphDestroy(graph);
        } else { (void)hipGetLastError(); c->use_graph = 0; }
    }
    if (!replayed && (rc = issue())) return rc;
    FTTE_HIP(c, hipEventRecord(Tm.stop, stream));
    if (timeline && !marks.empty()) {
        (void)hipEventSynchronize(Tm.stop);
        for (const Mark &m : marks) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, Tm.start, m.e);
            std::fprintf(stderr, "[ftte] hybrid timeline: %8.3f ms  pipeline %2d  %s\n", ms, m.pipe, m.what);
            (void)hipEventDestroy(m.e);
        }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, Tm.start, Tm.stop);
        std::fprintf(stderr, "[ftte] hybrid timeline: %8.3f ms  sweep done (means and merges in)\n", ms);
    }
    c->timing_used = 1;
    *done = true;
    return mark_sweep(c, stream);
}


} // namespace ftte
