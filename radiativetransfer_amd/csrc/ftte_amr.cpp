// ftte_amr.cpp -- see ftte_amr.h.  Host-side only; O(leaves) per direction, done once per (tree, direction
// list) and cached by the context: the reference's tree is static over a whole run, while it re-links every
// leaf for every direction in every iteration (equiSources.f90:1557-1566).
#include "ftte_amr.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>

#include "ftte_geometry.h"

namespace ftte {

std::string AmrTree::build(int n_, int64_t ncell_, const int32_t *lv)
{
    n = n_;
    ncell = ncell_;
    max_level = 0;
    const int64_t nbase = (int64_t)n * n * n;
    parent.assign(nbase, -1);
    child0.assign(nbase, -1);
    leaf.assign(nbase, -1);
    level.assign(nbase, 0);
    int64_t cur = 0;
    std::string err;
    // createFullyThreadedStructure, readCellArray.f90:154-187: a cell whose list entry is deeper than itself is
    // refined and its eight children consume the following entries
    std::function<void(int32_t, int)> grow = [&](int32_t node, int depth) {
        if (!err.empty()) return;
        if (cur >= ncell) { err = "error in levels: level list ends inside a refined cell"; return; }
        const int l = lv[cur];
        if (l == depth) {
            leaf[node] = (int32_t)cur++;
        } else if (l > depth) {
            if (depth + 1 > 60) { err = "error in levels: more than 60 levels"; return; }
            const int32_t c0 = (int32_t)parent.size();
            child0[node] = c0;
            for (int c = 0; c < 8; ++c) {
                parent.push_back(node);
                child0.push_back(-1);
                leaf.push_back(-1);
                level.push_back((int8_t)(depth + 1));
            }
            max_level = std::max(max_level, depth + 1);
            for (int c = 0; c < 8; ++c) grow(c0 + c, depth + 1);
        } else {
            err = "error in levels: level list is not a depth-first leaf list";
        }
    };
    for (int64_t b = 0; b < nbase && err.empty(); ++b) grow((int32_t)b, 0);
    if (err.empty() && cur != ncell) err = "error in levels: level list longer than the tree it describes";
    return err;
}

// where the top-ending segment of `b` leaves the cell = entry point of the (sub-)layer above
// (equiSources.f90:1507-1522, transportRoutinesModule.f90:167-182)
static int advance_entry_point(const ftte_pattern &b, double phi, double theta, double *x0, double *y0)
{
    if (b.xy_top == 1) {
        *x0 = b.xy_x0 + std::cos(phi) / std::tan(theta);
        *y0 = b.xy_y0 + std::sin(phi) / std::tan(theta);
    } else if (b.xy_top == 3) {
        *x0 = b.xz_x0 + b.xz_len * std::cos(theta) * std::cos(phi);
        *y0 = b.xz_len * std::cos(theta) * std::sin(phi);
    } else if (b.xy_top == 2) {
        *x0 = b.yz_len * std::cos(theta) * std::cos(phi);
        *y0 = b.yz_y0 + b.yz_len * std::cos(theta) * std::sin(phi);
    } else return 1;
    return (*x0 > 1.0 || *y0 > 1.0) ? 1 : 0;
}

int sub_layer_patterns(const ftte_pattern &pp, double phi, double theta, ftte_pattern *lo, ftte_pattern *hi)
{
    std::memset(lo, 0, sizeof *lo);
    std::memset(hi, 0, sizeof *hi);
    lo->xy_x0 = pp.xy_x0 < 0.5 ? 2.0 * pp.xy_x0 : 2.0 * pp.xy_x0 - 1.0;
    lo->xy_y0 = pp.xy_y0 < 0.5 ? 2.0 * pp.xy_y0 : 2.0 * pp.xy_y0 - 1.0;
    int bad = set_pattern(lo, phi, theta) ? 1 : 0;
    if (advance_entry_point(*lo, phi, theta, &hi->xy_x0, &hi->xy_y0) || set_pattern(hi, phi, theta)) bad = 1;
    return bad;
}

namespace {

constexpr int kBrickRowsHost = 8; // kBrickRows of ftte_internal.h

struct PatNode {
    ftte_pattern p;
    int32_t sub[2]; // patterns of the lower / upper sub-layer of a refined cell carrying this pattern
};

struct Builder {
    const AmrTree &T;
    AmrForest &F;
    int izone;
    double phi, theta;
    int cs[2][2][2]; // storage child index of the sweep child (i,j,k)
    std::vector<PatNode> pats;
    std::vector<int32_t> &node_pat;
    std::vector<int32_t> &depth;
    int32_t anc[64];
    int seq[64][3];
    int status = 0;
    std::string err;
    const ForestRegion *reg = nullptr;
    // fine blocks swept by bricks of their own (ForestRegion::has_fine): which nodes are their leaves and where in the block (fine
    // sweep coordinates, 1-based); which segments of the forest come after the blocks' bricks
    std::vector<uint8_t> &is_hole, &late;
    std::vector<int16_t> &fine_at; // [3 nodes]

    Builder(const AmrTree &t, AmrForest &f)
        : T(t), F(f), node_pat(f.scratch.node_pat), depth(f.scratch.depth), is_hole(f.scratch.is_hole), late(f.scratch.late), fine_at(f.scratch.fine_at) {}

    // Where the ray that enters fine cell (fi, fj, fk) [fine sweep frame of the region's block, 1-based; n_f + 1 on an axis: the
    // position just behind the block] through `face` waits in the direction's face block: what the fine bricks read at the block's
    // upstream faces and write at its downstream ones (BrickLaunch::sub; the same elements as brick_kernel's u_in/v_in/i_in)
    int32_t fine_element(int fi, int fj, int fk, int face) const
    {
        const ForestRegion::FineFaces &Q = reg->fine;
        const int u = reg->u_is_k ? fk : fj, v = reg->u_is_k ? fj : fk;
        const int ti = (fi - 1) / Q.chunk, il = (fi - 1) % Q.chunk, tu = (u - 1) / 64, tv = (v - 1) / kBrickRowsHost;
        if (face == 0) return (int32_t)(Q.base + Q.iface_off + ((int64_t)(ti % Q.nslot) * Q.vp + (v - 1)) * Q.up + (u - 1));
        const bool along_u = (face == 2) == reg->u_is_k;
        if (along_u) {
            const int64_t ring = tu > 0 ? tu - 1 : Q.ntu;
            return (int32_t)(Q.base + ((ring * Q.nslot + ti % Q.nslot) * Q.chunk + il) * ((int64_t)Q.ntv * Q.ut) + (int64_t)Q.ut * tv + (v - 1) % kBrickRowsHost);
        }
        const int64_t ring = tv > 0 ? tv - 1 : Q.ntv;
        return (int32_t)(Q.base + Q.vface_off + ((ring * Q.nslot + ti % Q.nslot) * Q.chunk + il) * Q.up + (u - 1));
    }

    // Where the ray that enters base cell (i, j, k) [sweep frame] through `face` (0: from layer i-1, 1: from j-1, 2: from k-1)
    // waits in the direction's face block when the cell on the other side belongs to a brick: the element the brick kernel
    // writes (uout / vout / the chunk's top) and reads (uin / vin / the chunk's bottom), ftte_brick.hip.
    int32_t face_element(int i, int j, int k, int face) const
    {
        const ForestRegion &R = *reg;
        const int u = R.u_is_k ? k : j, v = R.u_is_k ? j : k;
        const int ti = (i - 1) / R.chunk, il = (i - 1) % R.chunk, tu = (u - 1) / 64, tv = (v - 1) / kBrickRowsHost;
        if (face == 0) return (int32_t)(R.iface_off + ((int64_t)(ti % R.nslot) * R.vp + (v - 1)) * R.up + (u - 1));
        const bool along_u = (face == 2) == R.u_is_k; // face 2 steps along sweep-k
        if (along_u) {
            // between two bricks: the u-face ring of the brick on the left.  Inside a brick (the box's u-faces need not lie on
            // brick boundaries): the box's near face is ring 0, its far face ring 1 of the two rings at uqface_off
            const bool between = (u - 1) % 64 == 0;
            const int ju = R.u_is_k ? 2 : 1;
            const int64_t ring = between ? (int64_t)(tu - 1) : 2 * (int64_t)R.id + (u == R.lo[ju] ? 0 : 1);
            return (int32_t)((between ? 0 : R.uqface_off) + ((ring * R.nslot + ti % R.nslot) * R.chunk + il) * ((int64_t)R.ntv * R.ut) +
                             (int64_t)R.ut * tv + (v - 1) % kBrickRowsHost);
        }
        return (int32_t)(R.vface_off + (((int64_t)(tv - 1) * R.nslot + ti % R.nslot) * R.chunk + il) * R.up + (u - 1));
    }

    static int slot_of(int top) { return top == 1 ? 0 : (top == 3 ? 1 : 2); } // xyEnd, xzEnd, yzEnd -> 0, 1, 2

    // setRaysRefined, transportRoutinesModule.f90:150-187: the lower sub-layer starts at the parent's entry point
    // doubled (mod 1), the upper one where the lower one's top-ending segment leaves
    int32_t sub_pattern(int32_t parent, int which)
    {
        if (pats[parent].sub[0] < 0) {
            PatNode lo, hi;
            lo.sub[0] = lo.sub[1] = hi.sub[0] = hi.sub[1] = -1;
            if (sub_layer_patterns(pats[parent].p, phi, theta, &lo.p, &hi.p)) { status = FTTE_ERR_PATTERN; err = "ray pattern left the unit cell"; }
            const int32_t a = (int32_t)pats.size();
            pats.push_back(lo);
            pats.push_back(hi);
            pats[parent].sub[0] = a;
            pats[parent].sub[1] = a + 1;
        }
        return pats[parent].sub[which];
    }

    // get??Neighbour, transportRoutinesModule.f90:455-558: go down into `c`, at every level into the child that
    // holds the entry point (a, b) on the shared face; `.le. 0.5` picks the lower half.
    // face 0: (x, y) on the bottom face, children of the upper sub-layer;  face 1: (x, z), children of the far-y half;
    // face 2: (y, z), children of the far-x half.  Sweep frame: i <-> z, j <-> y, k <-> x.
    int32_t descend(int32_t c, int face, double a, double b) const
    {
        while (T.child0[c] >= 0) {
            const int ha = a <= 0.5 ? 0 : 1, hb = b <= 0.5 ? 0 : 1;
            int i, j, k;
            if (face == 0) { i = 1; k = ha; j = hb; }
            else if (face == 1) { j = 1; k = ha; i = hb; }
            else { k = 1; j = ha; i = hb; }
            c = T.child0[c] + cs[i][j][k];
            a = 2.0 * a - ha;
            b = 2.0 * b - hb;
        }
        return c;
    }

    // findNeighbours for one face, transportRoutinesModule.f90:264-418
    int32_t upstream_leaf(int lvl, int face, double a, double b) const
    {
        for (int lv = lvl; lv >= 0; --lv) {
            const int i = seq[lv][0], j = seq[lv][1], k = seq[lv][2];
            const int along = face == 0 ? i : (face == 1 ? j : k);
            if (along > 1) {
                const int si = i - (face == 0), sj = j - (face == 1), sk = k - (face == 2);
                int32_t sib;
                if (lv == 0) {
                    int ic, jc, kc;
                    rotate_indices(si, sj, sk, T.n, T.n, T.n, izone, &ic, &jc, &kc);
                    sib = (int32_t)(((int64_t)(ic - 1) * T.n + (jc - 1)) * T.n + (kc - 1));
                } else sib = T.child0[anc[lv - 1]] + cs[si - 1][sj - 1][sk - 1];
                return descend(sib, face, a, b);
            }
            // no sibling on the upstream side at this level: express the entry point in the parent's units
            if (face == 0) { b = b / 2.0 + (j == 1 ? 0.0 : 0.5); a = a / 2.0 + (k == 1 ? 0.0 : 0.5); }
            else if (face == 1) { b = b / 2.0 + (i == 1 ? 0.0 : 0.5); a = a / 2.0 + (k == 1 ? 0.0 : 0.5); }
            else { b = b / 2.0 + (i == 1 ? 0.0 : 0.5); a = a / 2.0 + (j == 1 ? 0.0 : 0.5); }
        }
        return -1;
    }

    // what leaf U hands over through the face it shares with a cell behind it at level `lvl`: its piece that ends there, or -- a coarser
    // leaf without one -- the mean of two of its pieces (:612-634)
    bool handed_over(int32_t U, int face, int lvl, int32_t *up, int32_t *up2)
    {
        const ftte_pattern &Q = pats[node_pat[U]].p;
        const int top = face == 0 ? Q.xy_top : (face == 1 ? Q.xz_top : Q.yz_top);
        const int32_t ub = 3 * T.leaf[U];
        *up2 = -1;
        if (top != 0) { *up = ub + slot_of(top); return true; }
        // the upstream leaf has no segment ending on the shared face: legal only behind a coarser leaf,
        // which then hands over the mean of its xy and xz (else yz) segments (:612-634)
        if (lvl <= T.level[U]) {
            status = FTTE_ERR_PATTERN;
            err = "upstream cell of the same or a finer level has no segment ending on the shared face "
                  "(the reference stops here: 'error in xzTop')";
            return false;
        }
        if (Q.xz_active) { *up = ub + 1; *up2 = ub; }
        else if (Q.yz_active) { *up = ub + 2; *up2 = ub; }
        else *up = ub;
        return true;
    }

    void leaf_segments(int32_t node, int lvl, double cell)
    {
        const ftte_pattern &P = pats[node_pat[node]].p;
        const int64_t base = 3 * (int64_t)T.leaf[node];
        // a leaf of a fine block that bricks of its own sweep: not part of the forest; what enters it from the forest is listed
        const bool hole = reg && lvl == 1 && reg->in_fine(seq[0][0], seq[0][1], seq[0][2]);
        int fc[3] = {0, 0, 0};
        if (hole) {
            is_hole[(size_t)node] = 1;
            for (int a = 0; a < 3; ++a) { fc[a] = 2 * (seq[0][a] - reg->flo[a]) + seq[1][a]; fine_at[3 * (size_t)node + (size_t)a] = (int16_t)fc[a]; }
        }
        for (int face = 0; face < 3; ++face) {
            const int64_t seg = base + face;
            const bool active = face == 0 || (face == 1 ? P.xz_active : P.yz_active) != 0;
            if (!active) { F.up[seg] = AmrForest::kInactive; continue; }
            double a, b, len;
            if (face == 0) { a = P.xy_x0; b = P.xy_y0; len = P.xy_len; }
            else if (face == 1) { a = P.xz_x0; b = P.xz_z0; len = P.xz_len; }
            else { a = P.yz_y0; b = P.yz_z0; len = P.yz_len; }
            F.dpath[seg] = cell * len;
            if (hole) {
                F.up[seg] = AmrForest::kInflow; // (active; never listed)
                // behind a sibling, or behind a child of the base cell before this one where that is the block's too: inside the
                // block, the bricks hand the ray over themselves (and the block's interior is not walked: no marks to ask)
                int before[3] = {seq[0][0], seq[0][1], seq[0][2]};
                --before[face];
                if (seq[1][face] == 2 || reg->in_fine(before[0], before[1], before[2])) continue;
            }
            const int32_t U = upstream_leaf(lvl, face, a, b);
            if (hole) {
                AmrForest::FineImport X{fine_element(fc[0], fc[1], fc[2], face), -1, -1};
                if (U >= 0) {
                    if (node_pat[U] < 0) { status = FTTE_ERR_STATE; err = "hybrid sweep: a fine block touches the region's surface"; return; }
                    if (!handed_over(U, face, lvl, &X.up, &X.up2)) return;
                }
                F.fine_imports.push_back(X);
                continue;
            }
            int32_t d = 0;
            if (U < 0) {
                F.up[seg] = AmrForest::kInflow;
            } else if (reg && !is_hole.empty() && is_hole[(size_t)U]) {
                // behind a fine cell that bricks sweep: the ray waits where the brick of that cell leaves what crosses this face
                int uc[3] = {fine_at[3 * (size_t)U], fine_at[3 * (size_t)U + 1], fine_at[3 * (size_t)U + 2]};
                ++uc[face];
                F.up[seg] = AmrForest::kImport;
                F.import_at[seg] = fine_element(uc[0], uc[1], uc[2], face);
                late[(size_t)seg] = 1;
            } else if (reg && node_pat[U] < 0) {
                // the upstream leaf was not visited: it lies outside the region, in a brick.  The rim of the region is made of
                // unrefined base cells, so this is a base cell behind a base cell and the ray waits in the brick's face buffer.
                if (lvl != 0 || T.child0[U] >= 0 || T.parent[U] >= 0) { status = FTTE_ERR_STATE; err = "hybrid sweep: a refined cell touches the region's surface"; return; }
                F.up[seg] = AmrForest::kImport;
                F.import_at[seg] = face_element(seq[0][0], seq[0][1], seq[0][2], face);
            } else {
                int32_t up, up2;
                if (!handed_over(U, face, lvl, &up, &up2)) return;
                F.up[seg] = up; F.up2[seg] = up2;
                if (!late.empty() && (late[(size_t)up] || (up2 >= 0 && late[(size_t)up2]))) {
                    // after the fine blocks' bricks: its depth counts from them (what it takes from before them is long there)
                    late[(size_t)seg] = 1;
                    if (late[(size_t)up]) d = 1 + depth[up];
                    if (up2 >= 0 && late[(size_t)up2]) d = std::max(d, 1 + depth[up2]);
                } else {
                    d = 1 + depth[up];
                    if (up2 >= 0) d = std::max(d, 1 + depth[up2]);
                }
            }
            depth[seg] = d;
        }
    }

    // every leaf under `node` belongs to a box swept in pass `pass`
    void mark_inside(int32_t node, uint8_t pass, std::vector<uint8_t> &pass_of)
    {
        if (T.child0[node] < 0) { F.inside[(size_t)T.leaf[node]] = is_hole.empty() || !is_hole[(size_t)node] ? 1 : 0; pass_of[(size_t)T.leaf[node]] = pass; return; }
        for (int c = 0; c < 8; ++c) mark_inside(T.child0[node] + c, pass, pass_of);
    }

    // transport's recursion order, transportRoutinesModule.f90:577-586: children in sweep order
    void visit(int32_t node, int32_t pat, int lvl, double cell)
    {
        if (status) return;
        node_pat[node] = pat;
        anc[lvl] = node;
        if (reg) F.scratch.nodes.push_back(node);
        if (T.child0[node] < 0) { if (reg) F.visited.push_back(T.leaf[node]); leaf_segments(node, lvl, cell); return; }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    seq[lvl + 1][0] = i + 1; seq[lvl + 1][1] = j + 1; seq[lvl + 1][2] = k + 1;
                    visit(T.child0[node] + cs[i][j][k], sub_pattern(pat, i), lvl + 1, cell / 2.0);
                }
    }
};

} // namespace

int build_forest(const AmrTree &tree, double phi, double theta, int izone, double box, AmrForest *out, std::string *err,
                 const ForestRegion *region)
{
    std::vector<ForestRegion> regions;
    if (region) regions.push_back(*region);
    return build_forest_regions(tree, phi, theta, izone, box, out, err, regions);
}

int build_forest_regions(const AmrTree &tree, double phi, double theta, int izone, double box, AmrForest *out, std::string *err,
                         const std::vector<ForestRegion> &regions)
{
    AmrForest &F = *out;
    const int n = tree.n;
    const int64_t nseg = 3 * tree.ncell;
    const bool restricted = !regions.empty();
    F.izone = izone; F.phi = phi; F.theta = theta;
    bool any_fine = false;
    for (const ForestRegion &R : regions) any_fine = any_fine || R.has_fine;
    AmrForest::Scratch &S = F.scratch;
    const int64_t nnode = (int64_t)tree.parent.size();
    // a forest of this tree that a restricted build left: everything but the leaves and nodes that build walked holds the defaults
    const bool again = restricted && S.ncell == tree.ncell && S.nnode == nnode && S.fine == any_fine && (int64_t)F.up.size() == nseg &&
                       (int64_t)F.inside.size() == tree.ncell;
    if (again) {
        for (int32_t leaf : F.visited) {
            for (int64_t s = 3 * (int64_t)leaf; s < 3 * (int64_t)leaf + 3; ++s) {
                F.up[(size_t)s] = AmrForest::kInactive; F.up2[(size_t)s] = -1; F.import_at[(size_t)s] = -1; S.depth[(size_t)s] = 0;
                if (any_fine) S.late[(size_t)s] = 0;
            }
            F.inside[(size_t)leaf] = 0; S.pass_of[(size_t)leaf] = 0;
        }
        for (int32_t node : S.nodes) { S.node_pat[(size_t)node] = -1; if (any_fine) S.is_hole[(size_t)node] = 0; }
    } else {
        F.up.assign(nseg, AmrForest::kInactive);
        F.up2.assign(nseg, -1);
        F.dpath.assign(nseg, 0.0);
        F.import_at.clear(); F.inside.clear();
        if (restricted) { F.import_at.assign(nseg, -1); F.inside.assign((size_t)tree.ncell, 0); }
        S.node_pat.assign((size_t)nnode, -1);
        S.depth.assign(nseg, 0);
        S.is_hole.clear(); S.fine_at.clear(); S.late.clear(); S.pass_of.clear();
        if (any_fine) {
            S.is_hole.assign((size_t)nnode, 0);
            S.fine_at.assign(3 * (size_t)nnode, 0);
            S.late.assign((size_t)nseg, 0);
        }
        if (restricted) S.pass_of.assign((size_t)tree.ncell, 0);
        S.ncell = restricted ? tree.ncell : -1; S.nnode = nnode; S.fine = any_fine;
    }
    F.visited.clear(); S.nodes.clear(); F.exports.clear();
    Builder B(tree, F);
    B.izone = izone; B.phi = phi; B.theta = theta; B.reg = nullptr;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                int a, b, c;
                rotate_indices(i + 1, j + 1, k + 1, 2, 2, 2, izone, &a, &b, &c); // equiSources.f90:1485-1491
                B.cs[i][j][k] = 4 * (a - 1) + 2 * (b - 1) + (c - 1);
            }
    std::vector<ftte_pattern> layers(n);
    if (layer_patterns(n, phi, theta, layers.data())) {
        *err = "ray pattern left the unit cell (setPattern consistency check)";
        return FTTE_ERR_PATTERN;
    }
    B.pats.resize(n);
    for (int i = 0; i < n; ++i) { B.pats[i].p = layers[i]; B.pats[i].sub[0] = B.pats[i].sub[1] = -1; }
    F.fine_imports.clear();
    const int per_box = any_fine ? 2 : 1; // passes per box: before and after its fine block's bricks

    const double cell = box / (double)n; // equiSources.f90:1570
    // pass of every leaf that belongs to a box (0 without boxes); exports carry their box's pass until they are sorted
    std::vector<uint8_t> &pass_of = S.pass_of;
    std::vector<int> export_pass;
    int npass = 1;
    if (restricted) {
        for (const ForestRegion &R : regions) {
            if (R.pass < 0 || R.pass > 254) { *err = "hybrid sweep: too many passes"; return FTTE_ERR_STATE; }
            npass = std::max(npass, per_box * (R.pass + 1));
        }
    }
    const size_t nreg = restricted ? regions.size() : 1;
    for (size_t r = 0; r < nreg && !B.status; ++r) {
        const ForestRegion *R = restricted ? &regions[r] : nullptr;
        B.reg = R;
        const int lo0 = R ? std::max(1, R->lo[0]) : 1, hi0 = R ? std::min(n, R->hi[0]) : n;
        const int lo1 = R ? std::max(1, R->lo[1]) : 1, hi1 = R ? std::min(n, R->hi[1]) : n;
        const int lo2 = R ? std::max(1, R->lo[2]) : 1, hi2 = R ? std::min(n, R->hi[2]) : n;
        for (int i = lo0; i <= hi0 && !B.status; ++i)
            for (int j = lo1; j <= hi1 && !B.status; ++j)
                for (int k = lo2; k <= hi2 && !B.status; ++k) {
                    int ic, jc, kc;
                    rotate_indices(i, j, k, n, n, n, izone, &ic, &jc, &kc);
                    B.seq[0][0] = i; B.seq[0][1] = j; B.seq[0][2] = k;
                    const int32_t node = (int32_t)(((int64_t)(ic - 1) * n + (jc - 1)) * n + (kc - 1));
                    if (R && B.node_pat[node] >= 0) { *err = "hybrid sweep: two boxes overlap"; return FTTE_ERR_STATE; }
                    // (a base cell in the interior of a fine block -- itself, what lies before it and what lies behind it all swept
                    // by the block's bricks -- has nothing to tell the forest: not walked)
                    if (R && R->in_fine(i, j, k) && R->in_fine(i - 1, j - 1, k - 1) && R->in_fine(i + 1, j + 1, k + 1)) continue;
                    B.visit(node, i - 1, 0, cell);
                }
        if (B.status) break;
        if (!R) continue;
        // the leaves of this box, and the rays that leave it: for every base cell just outside its far faces, the piece of the
        // cell inside that ends on the shared face (same rule as a link inside the forest; the rim cells are unrefined base cells)
        for (int i = lo0; i <= hi0; ++i)
            for (int j = lo1; j <= hi1; ++j)
                for (int k = lo2; k <= hi2; ++k) {
                    int ic, jc, kc;
                    if (R->in_fine(i, j, k) && R->in_fine(i - 1, j - 1, k - 1) && R->in_fine(i + 1, j + 1, k + 1)) continue; // (not walked: not the forest's)
                    rotate_indices(i, j, k, n, n, n, izone, &ic, &jc, &kc);
                    B.mark_inside((int32_t)(((int64_t)(ic - 1) * n + (jc - 1)) * n + (kc - 1)), (uint8_t)R->pass, pass_of);
                }
        for (int face = 0; face < 3; ++face) {
            if (R->hi[face] >= n) continue; // the box reaches the domain boundary: the rays leave the grid
            int lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
            lo[face] = hi[face] = R->hi[face]; // the box's last layer of cells along this axis
            for (int i = lo[0]; i <= hi[0]; ++i)
                for (int j = lo[1]; j <= hi[1]; ++j)
                    for (int k = lo[2]; k <= hi[2]; ++k) {
                        int ic, jc, kc;
                        rotate_indices(i, j, k, n, n, n, izone, &ic, &jc, &kc);
                        const int32_t U = (int32_t)(((int64_t)(ic - 1) * n + (jc - 1)) * n + (kc - 1));
                        if (tree.child0[U] >= 0) { *err = "hybrid sweep: a refined cell touches the region's surface"; return FTTE_ERR_STATE; }
                        const ftte_pattern &Q = B.pats[B.node_pat[U]].p;
                        const int top = face == 0 ? Q.xy_top : (face == 1 ? Q.xz_top : Q.yz_top);
                        if (top == 0) continue; // no piece of this layer's pattern ends on that face: nothing crosses it
                        const int di = i + (face == 0), dj = j + (face == 1), dk = k + (face == 2);
                        const int32_t xs = (int32_t)(3 * tree.leaf[U] + Builder::slot_of(top));
                        F.exports.push_back({B.face_element(di, dj, dk, face), xs});
                        export_pass.push_back(per_box * R->pass + (any_fine && B.late[(size_t)xs] ? 1 : 0));
                    }
        }
    }
    if (B.status) { *err = B.err; return B.status; }

    // counting sort of the active segments by (pass, depth)
    std::vector<int32_t> maxd((size_t)npass, -1);
    auto pass_of_seg = [&](int64_t s) { return restricted ? per_box * (int)pass_of[(size_t)(s / 3)] + (any_fine && B.late[(size_t)s] ? 1 : 0) : 0; };
    auto listed = [&](int64_t s) { return F.up[s] != AmrForest::kInactive && (!restricted || F.inside[(size_t)(s / 3)]); };
    // (restricted: only the leaves that were walked can be listed; in ascending order, as a run over all segments would meet them)
    if (restricted) std::sort(F.visited.begin(), F.visited.end());
    auto every_segment = [&](auto &&f) {
        if (restricted) { for (int32_t leaf : F.visited) for (int64_t s = 3 * (int64_t)leaf; s < 3 * (int64_t)leaf + 3; ++s) f(s); }
        else for (int64_t s = 0; s < nseg; ++s) f(s);
    };
    every_segment([&](int64_t s) { if (listed(s)) maxd[(size_t)pass_of_seg(s)] = std::max(maxd[(size_t)pass_of_seg(s)], B.depth[s]); });
    F.pass_first.assign((size_t)npass + 1, 0);
    for (int p = 0; p < npass; ++p) F.pass_first[(size_t)p + 1] = F.pass_first[(size_t)p] + (maxd[(size_t)p] + 1);
    const size_t ndepth = (size_t)F.pass_first[(size_t)npass];
    F.depth_off.assign(ndepth + 1, 0);
    every_segment([&](int64_t s) { if (listed(s)) ++F.depth_off[(size_t)F.pass_first[(size_t)pass_of_seg(s)] + (size_t)B.depth[s] + 1]; });
    for (size_t d = 1; d < F.depth_off.size(); ++d) F.depth_off[d] += F.depth_off[d - 1];
    F.order.resize((size_t)F.depth_off.back());
    std::vector<int64_t> cursor(F.depth_off.begin(), F.depth_off.end() - 1);
    every_segment([&](int64_t s) { if (listed(s)) F.order[(size_t)cursor[(size_t)F.pass_first[(size_t)pass_of_seg(s)] + (size_t)B.depth[s]]++] = (int32_t)s; });
    // the rays that leave the boxes, pass after pass
    F.export_first.assign((size_t)npass + 1, 0);
    if (!F.exports.empty()) {
        std::vector<AmrForest::Export> sorted(F.exports.size());
        for (int p : export_pass) ++F.export_first[(size_t)p + 1];
        for (int p = 0; p < npass; ++p) F.export_first[(size_t)p + 1] += F.export_first[(size_t)p];
        std::vector<int64_t> at(F.export_first.begin(), F.export_first.end() - 1);
        for (size_t q = 0; q < F.exports.size(); ++q) sorted[(size_t)at[(size_t)export_pass[q]]++] = F.exports[q];
        F.exports.swap(sorted);
    }
    return 0;
}

} // namespace ftte
