// tests/host_sanitize_harness.cpp -- TEST INFRASTRUCTURE.  The host planners of the library (csrc/ftte_amr.cpp: tree rebuild and
// per-direction segment forests, whole tree and restricted to a box; csrc/ftte_geometry.cpp; csrc/ftte_ingest.cpp) compiled with
// g++ -fsanitize=address,undefined and driven over random refined cell arrays: every index they produce is checked against the
// invariants the device kernels rely on (amr_level_kernel, amr_export_kernel), and the sanitizers watch the builders themselves.
// No GPU, no HIP: tests/test_host_sanitizers.py builds and runs it.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../radiativetransfer_amd/csrc/ftte_amr.h"
#include "../radiativetransfer_amd/csrc/ftte_geometry.h"

using namespace ftte;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double rnd()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(rng_state >> 11) / (double)(1ull << 53);
}

#define CHECK(cond, ...)                                                                                                              \
    do {                                                                                                                              \
        if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::exit(1); } \
    } while (0)

// depth-first level list of an n^3 base grid in which base cells are refined with probability p (children again with p / 2)
static void random_levels(int n, double p, int max_level, std::vector<int32_t> *levels)
{
    levels->clear();
    struct Rec { static void go(int level, double p, int max_level, std::vector<int32_t> *out) {
        if (level < max_level && rnd() < p) { for (int c = 0; c < 8; ++c) go(level + 1, p / 2, max_level, out); }
        else out->push_back(level);
    } };
    for (int b = 0; b < n * n * n; ++b) Rec::go(0, p, max_level, levels);
}

static bool same_forest(const AmrForest &A, const AmrForest &B)
{
    bool same = A.up == B.up && A.up2 == B.up2 && A.import_at == B.import_at && A.inside == B.inside && A.order == B.order &&
                A.depth_off == B.depth_off && A.pass_first == B.pass_first && A.export_first == B.export_first &&
                A.exports.size() == B.exports.size() && A.fine_imports.size() == B.fine_imports.size() && A.visited == B.visited;
    for (size_t q = 0; same && q < A.exports.size(); ++q) same = A.exports[q].at == B.exports[q].at && A.exports[q].seg == B.exports[q].seg;
    for (size_t q = 0; same && q < A.fine_imports.size(); ++q)
        same = A.fine_imports[q].at == B.fine_imports[q].at && A.fine_imports[q].up == B.fine_imports[q].up && A.fine_imports[q].up2 == B.fine_imports[q].up2;
    for (size_t q = 0; same && q < A.order.size(); ++q) same = A.dpath[(size_t)A.order[q]] == B.dpath[(size_t)B.order[q]];
    return same;
}

static void check_forest(const AmrTree &T, const AmrForest &F, const ForestRegion *R, int64_t face_elems)
{
    const int64_t nseg = 3 * T.ncell;
    CHECK((int64_t)F.up.size() == nseg && (int64_t)F.up2.size() == nseg && (int64_t)F.dpath.size() == nseg, "array sizes");
    // depth of every active segment, from `order`
    std::vector<int32_t> depth((size_t)nseg, -1);
    CHECK(!F.depth_off.empty() && F.depth_off.front() == 0 && F.depth_off.back() == (int64_t)F.order.size(), "depth_off ends");
    for (size_t d = 0; d + 1 < F.depth_off.size(); ++d) {
        CHECK(F.depth_off[d] <= F.depth_off[d + 1], "depth_off not monotone");
        for (int64_t q = F.depth_off[d]; q < F.depth_off[d + 1]; ++q) {
            const int32_t s = F.order[(size_t)q];
            CHECK(s >= 0 && s < nseg, "segment id %d out of range", s);
            CHECK(depth[(size_t)s] < 0, "segment %d listed twice", s);
            depth[(size_t)s] = (int32_t)d;
        }
    }
    int64_t active = 0;
    for (int64_t s = 0; s < nseg; ++s) {
        const bool in_region = !R || F.inside[(size_t)(s / 3)];
        const bool is_active = F.up[(size_t)s] != AmrForest::kInactive && in_region;
        if (!is_active) { CHECK(depth[(size_t)s] < 0 || !in_region, "inactive segment %lld in the order", (long long)s); continue; }
        ++active;
        CHECK(depth[(size_t)s] >= 0, "active segment %lld missing from the order", (long long)s);
        CHECK(std::isfinite(F.dpath[(size_t)s]) && F.dpath[(size_t)s] > 0, "segment length");
        const int32_t u = F.up[(size_t)s], u2 = F.up2[(size_t)s];
        if (u >= 0) {
            CHECK(u < nseg && depth[(size_t)u] >= 0 && depth[(size_t)u] < depth[(size_t)s], "upstream of %lld not earlier", (long long)s);
            if (R) CHECK(F.inside[(size_t)(u / 3)], "upstream outside the region without an import");
        } else if (u == AmrForest::kImport) {
            CHECK(R != nullptr, "import without a region");
            CHECK(F.import_at[(size_t)s] >= 0 && F.import_at[(size_t)s] < face_elems, "import element %d of %lld", F.import_at[(size_t)s], (long long)face_elems);
        } else CHECK(u == AmrForest::kInflow, "unknown upstream mark %d", u);
        if (u2 >= 0) CHECK(u >= 0 && u2 < nseg && depth[(size_t)u2] >= 0 && depth[(size_t)u2] < depth[(size_t)s], "second upstream of %lld", (long long)s);
    }
    CHECK(active == (int64_t)F.order.size(), "order holds %zu segments, %lld are active", F.order.size(), (long long)active);
    // the xy piece of every leaf (of the region) exists
    for (int64_t c = 0; c < T.ncell; ++c)
        if (!R || F.inside[(size_t)c]) CHECK(F.up[(size_t)(3 * c)] != AmrForest::kInactive, "leaf %lld without its first piece", (long long)c);
    for (const auto &X : F.exports) {
        CHECK(R != nullptr, "export without a region");
        CHECK(X.at >= 0 && X.at < face_elems, "export element %d of %lld", X.at, (long long)face_elems);
        CHECK(X.seg >= 0 && X.seg < nseg && depth[(size_t)X.seg] >= 0, "export of an inactive segment");
    }
}

int main()
{
    int cases = 0;
    for (int trial = 0; trial < 6; ++trial) {
        const int n = trial < 3 ? 6 + trial : 12;
        std::vector<int32_t> levels;
        random_levels(n, trial == 0 ? 0.0 : 0.08 + 0.05 * trial, trial % 2 ? 3 : 2, &levels);
        AmrTree T;
        const std::string why = T.build(n, (int64_t)levels.size(), levels.data());
        CHECK(why.empty(), "tree: %s", why.c_str());
        CHECK(T.ncell == (int64_t)levels.size(), "leaf count");
        // a truncated list must be refused, not read past
        if (levels.size() > 9) {
            AmrTree U;
            CHECK(!U.build(n, (int64_t)levels.size() - 1, levels.data()).empty() || T.max_level == 0, "truncated list accepted");
        }
        for (int pix = 0; pix < 48; pix += 5) {
            double phi_l, theta_l, phi, theta;
            int izone;
            CHECK(pix2ang_nest(2, pix, &phi_l, &theta_l) == 0, "pix2ang");
            CHECK(fold_direction(phi_l, theta_l, &phi, &theta, &izone) == 0, "fold");
            AmrForest F;
            std::string err;
            CHECK(build_forest(T, phi, theta, izone, 1.0, &F, &err) == 0, "forest: %s", err.c_str());
            check_forest(T, F, nullptr, 0);
            // the same, restricted to a box in the sweep frame, with the face layout of a brick plan over this grid
            ForestRegion R;
            R.chunk = 2; R.ut = 8; R.nslot = (n + R.chunk - 1) / R.chunk;
            const int ntu = (n + 63) / 64, ntv = (n + 7) / 8;
            R.ntv = ntv; R.up = 64 * ntu; R.vp = 8 * ntv;
            R.vface_off = (int64_t)ntu * R.nslot * R.chunk * ((int64_t)ntv * R.ut);
            R.iface_off = R.vface_off + (int64_t)ntv * R.nslot * R.chunk * R.up;
            R.uqface_off = R.iface_off + (int64_t)R.nslot * R.vp * R.up;
            const int64_t face_elems = R.uqface_off + 2 * (int64_t)R.nslot * R.chunk * ((int64_t)ntv * R.ut);
            R.u_is_k = (pix & 1) != 0;
            // i and the v axis on brick boundaries; the u axis whole, or (every other direction) ending inside the brick, where the
            // rays cross through the box's own two face rings
            const int iv = R.u_is_k ? 1 : 2, iu = R.u_is_k ? 2 : 1;
            R.lo[0] = 1 + R.chunk * (int)(rnd() * (n / R.chunk / 2)); R.hi[0] = std::min(n, R.lo[0] + R.chunk * (1 + (int)(rnd() * 2)) - 1);
            R.lo[iv] = 1; R.hi[iv] = n <= 8 ? n : 8;
            R.lo[iu] = (pix % 10 == 0 && n > 8) ? 3 : 1; R.hi[iu] = (pix % 10 == 0 && n > 8) ? n - 2 : n;
            // (refined cells outside the box are not this routine's concern: the planner only builds boxes that hold them all;
            // here the box is arbitrary, so only trees whose refined cells it holds are restricted)
            bool holds_all = true;
            for (int b = 0; b < n * n * n && holds_all; ++b)
                if (T.child0[(size_t)b] >= 0) {
                    const int c3[3] = {b / (n * n) + 1, (b / n) % n + 1, b % n + 1};
                    ZoneMap zm;
                    zone_map(izone, &zm);
                    int s3[3];
                    for (int a = 0; a < 3; ++a) s3[zm.src[a]] = zm.mirror[a] ? n + 1 - c3[a] : c3[a];
                    // one cell of margin: the box must hold the refined cells and the unrefined ones around them
                    for (int a = 0; a < 3; ++a) holds_all = holds_all && s3[a] - 1 >= R.lo[a] + (R.lo[a] > 1 ? 1 : 0) - 1 && s3[a] <= R.hi[a] - (R.hi[a] < n ? 1 : 0);
                }
            if (holds_all) {
                AmrForest G;
                CHECK(build_forest(T, phi, theta, izone, 1.0, &G, &err, &R) == 0, "restricted forest: %s", err.c_str());
                check_forest(T, G, &R, face_elems);
                ++cases;
            }
            ++cases;
        }
    }
    // two boxes that do not touch, swept in two passes: slabs along the march axis with a gap of two layers, on a tree whose
    // refined cells lie in the slabs whatever the izone does to the axes
    {
        const int n = 12;
        std::vector<int32_t> levels;
        for (int b = 0; b < n * n * n; ++b) {
            const int c3[3] = {b / (n * n) + 1, (b / n) % n + 1, b % n + 1};
            bool fine = true;
            for (int a = 0; a < 3; ++a) fine = fine && (c3[a] == 2 || c3[a] == 3 || c3[a] == 9 || c3[a] == 10);
            if (fine && (c3[0] + c3[1] + c3[2]) % 2 == 0) for (int c = 0; c < 8; ++c) levels.push_back(1); else levels.push_back(0);
        }
        AmrTree T;
        const std::string why2 = T.build(n, (int64_t)levels.size(), levels.data());
        CHECK(why2.empty() && T.max_level == 1, "two-slab tree: %s (max level %d)", why2.c_str(), T.max_level);
        for (int pix = 0; pix < 48; pix += 3) {
            double phi_l, theta_l, phi, theta;
            int izone;
            CHECK(pix2ang_nest(2, pix, &phi_l, &theta_l) == 0 && fold_direction(phi_l, theta_l, &phi, &theta, &izone) == 0, "direction");
            std::vector<ForestRegion> regions(2);
            int64_t face_elems = 0;
            for (int r = 0; r < 2; ++r) {
                ForestRegion &R = regions[(size_t)r];
                R.chunk = 1; R.ut = 8; R.nslot = n;
                const int ntu = 1, ntv = 2;
                R.ntv = ntv; R.up = 64; R.vp = 16;
                R.vface_off = (int64_t)ntu * R.nslot * R.chunk * ((int64_t)ntv * R.ut);
                R.iface_off = R.vface_off + (int64_t)ntv * R.nslot * R.chunk * R.up;
                R.uqface_off = R.iface_off + (int64_t)R.nslot * R.vp * R.up;
                face_elems = R.uqface_off + 4 * (int64_t)R.nslot * R.chunk * ((int64_t)ntv * R.ut);
                R.u_is_k = true;
                R.lo[0] = r == 0 ? 1 : 8; R.hi[0] = r == 0 ? 5 : 12;
                R.lo[1] = 1; R.hi[1] = n; R.lo[2] = 1; R.hi[2] = n;
                R.id = r; R.pass = r;
            }
            AmrForest F;
            std::string err;
            CHECK(build_forest_regions(T, phi, theta, izone, 1.0, &F, &err, regions) == 0, "two boxes: %s", err.c_str());
            check_forest(T, F, &regions[0], face_elems);
            CHECK(F.pass_first.size() == 3 && F.export_first.size() == 3 && F.pass_first[2] + 1 == (int32_t)F.depth_off.size(), "passes");
            // the first pass holds the leaves of the first slab only
            for (int64_t q = 0; q < F.depth_off[(size_t)F.pass_first[1]]; ++q) {
                const int32_t leaf = F.order[(size_t)q] / 3;
                CHECK(F.inside[(size_t)leaf], "pass 0 lists a leaf outside the boxes");
            }
            CHECK(F.export_first[1] > 0 && F.export_first[2] == (int64_t)F.exports.size(), "the first box hands rays on, the second reaches the boundary");
            // a forest kept from direction to direction (cleaned leaf by leaf, ftte_amr.h) comes out as a fresh one does
            static AmrForest kept;
            CHECK(build_forest_regions(T, phi, theta, izone, 1.0, &kept, &err, regions) == 0, "two boxes, kept forest: %s", err.c_str());
            CHECK(same_forest(kept, F), "a kept forest differs from a fresh one");
            ++cases;
        }
    }
    // a fully refined block swept by bricks of its own (ForestRegion::has_fine): a forest with a hole, two passes, imports into the
    // block and out of it; fresh and kept forests agree, every index lies inside the face block
    {
        const int n = 36, q = 32, lo = 3, hi = lo + q - 1; // symmetric: every izone sees the block at the same place; 2 q = one brick's lanes
        std::vector<int32_t> levels;
        for (int b = 0; b < n * n * n; ++b) {
            const int c3[3] = {b / (n * n) + 1, (b / n) % n + 1, b % n + 1};
            bool fine = true;
            for (int a = 0; a < 3; ++a) fine = fine && c3[a] >= lo && c3[a] <= hi;
            if (fine) for (int c = 0; c < 8; ++c) levels.push_back(1); else levels.push_back(0);
        }
        AmrTree T;
        const std::string why3 = T.build(n, (int64_t)levels.size(), levels.data());
        CHECK(why3.empty() && T.max_level == 1, "fine-block tree: %s", why3.c_str());
        AmrForest kept;
        for (int pix = 0; pix < 48; pix += 5) {
            double phi_l, theta_l, phi, theta;
            int izone;
            CHECK(pix2ang_nest(2, pix, &phi_l, &theta_l) == 0 && fold_direction(phi_l, theta_l, &phi, &theta, &izone) == 0, "direction");
            ForestRegion R;
            R.chunk = 2; R.ut = 8; R.nslot = 2;
            const int ntu = 1, ntv = 5;
            R.ntv = ntv; R.up = 64; R.vp = 40;
            R.vface_off = (int64_t)ntu * R.nslot * R.chunk * ((int64_t)ntv * R.ut);
            R.iface_off = R.vface_off + (int64_t)ntv * R.nslot * R.chunk * R.up;
            R.uqface_off = R.iface_off + (int64_t)R.nslot * R.vp * R.up;
            const int64_t base_elems = R.uqface_off + 2 * (int64_t)R.nslot * R.chunk * ((int64_t)ntv * R.ut);
            R.u_is_k = (pix & 2) != 0;
            // (the box: on brick boundaries along the march and along v -- here the whole extent --, one cell of rim along the lanes)
            const int iu = R.u_is_k ? 2 : 1;
            for (int a = 0; a < 3; ++a) { R.lo[a] = a == iu ? lo - 1 : 1; R.hi[a] = a == iu ? hi + 1 : n; R.flo[a] = lo; R.fhi[a] = hi; }
            R.has_fine = true;
            ForestRegion::FineFaces &Q = R.fine; // a sub-grid of 64 fine cells a side: one brick across, a ring / slot more at the edges
            Q.chunk = 16; Q.ut = 8; Q.nslot = 2 * q / Q.chunk + 1; Q.ntu = 1; Q.ntv = 8; Q.up = 64; Q.vp = 64;
            Q.vface_off = (int64_t)(Q.ntu + 1) * Q.nslot * Q.chunk * ((int64_t)Q.ntv * Q.ut);
            Q.iface_off = Q.vface_off + (int64_t)(Q.ntv + 1) * Q.nslot * Q.chunk * Q.up;
            Q.base = base_elems;
            const int64_t face_elems = base_elems + Q.iface_off + (int64_t)Q.nslot * Q.vp * Q.up;
            AmrForest F;
            std::string err;
            CHECK(build_forest_regions(T, phi, theta, izone, 1.0, &F, &err, {R}) == 0, "fine block: %s", err.c_str());
            CHECK(build_forest_regions(T, phi, theta, izone, 1.0, &kept, &err, {R}) == 0, "fine block, kept forest: %s", err.c_str());
            CHECK(same_forest(kept, F), "fine block: a kept forest differs from a fresh one");
            CHECK(F.pass_first.size() == 3, "a box with a fine block is swept in two passes");
            CHECK(!F.fine_imports.empty(), "rays enter the block from the forest");
            for (const auto &X : F.fine_imports) {
                CHECK(X.at >= base_elems && X.at < face_elems, "an import lands outside the fine bricks' face block");
                CHECK(X.up >= -1 && X.up < 3 * T.ncell && X.up2 >= -1 && X.up2 < 3 * T.ncell, "an import reads a segment that is not there");
                if (X.up >= 0) CHECK(F.inside[(size_t)(X.up / 3)], "an import reads a segment outside the forest");
            }
            size_t behind = 0;
            for (int32_t sg : F.order) {
                CHECK(F.inside[(size_t)(sg / 3)], "a fine cell is listed in the forest");
                if (F.up[(size_t)sg] == AmrForest::kImport) {
                    CHECK(F.import_at[(size_t)sg] >= 0 && F.import_at[(size_t)sg] < face_elems, "a forest segment takes its ray from outside the face block: %d of %lld (base %lld), u_is_k %d", F.import_at[(size_t)sg], (long long)face_elems, (long long)base_elems, (int)R.u_is_k);
                    if (F.import_at[(size_t)sg] >= base_elems) ++behind;
                }
            }
            CHECK(behind > 0, "no forest segment behind the block");
            // the leaves of the block: q^3 base cells refined once, none of them the forest's
            int64_t in_forest = 0;
            for (int64_t l = 0; l < T.ncell; ++l) in_forest += F.inside[(size_t)l];
            CHECK(in_forest == (int64_t)n * n * (q + 2) - (int64_t)q * q * q, "the forest holds the cells around the block and nothing of it");
            ++cases;
        }
    }
    // per-layer patterns of every direction of level 3
    for (int pix = 0; pix < 192; ++pix) {
        double phi_l, theta_l, phi, theta;
        int izone;
        CHECK(pix2ang_nest(4, pix, &phi_l, &theta_l) == 0 && fold_direction(phi_l, theta_l, &phi, &theta, &izone) == 0, "direction %d", pix);
        std::vector<ftte_pattern> layers(37);
        CHECK(layer_patterns(37, phi, theta, layers.data()) == 0, "layer patterns");
        for (int i = 1; i <= 5; ++i) {
            int ic, jc, kc;
            CHECK(rotate_indices(i, 6 - i, 3, 5, 5, 5, izone, &ic, &jc, &kc) == 0 && ic >= 1 && ic <= 5 && jc >= 1 && jc <= 5 && kc >= 1 && kc <= 5, "rotateIndices");
        }
    }
    // grid ingest: two levels, the second covering one octant of one base cell partly
    {
        const int n = 4;
        std::vector<float> pos(3 * n * n * n), lT(n * n * n, 4.f), lnH(n * n * n, -2.f), lx(n * n * n, -3.f), ab(4 * n * n * n, 0.01f);
        for (int i = 0, q = 0; i < n; ++i) for (int j = 0; j < n; ++j) for (int k = 0; k < n; ++k, ++q) {
            pos[q] = (i + 0.5f) * 10.f; pos[n * n * n + q] = (j + 0.5f) * 10.f; pos[2 * n * n * n + q] = (k + 0.5f) * 10.f;
        }
        std::vector<float> pos2 = {12.5f, 17.5f, 12.5f, 12.5f, 12.5f, 17.5f}; // two children of base cell (2,2,2): (x,x | y,y | z,z)
        std::vector<float> lT2(2, 3.f), lnH2(2, -1.f), lx2(2, -1.f), ab2(8, 0.03f);
        ftte_level_list L[2] = {{(int64_t)n * n * n, pos.data(), lT.data(), lnH.data(), lx.data(), nullptr, ab.data()},
                                {2, pos2.data(), lT2.data(), lnH2.data(), lx2.data(), nullptr, ab2.data()}};
        ftte_cellarray *A = nullptr;
        CHECK(ftte_ingest_levels(2, L, &A) == 0 && A, "ingest");
        int nx = 0, hv = 0, hm = 0;
        int64_t ncell = 0;
        double box = 0;
        CHECK(ftte_cellarray_info(A, &nx, &ncell, &box, &hv, &hm) == 0 && nx == n && ncell == (int64_t)n * n * n + 7 && hm == 1 && hv == 0, "ingest info");
        std::vector<int32_t> lev((size_t)ncell);
        std::vector<double> f((size_t)ncell * 9);
        CHECK(ftte_cellarray_fields(A, lev.data(), &f[0], &f[(size_t)ncell], &f[2 * (size_t)ncell], &f[3 * (size_t)ncell], &f[4 * (size_t)ncell],
                                    &f[5 * (size_t)ncell], &f[6 * (size_t)ncell], &f[7 * (size_t)ncell], &f[8 * (size_t)ncell]) == 0, "ingest fields");
        AmrTree T;
        CHECK(T.build(nx, ncell, lev.data()).empty() && T.max_level == 1, "the ingested level list is a tree");
        ftte_cellarray_free(A);
        ++cases;
    }
    std::printf("host planners under the sanitizers: %d cases\n", cases);
    return 0;
}
