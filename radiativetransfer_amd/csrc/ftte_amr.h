// ftte_amr.h -- host planner of the sweep on refined cell arrays (the fully threaded tree of the reference).
//
// On an AMR tree the per-direction dependency structure of the reference (setRaysRefined, findNeighbours,
// get??Neighbour, transport: transportRoutinesModule.f90:121-218, 264-558, 560-963) is a forest over ray
// *segments*: every segment (leaf, xy|xz|yz) takes its incoming intensity from exactly one upstream segment --
// the one of the upstream leaf that ends on the shared face -- or from the inflow, or (a fine cell behind a
// coarser one that has no segment ending on that face, :612-634) from the mean of two segments of that leaf.
// The planner walks the tree once per direction in the reference's sweep order, resolves those links with
// the reference's rules, and orders the segments by their depth in the forest.  The device then processes one
// depth after the other (ftte_kernels.hip: amr_level_kernel), every segment of a depth in parallel.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/ftte.h"

namespace ftte {

// The tree rebuilt from the depth-first leaf list (readCellArray.f90:154-187).  Nodes 0 .. n^3-1 are the base
// cells in storage order; the 8 children of a refined node are contiguous, indexed 4(a-1)+2(b-1)+(c-1) by their
// storage position (a,b,c), which is also their order in the cell array (equiSources.f90:4044-4079).
struct AmrTree {
    int n = 0;
    int64_t ncell = 0;
    int max_level = 0;
    std::vector<int32_t> parent; // node -> parent node (-1 for base cells)
    std::vector<int32_t> child0; // node -> first child node, -1 for a leaf
    std::vector<int32_t> leaf;   // node -> cell-array index, -1 for a refined node
    std::vector<int8_t> level;

    // returns "" or the reference's complaint ('error in levels', readCellArray.f90:182)
    std::string build(int n, int64_t ncell, const int32_t *levels);
    bool refined() const { return max_level > 0; }
};

// A box of base cells in the sweep frame of one izone (1-based, inclusive; i = march axis, j, k) to which a forest can be
// restricted, and where the rays that cross its surface live in the face buffers of the brick sweep (ftte_internal.h:
// BrickLaunch).  Hybrid sweep of a refined cell array: the bricks outside the box are swept by the brick kernel, the box --
// every refined cell plus a rim of unrefined base cells, so that its surface separates unrefined base cells only -- by the forest.
struct ForestRegion {
    int lo[3] = {1, 1, 1}, hi[3] = {0, 0, 0}; // i, j, k
    bool u_is_k = true;                        // which of sweep-j / sweep-k is the lane axis u of the bricks
    int chunk = 1, ut = 8, nslot = 1;          // layers per brick, u-face doubles per brick and layer, face slots along the march
    int ntv = 1, up = 64, vp = 8;              // v bricks, padded extents
    int64_t vface_off = 0, iface_off = 0;      // as BrickLaunch
    int64_t uqface_off = 0;                    // as BrickLaunch: the boxes' own u-face rings (two per box: 2 * id, 2 * id + 1),
                                               // used where a box's u-faces lie inside a brick
    int id = 0;                                // which box of the direction this is (its pair of rings)
    int pass = 0;                              // the boxes of a direction are swept in passes (a box behind another one waits
                                               // for the bricks in between): this box's pass
    bool contains(int i, int j, int k) const { return i >= lo[0] && i <= hi[0] && j >= lo[1] && j <= hi[1] && k >= lo[2] && k <= hi[2]; }

    // A fully refined block inside the box -- every base cell of [flo, fhi] (sweep frame, inclusive) refined exactly once -- whose
    // fine cells are swept by bricks of their own on the fine level (ftte_hybrid.cpp): inside the block the fine cells are a uniform
    // grid with a pattern per sub-layer (setRaysRefined, transportRoutinesModule.f90:150-187).  Its leaves are left out of the forest:
    // a forest segment behind a fine cell takes its ray from the fine bricks' face rings (kImport), a fine cell behind a forest leaf
    // gets it from there (AmrForest::fine_imports), and the forest comes in two passes per box: what does not depend on the block
    // before its bricks, what does after them.
    bool has_fine = false;
    int flo[3] = {1, 1, 1}, fhi[3] = {0, 0, 0};
    struct FineFaces {                         // the fine bricks' face block (BrickLaunch of a sub-grid: one ring / slot more at the edges)
        int chunk = 1, ut = 8, nslot = 1, ntu = 1, ntv = 1, up = 64, vp = 8;
        int64_t vface_off = 0, iface_off = 0;
        int64_t base = 0;                      // where it starts inside the direction's face block
    } fine;
    bool in_fine(int i, int j, int k) const { return has_fine && i >= flo[0] && i <= fhi[0] && j >= flo[1] && j <= fhi[1] && k >= flo[2] && k <= fhi[2]; }
};

// One direction's segment forest.  Segment id = 3 * leaf + slot, slot 0 xy, 1 xz, 2 yz (the order in which the
// reference adds them into the cell's mean).
struct AmrForest {
    static constexpr int32_t kInflow = -1;   // upstream is the domain boundary
    static constexpr int32_t kInactive = -2; // the leaf's pattern has no such segment
    static constexpr int32_t kImport = -3;   // upstream lies outside the region: the ray waits in a face buffer (import_at)
    std::vector<int32_t> up;     // [3 ncell] upstream segment, kInflow, or kInactive
    std::vector<int32_t> up2;    // [3 ncell] second upstream segment of the mean-of-two rule, else -1
    std::vector<double> dpath;   // [3 ncell] cell size * segment length (transportRoutinesModule.f90:651)
    std::vector<int32_t> import_at; // [3 ncell] element offset in the direction's face block, for kImport segments (region only)
    struct Export { int32_t at, seg; };  // face element <- outgoing intensity of segment `seg`
    std::vector<Export> exports;    // the rays that leave the region into a brick (region only)
    std::vector<uint8_t> inside;    // [ncell] the leaf belongs to the region (region only; empty = all)
    std::vector<int32_t> order;  // active segments sorted by pass, then depth
    std::vector<int64_t> depth_off; // [ndepth + 1] ranges of `order`; the depths of all passes one after the other
    struct FineImport { int32_t at, up, up2; }; // face element of a fine block's bricks <- segment up (mean with up2; -1: the inflow)
    std::vector<FineImport> fine_imports;       // the rays that enter fine blocks from the forest (regions with has_fine)
    std::vector<int32_t> pass_first;   // [npass + 1] where each pass starts in depth_off ({0, ndepth} without regions)
    std::vector<int64_t> export_first; // [npass + 1] where each pass's rays start in `exports` (sorted by pass)
    int izone = 0;
    double phi = 0, theta = 0;
    // Builds restricted to boxes touch a small part of the tree.  `visited` lists the leaves the last build walked (ascending) --
    // all the others hold the defaults: up = kInactive, up2 = import_at = -1, inside = 0 -- and an AmrForest handed to
    // build_forest_regions again for the same tree is cleaned leaf by leaf instead of being filled anew: the builder's work then
    // follows the boxes, not the tree.  (Empty after a build over the whole tree.)
    std::vector<int32_t> visited;
    struct Scratch {                         // the builder's own arrays, kept with the forest for the same reason
        std::vector<int32_t> node_pat, depth, nodes;
        std::vector<uint8_t> is_hole, late, pass_of;
        std::vector<int16_t> fine_at;
        int64_t ncell = -1, nnode = -1;      // the tree they are sized for, and clean for (apart from `visited` / `nodes`)
        bool fine = false;
    } scratch;
};

// Returns 0, or an ftte_status (FTTE_ERR_PATTERN where the reference stops: a pattern leaving the unit cell,
// or a same-level upstream leaf without a segment on the shared face, transportRoutinesModule.f90:613-616).
// region (may be null): restrict the forest to the leaves of that box; links across its surface become imports / exports.
int build_forest(const AmrTree &tree, double phi_folded, double theta_folded, int izone, double box, AmrForest *out,
                 std::string *err, const ForestRegion *region = nullptr);
// The same restricted to several boxes that do not touch each other (a rim of unrefined base cells and at least one brick
// between any two): one forest per box, ordered by the boxes' passes.
int build_forest_regions(const AmrTree &tree, double phi_folded, double theta_folded, int izone, double box, AmrForest *out,
                         std::string *err, const std::vector<ForestRegion> &regions);
// setRaysRefined, transportRoutinesModule.f90:150-187: the patterns of the lower and the upper sub-layer of a refined cell that
// carries `parent`.  Returns 0 or 1 (a pattern left the unit cell).
int sub_layer_patterns(const ftte_pattern &parent, double phi_folded, double theta_folded, ftte_pattern *lower, ftte_pattern *upper);

} // namespace ftte
