// ftte_kernels.h -- launch wrappers of ftte_kernels.hip (all asynchronous on `stream`;
// return 0, -1 bad argument, -2 launch failure)
#pragma once
#include <hip/hip_runtime_api.h>

#include "ftte_internal.h"

namespace ftte {

// rows x stack: 4x{1,4,8}, 8x{1,2,4}, 16x1; waves: 2, 3, 4, 6
void set_lds_pad(int bytes); // diagnostic: dynamic LDS per workgroup, to cap residency
int lds_pad();
int launch_sweep(const LaunchRec &L, int rows, int waves, int stack, int nnu, hipStream_t stream);
// one stage of the cell-fixed brick sweep; max_dirs: directions of the launch's largest group (sizes the LDS); waves 2..4
// persistent > 0 (BrickLaunch::queue set): the whole sweep in one launch of that many workgroups, a task queue per XCD
int launch_brick(const BrickLaunch &L, int max_dirs, int waves, hipStream_t stream, bool masked = false, int persistent = 0);
int launch_xcc_census(unsigned *mask_dev, hipStream_t stream); // bit x of *mask_dev set: some workgroup ran on the XCD whose HW_REG_XCC_ID is x
int launch_brick_pair(const BrickLaunch &L, int max_dirs, int waves, hipStream_t stream); // two wavefronts per brick, four rows each
// cell-array order -> layout 1 ([jc][ic][kc]) or 2 ([kc][ic][jc]); nnu groups, group_stride apart
int launch_to_layout(int layout, const double *src, double *dst, int n, int nnu, long group_stride, hipStream_t stream, bool tiled = false, int tchunk = 0); // tiled: brick order (BrickLaunch::tiled), layout 0 too; tchunk: layers per piece
// cell-array order -> the three layouts in one pass (dst0: a copy)
int launch_set_layouts(const double *src, double *dst0, double *dst1, double *dst2, int n, int nnu, long group_stride, hipStream_t stream);
// J (cell-array order) = acc[0] + acc[1] + ... in list order; layout[a] in {0,1,2}
int launch_merge(const double *const *acc, const int *layout, int count, double *J, int n, int nnu, long group_stride,
                 bool accumulate, hipStream_t stream, const int32_t *leaf_of_base = nullptr, long j_stride = 0, bool tiled = false, int tchunk = 0);
// out[q] = parts[0][q] + parts[1][q] + ... (in that order), q < count: the pieces of J the devices of one context swept for different
// directions (ftte_multi.cpp; the partners' buffers are read where they lie)
int launch_sum_parts(const double *const *parts, int nparts, double *out, long count, hipStream_t stream);
// hybrid sweep of a refined cell array: leaf-ordered values -> values of the base cells; the rays leaving the forest's region
int launch_base_cells(const double *leaf_values, const int32_t *leaf_of_base, double *base_values, long nbase, long ncell, int nnu,
                      hipStream_t stream);
int launch_amr_export(const AmrLevelRec &A, int64_t most_exports, hipStream_t stream);
int launch_amr_fine_import(const AmrLevelRec &A, int64_t most_imports, hipStream_t stream); // forest -> the face rings of a fine block's bricks
int launch_opacity(const double *HI, const double *HeI, const double *HeII, const double *beta, double *kappa, long ncell,
                   int nnu, hipStream_t stream);

// refined cell arrays: one depth of the segment forest of up to kAmrBatch directions; then the per-leaf means into J
int launch_cell_major(const double *src, double *dst, long ncell, int nnu, hipStream_t stream, const int32_t *cells = nullptr, long count = 0);
int launch_amr_level(const AmrLevelRec &A, hipStream_t stream);
// levels depth0 .. depth0 + ndepth - 1 in one launch, a workgroup per direction (tables: those levels' [count[ndir], begin[ndir]] pairs)
int launch_amr_levels(const AmrLevelRec &A, const int64_t *tables, int ndepth, hipStream_t stream);
int launch_amr_combine(const AmrLevelRec &A, double *J, bool zero_first, hipStream_t stream);

// point sources: table accumulation (P3), logs of user tables, one pixel level of the tracer (P1 + P2)
int launch_rate_table(const FreqBin *bins, int nbins, double *tables, double *logtab, hipStream_t stream);
int launch_rate_lookup(const double *logtab, int dust, int nsample, const double *tau, double *out, hipStream_t stream);
int launch_log_table(const double *tables, double *logtab, hipStream_t stream);
int launch_point_trace(const TraceRec &T, hipStream_t stream);
int launch_pack_medium(const double *const field[5], double *packed, long ncell, hipStream_t stream);
int launch_repack_rates(double *planes, double *packed, long ncell, bool to_packed, hipStream_t stream);

// ionisation equilibrium of every leaf (solveRateEquations)
int launch_rate_equations(const ChemRec &R, hipStream_t stream);
int launch_thin_limit(const double *HI, const double *HeI, const double *HeII, const double *rho, const double *uvb, double threshold,
                      double *J, long ncell, int nnu, hipStream_t stream);

} // namespace ftte
