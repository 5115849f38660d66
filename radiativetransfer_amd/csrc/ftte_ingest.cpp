// ftte_ingest.cpp -- grid ingest: per-level lists of SPH-projected cells -> the cell array.
//
// What the reference does between reading its grid file and the first transfer, equiSources.f90:427-618 with
// placeCellProjectWithVelocity (:1870-1974) and the cell-array order of writeCell (:4044-4079), as host code on flat
// arrays: no pointer tree of 680-byte nodes, the octree is three index arrays.  No device is involved.
//
// The reference's arithmetic is kept operation by operation, including where it is single precision:
//   * positions are normalised to the unit box in double and stored back into the real*4 list (:484-490);
//   * the smoothing pass finds a level-1 cell's base cell with a SINGLE-precision product int(pos*nx) (:531-533), the placing
//     loop with a double one (:587-589);
//   * 10.**readArray(..) is real*4 ** real*4 (:1930-1933).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ftte.h"

namespace {

inline double W(float x) { return (double)x; } // a default-real literal widened to the reference's RealKind

struct Node {
    int32_t child0 = -1;              // first of eight children (i, j, k in 1..2, k fastest), or -1
    double tgas = 0, rho = 0, HI = 0, HeI = 0, HeII = 0, velx = 0, vely = 0, velz = 0, abun2 = 0;
};

} // namespace

struct ftte_cellarray {
    int nx = 0;
    double box = 0;
    bool kinematics = false, metals = false;
    std::vector<int32_t> level;
    std::vector<double> field[9]; // HI, HeI, HeII, tgas, rho, velx, vely, velz, abun2
};

namespace {

void emit(const std::vector<Node> &T, int32_t v, int depth, ftte_cellarray &A)
{
    const Node &N = T[(size_t)v];
    if (N.child0 >= 0) {
        for (int c = 0; c < 8; ++c) emit(T, N.child0 + c, depth + 1, A);
        return;
    }
    A.level.push_back(depth);
    const double f[9] = {N.HI, N.HeI, N.HeII, N.tgas, N.rho, N.velx, N.vely, N.velz, N.abun2};
    for (int q = 0; q < 9; ++q) A.field[q].push_back(f[q]);
}

} // namespace

extern "C" {

int ftte_ingest_levels(int nlevels, const ftte_level_list *lists, ftte_cellarray **out)
{
    if (!out) return FTTE_ERR_ARG;
    *out = nullptr;
    if (nlevels < 1 || !lists) return FTTE_ERR_ARG;
    for (int l = 0; l < nlevels; ++l)
        if (lists[l].ncell < 0 || (lists[l].ncell > 0 && (!lists[l].pos || !lists[l].lT || !lists[l].lnH || !lists[l].lx))) return FTTE_ERR_ARG;
    const bool kin = lists[0].vel != nullptr, met = lists[0].abun != nullptr;
    for (int l = 0; l < nlevels; ++l)
        if (lists[l].ncell > 0 && ((lists[l].vel != nullptr) != kin || (lists[l].abun != nullptr) != met)) return FTTE_ERR_ARG;

    // base grid size from the number of level-1 cells, :427-441
    const int64_t n1 = lists[0].ncell;
    int nx = 0;
    int64_t cube = 0;
    while (cube < n1) { ++nx; cube = (int64_t)nx * nx * nx; }
    if (cube != n1 || nx < 2) return FTTE_ERR_NOT_CUBIC;

    // bounding box of the level-1 cell centres, widened by nx/(nx-1): :447-480
    double lo[3] = {1.e10, 1.e10, 1.e10}, hi[3] = {-1.e10, -1.e10, -1.e10};
    for (int64_t c = 0; c < n1; ++c)
        for (int q = 0; q < 3; ++q) {
            const double p = (double)lists[0].pos[(size_t)q * n1 + c];
            lo[q] = std::fmin(lo[q], p);
            hi[q] = std::fmax(hi[q], p);
        }
    for (int q = 0; q < 3; ++q) {
        const double mid = 0.5 * (lo[q] + hi[q]);
        const double half = 0.5 * (hi[q] - lo[q]) * (double)(float)nx / (double)(float)(nx - 1);
        lo[q] = mid - half;
        hi[q] = mid + half;
    }
    ftte_cellarray *A = new ftte_cellarray;
    A->nx = nx; A->kinematics = kin; A->metals = met;
    A->box = std::fabs(lo[0] - hi[0]) * (W(1.e3f) * W(3.08568025e18f)); // physicalBoxSize = abs(xa-xb)*kpc, :481

    // positions in the unit box, stored back in single precision: :483-490
    std::vector<std::vector<float>> pos((size_t)nlevels);
    for (int l = 0; l < nlevels; ++l) {
        const int64_t nc = lists[l].ncell;
        pos[(size_t)l].resize((size_t)3 * nc);
        for (int q = 0; q < 3; ++q)
            for (int64_t c = 0; c < nc; ++c)
                pos[(size_t)l][(size_t)q * nc + c] = (float)(((double)lists[l].pos[(size_t)q * nc + c] - lo[q]) / (hi[q] - lo[q]));
    }

    // smoothing of the level-1 abundances (two 1-2-1 passes along every axis), :526-578.  The reference runs this block
    // whether or not the grid carries metals (without them it dereferences an unallocated array); here it needs them.
    std::vector<float> abun2_l1;
    if (met) {
        const size_t nn = (size_t)nx * nx * nx;
        std::vector<double> u(nn, 0.0), t(nn);
        std::vector<size_t> where((size_t)n1);
        for (int64_t c = 0; c < n1; ++c) {
            int ijk[3];
            for (int q = 0; q < 3; ++q) {
                ijk[q] = (int)(pos[0][(size_t)q * n1 + c] * (float)nx); // single-precision product, :531-533
                if (ijk[q] < 0 || ijk[q] >= nx) { delete A; return FTTE_ERR_ARG; }
            }
            where[(size_t)c] = ((size_t)ijk[0] * nx + ijk[1]) * nx + ijk[2];
            u[where[(size_t)c]] = (double)lists[0].abun[(size_t)1 * n1 + c];
        }
        const size_t stride[3] = {(size_t)nx * nx, (size_t)nx, 1};
        for (int pass = 0; pass < 2; ++pass)
            for (int axis = 0; axis < 3; ++axis) {
                // tmp(i) receives 0.25 u(i-1), then 0.5 u(i), then 0.25 u(i+1), in that order (loops ascending)
                for (size_t at = 0; at < nn; ++at) {
                    const int i = (int)((at / stride[axis]) % (size_t)nx);
                    double s = 0.0;
                    if (i > 0) s = s + 0.25 * u[at - stride[axis]];
                    s = s + 0.5 * u[at];
                    if (i < nx - 1) s = s + 0.25 * u[at + stride[axis]];
                    t[at] = s;
                }
                u.swap(t);
            }
        abun2_l1.resize((size_t)n1);
        for (int64_t c = 0; c < n1; ++c) abun2_l1[(size_t)c] = (float)u[where[(size_t)c]];
    }

    // the tree: base cells zeroed (:495-521), every listed cell placed (:580-618)
    std::vector<Node> T((size_t)nx * nx * nx);
    const double mp = W(1.6726231e-24f), mn = W(1.67492728e-24f), psi = W(0.76f);
    const double mh = mp, mhe = 2. * (mp + mn);
    for (int l = 0; l < nlevels; ++l) {
        const int64_t nc = lists[l].ncell;
        for (int64_t c = 0; c < nc; ++c) {
            double x[3];
            int ijk[3];
            for (int q = 0; q < 3; ++q) {
                const double x0 = (double)pos[(size_t)l][(size_t)q * nc + c];
                ijk[q] = (int)(x0 * nx);
                if (ijk[q] < 0 || ijk[q] >= nx) { delete A; return FTTE_ERR_ARG; }
                x[q] = x0 * (double)(float)nx - (double)(float)ijk[q];
            }
            int32_t v = (int32_t)(((size_t)ijk[0] * nx + ijk[1]) * nx + ijk[2]);
            // placeCellProjectWithVelocity: level l+1 descends l times, refining on the way (:1883-1931)
            for (int d = 0; d < l; ++d) {
                if (T[(size_t)v].child0 < 0) {
                    const int32_t first = (int32_t)T.size();
                    if (T.size() + 8 > (size_t)0x7fffffff) { delete A; return FTTE_ERR_UNSUPPORTED; }
                    Node kid; // children inherit tgas, rho, HI, HeI, HeII; velocities and abun2 start at zero
                    kid.tgas = T[(size_t)v].tgas; kid.rho = T[(size_t)v].rho; kid.HI = T[(size_t)v].HI;
                    kid.HeI = T[(size_t)v].HeI; kid.HeII = T[(size_t)v].HeII;
                    T.insert(T.end(), 8, kid);
                    T[(size_t)v].child0 = first;
                }
                int h[3];
                for (int q = 0; q < 3; ++q) {
                    if (x[q] < 0.5) { h[q] = 0; x[q] = 2. * x[q]; }
                    else { h[q] = 1; x[q] = 2. * x[q] - 1.; }
                }
                v = T[(size_t)v].child0 + 4 * h[0] + 2 * h[1] + h[2];
            }
            Node &N = T[(size_t)v];
            // :1933-1962; 10.**readArray(i) is real*4 ** real*4
            N.tgas = (double)std::pow(10.f, lists[l].lT[c]);
            const double nh = (double)std::pow(10.f, lists[l].lnH[c]);
            const double xneu = (double)std::pow(10.f, lists[l].lx[c]);
            N.rho = nh * mh / psi;
            N.HI = nh * xneu;
            const double nhe = (W(1.f) - psi) * N.rho / mhe;
            N.HeI = nhe * W(1.f);
            N.HeII = nhe * W(0.f);
            if (kin) {
                N.velx = (double)lists[l].vel[(size_t)0 * nc + c];
                N.vely = (double)lists[l].vel[(size_t)1 * nc + c];
                N.velz = (double)lists[l].vel[(size_t)2 * nc + c];
            }
            if (met) N.abun2 = (double)(l == 0 ? abun2_l1[(size_t)c] : lists[l].abun[(size_t)1 * nc + c]);
            else N.abun2 = W(0.02f);
        }
    }

    // the cell array: base cells i, j, k (k fastest), a refined cell replaced by its children, writeCell :4044-4079
    const size_t nbase = (size_t)nx * nx * nx;
    for (size_t b = 0; b < nbase; ++b) emit(T, (int32_t)b, 0, *A);
    *out = A;
    return FTTE_OK;
}

int ftte_cellarray_info(const ftte_cellarray *a, int *nx, int64_t *ncell, double *box_cm, int *has_velocity, int *has_metals)
{
    if (!a) return FTTE_ERR_ARG;
    if (nx) *nx = a->nx;
    if (ncell) *ncell = (int64_t)a->level.size();
    if (box_cm) *box_cm = a->box;
    if (has_velocity) *has_velocity = a->kinematics ? 1 : 0;
    if (has_metals) *has_metals = a->metals ? 1 : 0;
    return FTTE_OK;
}

int ftte_cellarray_fields(const ftte_cellarray *a, int32_t *level, double *HI, double *HeI, double *HeII, double *tgas, double *rho,
                          double *velx, double *vely, double *velz, double *abun2)
{
    if (!a) return FTTE_ERR_ARG;
    const size_t n = a->level.size();
    if (level) std::memcpy(level, a->level.data(), sizeof(int32_t) * n);
    double *dst[9] = {HI, HeI, HeII, tgas, rho, velx, vely, velz, abun2};
    for (int q = 0; q < 9; ++q)
        if (dst[q]) std::memcpy(dst[q], a->field[q].data(), sizeof(double) * n);
    return FTTE_OK;
}

void ftte_cellarray_free(ftte_cellarray *a) { delete a; }

} // extern "C"
