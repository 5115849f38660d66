/* ftte_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the reference's diffuse radiative-transfer
 * sweep (razoumov/radiativeTransfer: transportRoutinesModule.f90,
 * rotateIndicesModule.f90 and the runUVBTransfer block of equiSources.f90).
 * It exists to check the HIP path; nothing under radiativetransfer_amd/ may
 * include, link or call it.  Allowed users: tests/, __graft_entry__.smoke(),
 * and the cpu_baseline leg of bench.py.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against
 * vectors produced by the reference's own compiled modules
 * (oracle/_ref/ref_harness, tests/golden/make_golden.py).
 */
#ifndef FTTE_ORACLE_H
#define FTTE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* segment-end codes, definitionsModule.f90:159 */
enum { FO_XY_END = 1, FO_YZ_END = 2, FO_XZ_END = 3 };

/* one layer's ray pattern, definitionsModule.f90:123-152 */
typedef struct {
    double xy_x0, xy_y0, xy_len;
    double xz_x0, xz_z0, xz_len;
    double yz_y0, yz_z0, yz_len;
    int32_t xz_active, yz_active;
    int32_t xy_top, xz_top, yz_top;
    int32_t pad_;
} fo_pattern;

/* arithmetic flavour of the sweep */
enum {
    FO_ARITH_REFERENCE = 0, /* libm exp/log, (Iin-Iout)/log(Iin/Iout): the reference's formulae */
    FO_ARITH_DEVICE = 1,    /* radiativetransfer_amd/csrc/ftte_math.h: what the GPU evaluates    */
    FO_ARITH_EXACT = 2      /* every segment in extended precision (long double, 64-bit mantissa: expl, expm1l), rounded to double
                               once: Iout = Iin exp(-tau), mean = Iin (1 - exp(-tau))/tau -- what the reference's log-mean IS when
                               nothing is emitted --, with a source function the exact path mean, with the reference's emissivity
                               term its formulae as they stand.  Shares nothing with ftte_math.h: the yardstick both the device
                               arithmetic and the reference's own double-precision evaluation are measured against in the tests. */
};
/* order in which directions are summed into J */
enum {
    FO_ORDER_SERIAL = 0, /* one accumulator, directions in list order (the reference)          */
    FO_ORDER_CLASSED = 1 /* three accumulators by march axis (storage i, j, k), summed at the
                            end as (Ji + Jj) + Jk: the order the device uses                    */
};

double fo_pi(void);      /* definitionsModule.f90:8  (float32-rounded literal)  */
double fo_half_pi(void); /* :9 */
double fo_two_pi(void);  /* :10 */

/* rotateIndicesModule.f90:7-113.  1-based indices, returns 0 or -1 (bad izone). */
int fo_rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *ic, int *jc, int *kc);

/* equiSources.f90:2118-2231 (+ rotateAngles :2297-2335, getAngle :2337-2361). */
int fo_pix2ang_nest(int nside, int64_t ipix, double *phi, double *theta);

/* equiSources.f90:1395-1454.  Returns 0, or -1/-2/-3 where the reference stops
 * (phi on a quadrant boundary / theta on a boundary / tie of dominant axes). */
int fo_fold_direction(double phi_large, double theta_large, double *phi, double *theta, int *izone);

/* transportRoutinesModule.f90:7-85.  p->xy_x0, p->xy_y0 are inputs.
 * Returns 0, or -1 where the reference stops (:33-36, :60-63). */
int fo_set_pattern(fo_pattern *p, double phi, double theta);

/* equiSources.f90:1495-1534: patterns of layers 1..n for a folded direction. */
int fo_layer_patterns(int n, double phi, double theta, fo_pattern *layers);

/* One diffuse-transfer iteration on a uniform n^3 grid (equiSources.f90:1385-1806
 * with every base cell unrefined).  kappa, J: [nnu][n^3] in cell-array order
 * (flat = ((i-1)*n + (j-1))*n + (k-1), definitionsModule.f90:323-326).
 * J is overwritten (the reference zeroes it in computeOpacities,
 * equiSources.f90:4964-4966).  eta may be NULL (the reference's hard-wired
 * zero emissivity, transportRoutinesModule.f90:673-675); non-NULL: the reference's emission term
 * Iout = Iin*tmpabs + nemi*tmpemi/dpath (:676).  src (may be NULL) is NOT in the reference: a source
 * function S per cell adding S*(1-exp(-tau)), the form source iterations need.
 * noise (may be NULL): [nnu][n^3], receives a per-cell bound on the rounding noise the
 * REFERENCE formula (Iin-Iout)/log(Iin/Iout) carries: sum over directions and segments of
 * (w/nseg) * Iin * (eps/2)/tau_seg, eps/2 = 2^-53.  The quotient Iin/Iout is rounded before the
 * logarithm is taken, which perturbs log by eps/2 absolute and the mean by a relative eps/(2 tau);
 * the parity tests use it as the tau-aware part of their tolerance.
 * Returns 0 or the negative code of the first direction that cannot be folded. */
int fo_diffuse_sweep_uniform(int n, int nnu, const double *kappa, const double *eta, const double *src, double box,
                             int ndir, const double *phi, const double *theta, const double *w, const double *uvb,
                             double *J, int arith, int order, double *noise);

/* Same on an AMR cell array: level[ncell] is the depth-first leaf list
 * (readCellArray.f90:154-187); kappa, J: [nnu][ncell].  Restates
 * setRaysRefined / findNeighbours / get??Neighbour / transport
 * (transportRoutinesModule.f90:121-218, 264-558, 560-963). */
int fo_diffuse_sweep_tree(int n, int64_t ncell, const int32_t *level, int nnu, const double *kappa, const double *eta,
                          const double *src, double box, int ndir, const double *phi, const double *theta,
                          const double *w, const double *uvb, double *J, int arith, int order, double *noise);

/* equiSources.f90:4956-4983 generalised to nnu groups: kappa[g][c] = sum_s n_s[c]*beta[s][g],
 * summed in species order HI, HeI, HeII. beta: [3][nnu]. */
void fo_compute_opacities(int64_t ncell, int nnu, const double *HI, const double *HeI, const double *HeII,
                          const double *beta, double *kappa);

/* ---- point sources (ftte_oracle_point.c) -------------------------------------------------------------------------- */
/* dustCrossSection, dustModule.f90:30-73 (SMC fit); a_smc: [7][5] row-major */
double fo_dust_cross_section(double lambda_um, const double *a_smc);
/* stellarBetaTable.f90: tables [6][11^4] = reactionRate1..3, energyRate1..3, flat index ((idust*11+i3)*11+i2)*11+i1;
 * spec [5][37][1221] log10(erg/s/A), wavelength [1221] cm ascending; iSpectrum, iMetal 1-based;
 * output_sigma [4][300] (24, 25, 26, dust) or NULL */
void fo_stellar_beta_table(const double *a_smc, const double *wavelength, const double *spec, int iSpectrum, double coefSpectrum,
                           int iMetal, double coefMetal, double *tables, double *total_integral, double *output_sigma);
/* getRatesHydrogenHelium, equiSources.f90:4157-4311 */
void fo_get_rates(const double *tables, int dust, int reaction, double tau1, double tau2, double tau3, double tau_dust,
                  double *number_rate, double *heating_rate);
/* rmax(1:30), equiSources.f90:304-309 */
void fo_rmax(double *rmax);
/* the per-source loop equiSources.f90:1256-1329 with startNewLongRay (:3120-3385); rates [6][ncell] = krate24, krate25,
 * krate26, crate24, crate25, crate26, overwritten; src_leaf 0-based cell-array indices; src_ndot = float(weight) */
int fo_point_sources(int n, int64_t ncell, const int32_t *level, const double *HI, const double *HeI, const double *HeII,
                     const double *rho, const double *abun2, double box, int dust, int nsrc, const int64_t *src_leaf,
                     const double *src_ndot, const double *tables, double *rates, int *highest_pixel_level,
                     const double *pix /* may be NULL: (phi,theta) of all pixels of levels 1..pix_levels, concatenated */,
                     int pix_levels);

/* the same with the escape bookkeeping of startNewLongRay (:3198-3233, 3336-3345): escape [nsrc][315] = per star
 * ndotRemaining[7], ndotBoundary[7], ndotDust, ndotSpectrum[300]; out_sigma [4][300] as fo_stellar_beta_table returns it or
 * NULL (spectrum left zero); fraction [nsrc][7] of the `src:` line (:1342-1348); escape, fraction may be NULL */
int fo_point_sources_escape(int n, int64_t ncell, const int32_t *level, const double *HI, const double *HeI, const double *HeII,
                            const double *rho, const double *abun2, double box, int dust, int nsrc, const int64_t *src_leaf,
                            const double *src_ndot, const double *tables, double *rates, int *highest_pixel_level,
                            const double *pix, int pix_levels, const double *out_sigma, double *escape, double *fraction);

/* radiativetransfer_amd/csrc/ftte_math.h evaluated on the host, element-wise (for tests of the
 * device arithmetic itself): e = exp(-tau), g = (1-exp(-tau))/tau; out = (acc/nseg)*w. */
void fo_device_attenuation(int64_t count, const double *tau, double *e, double *g);
void fo_device_log(int64_t count, const double *x, double *out);
void fo_device_cell_mean(int64_t count, const double *acc, int nseg, double w, double *out);
void fo_device_segment_emit(int64_t count, double *I, const double *tau, const double *eta, const double *src, double *mean);
/* ftte_segment_source element-wise (a source function: the exact path mean) */
void fo_device_segment_source(int64_t count, double *I, const double *tau, const double *src, double *mean);

/* ---- ionisation equilibrium (ftte_oracle_chem.c): solveRateEquations, equiSources.f90:3459-3677 ---- */
long fo_solve_rate_equations(int n, long ncell, const int32_t *level, double box, const double *rho, const double *tgas,
                             double *HI_io, double *HeI_io, double *HeII_io, const double *krate, int run_uvb, const double *J,
                             const double *ksi, const double *uniform, double threshold, int nratec, double logtem0,
                             double logtem9, double dlogtem, const double *k, long *iterations);

void fo_uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma);

void fo_assign_uvb_radiation(long ncell, int nnu, const double *HI, const double *HeI, const double *HeII, const double *rho,
                             const double *uvb, double threshold, double *J);

void fo_uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma);

void fo_coll_rates(double T, int recombination_type, double *k);
void fo_rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                                double *logtem9, double *dlogtem);

#ifdef __cplusplus
}
#endif
#endif
