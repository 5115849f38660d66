/* ftte.h -- C ABI of the MI355X diffuse radiative-transfer sweep.
 *
 * The reference (razoumov/radiativeTransfer, "FTTE") has no FFI of its own: its hot path
 * is reached through `use transportRoutinesModule` (equiSources.f90:25) and through code
 * inlined in the main program (equiSources.f90:1372-1808), with state in module globals
 * (definitionsModule.f90:55-62,104,182,256).  This header is the seam cut at
 * equiSources.f90:1383-1806: everything between `computeOpacities` and the chemistry.
 * Each entry point names the reference code it stands in for.
 *
 *   in : the cell array (definitionsModule.f90:323-326: depth-first leaf list, `level` per
 *        leaf, k fastest on the base grid), per-leaf opacities kappa_nu (or species densities
 *        + cross-sections), the inflow uvb_nu, a direction list (phi, theta, weight) BEFORE
 *        folding, the box size
 *   out: J_nu per leaf, same order, overwritten (the reference zeroes J in computeOpacities,
 *        equiSources.f90:4964-4966, then accumulates one term per direction,
 *        transportRoutinesModule.f90:953-955)
 *
 * Conventions: plain C types, caller owns every host array, the library owns its device
 * buffers, every function returns 0 on success or a negative ftte_status (it never calls
 * exit: where the reference executes `stop`, the status says which `stop`).  All arithmetic
 * is IEEE binary64.  A context is not thread-safe; use one per host thread / per GPU.
 *
 * The library is linked against the HIP runtime and needs a gfx950 device for every call
 * that computes on the grid; the host-only geometry helpers at the end work without one.
 */
#ifndef FTTE_H
#define FTTE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ftte_ctx ftte_ctx;

typedef enum {
    FTTE_OK = 0,
    FTTE_ERR_ARG = -1,            /* null pointer, non-positive size, ...                         */
    FTTE_ERR_STATE = -2,          /* call order: grid / opacity not set                           */
    FTTE_ERR_NO_DEVICE = -3,      /* no usable HIP device, or a HIP call failed                    */
    FTTE_ERR_UNSUPPORTED = -4,    /* valid request this build does not implement (see message)     */
    FTTE_ERR_NOT_CUBIC = -5,      /* equiSources.f90:436-439 'base grid needs to be of size n^3'   */
    FTTE_ERR_LEVELS = -6,         /* readCellArray.f90:182 'error in levels'                       */
    FTTE_ERR_PHI = -7,            /* equiSources.f90:1412 'error in phi' (on a quadrant boundary)  */
    FTTE_ERR_THETA = -8,          /* equiSources.f90:1426 'error in theta'                         */
    FTTE_ERR_DOMINANT_AXIS = -9,  /* equiSources.f90:1450 'error in theta or phi' (tie)            */
    FTTE_ERR_PATTERN = -10,       /* transportRoutinesModule.f90:33-36,60-63; equiSources.f90:1523 */
    FTTE_ERR_IZONE = -11,         /* rotateIndices called with izone outside 1..24                 */
    FTTE_ERR_PIXEL = -12,         /* equiSources.f90:2152-2160 'nside/ipix out of range'           */
    FTTE_ERR_RATES = -13,         /* equiSources.f90:3637-3654: a species fraction left [0, 1]     */
    FTTE_ERR_MEMORY = -14,        /* not enough device memory for the request (see message)         */
    FTTE_ERR_STALLED = -15        /* a one-launch sweep (option "dataflow") stopped making progress; its J is not valid */
} ftte_status;

/* ---- lifetime ------------------------------------------------------------------------------ */

/* ndev = 1: one context drives one device (dev_ids[0] = HIP device ordinal; dev_ids may be NULL for the current device).
 *
 * ndev > 1: ONE context, driven by the one host thread the reference is (its direction loop, equiSources.f90:1385-1391, is serial;
 * the directions' only coupling is the sum transportRoutinesModule.f90:953-955), sweeps on dev_ids[0 .. ndev-1].  Such a context
 * takes HOST arrays: ftte_set_grid, ftte_set_opacity, ftte_set_emissivity / ftte_set_source_function, ftte_diffuse_sweep,
 * ftte_diffuse_iteration (plus ftte_set_option, ftte_host_register / _unregister, ftte_counter, ftte_launch_info for the first
 * device's share, ftte_last_error, ftte_destroy); every other entry point returns FTTE_ERR_UNSUPPORTED on it.  The work is split
 * frequency groups first, then directions: with r_nu = gcd(ndev, groups) and r_dir = ndev / r_nu, device k sweeps the groups of
 * frequency slice k mod r_nu for the directions of slice k div r_nu.  A frequency-sharded J_nu is complete where it is computed
 * (8 groups on 8 devices: nothing is exchanged, every device sends its groups home); where directions are split too, the r_dir
 * devices that hold the same groups sum their J by a reduce-scatter over RCCL (librccl.so, loaded on first need) and each sends
 * its piece home.  Where RCCL cannot serve (two entries of dev_ids name the same device -- a test on a one-GPU box --, or no
 * library) the pieces are summed by a kernel that reads the partners' buffers in place; ftte_multi_info says which.
 * J equals the single-device J to the rounding of the sum over directions.  The other way to many GPUs -- one process and one
 * context per GPU, J combined by the host driver (radiativetransfer_amd/distributed.py over torch.distributed) -- stays. */
int ftte_create(ftte_ctx **ctx, int ndev, const int *dev_ids);
/* How the last sweep of a multi-device context combined its devices' J, in words ("" for a single-device context). */
const char *ftte_multi_info(const ftte_ctx *ctx);
int ftte_destroy(ftte_ctx *ctx);
/* Message of the last failing call on this context ("" if none); ctx may be NULL for the
 * message of the last failing ftte_create. */
const char *ftte_last_error(const ftte_ctx *ctx);

/* ---- inputs -------------------------------------------------------------------------------- */

/* The grid: stands in for the tree the reference builds from the cell array
 * (readCellArray.f90:154-187 createFullyThreadedStructure) and for physicalBoxSize
 * (definitionsModule.f90:256).  nx == ny == nz is required, as in the reference.
 * level[ncell]: depth-first leaf list, 0 = base cell.  A list of all zeros (ncell = nx^3) is a
 * uniform grid and takes the tiled sweep kernel; a refined cell array takes the general
 * segment-forest path (setRaysRefined / findNeighbours / transport of the reference).
 * The reference's tree is static over a run: a call with the list the context already holds returns at once and
 * keeps the tree, the sweep plans and the device-resident medium (only box_cm is taken over). */
int ftte_set_grid(ftte_ctx *ctx, int nx, int ny, int nz, int64_t ncell, const int32_t *level, double box_cm);

/* Opacities kappa[nnu][ncell] in cell-array order (host memory), cm^-1.  Stands in for the
 * kappa1..3 fields filled by computeOpacities (equiSources.f90:4977-4980); nnu is free
 * (the reference hard-wires 3 groups, definitionsModule.f90:169-171). */
int ftte_set_opacity(ftte_ctx *ctx, int nnu, const double *kappa);
/* Same, kappa already resident in device memory (not retained beyond the call).  The data must be complete when
 * the call is made (synchronise the stream that produced it); the same holds for the other *_device setters. */
int ftte_set_opacity_device(ftte_ctx *ctx, int nnu, const double *kappa_dev);
/* computeOpacities itself (equiSources.f90:4956-4983) for nnu groups, on the device:
 * kappa_g = HI*beta[0][g] + HeI*beta[1][g] + HeII*beta[2][g] (left to right).
 * HI/HeI/HeII: [ncell] host arrays, beta: [3][nnu] host array (rows: HI = beta24,
 * HeI = beta26, HeII = beta25 of the reference's group tables). */
int ftte_set_species(ftte_ctx *ctx, int nnu, const double *HI, const double *HeI, const double *HeII,
                     const double *beta);
/* Emissivity eta[nnu][ncell] (call after the opacities are set; sized by their nnu).  NULL selects the reference's
 * hard-wired zero emissivity (transportRoutinesModule.f90:673-675).  Non-NULL switches the sweep to the reference's
 * emission term as written at :676,   Iout = Iin*tmpabs + nemi*tmpemi/dpath   with tmpemi = (1-tmpabs)/kappa
 * (dpath below tau = 1e-10), and to its log-mean (:1044-1048) for the cell intensity.  Replaces any source function. */
int ftte_set_emissivity(ftte_ctx *ctx, const double *eta);
int ftte_set_emissivity_device(ftte_ctx *ctx, const double *eta_dev);
/* Source function S[nnu][ncell]: NOT in the reference (whose emission term is never enabled and is not an emissivity
 * per unit length): Iout = Iin*exp(-tau) + S*(1 - exp(-tau)), the form a source iteration S = (1-eps) J + eps B needs
 * (BASELINE configs[4]).  NULL switches emission off.  Replaces any emissivity. */
int ftte_set_source_function(ftte_ctx *ctx, const double *S);
int ftte_set_source_function_device(ftte_ctx *ctx, const double *S_dev);

/* ---- the sweep ----------------------------------------------------------------------------- */

/* One diffuse-transfer iteration = equiSources.f90:1385-1806 for a caller-supplied direction
 * list: for every direction, fold it (:1395-1454), build the per-layer ray patterns
 * (:1495-1534, setPattern), and sweep all cells (:1572-1796 / transport), accumulating
 * J_nu += w * mean-over-segments(log-mean intensity).
 * phi[ndir] in (0, 2 pi), theta[ndir] in (-pi/2, pi/2) (the reference's un-folded angles, as
 * pix2ang_nest returns them), w[ndir], uvb[nnu] (inflow on every upstream boundary face,
 * definitionsModule.f90:55-56), J[nnu][ncell] host memory, overwritten. */
int ftte_diffuse_sweep(ftte_ctx *ctx, int ndir, const double *phi, const double *theta, const double *w,
                       const double *uvb, double *J);
/* ftte_set_opacity + ftte_diffuse_sweep in one call (the pair the drop-in for the reference's runUVBTransfer block makes on
 * every outer iteration), and faster than the two: on a uniform grid the frequency groups travel in lanes --
 * the first lane is swept while the second one's opacities are still crossing PCIe, and its J goes back while the second is
 * swept.  Same J as the two calls; refined cell arrays and the options that exclude lanes take the two calls internally.
 * kappa[nnu][ncell], J[nnu][ncell] host memory (pageable, or registered with ftte_host_register: DMA in place). */
int ftte_diffuse_iteration(ftte_ctx *ctx, int nnu, const double *kappa, int ndir, const double *phi, const double *theta,
                           const double *w, const double *uvb, double *J);
/* Same with J in device memory.  `stream` is a hipStream_t; NULL = the context's own stream, a blocking stream,
 * i.e. one that is implicitly ordered with the legacy default stream.  The call is asynchronous with respect to the
 * host, ordered on that stream. */
int ftte_diffuse_sweep_device(ftte_ctx *ctx, int ndir, const double *phi, const double *theta, const double *w,
                              const double *uvb, double *J_dev, void *stream);

/* ---- point sources ---------------------------------------------------------------------------
 * The `runStellarTransfer` block, equiSources.f90:1256-1370: for each star particle build the rate
 * tables of its population (stellarBetaTable), then send 12 HEALPix rays from the centre of its host
 * cell; a ray splits into its four daughter pixels after rmax(level) cells (startNewLongRay,
 * :3120-3385), crosses cells with drawSegment (:2412-2595) and the neighbour search
 * (find/zoom??Neighbour, :2647-2960), and deposits in every cell it crosses
 * ndot * (R(depth) - R(depth + tau)) for the three photo-reactions, R from the tables
 * (getRatesHydrogenHelium, :4157-4311).
 *
 * Host sequence, mirroring the reference:
 *   ftte_set_grid; ftte_set_medium; ftte_set_zero_rates;
 *   per population: ftte_stellar_beta_table (or ftte_set_rate_tables); ftte_point_sources(stars of it);
 *   ftte_get_point_rates.
 * Rates accumulate on the device between ftte_set_zero_rates and ftte_get_point_rates. */

#define FTTE_TABLE_SIZE 14641 /* (ndepth+1)^4 = 11^4, definitionsModule.f90:72-77 */
#define FTTE_MAX_PIXEL_LEVEL 6 /* maxPixelLevel, equiSources.f90:9 */

/* stellarBetaTable(nfbins, frequencyBinWidth, totalIntegral, iSpectrum, coefSpectrum, iMetal, coefMetal),
 * stellarBetaTable.f90:3-289, with the module data it reads passed explicitly, all as the Fortran
 * arrays lie in memory: a_smc(7,5) (read at dustModule.f90:16-21), wavelength(nwave) [cm, ascending],
 * specificLuminosity(nmetal, nspectrum, nwave) (definitionsModule.f90:270; 5 x 37 x 1221 in the
 * reference).  iSpectrum, iMetal are 1-based.  The six tables reactionRate1..3, energyRate1..3 are
 * accumulated on the device over the 399 frequency bins and stay there; total_integral (may be NULL)
 * is the reference's totalIntegral. */
int ftte_stellar_beta_table(ftte_ctx *ctx, const double *a_smc, int nwave, const double *wavelength_cm, int nspectrum,
                            int nmetal, const double *specific_luminosity, int iSpectrum, double coefSpectrum, int iMetal,
                            double coefMetal, double *total_integral);
/* Tables computed elsewhere (e.g. by the reference itself): tables[6][FTTE_TABLE_SIZE], order
 * reactionRate1, 2, 3, energyRate1, 2, 3, each the Fortran array (0:ndepth,0:ndepth,0:ndepth,0:ndepth)
 * (tau1, tau2, tau3, tauDust) as it lies in memory. */
int ftte_set_rate_tables(ftte_ctx *ctx, const double *tables);
int ftte_get_rate_tables(ftte_ctx *ctx, double *tables);
/* getRatesHydrogenHelium(reaction, tau1, tau2, tau3, tauDust, numberRate, heatingRate),
 * equiSources.f90:4157-4311, for nsample depth tuples and all three reactions, evaluated on the device:
 * tau[nsample][4], rates[nsample][3][2] = (numberRate, heatingRate) per reaction.
 * dust_approximation: 0 noDust, 1 dust ~ HI, 2 dust ~ total hydrogen (definitionsModule.f90:87). */
int ftte_get_rates_hydrogen_helium(ftte_ctx *ctx, int dust_approximation, int nsample, const double *tau, double *rates);
/* Cell fields the tracer reads (zoneType HI, HeI, HeII, rho, abun2, definitionsModule.f90:163-168), cell-array
 * order, ncell each.  rho / abun2 may be NULL when the dust approximation does not read them. */
int ftte_set_medium(ftte_ctx *ctx, const double *HI, const double *HeI, const double *HeII, const double *rho,
                    const double *abun2, int dust_approximation);
int ftte_set_medium_device(ftte_ctx *ctx, const double *HI, const double *HeI, const double *HeII, const double *rho,
                           const double *abun2, int dust_approximation);
/* setZeroRates, equiSources.f90:4128-4155 */
int ftte_set_zero_rates(ftte_ctx *ctx);
/* localizeCellFromStar, equiSources.f90:2597-2620: the star's call sequence position(1:3*(level+1)) (base
 * indices 1..n, then child indices 1..2 per level) -> 0-based cell-array index of its host cell. */
int ftte_locate_cell(ftte_ctx *ctx, int level, const int32_t *position, int64_t *cell);
/* The per-star loop body, equiSources.f90:1268-1329, for nsrc stars that share the current tables:
 * src_cell[nsrc] host cells (0-based cell-array index), src_ndot[nsrc] = float(weight).  Adds into the device
 * rates.  highest_pixel_level (may be NULL): the largest value of the reference's highestPixelLevel over
 * these stars.  Deposition uses fp64 atomics: the last bits of the sums depend on the run. */
int ftte_point_sources(ftte_ctx *ctx, int nsrc, const int64_t *src_cell, const double *src_ndot, int *highest_pixel_level);
/* The escape bookkeeping startNewLongRay keeps per star (equiSources.f90:3198-3233, :3336-3345; reset per star at :1267-1270), for
 * the nsrc stars of the LAST ftte_point_sources call, in its order: remaining[nsrc][7] = ndotRemaining and boundary[nsrc][7] =
 * ndotBoundary at outputRadius = 0.1, 0.3, 1, 3, 10, 30, 100 kpc (equiSources.f90:10), dust[nsrc] = ndotDust, spectrum[nsrc][300] =
 * ndotSpectrum at the last radius (zero unless the output cross-sections are known: ftte_stellar_beta_table computes them,
 * ftte_set_output_sigma hands them over), fraction[nsrc][7] as the main program forms it for its `src:` line (:1342-1348:
 * remaining / (ndot - boundary), or 0 once a whole photon unit has left through the box).  Any pointer may be NULL.  The sums are
 * accumulated with fp64 atomics like the rates. */
int ftte_point_escape(ftte_ctx *ctx, int nsrc, double *remaining, double *boundary, double *dust, double *spectrum, double *fraction);
/* outputSigma24, outputSigma25, outputSigma26, outputSigmaDust (definitionsModule.f90:293-294; stellarBetaTable.f90:119-152),
 * sigma[4][300] in that order, for tables that came in through ftte_set_rate_tables */
int ftte_set_output_sigma(ftte_ctx *ctx, const double *sigma);
/* rates[6][ncell]: krate24, krate25, krate26, crate24, crate25, crate26 (zoneType, definitionsModule.f90:166) */
int ftte_get_point_rates(ftte_ctx *ctx, double *rates);
/* The same [6][ncell] in device memory: a copy in the interface's layout (the tracer accumulates into a packed array),
 * valid until the rates change */
int ftte_point_rates_device(ftte_ctx *ctx, double **rates_dev);
/* Replace the device-resident rates, e.g. by the sum over the GPUs that traced different stars; same layout */
int ftte_set_point_rates(ftte_ctx *ctx, const double *rates);
/* cell crossings (ray segments drawn) of the last ftte_point_sources, all rays of all stars */
long long ftte_point_ray_steps(const ftte_ctx *ctx);
/* rmax(1:30), equiSources.f90:296-309 (formula, halved) */
int ftte_rmax(double *rmax30);
/* uvbBetaTable(nfreq, freqdel, alpha), uvbBetaTable.f90:3-305 (host): the three frequency groups [nu1, nu2], [nu2, nu3],
 * [nu3, inf) with power-law slopes alpha[3].  beta[3][3] = [species HI, HeI, HeII][group] (group%beta24, beta26, beta25:
 * the matrix ftte_set_species / ftte_compute_opacities take; computeOpacities uses its lower triangle,
 * equiSources.f90:4977-4980), ksi[3][3] = [group][ksi24, ksi25, ksi26] (what ftte_solve_rate_equations takes),
 * gamma[3][3] = [group][gammaHI, gammaHeI, gammaHeII].  The reference calls it with nfbins = 400, frequencyBinWidth = 0.02
 * (equiSources.f90:253). */
int ftte_uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma);
/* coll_rates(T, k1, ..., k19, recombinationType), coll_rates.f:42-150 (host): the six rate coefficients the equilibrium uses,
 * k[6] = k1..k6 [cm^3/s]; recombination_type 1 = case A, 2 = case B (the reference's setting, definitionsModule.f90:48) */
int ftte_coll_rates(double T, int recombination_type, double *k);
/* k1a..k6a(nratec) as calc_rates.f:324-337 fills them, k[6][nratec], and logtem0, logtem9, dlogtem of equiSources.f90:174-176:
 * what ftte_set_rate_coefficients takes (the reference: nratec = 5000, temstart = 1., temend = 1.e8) */
int ftte_rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                                 double *logtem9, double *dlogtem);
/* uniformTable(nfreq, freqdel, alphaQuasar, alphaStellar), uniformTable.f90:1-200 (host): the two power-law components of the
 * uniform background.  ksi[2][3] = [quasar, stellar][ksi24, ksi25, ksi26], gamma[2][3] = [quasar, stellar][gammaHI, gammaHeI,
 * gammaHeII]; uniformQuasar*ksi[0][x] + uniformStellar*ksi[1][x] is the `uniform` argument of ftte_solve_rate_equations. */
int ftte_uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma);
/* dustCrossSection(lambda [micron]), dustModule.f90:30-73, SMC curve; a_smc(7,5) Fortran order */
double ftte_dust_cross_section(double lambda_micron, const double *a_smc);

/* ---- ionisation equilibrium --------------------------------------------------------------------
 * solveRateEquations(currentCell, nx, runUVBTransfer), equiSources.f90:3459-3677, for every leaf: the
 * consumer of J (Jmean1..3) and of the point-source rates (krate24..26), and the producer of the next
 * iteration's HI, HeI, HeII (driver loop equiSources.f90:1811-1819).  Per cell: photo-rates per
 * absorber, collisional rate coefficients interpolated in log T, bisection on the electron density.
 * The update works on the device-resident medium of ftte_set_medium (which must have been given rho)
 * and leaves the new HI, HeI, HeII there; every operation is the reference's, in its order. */

/* k1a..k6a(nratec) of calc_rates / coll_rates (calc_rates.f:324-337) and the table's logtem0, logtem9,
 * dlogtem (equiSources.f90:174-176) */
int ftte_set_rate_coefficients(ftte_ctx *ctx, int nratec, double logtem0, double logtem9, double dlogtem, const double *k1a,
                               const double *k2a, const double *k3a, const double *k4a, const double *k5a, const double *k6a);
/* zoneType%tgas per leaf [K] */
int ftte_set_temperature(ftte_ctx *ctx, const double *tgas);
/* run_uvb_transfer != 0 (:3545-3553): J[3][ncell] (Jmean1..3) and ksi[3][3] = (ksi24, ksi25, ksi26) of group1..3
 * (normCrossSectionType, definitionsModule.f90:94-103); uniform may be NULL.
 * run_uvb_transfer == 0 (:3554-3562): uniform[3] = uniformQuasar*quasar%ksi2x + uniformStellar*stellar%ksi2x for
 * x = 4, 5, 6, applied where the Lyman-limit mean free path is at least self_shielding_threshold; J, ksi may be NULL.
 * use_point_rates != 0: krate24..26 of the device-resident point-source rates enter (:3520-3542); 0: they are zero.
 * max_change (may be NULL): largest change of HI/nH, HeI/nHe, HeII/nHe over the cells.
 * FTTE_ERR_RATES where the reference prints the species and stops; the medium is then left unchanged. */
int ftte_solve_rate_equations(ftte_ctx *ctx, int run_uvb_transfer, const double *J, const double *ksi, const double *uniform,
                              double self_shielding_threshold, int use_point_rates, double *max_change);
/* Same with J in device memory, e.g. the output of ftte_diffuse_sweep_device: the iteration stays on the device */
int ftte_solve_rate_equations_device(ftte_ctx *ctx, int run_uvb_transfer, const double *J_dev, const double *ksi,
                                     const double *uniform, double self_shielding_threshold, int use_point_rates,
                                     double *max_change);
/* HI, HeI, HeII of the device-resident medium, ncell each */
int ftte_get_medium(ftte_ctx *ctx, double *HI, double *HeI, double *HeII);
/* computeOpacities (equiSources.f90:4956-4983) on the device-resident medium: kappa[g] = HI beta[0][g] + HeI beta[1][g] +
 * HeII beta[2][g]; equivalent to ftte_set_species with the medium's arrays, without the host round trip */
int ftte_compute_opacities(ftte_ctx *ctx, int nnu, const double *beta);
/* assignUvbRadiation(currentCell), transportRoutinesModule.f90:1056-1093: the optically thin alternative to the sweep (not called
 * in the reference's loop): J[g][cell] = uvb[g] where the Lyman-limit mean free path 1/(min(HI, psi rho/mh) 6.3e-18 + HeI 7.42e-18 +
 * HeII 1.58e-18) of the device-resident medium is at least the threshold, else 0.  J[nnu][ncell]. */
int ftte_assign_uvb_radiation(ftte_ctx *ctx, int nnu, const double *uvb, double self_shielding_threshold, double *J);
int ftte_assign_uvb_radiation_device(ftte_ctx *ctx, int nnu, const double *uvb, double self_shielding_threshold, double *J_dev);
/* bisection steps of the last ftte_solve_rate_equations*, summed over the cells */
long long ftte_rate_equation_steps(const ftte_ctx *ctx);

/* ---- grid ingest (no device needed) -----------------------------------------------------------------
 * What the reference does between reading its grid file and the first transfer (equiSources.f90:427-618,
 * placeCellProjectWithVelocity :1870-1974): per refinement level a list of SPH-projected cells -- position [kpc], log10 of
 * temperature, hydrogen density and neutral fraction, optionally velocities and four abundances -- becomes the octree, and the
 * octree the cell array in writeCell's order (:4044-4079), the order ftte_set_grid, ftte_set_medium and the .dat file take.
 * Level 1 must fill an n^3 base grid (FTTE_ERR_NOT_CUBIC otherwise, :436-439); a cell of level L lies L-1 refinements deep;
 * cells of a refined cell that no list covers keep their parent's state, as in the reference.  The HDF4 container the
 * reference reads the lists from (:316-423) is not read here: the lists are handed over as arrays. */
typedef struct {
    int64_t ncell;
    const float *pos;            /* (ncell,3) as the Fortran array lies: all x, then all y, then all z [kpc] */
    const float *lT, *lnH, *lx;  /* log10 T [K], log10 n_H [cm^-3], log10 neutral fraction */
    const float *vel;            /* (ncell,3) or NULL (the reference's readKinematics) */
    const float *abun;           /* (ncell,4) or NULL (readMetals); column 2 becomes abun2, smoothed on level 1 (:526-578) */
} ftte_level_list;
typedef struct ftte_cellarray ftte_cellarray;
int ftte_ingest_levels(int nlevels, const ftte_level_list *lists, ftte_cellarray **out);
/* base grid size, number of leaves, physicalBoxSize [cm] (:481) */
int ftte_cellarray_info(const ftte_cellarray *a, int *nx, int64_t *ncell, double *box_cm, int *has_velocity, int *has_metals);
/* the leaf arrays, cell-array order, as the tree holds them (binary64); any pointer may be NULL.  tgas and rho are zoneType's
 * fields; abun2 is 0.02 everywhere without metals (:1958). */
int ftte_cellarray_fields(const ftte_cellarray *a, int32_t *level, double *HI, double *HeI, double *HeII, double *tgas, double *rho,
                          double *velx, double *vely, double *velz, double *abun2);
void ftte_cellarray_free(ftte_cellarray *a);

/* ---- host arrays ------------------------------------------------------------------------------
 * The reference keeps its fields in host memory (the zoneType tree, definitionsModule.f90:163-180); the drop-ins
 * flatten them into cell-array order and hand them over.  Pageable arrays cross PCIe through pinned staging blocks
 * inside the library; an array registered here is pinned in place and moved by DMA directly, for as long as it stays
 * registered (the caller unregisters it before freeing it).  bytes counts from ptr. */
int ftte_host_register(ftte_ctx *ctx, void *ptr, size_t bytes);
int ftte_host_unregister(ftte_ctx *ctx, void *ptr);

/* ---- tuning and instrumentation ------------------------------------------------------------- */

/* How often the expensive host-side builds have run on this context: "grid_builds" (tree rebuilt from a level list:
 * ftte_set_grid with an unchanged list keeps the tree, the plans and the resident medium), "plan_builds" (tiled-sweep
 * planner), "forest_builds" (per-direction segment forests of a refined cell array); of the hybrid sweep's current plan, "hybrid_boxes"
 * (boxes around clusters of refined cells, of the izone that has most) and "hybrid_passes" (passes their forests are swept in), 0
 * when the last sweep did not take the hybrid path; "fine_block": fine cells a side of the refined block that plan sweeps with bricks of
 * its own (0: none); "brick_form": the form of the brick kernel the last uniform-grid sweep of the
 * brick engine took (option "team": 0 or 2; -1 before the first); "devices"; of a multi-device context also "frequency_slices",
 * "direction_slices" and "multi_rccl" (1: the last direction-split sweep was summed over RCCL), "rccl_loadable", "rccl_selftest" (runs the
 * direction sum's RCCL calls on a clique of one rank, the first device: 1 = the piece came back unchanged; negative = it could not
 * run), the rest from its first device.
 * -1 for an unknown name. */
long long ftte_counter(const ftte_ctx *ctx, const char *name);

/* Tuning knobs.  0 means "automatic" where noted.  Results do not depend on any of them except through the order in which the
 * directions' contributions are added into J (last bits); unknown keys return FTTE_ERR_ARG.
 *   uniform grids
 *     "engine"      0 automatic = 2 cell-fixed bricks, 1 ray-following tiles (3a)
 *     "chunk"       bricks: layers per brick (0: 16, fewer with few frequency groups)
 *     "group"       bricks: most directions of one izone that share a brick pass, 1..8 (0: 3)
 *     "share"       bricks: groups sharing a J accumulator: 0 none, 1 the passes of one izone, 2 and izone pairs (default)
 *     "lanes"       bricks: streams the frequency groups (or, with fewer groups than lanes, the direction groups) are spread over
 *     "brick_waves" bricks: waves per SIMD the kernel is compiled for, 2..4
 *     "dataflow"    bricks: 0 a launch per stage, 1 one launch whose bricks wait for each other, 2 the same with
 *                   write-through stores, 3 one launch of persistent workgroups with a task queue per XCD
 *     "tiled"       bricks: 1 = opacities and accumulators stored brick by brick (a brick's layer in one piece; grids of whole
 *                   bricks), 2 = the whole brick in one piece; same results, measured without gain; 0 = in frames (default)
 *     "team"        bricks: 0 = one wavefront per brick, 2 = two wavefronts per
 *                   brick, four rows each ("pair_waves": workgroups per SIMD that form is compiled for, 2..4); -1 (default):
 *                   2 with up to four frequency groups on this GPU, else 0
 *     "rows", "stack", "slots", "waves"  tiles: rays per lane (4, 8, 16), wavefronts per workgroup (1, 2, 4, 8), directions in flight
 *                   per launch (1..16), waves per SIMD (2..6); built variants rows x stack = 4x{1,4,8}, 8x{1,2,4}, 16x1
 *   refined cell arrays
 *     "hybrid"      1 bricks outside boxes around the refined cells, segment forests inside (default; 3b), 0 forests for the whole tree
 *     "pipelines"   hybrid: independent bricks-forests-bricks sequences on streams of their own, 1..4 (default 3)
 *     "box_lanes"   hybrid: along a brick's 64 lanes the boxes end on multiples of this (a divisor of 64; default 1: one cell beyond
 *                   the refined cells)
 *     "graph"       hybrid: 1 = the launches of a sweep are captured once into a hipGraph and replayed (default 0: measured slower
 *                   than issuing them on this runtime)
 *     "fine_bricks" hybrid: 1 (default) = a fully refined block -- one cluster that is a cube of base cells refined exactly once, twice
 *                   its side a multiple of 64 -- is swept by bricks of its own on the fine level, the forests keep what lies around it
 *                   (DESIGN.md 3b); 0 = every leaf of a box through the forests.  "fine_chunk": layers per fine brick (0: as "chunk")
 *     "forest_batch" most directions per launch of the segment forests (0: what the path and the device memory allow)
 *     "forest_fuse" runs of forest levels with at most this many (segment, frequency group) pairs in the fullest direction go in ONE
 *                   launch, a workgroup per direction and a barrier per level (default 4096; 0: a launch per level).  Same bits.
 *     "forest"      1: use the segment forests on a uniform grid too, for cross-checks
 *   "ldspad"        diagnostic: extra dynamic LDS per workgroup (bytes), to cap residency
 *   "atomic_acc"    bricks: 1 = later visitors of a shared accumulator add with fp64 atomics instead of read-add-store (default 0)
 *   "queue_mix"     bricks, dataflow = 3: how (frequency group, accumulator) units are dealt to the XCDs' queues: 0 a frequency
 *                   group per queue where they divide, 1 by load, 2 (group + accumulator) mod queues
 *   "ablate"        diagnostic, WRONG RESULTS, timing only: mask of parts of the brick kernel's memory traffic to leave out
 *   multi-device contexts: every option goes to every device; "multi_reduce" 1 = always sum direction slices with the peer kernel */
int ftte_set_option(ftte_ctx *ctx, const char *key, int value);
/* Launch records of the last sweep (valid after the sweep's stream has been synchronised).
 * Each sweep-kernel launch is bracketed by HIP events on the stream it runs on. */
int ftte_launch_count(const ftte_ctx *ctx);
/* ms: device time of launch `idx`; updates: cell.direction.frequency updates it performed. */
int ftte_launch_info(ftte_ctx *ctx, int idx, double *ms, int64_t *updates);

/* ---- host geometry, the reference's callable surface (no device needed) ---------------------- */

/* one layer's ray pattern: patternType of definitionsModule.f90:141-152 without the tree link */
typedef struct {
    double xy_x0, xy_y0, xy_len;
    double xz_x0, xz_z0, xz_len;
    double yz_y0, yz_z0, yz_len;
    int32_t xz_active, yz_active;
    int32_t xy_top, xz_top, yz_top; /* 0 none, 1 xy segment, 2 yz segment, 3 xz segment (:159) */
    int32_t reserved_;
} ftte_pattern;

/* rotateIndices(i,j,k,nx,ny,nz,izone,icell,jcell,kcell), rotateIndicesModule.f90:7-113 */
int ftte_rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *icell, int *jcell, int *kcell);
/* pix2ang_nest + rotateAngles, equiSources.f90:2118-2231, 2297-2335: rotated centre of NESTED
 * pixel ipix; phi in [0, 2 pi), theta = colatitude - pi/2 */
int ftte_pix2ang_nest(int nside, int64_t ipix, double *phi, double *theta);
/* direction folding, equiSources.f90:1395-1454: canonical (phi, theta) and izone 1..24 */
int ftte_fold_direction(double phi_large, double theta_large, double *phi, double *theta, int *izone);
/* setPattern(pattern, phi, theta), transportRoutinesModule.f90:7-85; xy_x0, xy_y0 are inputs */
int ftte_set_pattern(ftte_pattern *pattern, double phi, double theta);
/* patterns of layers 1..n of a folded direction, equiSources.f90:1495-1534 */
int ftte_layer_patterns(int n, double phi, double theta, ftte_pattern *layers);
/* computeCellIntensity(Jmean, Iin, Iout), transportRoutinesModule.f90:1036-1054 */
void ftte_compute_cell_intensity(double *Jmean, double Iin, double Iout);

#ifdef __cplusplus
}
#endif
#endif /* FTTE_H */
