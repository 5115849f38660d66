// ftte_point.h -- host side of the point-source path (rows P1-P3 of the scope table): the stellar rate
// tables (stellarBetaTable.f90), the table look-up (getRatesHydrogenHelium, equiSources.f90:4157-4311) and
// the long-characteristics tracer with HEALPix ray splitting (startNewLongRay, equiSources.f90:3120-3385).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "ftte_amr.h"
#include "ftte_internal.h"

namespace ftte {

constexpr int kFrequencies = 400;    // nfreq, stellarBetaTable.f90:14
constexpr int kSplitBatch = 1024;    // sources traced together; bounds the split queues (3072 records each)

// Everything the point-source path keeps on the device.  Owned by the context.
struct PointState {
    // the tree, uploaded on first use after ftte_set_grid (nothing is uploaded for a uniform grid)
    NodeRec *node = nullptr;
    bool tree_ready = false;
    std::vector<int32_t> node_of_leaf; // cell-array index -> node
    // HI, HeI, HeII, rho, abun2 in cell-array order
    double *medium[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t medium_cells = 0;
    int dust = 0;
    bool medium_ready = false;
    double *packed = nullptr;   // [ncell][kCellRec] copy the tracer reads, rebuilt when the medium changes
    bool packed_ready = false;
    bool rho_given = false;   // ftte_set_medium received a density (the equilibrium update needs it)
    // [6][11^4] rate tables and their logarithms
    double *tables = nullptr, *logtab = nullptr;
    bool tables_ready = false;
    FreqBin *bins = nullptr;
    double *pixdir = nullptr; // [kPixelCount][3]
    double rmax[30];
    // [ncell][kCellRec] krate24, krate25, krate26, crate24, crate25, crate26, 0, 0: what the tracer adds into
    double *rates = nullptr;
    int64_t rates_cells = 0;
    double *rate_planes = nullptr; // [6][ncell]: the layout of the interface, filled on request
    // tracer scratch
    SplitRec *queue[2] = {nullptr, nullptr};
    int32_t queue_capacity = 0;
    int32_t *counters = nullptr; // [0] queue length, [1] highest pixel level, [2] error, [4..5] 64-bit count of cell crossings
    long long ray_steps = 0;     // of the last trace
    int32_t *src_node = nullptr;
    double *src_ndot = nullptr;
    int32_t src_capacity = 0;
    double *sample_in = nullptr, *sample_out = nullptr;
    int32_t sample_capacity = 0;
    // escape bookkeeping of the last trace (startNewLongRay, equiSources.f90:3198-3233, 3336-3345): per star ndotRemaining[7],
    // ndotBoundary[7], ndotDust, ndotSpectrum[300]
    double *escape = nullptr;            // device, [stars of the call][kEscapeRec]
    size_t escape_capacity = 0;
    std::vector<double> escape_host, escape_ndot; // the same on the host after the trace; the stars' photon rates
    double *sigma_ratio = nullptr;       // device [4][300]: outputSigma* / threshold cross-section (stellarBetaTable.f90:119-152)
    bool sigma_ready = false;

    void release();
    void drop_grid(); // after ftte_set_grid: tree, medium and rates belong to the old grid
};

// dustCrossSection, dustModule.f90:30-73 (SMC branch); a_smc is the Fortran array a_smc(7,5), lambda in micron
double dust_cross_section(double lambda_um, const double *a_smc);
// stellarPopulation, stellarPopulationModule.f90:7-50.  spec is the Fortran array specificLuminosity(nmetal,nspectrum,nwave)
double stellar_population(const double *spec, int nmetal, int nspectrum, int nwave, const double *wavelength, int iSpectrum,
                          double coefSpectrum, int iMetal, double coefMetal, double freq_ev);
// uvbBetaTable(nfreq, freqdel, alpha), uvbBetaTable.f90:3-305: group-averaged cross-sections beta[species HI, HeI, HeII][group],
// photo-rate coefficients ksi[group][24, 25, 26] and heating coefficients gamma[group][HI, HeI, HeII] of the three frequency groups
void uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma);
// uniformTable(nfreq, freqdel, alphaQuasar, alphaStellar), uniformTable.f90:1-200: ksi[component quasar, stellar][24, 25, 26] and
// gamma[component][HI, HeI, HeII] of the two power-law components of the uniform background
void uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma);
// coll_rates (coll_rates.f:42-150): k1..k6 at temperature T; recombination_type 1 = case A, 2 = case B (definitionsModule.f90:48)
void coll_rates(double T, int recombination_type, double *k);
// k1a..k6a(nratec) as calc_rates.f:324-337 fills them, with the table bounds of equiSources.f90:174-176; k[6][nratec]
void rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                             double *logtem9, double *dlogtem);
// rmax(1:30), equiSources.f90:296-309
void rmax_table(double *rmax30);

// Each returns 0 or an ftte_status and fills *err.
int point_stellar_beta_table(PointState &P, hipStream_t stream, const double *a_smc, int nwave, const double *wavelength,
                             int nspectrum, int nmetal, const double *spec, int iSpectrum, double coefSpectrum, int iMetal,
                             double coefMetal, double *total_integral, std::string *err);
int point_set_tables(PointState &P, hipStream_t stream, const double *tables, std::string *err);
int point_get_tables(PointState &P, hipStream_t stream, double *tables, std::string *err);
int point_lookup(PointState &P, hipStream_t stream, int dust, int nsample, const double *tau, double *rates, std::string *err);
int point_set_medium(PointState &P, hipStream_t stream, int64_t ncell, const double *const field[5], bool on_device, int dust,
                     std::string *err);
int point_zero_rates(PointState &P, hipStream_t stream, int64_t ncell, std::string *err);
// rates in the interface's layout [6][ncell], in device memory (valid until the next trace)
int point_rate_planes(PointState &P, hipStream_t stream, double **planes, std::string *err);
int point_set_rates(PointState &P, hipStream_t stream, int64_t ncell, const double *planes_host, std::string *err);
int point_trace(PointState &P, hipStream_t stream, const AmrTree &tree, double box, int nsrc, const int64_t *src_cell,
                const double *src_ndot, int *highest_pixel_level, std::string *err);
// outputSigma24, 25, 26, Dust [4][300] (absolute cross-sections, as the reference's module arrays hold them)
int point_set_output_sigma(PointState &P, hipStream_t stream, const double *sigma, std::string *err);

} // namespace ftte
