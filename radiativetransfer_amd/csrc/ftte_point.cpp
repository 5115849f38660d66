// ftte_point.cpp -- host side of the point-source path.  See ftte_point.h.
#include "ftte_point.h"

#include <cmath>
#include <cstring>

#include "ftte_geometry.h"
#include "ftte_kernels.h"

namespace ftte {

namespace {

// The reference writes most constants as default-real literals; they reach the double-precision arithmetic widened
// from single precision.  W() spells that out.
#define W(x) ((double)(x##f))

const double kHydrogen = W(13.598), kHeI = W(24.587), kHeII = W(54.418); // definitionsModule.f90:30-32
inline double c_light() { return W(2.99792458e10); }
inline double ev_to_erg() { return 1.60217646e-12; }
inline double ev_to_hz() { return 1.60217646e-12 / W(6.6260693e-27); }
inline double pi_ref() { return (double)3.141592654f; }

inline double pow4(double x) { return x * x * x * x; }

int hip_fail(std::string *err, const char *what, hipError_t e)
{
    *err = std::string(what) + ": " + hipGetErrorString(e);
    return FTTE_ERR_NO_DEVICE;
}

#define POINT_HIP(call)                                                                                            \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return hip_fail(err, #call, e_);                                                     \
    } while (0)

template <class T> int ensure(T *&p, size_t count, std::string *err)
{
    if (p) return 0;
    POINT_HIP(hipMalloc((void **)&p, sizeof(T) * count));
    return 0;
}

template <class T> void drop(T *&p)
{
    if (p) (void)hipFree(p);
    p = nullptr;
}

// hydrogenic photoionisation cross-section above threshold, stellarBetaTable.f90:40-44 and :52-56
inline double hydrogenic(double sigma0, double threshold, double nu)
{
    const double dum = std::sqrt(nu / threshold - 1);
    return sigma0 * pow4(threshold / nu) * std::exp(4.0 - 4.0 * std::atan(dum) / dum) / (1 - std::exp(-2.0 * pi_ref() / dum));
}

} // namespace

void PointState::drop_grid()
{
    drop(node);
    tree_ready = false;
    for (auto &m : medium) drop(m);
    drop(packed);
    packed_ready = false;
    medium_cells = 0;
    medium_ready = false;
    rho_given = false;
    drop(rates); drop(rate_planes);
    rates_cells = 0;
    std::vector<int32_t>().swap(node_of_leaf);
}

void PointState::release()
{
    drop_grid();
    drop(tables); drop(logtab); drop(bins); drop(pixdir);
    drop(queue[0]); drop(queue[1]); drop(counters); drop(src_node); drop(src_ndot); drop(sample_in); drop(sample_out);
    drop(escape); drop(sigma_ratio);
    escape_capacity = 0; sigma_ready = false;
    tables_ready = false;
    queue_capacity = src_capacity = sample_capacity = 0;
}

double dust_cross_section(double lambda_um, const double *a_smc)
{
    double sigma = 0.0;
    for (int i = 0; i < 7; ++i) {
        // a_smc(i,1:5): lambda_i, a_i, b_i, p_i, q_i
        const double l = a_smc[i], a = a_smc[i + 7], b = a_smc[i + 14], p = a_smc[i + 21], q = a_smc[i + 28];
        const double x = lambda_um / l;
        sigma = sigma + a / (std::pow(x, p) + std::pow(x, -q) + b);
    }
    return W(1.1) * sigma * (double)0.9210340372f;
}

double stellar_population(const double *spec, int nmetal, int nspectrum, int nwave, const double *wavelength, int iSpectrum,
                          double cS, int iMetal, double cM, double freq_ev)
{
    auto SL = [&](int m, int s, int w) { return spec[(size_t)(m - 1) + (size_t)nmetal * ((size_t)(s - 1) + (size_t)nspectrum * (size_t)(w - 1))]; };
    const double lam = c_light() / (freq_ev * ev_to_hz());
    int iw = 1;
    while (iw + 1 < nwave && lam > wavelength[iw]) ++iw;
    double cw = (lam - wavelength[iw - 1]) / (wavelength[iw] - wavelength[iw - 1]);
    cw = std::fmin(std::fmax(0.0, cw), 1.0);
    const double sp1 = cS * ((1.0 - cw) * SL(iMetal, iSpectrum + 1, iw) + cw * 1.0 * SL(iMetal, iSpectrum + 1, iw + 1)) +
                       (1.0 - cS) * ((1.0 - cw) * SL(iMetal, iSpectrum, iw) + cw * SL(iMetal, iSpectrum, iw + 1));
    const double sp2 = cS * ((1.0 - cw) * SL(iMetal + 1, iSpectrum + 1, iw) + cw * 1.0 * SL(iMetal + 1, iSpectrum + 1, iw + 1)) +
                       (1.0 - cS) * ((1.0 - cw) * SL(iMetal + 1, iSpectrum, iw) + cw * SL(iMetal + 1, iSpectrum, iw + 1));
    double sp = (1.0 - cM) * sp1 + cM * sp2;
    const double nu_hz = freq_ev * ev_to_hz();
    sp = std::pow(10.0, sp) / W(1.e-8) * c_light() / (nu_hz * nu_hz);
    return sp;
}

void uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma)
{
    // uvbBetaTable.f90:40-66: the frequency grid and the three cross-sections on it
    std::vector<double> nu(nfreq), s24(nfreq), s25(nfreq), s26(nfreq);
    for (int i = 0; i < nfreq; ++i) {
        nu[i] = std::pow(10.0, (double)i * freqdel);
        s24[i] = nu[i] > kHydrogen ? hydrogenic(W(6.3e-18), kHydrogen, nu[i]) : 0.0;
        s25[i] = nu[i] > kHeII ? hydrogenic(W(1.58e-18), kHeII, nu[i]) : 0.0;
        s26[i] = nu[i] > kHeI ? W(7.42e-18) * (W(1.66) * std::pow(nu[i] / kHeI, (double)(-2.05f)) -
                                               W(0.66) * std::pow(nu[i] / kHeI, (double)(-3.05f)))
                              : 0.0;
    }
    const double nu1 = kHydrogen, nu2 = kHeI, nu3 = kHeII;
    const double lo[3] = {nu1, nu2, nu3}, hi[3] = {nu2, nu3, HUGE_VAL};
    double b[3][3] = {}, k[3][3] = {}, g[3][3] = {}; // [group][24, 25, 26] / [group][HI, HeI, HeII]
    for (int i = 1; i < nfreq; ++i) { // :171-252
        const double freq = nu[i], delta_nu = nu[i] - nu[i - 1];
        for (int q = 0; q < 3; ++q) {
            if (!(freq >= lo[q] && freq <= hi[q])) continue;
            const double dtmp = std::pow(freq / lo[q], -alpha[q]) * delta_nu;
            const double over = dtmp * ev_to_hz() / (freq * ev_to_erg());
            b[q][0] = b[q][0] + dtmp * s24[i]; b[q][1] = b[q][1] + dtmp * s25[i]; b[q][2] = b[q][2] + dtmp * s26[i];
            k[q][0] = k[q][0] + over * s24[i]; k[q][1] = k[q][1] + over * s25[i]; k[q][2] = k[q][2] + over * s26[i];
            g[q][0] = g[q][0] + over * (freq - nu1) * ev_to_erg() * s24[i];
            if (q >= 1) g[q][1] = g[q][1] + over * (freq - nu2) * ev_to_erg() * s26[i];
            if (q == 2) g[q][2] = g[q][2] + over * (freq - nu3) * ev_to_erg() * s25[i];
        }
    }
    // :254-296: beta is normalised by the group's energy shape
    const double shape[3] = {(1. - std::pow(nu2 / nu1, 1. - alpha[0])) / (alpha[0] - 1.),
                             (1. - std::pow(nu3 / nu2, 1. - alpha[1])) / (alpha[1] - 1.), 1. / (alpha[2] - 1.)};
    for (int q = 0; q < 3; ++q) {
        const double energy_shape = shape[q] * lo[q];
        // out: beta[species HI, HeI, HeII][group]; ksi[group][24, 25, 26]; gamma[group][HI, HeI, HeII]
        beta[0 * 3 + q] = b[q][0] / energy_shape;
        beta[1 * 3 + q] = b[q][2] / energy_shape;
        beta[2 * 3 + q] = b[q][1] / energy_shape;
        for (int r = 0; r < 3; ++r) { ksi[q * 3 + r] = k[q][r]; gamma[q * 3 + r] = g[q][r]; }
    }
}

void uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma)
{
    const double nu1 = kHydrogen, nu2 = kHeI, nu3 = kHeII;
    const double alpha[2] = {alpha_quasar, alpha_stellar};
    for (int q = 0; q < 6; ++q) ksi[q] = gamma[q] = 0.0;
    double prev = 0.0;
    for (int i = 0; i < nfreq; ++i) {
        const double nu = std::pow(10.0, (double)i * freqdel);
        const double s24 = nu > kHydrogen ? hydrogenic(W(6.3e-18), kHydrogen, nu) : 0.0;
        const double s25 = nu > kHeII ? hydrogenic(W(1.58e-18), kHeII, nu) : 0.0;
        const double s26 = nu > kHeI ? W(7.42e-18) * (W(1.66) * std::pow(nu / kHeI, (double)(-2.05f)) - W(0.66) * std::pow(nu / kHeI, (double)(-3.05f)))
                                     : 0.0;
        if (i >= 1) { // uniformTable.f90:136-190
            const double delta_nu = nu - prev;
            for (int c = 0; c < 2; ++c) {
                const double dtmp = std::pow(nu / nu1, -alpha[c]) * delta_nu;
                const double over = dtmp * ev_to_hz() / (nu * ev_to_erg());
                if (nu >= nu1) {
                    ksi[3 * c + 0] = ksi[3 * c + 0] + over * s24;
                    ksi[3 * c + 1] = ksi[3 * c + 1] + over * s25;
                    ksi[3 * c + 2] = ksi[3 * c + 2] + over * s26;
                    gamma[3 * c + 0] = gamma[3 * c + 0] + over * (nu - nu1) * ev_to_erg() * s24;
                }
                if (nu >= nu2) gamma[3 * c + 1] = gamma[3 * c + 1] + over * (nu - nu2) * ev_to_erg() * s26;
                if (nu >= nu3) gamma[3 * c + 2] = gamma[3 * c + 2] + over * (nu - nu3) * ev_to_erg() * s25;
            }
        }
        prev = nu;
    }
}

namespace {
inline double powi_left(double x, int n)
{
    double r = x;
    for (int i = 1; i < n; ++i) r = r * x;
    return r;
}
} // namespace

void coll_rates(double T, int recombination_type, double *k)
{
    // coll_rates.f:62-150 (fits of Abel et al. 1997 and Hui & Gnedin 1997); its literals are single precision
    const double T_eV = T / W(11605.);
    const double L = std::log(T_eV);
    if (T_eV > W(0.8)) {
        k[0] = std::exp(W(-32.71396786375) + W(13.53655609057) * L - W(5.739328757388) * powi_left(L, 2) + W(1.563154982022) * powi_left(L, 3) -
                        W(0.2877056004391) * powi_left(L, 4) + W(0.03482559773736999) * powi_left(L, 5) - W(0.00263197617559) * powi_left(L, 6) +
                        W(0.0001119543953861) * powi_left(L, 7) - W(2.039149852002e-6) * powi_left(L, 8));
        k[2] = std::exp(W(-44.09864886561001) + W(23.91596563469) * L - W(10.75323019821) * powi_left(L, 2) + W(3.058038757198) * powi_left(L, 3) -
                        W(0.5685118909884001) * powi_left(L, 4) + W(0.06795391233790001) * powi_left(L, 5) -
                        W(0.005009056101857001) * powi_left(L, 6) + W(0.0002067236157507) * powi_left(L, 7) -
                        W(3.649161410833e-6) * powi_left(L, 8));
        k[4] = std::exp(W(-68.71040990212001) + W(43.93347632635) * L - W(18.48066993568) * powi_left(L, 2) + W(4.701626486759002) * powi_left(L, 3) -
                        W(0.7692466334492) * powi_left(L, 4) + W(0.08113042097303) * powi_left(L, 5) - W(0.005324020628287001) * powi_left(L, 6) +
                        W(0.0001975705312221) * powi_left(L, 7) - W(3.165581065665e-6) * powi_left(L, 8));
    } else {
        k[0] = k[2] = k[4] = W(1.0e-20);
    }
    const double kb = 1.3806503e-16, ev = 1.60217646e-12;
    if (recombination_type == 1) { // case A
        if (T_eV > W(0.8))
            k[3] = W(1.54e-9) * (1. + W(0.3) / std::exp(W(8.099328789667) / T_eV)) / (std::exp(W(40.49664394833662) / T_eV) * std::pow(T_eV, W(1.5))) +
                   W(3.92e-13) / std::pow(T_eV, W(0.6353));
        else k[3] = W(3.92e-13) / std::pow(T_eV, W(0.6353));
        if (T > W(5500.0))
            k[1] = std::exp(W(-28.61303380689232) - W(0.7241125657826851) * L - W(0.02026044731984691) * powi_left(L, 2) -
                            W(0.002380861877349834) * powi_left(L, 3) - W(0.0003212605213188796) * powi_left(L, 4) -
                            W(0.00001421502914054107) * powi_left(L, 5) + W(4.989108920299513e-6) * powi_left(L, 6) +
                            W(5.755614137575758e-7) * powi_left(L, 7) - W(1.856767039775261e-8) * powi_left(L, 8) -
                            W(3.071135243196595e-9) * powi_left(L, 9));
        else k[1] = k[3];
        k[5] = W(3.36e-10) / std::sqrt(T) / std::pow(T / W(1.e3), W(0.2)) / (1. + std::pow(T / W(1.e6), W(0.7)));
    } else { // case B
        double tmp = (double)(2.f * 24.587f) * ev / (kb * T);
        k[3] = W(1.26e-14) * (std::sqrt(tmp) * std::sqrt(std::sqrt(tmp))); // tmp**0.750 as the reference's compiler forms it
        tmp = (double)(2.f * 13.598f) * ev / (kb * T);
        k[1] = W(2.753e-14) * std::pow(tmp, W(1.500)) / std::pow(1. + std::pow(tmp / W(2.740), W(0.407)), W(2.242));
        tmp = (double)(2.f * 54.418f) * ev / (kb * T);
        k[5] = (double)(2.f * 2.753e-14f) * std::pow(tmp, W(1.500)) / std::pow(1. + std::pow(tmp / W(2.740), W(0.407)), W(2.242));
    }
}

void rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                             double *logtem9, double *dlogtem)
{
    *logtem0 = std::log(temstart);
    *logtem9 = std::log(temend);
    *dlogtem = (std::log(temend) - std::log(temstart)) / (double)(float)(nratec - 1);
    for (int i = 1; i <= nratec; ++i) { // calc_rates.f:324-337
        const double ttt = std::exp(std::log(temstart) + (double)(float)(i - 1) * *dlogtem);
        double six[6];
        coll_rates(ttt, recombination_type, six);
        for (int r = 0; r < 6; ++r) k[(size_t)r * nratec + (i - 1)] = six[r];
    }
}

void rmax_table(double *rmax30)
{
    for (int ir = 1; ir <= 30; ++ir) {
        const float v = std::sqrt(3.f) * (std::sqrt(0.5f * std::pow(4.f, (float)(ir - 1)) - 1.f / 12.f) + 0.5f);
        rmax30[ir - 1] = (double)v / 2.0;
    }
}

int point_stellar_beta_table(PointState &P, hipStream_t stream, const double *a_smc, int nwave, const double *wavelength,
                             int nspectrum, int nmetal, const double *spec, int iSpectrum, double cS, int iMetal, double cM,
                             double *total_integral, std::string *err)
{
    // the frequency grid and the cross-sections on it, stellarBetaTable.f90:27-66
    static thread_local double nu[kFrequencies], s24[kFrequencies], s25[kFrequencies], s26[kFrequencies], sd[kFrequencies];
    const double freqdel = W(0.02);
    for (int i = 0; i < kFrequencies; ++i) {
        nu[i] = std::pow(10.0, (double)i * freqdel);
        const double lambda = c_light() / (nu[i] * ev_to_hz()) * W(1.e8); // Angstrom
        sd[i] = dust_cross_section(lambda / W(1.e4), a_smc) * W(1.e-22);
        s24[i] = nu[i] > kHydrogen ? hydrogenic(W(6.3e-18), kHydrogen, nu[i]) : 0.0;
        s25[i] = nu[i] > kHeII ? hydrogenic(W(1.58e-18), kHeII, nu[i]) : 0.0;
        s26[i] = nu[i] > kHeI ? W(7.42e-18) * (W(1.66) * std::pow(nu[i] / kHeI, (double)(-2.05f)) -
                                               W(0.66) * std::pow(nu[i] / kHeI, (double)(-3.05f)))
                              : 0.0;
    }
    // what every depth tuple needs from a frequency bin, :217-232 and :243-246
    std::vector<FreqBin> bins(kFrequencies - 1);
    double total = 0.0;
    const double thr[3] = {kHydrogen, kHeI, kHeII};
    for (int i = 1; i < kFrequencies; ++i) {
        const double freq = nu[i], delta_nu = nu[i] - nu[i - 1];
        const double lum = stellar_population(spec, nmetal, nspectrum, nwave, wavelength, iSpectrum, cS, iMetal, cM, freq);
        FreqBin &B = bins[i - 1];
        B.dtmp = lum / (freq * ev_to_erg()) * delta_nu * ev_to_hz();
        if (freq >= kHydrogen) total = total + B.dtmp;
        B.r24 = s24[i] / W(6.3e-18);
        B.r26 = s26[i] / W(7.42e-18);
        B.r25 = s25[i] / W(1.58e-18);
        B.rdust = sd[i] / W(5.4116737e-22);
        for (int r = 0; r < 3; ++r) B.excess[r] = freq >= thr[r] ? (freq - thr[r]) * ev_to_erg() : -1.0;
    }
    if (total_integral) *total_integral = total;

    int rc;
    {   // the output energies and the cross-sections on them, stellarBetaTable.f90:119-152 (nenergy = 300 between lowerEnergy
        // and upperEnergy, definitionsModule.f90:290-292)
        double sigma[4 * kOutputEnergies];
        const double lower = kHydrogen, upper = W(10.) * kHydrogen; // lowerEnergy, upperEnergy
        for (int ie = 1; ie <= kOutputEnergies; ++ie) {
            // float(ienergy-1)/float(nenergy-1) is a single-precision quotient, :122
            const double freq = lower * std::exp((double)((float)(ie - 1) / (float)(kOutputEnergies - 1)) * (std::log(upper) - std::log(lower)));
            const double lambda = c_light() / (freq * ev_to_hz()) * W(1.e8);
            sigma[3 * kOutputEnergies + ie - 1] = dust_cross_section(lambda / W(1.e4), a_smc) * W(1.e-22);
            sigma[0 * kOutputEnergies + ie - 1] = freq > kHydrogen ? hydrogenic(W(6.3e-18), kHydrogen, freq) : freq == kHydrogen ? W(6.3e-18) : 0.0;
            sigma[1 * kOutputEnergies + ie - 1] = freq > kHeII ? hydrogenic(W(1.58e-18), kHeII, freq) : 0.0;
            sigma[2 * kOutputEnergies + ie - 1] = freq > kHeI ? W(7.42e-18) * (W(1.66) * std::pow(freq / kHeI, (double)(-2.05f)) -
                                                                               W(0.66) * std::pow(freq / kHeI, (double)(-3.05f)))
                                                              : 0.0;
        }
        if ((rc = point_set_output_sigma(P, stream, sigma, err))) return rc;
    }
    if ((rc = ensure(P.tables, (size_t)6 * kTableSize, err))) return rc;
    if ((rc = ensure(P.logtab, (size_t)6 * kTableSize, err))) return rc;
    if ((rc = ensure(P.bins, (size_t)kFrequencies, err))) return rc;
    POINT_HIP(hipMemcpyAsync(P.bins, bins.data(), sizeof(FreqBin) * bins.size(), hipMemcpyHostToDevice, stream));
    if (launch_rate_table(P.bins, (int)bins.size(), P.tables, P.logtab, stream)) { *err = "rate table kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
    POINT_HIP(hipStreamSynchronize(stream)); // `bins` leaves scope
    P.tables_ready = true;
    return 0;
}

int point_set_output_sigma(PointState &P, hipStream_t stream, const double *sigma, std::string *err)
{
    // the tracer multiplies the threshold depths by sigma(E) / sigma(threshold), :3216-3219
    double ratio[4 * kOutputEnergies];
    const double thr[4] = {W(6.30e-18), W(1.58e-18), W(7.42e-18), W(5.4116737e-22)};
    for (int q = 0; q < 4; ++q)
        for (int ie = 0; ie < kOutputEnergies; ++ie) ratio[q * kOutputEnergies + ie] = sigma[q * kOutputEnergies + ie] / thr[q];
    int rc;
    if ((rc = ensure(P.sigma_ratio, (size_t)4 * kOutputEnergies, err))) return rc;
    POINT_HIP(hipMemcpyAsync(P.sigma_ratio, ratio, sizeof ratio, hipMemcpyHostToDevice, stream));
    POINT_HIP(hipStreamSynchronize(stream));
    P.sigma_ready = true;
    return 0;
}

int point_set_tables(PointState &P, hipStream_t stream, const double *tables, std::string *err)
{
    P.sigma_ready = false; // the cross-sections belong to the population whose tables these replace: ftte_set_output_sigma
    int rc;
    if ((rc = ensure(P.tables, (size_t)6 * kTableSize, err))) return rc;
    if ((rc = ensure(P.logtab, (size_t)6 * kTableSize, err))) return rc;
    POINT_HIP(hipMemcpyAsync(P.tables, tables, sizeof(double) * 6 * kTableSize, hipMemcpyHostToDevice, stream));
    if (launch_log_table(P.tables, P.logtab, stream)) { *err = "table kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
    POINT_HIP(hipStreamSynchronize(stream));
    P.tables_ready = true;
    return 0;
}

int point_get_tables(PointState &P, hipStream_t stream, double *tables, std::string *err)
{
    if (!P.tables_ready) { *err = "no rate tables: call ftte_stellar_beta_table or ftte_set_rate_tables first"; return FTTE_ERR_STATE; }
    POINT_HIP(hipMemcpyAsync(tables, P.tables, sizeof(double) * 6 * kTableSize, hipMemcpyDeviceToHost, stream));
    POINT_HIP(hipStreamSynchronize(stream));
    return 0;
}

int point_lookup(PointState &P, hipStream_t stream, int dust, int nsample, const double *tau, double *rates, std::string *err)
{
    if (!P.tables_ready) { *err = "no rate tables: call ftte_stellar_beta_table or ftte_set_rate_tables first"; return FTTE_ERR_STATE; }
    if (nsample > P.sample_capacity) {
        drop(P.sample_in); drop(P.sample_out);
        P.sample_capacity = 0;
        POINT_HIP(hipMalloc((void **)&P.sample_in, sizeof(double) * 4 * nsample));
        POINT_HIP(hipMalloc((void **)&P.sample_out, sizeof(double) * 6 * nsample));
        P.sample_capacity = nsample;
    }
    POINT_HIP(hipMemcpyAsync(P.sample_in, tau, sizeof(double) * 4 * nsample, hipMemcpyHostToDevice, stream));
    if (launch_rate_lookup(P.logtab, dust, nsample, P.sample_in, P.sample_out, stream)) { *err = "look-up kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
    POINT_HIP(hipMemcpyAsync(rates, P.sample_out, sizeof(double) * 6 * nsample, hipMemcpyDeviceToHost, stream));
    POINT_HIP(hipStreamSynchronize(stream));
    return 0;
}

int point_set_medium(PointState &P, hipStream_t stream, int64_t ncell, const double *const field[5], bool on_device, int dust,
                     std::string *err)
{
    if (P.medium_cells != ncell) {
        for (auto &m : P.medium) drop(m);
        drop(P.packed);
        P.medium_cells = 0;
        P.medium_ready = false;
    }
    P.packed_ready = false;
    for (int f = 0; f < 5; ++f) {
        int rc;
        if ((rc = ensure(P.medium[f], (size_t)ncell, err))) return rc;
        if (field[f])
            POINT_HIP(hipMemcpyAsync(P.medium[f], field[f], sizeof(double) * ncell, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream));
        else
            POINT_HIP(hipMemsetAsync(P.medium[f], 0, sizeof(double) * ncell, stream)); // rho, abun2 are only read with dust
    }
    POINT_HIP(hipStreamSynchronize(stream));
    P.medium_cells = ncell;
    P.dust = dust;
    P.medium_ready = true;
    P.rho_given = field[3] != nullptr;
    return 0;
}

int point_zero_rates(PointState &P, hipStream_t stream, int64_t ncell, std::string *err)
{
    if (P.rates_cells != ncell) { drop(P.rates); drop(P.rate_planes); P.rates_cells = 0; }
    int rc;
    if ((rc = ensure(P.rates, (size_t)kCellRec * ncell, err))) return rc;
    P.rates_cells = ncell;
    POINT_HIP(hipMemsetAsync(P.rates, 0, sizeof(double) * kCellRec * ncell, stream));
    return 0;
}

int point_rate_planes(PointState &P, hipStream_t stream, double **planes, std::string *err)
{
    int rc;
    if ((rc = ensure(P.rate_planes, (size_t)6 * P.rates_cells, err))) return rc;
    if (launch_repack_rates(P.rate_planes, P.rates, (long)P.rates_cells, false, stream)) { *err = "layout kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
    *planes = P.rate_planes;
    return 0;
}

int point_set_rates(PointState &P, hipStream_t stream, int64_t ncell, const double *planes_host, std::string *err)
{
    int rc;
    if ((rc = point_zero_rates(P, stream, ncell, err))) return rc;
    if ((rc = ensure(P.rate_planes, (size_t)6 * ncell, err))) return rc;
    POINT_HIP(hipMemcpyAsync(P.rate_planes, planes_host, sizeof(double) * 6 * ncell, hipMemcpyHostToDevice, stream));
    if (launch_repack_rates(P.rate_planes, P.rates, (long)ncell, true, stream)) { *err = "layout kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
    POINT_HIP(hipStreamSynchronize(stream));
    return 0;
}

int point_trace(PointState &P, hipStream_t stream, const AmrTree &tree, double box, int nsrc, const int64_t *src_cell,
                const double *src_ndot, int *highest_pixel_level, std::string *err)
{
    if (!P.tables_ready) { *err = "no rate tables: call ftte_stellar_beta_table or ftte_set_rate_tables first"; return FTTE_ERR_STATE; }
    if (!P.medium_ready || P.medium_cells != tree.ncell) { *err = "no medium: call ftte_set_medium after ftte_set_grid"; return FTTE_ERR_STATE; }
    int rc;
    if (!P.rates || P.rates_cells != tree.ncell)
        if ((rc = point_zero_rates(P, stream, tree.ncell, err))) return rc;

    const size_t nnode = tree.parent.size();
    if (!P.tree_ready) {
        drop(P.node);
        if (tree.refined()) {
            std::vector<NodeRec> nodes(nnode);
            for (size_t v = 0; v < nnode; ++v) nodes[v] = NodeRec{tree.child0[v], tree.leaf[v], tree.parent[v], (int32_t)tree.level[v]};
            if ((rc = ensure(P.node, nnode, err))) return rc;
            POINT_HIP(hipMemcpyAsync(P.node, nodes.data(), sizeof(NodeRec) * nnode, hipMemcpyHostToDevice, stream));
            POINT_HIP(hipStreamSynchronize(stream));
        }
        P.tree_ready = true;
    }
    if (!P.packed_ready) {
        if ((rc = ensure(P.packed, (size_t)kCellRec * tree.ncell, err))) return rc;
        if (launch_pack_medium(P.medium, P.packed, (long)tree.ncell, stream)) { *err = "layout kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
        P.packed_ready = true;
    }
    if (!P.pixdir) {
        // unit vectors of every pixel of levels 1..6.  The reference evaluates cos(phi)*cos(theta), sin(phi)*cos(theta),
        // sin(theta) where it needs them (equiSources.f90:2437-2439, :3331-3333); the products are formed here once.
        std::vector<double> dir((size_t)3 * kPixelCount);
        size_t at = 0;
        for (int L = 1; L <= kMaxPixelLevel; ++L) {
            const int nside = 1 << (L - 1);
            const int64_t npix = (int64_t)12 * nside * nside;
            for (int64_t ip = 0; ip < npix; ++ip, ++at) {
                double phi, theta;
                if (pix2ang_nest(nside, ip, &phi, &theta)) { *err = "pix2ang_nest failed"; return FTTE_ERR_PIXEL; }
                dir[3 * at + 0] = std::cos(phi) * std::cos(theta);
                dir[3 * at + 1] = std::sin(phi) * std::cos(theta);
                dir[3 * at + 2] = std::sin(theta);
            }
        }
        if ((rc = ensure(P.pixdir, dir.size(), err))) return rc;
        POINT_HIP(hipMemcpyAsync(P.pixdir, dir.data(), sizeof(double) * dir.size(), hipMemcpyHostToDevice, stream));
        POINT_HIP(hipStreamSynchronize(stream));
        rmax_table(P.rmax);
    }
    if (!P.counters) {
        if ((rc = ensure(P.counters, 8, err))) return rc;
    }
    const int batch_max = nsrc < kSplitBatch ? nsrc : kSplitBatch;
    const int32_t need = batch_max * 3072; // at most 12 * 4^4 rays of one source split into level 6
    if (need > P.queue_capacity) {
        drop(P.queue[0]); drop(P.queue[1]);
        P.queue_capacity = 0;
        POINT_HIP(hipMalloc((void **)&P.queue[0], sizeof(SplitRec) * need));
        POINT_HIP(hipMalloc((void **)&P.queue[1], sizeof(SplitRec) * need));
        P.queue_capacity = need;
    }
    if (batch_max > P.src_capacity) {
        drop(P.src_node); drop(P.src_ndot);
        P.src_capacity = 0;
        POINT_HIP(hipMalloc((void **)&P.src_node, sizeof(int32_t) * batch_max));
        POINT_HIP(hipMalloc((void **)&P.src_ndot, sizeof(double) * batch_max));
        P.src_capacity = batch_max;
    }

    // cell-array index -> tree node (kept until the grid changes; the identity on a uniform grid)
    if (tree.refined() && P.node_of_leaf.size() != (size_t)tree.ncell) {
        P.node_of_leaf.assign((size_t)tree.ncell, -1);
        for (size_t v = 0; v < nnode; ++v)
            if (tree.leaf[v] >= 0) P.node_of_leaf[(size_t)tree.leaf[v]] = (int32_t)v;
    }
    std::vector<int32_t> node_of(nsrc);
    for (int s = 0; s < nsrc; ++s) {
        if (src_cell[s] < 0 || src_cell[s] >= tree.ncell) { *err = "ftte_point_sources: source cell outside the cell array"; return FTTE_ERR_ARG; }
        node_of[s] = tree.refined() ? P.node_of_leaf[(size_t)src_cell[s]] : (int32_t)src_cell[s];
    }

    if ((size_t)nsrc * kEscapeRec > P.escape_capacity) {
        drop(P.escape);
        P.escape_capacity = 0;
        POINT_HIP(hipMalloc((void **)&P.escape, sizeof(double) * (size_t)nsrc * kEscapeRec));
        P.escape_capacity = (size_t)nsrc * kEscapeRec;
    }
    POINT_HIP(hipMemsetAsync(P.escape, 0, sizeof(double) * (size_t)nsrc * kEscapeRec, stream)); // :1267-1270

    TraceRec T;
    std::memset(&T, 0, sizeof(T));
    T.sigma_ratio = P.sigma_ready ? P.sigma_ratio : nullptr;
    {
        static const float radii[kOutputRadii] = {0.1f, 0.3f, 1.f, 3.f, 10.f, 30.f, 100.f}; // outputRadius [kpc], equiSources.f90:10
        for (int ir = 0; ir < kOutputRadii; ++ir) T.out_radius_kpc[ir] = (double)radii[ir];
        T.kpc = W(1.e3) * W(3.08568025e18); // definitionsModule.f90:21-22
    }
    T.node = tree.refined() ? P.node : nullptr;
    T.n = tree.n; T.dust = P.dust; T.ncell = tree.ncell; T.box = box;
    T.medium = P.packed;
    T.logtab = P.logtab; T.pixdir = P.pixdir;
    T.rmax[0] = 0.0;
    for (int L = 1; L <= kMaxPixelLevel; ++L) T.rmax[L] = P.rmax[L - 1];
    T.rates = P.rates;
    T.src_node = P.src_node; T.src_ndot = P.src_ndot;
    T.out_count = P.counters; T.highest_level = P.counters + 1; T.error = P.counters + 2;
    T.steps = reinterpret_cast<unsigned long long *>(P.counters + 4);
    T.out_capacity = P.queue_capacity;

    int32_t host_counters[4] = {0, 0, 0, 0};
    int highest = 0;
    for (int s0 = 0; s0 < nsrc; s0 += batch_max) {
        const int ns = nsrc - s0 < batch_max ? nsrc - s0 : batch_max;
        POINT_HIP(hipMemcpyAsync(P.src_node, node_of.data() + s0, sizeof(int32_t) * ns, hipMemcpyHostToDevice, stream));
        POINT_HIP(hipMemcpyAsync(P.src_ndot, src_ndot + s0, sizeof(double) * ns, hipMemcpyHostToDevice, stream));
        int32_t nrec = 0;
        T.escape = P.escape + (size_t)s0 * kEscapeRec;
        for (int L = 1; L <= kMaxPixelLevel; ++L) {
            T.pixel_level = L;
            T.nrays = L == 1 ? 12 * ns : 4 * nrec;
            if (T.nrays == 0) break;
            T.in = P.queue[L & 1];
            T.out = P.queue[(L + 1) & 1];
            // queue length back to zero, highest level and error carried over
            host_counters[0] = 0;
            host_counters[1] = highest;
            POINT_HIP(hipMemcpyAsync(P.counters, host_counters, sizeof(int32_t) * 2, hipMemcpyHostToDevice, stream));
            if (L == 1 && s0 == 0) POINT_HIP(hipMemsetAsync(P.counters + 2, 0, sizeof(int32_t) * 6, stream));
            if (launch_point_trace(T, stream)) { *err = "tracer kernel failed to launch"; return FTTE_ERR_NO_DEVICE; }
            POINT_HIP(hipMemcpyAsync(host_counters, P.counters, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, stream));
            POINT_HIP(hipStreamSynchronize(stream));
            nrec = host_counters[0];
            highest = host_counters[1];
            if (host_counters[2]) {
                static const char *what[] = {"", "continuation cell outside the base grid", "error in checkPoint (start)",
                                             "error in checkPoint", "ray did not terminate", "split queue overflow"};
                const int e = host_counters[2] < 6 ? host_counters[2] : 3;
                *err = std::string("ftte_point_sources: ") + what[e];
                return FTTE_ERR_PATTERN;
            }
        }
    }
    unsigned long long steps = 0;
    POINT_HIP(hipMemcpyAsync(&steps, P.counters + 4, sizeof steps, hipMemcpyDeviceToHost, stream));
    P.escape_host.resize((size_t)nsrc * kEscapeRec);
    P.escape_ndot.assign(src_ndot, src_ndot + nsrc);
    POINT_HIP(hipMemcpyAsync(P.escape_host.data(), P.escape, sizeof(double) * (size_t)nsrc * kEscapeRec, hipMemcpyDeviceToHost, stream));
    POINT_HIP(hipStreamSynchronize(stream));
    P.ray_steps = (long long)steps;
    if (highest_pixel_level) *highest_pixel_level = highest;
    return 0;
}

} // namespace ftte
