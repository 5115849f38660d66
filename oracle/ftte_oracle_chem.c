/* ftte_oracle_chem.c -- TEST INFRASTRUCTURE ONLY (see ftte_oracle.h).
 *
 * Plain-C restatement of the reference's ionisation-equilibrium update, the consumer of J and of the point-source rates
 * (SURVEY.md 8(f) row F1): solveRateEquations, equiSources.f90:3459-3677, with `opposite` (:5044-5058).  Operation order
 * and float32-widened literals follow the reference so that the vectors produced by its own compiled code
 * (tests/golden/chem*.npz, oracle/chem_harness.f90) are reproduced.
 */
#include "ftte_oracle.h"

#include <math.h>
#include <stddef.h>

#define F(x) ((double)(x##f)) /* a default-real literal of the reference, widened */

static int opposite(double a, double b) { return ((a > 0.) && (b < 0.)) || ((a < 0.) && (b > 0.)); }

typedef struct { double k1, k2, k3, k4, k5, k6, nh, nhe, kr24, kr25, kr26; } eq_t;

/* one evaluation of the two statements the reference repeats three times (:3592-3596, :3600-3604, :3618-3622) */
static double residual(const eq_t *q, double de, double *HeI_out)
{
    const double X = q->k3 * de + q->kr26, Y = q->k4 * de;
    const double HeI = (de - q->nh / (1. + q->k2 * de / (q->k1 * de + q->kr24)) - 2. * q->nhe) / (X / Y - 2. - 2. * X / Y);
    const double res = q->k3 * HeI * de + q->k6 * (q->nhe - HeI - HeI * X / Y) * de + q->kr26 * HeI -
                       HeI * X / Y * (q->k4 * de + q->k5 * de + q->kr25);
    *HeI_out = HeI;
    return res;
}

/* All leaves.  krate: [3][ncell] (krate24, krate25, krate26 per cell, 1/s per cell as the tracer leaves them) or NULL;
 * run_uvb != 0: J [3][ncell] and ksi[3 groups][3] (ksi24, ksi25, ksi26); else uniform[3] (uniformQuasar*quasar%ksi +
 * uniformStellar*stellar%ksi per reaction) and the self-shielding threshold.  k: [6][nratec].  HI, HeI, HeII in/out.
 * Returns 0, or 1 + the index of the first cell at which the reference stops (:3637-3654). */
long fo_solve_rate_equations(int n, long ncell, const int32_t *level, double box, const double *rho, const double *tgas,
                             double *HI_io, double *HeI_io, double *HeII_io, const double *krate, int run_uvb, const double *J,
                             const double *ksi, const double *uniform, double threshold, int nratec, double logtem0,
                             double logtem9, double dlogtem, const double *k, long *iterations)
{
    const double psi = F(0.76), mp = F(1.6726231e-24), mn = F(1.67492728e-24);
    const double mh = mp, mhe = 2. * (mp + mn), pi = F(3.141592654);
    long its = 0;
    for (long c = 0; c < ncell; ++c) {
        eq_t q;
        q.nh = psi * rho[c] / mh;
        q.nhe = (1. - psi) * rho[c] / mhe;
        double HI = fmin(HI_io[c], q.nh);
        double HeI = HeI_io[c], HeII = HeII_io[c];
        double HeIII = q.nhe - HeI_io[c] - HeII_io[c];
        if (HeIII < 0.) {
            HeIII = 0.;
            if (HeII < 0.) HeII = 0.;
        }
        const double size = box / (double)((float)(1 << level[c]) * (float)n);
        q.kr24 = (krate && HI > 0.) ? krate[c] / (size * size * size * HI) : 0.;
        q.kr25 = (krate && HeII > 0.) ? krate[ncell + c] / (size * size * size * HeII) : 0.;
        q.kr26 = (krate && HeI > 0.) ? krate[2 * ncell + c] / (size * size * size * HeI) : 0.;
        q.kr24 = fmax(q.kr24, 0.); q.kr25 = fmax(q.kr25, 0.); q.kr26 = fmax(q.kr26, 0.);
        if (run_uvb) {
            const double t1 = 4. * pi * J[c], t2 = 4. * pi * J[ncell + c], t3 = 4. * pi * J[2 * ncell + c];
            q.kr24 = q.kr24 + t1 * ksi[0] + t2 * ksi[3] + t3 * ksi[6];
            q.kr25 = q.kr25 + t3 * ksi[7];
            q.kr26 = q.kr26 + t2 * ksi[5] + t3 * ksi[8];
        } else {
            const double mfp = 1. / (HI * F(6.3e-18) + HeI * F(7.42e-18) + HeII * F(1.58e-18));
            if (mfp >= threshold) {
                q.kr24 = q.kr24 + 4. * pi * uniform[0];
                q.kr25 = q.kr25 + 4. * pi * uniform[1];
                q.kr26 = q.kr26 + 4. * pi * uniform[2];
            }
        }
        double logtem = log(tgas[c]);
        logtem = fmax(logtem, logtem0);
        logtem = fmin(logtem, logtem9);
        int ix = (int)((logtem - logtem0) / dlogtem) + 1;
        ix = ix < 1 ? 1 : ix;
        ix = ix > nratec - 1 ? nratec - 1 : ix;
        const double t1 = logtem0 + (double)(ix - 1) * dlogtem, t2 = logtem0 + (double)ix * dlogtem, tdef = t2 - t1;
        double kk[6];
        for (int r = 0; r < 6; ++r) {
            const double *ka = k + (size_t)r * nratec;
            kk[r] = ka[ix - 1] + (logtem - t1) * (ka[ix] - ka[ix - 1]) / tdef;
        }
        q.k1 = kk[0]; q.k2 = kk[1]; q.k3 = kk[2]; q.k4 = kk[3]; q.k5 = kk[4]; q.k6 = kk[5];

        double de1 = F(1.e-30), de2 = q.nh + 2. * q.nhe, res1, res, de;
        res1 = residual(&q, de1, &HeI);
        de = de2;
        (void)residual(&q, de2, &HeI);
        double HeIprev = -1.;
        while (fabs(HeI - HeIprev) / q.nhe > 1.e-10) {
            HeIprev = HeI;
            de = 0.5 * (de1 + de2);
            res = residual(&q, de, &HeI);
            if (opposite(res, res1)) de2 = de;
            else { de1 = de; res1 = res; }
            ++its;
        }
        const double X = q.k3 * de + q.kr26, Y = q.k4 * de;
        HeII = HeI * X / Y;
        const double HII = q.nh / (1. + q.k2 * de / (q.k1 * de + q.kr24));
        HI = q.k2 * HII * de / (q.k1 * de + q.kr24);
        if (!(HI / q.nh >= 0. && HI / q.nh <= 1.)) return c + 1;
        if (!(HeI / q.nhe >= 0. && HeI / q.nhe <= 1.)) return c + 1;
        HI_io[c] = HI; HeI_io[c] = HeI; HeII_io[c] = HeII;
    }
    if (iterations) *iterations = its;
    return 0;
}

/* uvbBetaTable, uvbBetaTable.f90:3-305.  out: beta[3 groups][3] (beta24, beta25, beta26), ksi[3][3] (ksi24, ksi25, ksi26),
 * gamma[3][3] (gammaHI, gammaHeI, gammaHeII). */
void fo_uvb_beta_table(int nfreq, double freqdel, const double *alpha, double *beta, double *ksi, double *gamma)
{
    const double nu1 = F(13.598), nu2 = F(24.587), nu3 = F(54.418), pi = F(3.141592654);
    const double ev_to_erg = 1.60217646e-12, ev_to_hz = ev_to_erg / F(6.6260693e-27);
    for (int q = 0; q < 9; ++q) beta[q] = ksi[q] = gamma[q] = 0.0;
    double prev = 0.0;
    for (int i = 0; i < nfreq; ++i) {
        const double nu = pow(10.0, (double)i * freqdel);
        double s24 = 0.0, s25 = 0.0, s26 = 0.0;
        if (nu > nu1) {
            const double dum = sqrt(nu / nu1 - 1);
            const double r = nu1 / nu;
            s24 = F(6.3e-18) * (r * r * r * r) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        }
        if (nu > nu3) {
            const double dum = sqrt(nu / nu3 - 1);
            const double r = nu3 / nu;
            s25 = F(1.58e-18) * (r * r * r * r) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        }
        if (nu > nu2) s26 = F(7.42e-18) * (F(1.66) * pow(nu / nu2, (double)(-2.05f)) - F(0.66) * pow(nu / nu2, (double)(-3.05f)));
        if (i >= 1) {
            const double delta_nu = nu - prev;
            const double lo[3] = {nu1, nu2, nu3};
            const int in[3] = {nu >= nu1 && nu <= nu2, nu >= nu2 && nu <= nu3, nu >= nu3};
            for (int q = 0; q < 3; ++q) {
                if (!in[q]) continue;
                const double dtmp = pow(nu / lo[q], -alpha[q]) * delta_nu;
                const double over = dtmp * ev_to_hz / (nu * ev_to_erg);
                beta[3 * q + 0] = beta[3 * q + 0] + dtmp * s24;
                beta[3 * q + 1] = beta[3 * q + 1] + dtmp * s25;
                beta[3 * q + 2] = beta[3 * q + 2] + dtmp * s26;
                ksi[3 * q + 0] = ksi[3 * q + 0] + over * s24;
                ksi[3 * q + 1] = ksi[3 * q + 1] + over * s25;
                ksi[3 * q + 2] = ksi[3 * q + 2] + over * s26;
                gamma[3 * q + 0] = gamma[3 * q + 0] + over * (nu - nu1) * ev_to_erg * s24;
                if (q >= 1) gamma[3 * q + 1] = gamma[3 * q + 1] + over * (nu - nu2) * ev_to_erg * s26;
                if (q == 2) gamma[3 * q + 2] = gamma[3 * q + 2] + over * (nu - nu3) * ev_to_erg * s25;
            }
        }
        prev = nu;
    }
    const double shape[3] = {(1. - pow(nu2 / nu1, 1. - alpha[0])) / (alpha[0] - 1.), (1. - pow(nu3 / nu2, 1. - alpha[1])) / (alpha[1] - 1.),
                             1. / (alpha[2] - 1.)};
    const double lo[3] = {nu1, nu2, nu3};
    for (int q = 0; q < 3; ++q)
        for (int r = 0; r < 3; ++r) beta[3 * q + r] = beta[3 * q + r] / (shape[q] * lo[q]);
}

/* assignUvbRadiation, transportRoutinesModule.f90:1056-1093.  J: [nnu][ncell]. */
void fo_assign_uvb_radiation(long ncell, int nnu, const double *HI, const double *HeI, const double *HeII, const double *rho,
                             const double *uvb, double threshold, double *J)
{
    const double psi = F(0.76), mh = F(1.6726231e-24);
    for (long c = 0; c < ncell; ++c) {
        const double hi = fmin(HI[c], psi * rho[c] / mh);
        const double mfp = 1. / (hi * F(6.3e-18) + HeI[c] * F(7.42e-18) + HeII[c] * F(1.58e-18));
        for (int g = 0; g < nnu; ++g) J[(size_t)g * ncell + c] = (mfp >= threshold) ? uvb[g] : 0.0;
    }
}

/* uniformTable, uniformTable.f90:1-200.  ksi[2][3] (quasar, stellar) x (24, 25, 26); gamma[2][3] x (HI, HeI, HeII). */
void fo_uniform_table(int nfreq, double freqdel, double alpha_quasar, double alpha_stellar, double *ksi, double *gamma)
{
    const double nu1 = F(13.598), nu2 = F(24.587), nu3 = F(54.418), pi = F(3.141592654);
    const double ev_to_erg = 1.60217646e-12, ev_to_hz = ev_to_erg / F(6.6260693e-27);
    const double alpha[2] = {alpha_quasar, alpha_stellar};
    for (int q = 0; q < 6; ++q) ksi[q] = gamma[q] = 0.0;
    double prev = 0.0;
    for (int i = 0; i < nfreq; ++i) {
        const double nu = pow(10.0, (double)i * freqdel);
        double s24 = 0.0, s25 = 0.0, s26 = 0.0;
        if (nu > nu1) {
            const double dum = sqrt(nu / nu1 - 1), r = nu1 / nu;
            s24 = F(6.3e-18) * (r * r * r * r) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        }
        if (nu > nu3) {
            const double dum = sqrt(nu / nu3 - 1), r = nu3 / nu;
            s25 = F(1.58e-18) * (r * r * r * r) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        }
        if (nu > nu2) s26 = F(7.42e-18) * (F(1.66) * pow(nu / nu2, (double)(-2.05f)) - F(0.66) * pow(nu / nu2, (double)(-3.05f)));
        if (i >= 1) {
            const double delta_nu = nu - prev;
            for (int c = 0; c < 2; ++c) {
                const double dtmp = pow(nu / nu1, -alpha[c]) * delta_nu;
                const double over = dtmp * ev_to_hz / (nu * ev_to_erg);
                if (nu >= nu1) {
                    ksi[3 * c + 0] = ksi[3 * c + 0] + over * s24;
                    ksi[3 * c + 1] = ksi[3 * c + 1] + over * s25;
                    ksi[3 * c + 2] = ksi[3 * c + 2] + over * s26;
                    gamma[3 * c + 0] = gamma[3 * c + 0] + over * (nu - nu1) * ev_to_erg * s24;
                }
                if (nu >= nu2) gamma[3 * c + 1] = gamma[3 * c + 1] + over * (nu - nu2) * ev_to_erg * s26;
                if (nu >= nu3) gamma[3 * c + 2] = gamma[3 * c + 2] + over * (nu - nu3) * ev_to_erg * s25;
            }
        }
        prev = nu;
    }
}

/* coll_rates (coll_rates.f:42-150): the six rate coefficients the equilibrium uses.  Fixed-form literals without a `d`
 * exponent are single precision; integer powers are products taken from the left.  recombination_type 1 = case A, 2 = case B. */
static double powi_left(double x, int n) { double r = x; for (int i = 1; i < n; ++i) r = r * x; return r; }

void fo_coll_rates(double T, int recombination_type, double *k /* [6] */)
{
    const double T_eV = T / F(11605.);
    const double L = log(T_eV);
    if (T_eV > F(0.8)) {
        k[0] = exp(F(-32.71396786375) + F(13.53655609057) * L - F(5.739328757388) * powi_left(L, 2) + F(1.563154982022) * powi_left(L, 3) -
                   F(0.2877056004391) * powi_left(L, 4) + F(0.03482559773736999) * powi_left(L, 5) - F(0.00263197617559) * powi_left(L, 6) +
                   F(0.0001119543953861) * powi_left(L, 7) - F(2.039149852002e-6) * powi_left(L, 8));
        k[2] = exp(F(-44.09864886561001) + F(23.91596563469) * L - F(10.75323019821) * powi_left(L, 2) + F(3.058038757198) * powi_left(L, 3) -
                   F(0.5685118909884001) * powi_left(L, 4) + F(0.06795391233790001) * powi_left(L, 5) -
                   F(0.005009056101857001) * powi_left(L, 6) + F(0.0002067236157507) * powi_left(L, 7) -
                   F(3.649161410833e-6) * powi_left(L, 8));
        k[4] = exp(F(-68.71040990212001) + F(43.93347632635) * L - F(18.48066993568) * powi_left(L, 2) + F(4.701626486759002) * powi_left(L, 3) -
                   F(0.7692466334492) * powi_left(L, 4) + F(0.08113042097303) * powi_left(L, 5) - F(0.005324020628287001) * powi_left(L, 6) +
                   F(0.0001975705312221) * powi_left(L, 7) - F(3.165581065665e-6) * powi_left(L, 8));
    } else {
        k[0] = F(1.0e-20); k[2] = F(1.0e-20); k[4] = F(1.0e-20);
    }
    const double kb = 1.3806503e-16, ev = 1.60217646e-12;
    if (recombination_type == 1) {
        if (T_eV > F(0.8))
            k[3] = F(1.54e-9) * (1. + F(0.3) / exp(F(8.099328789667) / T_eV)) / (exp(F(40.49664394833662) / T_eV) * pow(T_eV, F(1.5))) +
                   F(3.92e-13) / pow(T_eV, F(0.6353));
        else k[3] = F(3.92e-13) / pow(T_eV, F(0.6353));
        if (T > F(5500.0))
            k[1] = exp(F(-28.61303380689232) - F(0.7241125657826851) * L - F(0.02026044731984691) * powi_left(L, 2) -
                       F(0.002380861877349834) * powi_left(L, 3) - F(0.0003212605213188796) * powi_left(L, 4) -
                       F(0.00001421502914054107) * powi_left(L, 5) + F(4.989108920299513e-6) * powi_left(L, 6) +
                       F(5.755614137575758e-7) * powi_left(L, 7) - F(1.856767039775261e-8) * powi_left(L, 8) -
                       F(3.071135243196595e-9) * powi_left(L, 9));
        else k[1] = k[3];
        k[5] = F(3.36e-10) / sqrt(T) / pow(T / F(1.e3), F(0.2)) / (1. + pow(T / F(1.e6), F(0.7)));
    } else {
        double tmp = (double)(2.f * 24.587f) * ev / (kb * T);
        k[3] = F(1.26e-14) * (sqrt(tmp) * sqrt(sqrt(tmp))); /* tmp**0.750 as the reference's compiler forms it */
        tmp = (double)(2.f * 13.598f) * ev / (kb * T);
        k[1] = F(2.753e-14) * pow(tmp, F(1.500)) / pow(1. + pow(tmp / F(2.740), F(0.407)), F(2.242));
        tmp = (double)(2.f * 54.418f) * ev / (kb * T);
        k[5] = (double)(2.f * 2.753e-14f) * pow(tmp, F(1.500)) / pow(1. + pow(tmp / F(2.740), F(0.407)), F(2.242));
    }
}

/* the table loop of calc_rates.f:324-337 with the bounds of equiSources.f90:174-176: k[6][nratec] */
void fo_rate_coefficient_tables(int nratec, double temstart, double temend, int recombination_type, double *k, double *logtem0,
                                double *logtem9, double *dlogtem)
{
    *logtem0 = log(temstart);
    *logtem9 = log(temend);
    *dlogtem = (log(temend) - log(temstart)) / (double)(float)(nratec - 1);
    for (int i = 1; i <= nratec; ++i) {
        const double ttt = exp(log(temstart) + (double)(float)(i - 1) * *dlogtem);
        double six[6];
        fo_coll_rates(ttt, recombination_type, six);
        for (int r = 0; r < 6; ++r) k[(size_t)r * nratec + (i - 1)] = six[r];
    }
}
