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
