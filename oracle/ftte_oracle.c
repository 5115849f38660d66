/* ftte_oracle.c -- TEST INFRASTRUCTURE ONLY (see ftte_oracle.h).
 *
 * Plain-C restatement of the reference diffuse sweep.  Each function cites the
 * reference lines it follows.  Written for clarity, not speed; FO_ARITH_REFERENCE
 * keeps the reference's operation order and its float32-widened literals.
 */
#include "ftte_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* the device arithmetic, for FO_ARITH_DEVICE only */
#include "../radiativetransfer_amd/csrc/ftte_math.h"
static const ftte_consts fo_device_consts = FTTE_CONSTS_INIT;

/* ------------------------------------------------------------------ constants
 * definitionsModule.f90:8-10: `pi = 3.141592654` is a default-real literal, so
 * the double constant holds the float32-rounded value 3.1415927410125732. */
double fo_pi(void) { return (double)3.141592654f; }
double fo_half_pi(void) { return 0.5 * fo_pi(); }
double fo_two_pi(void) { return 2.0 * fo_pi(); }

/* ------------------------------------------------------------- rotateIndices
 * rotateIndicesModule.f90:14-111.  The table there is 12 base cases
 * (permutation of (i,j,k) + reflections of the 2nd/3rd component) and the same
 * 12 again with the first component reflected. */
int fo_rotate_indices(int i, int j, int k, int nx, int ny, int nz, int izone, int *ic, int *jc, int *kc)
{
    if (izone < 1 || izone > 24) return -1;
    const int base = (izone - 1) % 12;
    int a, b, c;
    switch (base % 3) { /* cyclic permutation: 0 (i,j,k)  1 (j,k,i)  2 (k,i,j) */
    case 0: a = i; b = j; c = k; break;
    case 1: a = j; b = k; c = i; break;
    default: a = k; b = i; c = j; break;
    }
    switch (base / 3) {
    case 0: break;                                   /* izone 1-3   */
    case 1: {                                        /* izone 4-6: (a, c', b) with third reflected */
        /* 4: (i,k,nz+1-j)  5: (j,i,nz+1-k)  6: (k,j,nz+1-i) */
        int p = a, q, r;
        if (base % 3 == 0) { q = k; r = nz + 1 - j; }
        else if (base % 3 == 1) { q = i; r = nz + 1 - k; }
        else { q = j; r = nz + 1 - i; }
        a = p; b = q; c = r;
        break;
    }
    case 2:                                          /* izone 7-9: second and third reflected */
        b = ny + 1 - b; c = nz + 1 - c; break;
    default: {                                       /* izone 10-12 */
        /* 10: (i,ny+1-k,j)  11: (j,ny+1-i,k)  12: (k,ny+1-j,i) */
        int p = a, q, r;
        if (base % 3 == 0) { q = ny + 1 - k; r = j; }
        else if (base % 3 == 1) { q = ny + 1 - i; r = k; }
        else { q = ny + 1 - j; r = i; }
        a = p; b = q; c = r;
        break;
    }
    }
    if (izone > 12) a = nx + 1 - a;
    *ic = a; *jc = b; *kc = c;
    return 0;
}

/* --------------------------------------------------------------- directions */
static double fo_arcsin(double x) /* equiSources.f90:2277-2295 */
{
    if (x > 1.0) return fo_half_pi();
    if (x < -1.0) return -fo_half_pi();
    return asin(x);
}

static double fo_get_angle(double cosphi, double sinphi) /* equiSources.f90:2337-2361 */
{
    double phi = fo_arcsin(sinphi);
    if (cosphi > 0.0) return (sinphi > 0.0) ? phi : fo_two_pi() + phi;
    return fo_pi() - phi;
}

static void fo_rotate_angles(double *phi, double *theta) /* equiSources.f90:2297-2335 */
{
    double phi0 = *phi, theta0 = *theta, rot, cosphi, sinphi, th;
    rot = (double)0.111f; /* about x */
    th = fo_arcsin(cos(theta0) * sin(phi0) * sin(rot) + sin(theta0) * cos(rot));
    cosphi = cos(theta0) * cos(phi0) / cos(th);
    sinphi = (cos(theta0) * sin(phi0) * cos(rot) - sin(theta0) * sin(rot)) / cos(th);
    phi0 = fo_get_angle(cosphi, sinphi);
    theta0 = th;
    rot = (double)0.222f; /* about y */
    th = fo_arcsin(cos(theta0) * cos(phi0) * sin(rot) + sin(theta0) * cos(rot));
    cosphi = (cos(theta0) * cos(phi0) * cos(rot) - sin(theta0) * sin(rot)) / cos(th);
    sinphi = cos(theta0) * sin(phi0) / cos(th);
    *phi = fo_get_angle(cosphi, sinphi);
    *theta = th;
}

/* bit de-interleave of the NESTED index inside a face, equiSources.f90:2233-2275 */
static void fo_pix2xy(int64_t ipf, int *ix, int *iy)
{
    int x = 0, y = 0, bit = 0;
    while (ipf) {
        x |= (int)(ipf & 1) << bit; ipf >>= 1;
        y |= (int)(ipf & 1) << bit; ipf >>= 1;
        ++bit;
    }
    *ix = x; *iy = y;
}

int fo_pix2ang_nest(int nside, int64_t ipix, double *phi_out, double *theta_out)
{
    static const int jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
    static const int jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};
    if (nside < 1 || nside > 8192 * 4) return -1;
    const int64_t npface = (int64_t)nside * nside;
    if (ipix < 0 || ipix > 12 * npface - 1) return -2;

    const double fn = (double)(float)nside;
    const double fact1 = 1.0 / (3.0 * fn * fn);
    const double fact2 = 2.0 / (3.0 * fn);
    const int nl4 = 4 * nside;
    const int face = (int)(ipix / npface);
    int ix, iy;
    fo_pix2xy(ipix % npface, &ix, &iy);
    const int jrt = ix + iy, jpt = ix - iy;
    const int jr = jrll[face] * nside - jrt - 1;
    int nr = nside, kshift = (jr - nside) % 2; /* Fortran MOD keeps the sign of the dividend, as C % */
    double z = (double)(float)(2 * nside - jr) * fact2;
    if (jr < nside) {
        nr = jr; z = 1.0 - (double)((float)nr * (float)nr) * fact1; kshift = 0;
    } else if (jr > 3 * nside) {
        nr = nl4 - jr; z = -1.0 + (double)((float)nr * (float)nr) * fact1; kshift = 0;
    }
    double theta = acos(z) - fo_half_pi();
    int jp = (jpll[face] * nr + jpt + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    if (jp < 1) jp += nl4;
    double phi = (double)((float)jp - (float)(kshift + 1) * 0.5f) * fo_half_pi() / (double)(float)nr;
    while (phi > fo_two_pi()) phi -= fo_two_pi();
    while (phi < 0.0) phi += fo_two_pi();
    fo_rotate_angles(&phi, &theta);
    if (phi > 2.0 * fo_pi()) return -3;
    *phi_out = phi; *theta_out = theta;
    return 0;
}

int fo_fold_direction(double phiL, double thetaL, double *phi, double *theta, int *izone)
{
    const double pi = fo_pi();
    int zone = 1;
    double p1, t1;
    if (phiL > 0.0 && phiL < 0.5 * pi) { p1 = phiL; }
    else if (phiL > 0.5 * pi && phiL < pi) { p1 = phiL - 0.5 * pi; zone += 3; }
    else if (phiL > pi && phiL < 1.5 * pi) { p1 = phiL - pi; zone += 6; }
    else if (phiL > 1.5 * pi && phiL < 2.0 * pi) { p1 = phiL - 1.5 * pi; zone += 9; }
    else return -1;
    if (thetaL > 0.0 && thetaL < 0.5 * pi) { t1 = thetaL; }
    else if (thetaL > -0.5 * pi && thetaL < 0.0) { t1 = -thetaL; zone += 12; }
    else return -2;

    const double tz = 1.0 / sin(t1);
    const double tx = 1.0 / (cos(p1) * cos(t1));
    const double ty = 1.0 / (sin(p1) * cos(t1));
    if (tz < fmin(tx, ty)) {
        *theta = t1; *phi = p1;
    } else if (tx < fmin(tz, ty)) {
        *theta = fo_arcsin(cos(t1) * cos(p1));
        *phi = fo_arcsin(sin(t1) / cos(*theta));
        zone += 1;
    } else if (ty < fmin(tz, tx)) {
        *theta = fo_arcsin(cos(t1) * sin(p1));
        *phi = acos(sin(t1) / cos(*theta));
        zone += 2;
    } else return -3;
    *izone = zone;
    return 0;
}

/* ------------------------------------------------------------------ patterns */
int fo_set_pattern(fo_pattern *p, double phi, double theta) /* transportRoutinesModule.f90:7-85 */
{
    const double t_top = 1.0 / sin(theta);
    const double t_x = (1.0 - p->xy_x0) / (cos(phi) * cos(theta));
    const double t_y = (1.0 - p->xy_y0) / (sin(phi) * cos(theta));

    if (t_top < fmin(t_x, t_y)) { /* straight to the top face */
        p->xy_len = t_top;
        p->xz_active = 0; p->yz_active = 0;
        p->xy_top = FO_XY_END; p->xz_top = 0; p->yz_top = 0;
        return 0;
    }
    if (t_x < fmin(t_top, t_y)) { /* leaves through x = 1, continues as the yz ray */
        p->xy_len = t_x;
        p->yz_active = 1;
        p->yz_y0 = (1.0 - p->xy_x0) * tan(phi) + p->xy_y0;
        p->yz_z0 = p->xy_len * sin(theta);
        if (p->yz_y0 > 1.0 || p->yz_z0 > 1.0) return -1;
        const double a_top = (1.0 - p->yz_z0) / sin(theta);
        const double a_y = (1.0 - p->yz_y0) / (sin(phi) * cos(theta));
        if (a_top < a_y) {
            p->yz_len = a_top;
            p->xz_active = 0;
            p->xy_top = FO_YZ_END; p->xz_top = 0; p->yz_top = FO_XY_END;
        } else {
            p->yz_len = a_y;
            p->xz_active = 1;
            p->xz_x0 = (1.0 - p->yz_y0) / tan(phi);
            p->xz_z0 = p->yz_z0 + a_y * sin(theta);
            p->xz_len = (1.0 - p->xz_z0) / sin(theta);
            p->xy_top = FO_XZ_END; p->xz_top = FO_YZ_END; p->yz_top = FO_XY_END;
        }
        return 0;
    }
    /* leaves through y = 1, continues as the xz ray */
    p->xy_len = t_y;
    p->xz_active = 1;
    p->xz_x0 = (1.0 - p->xy_y0) / tan(phi) + p->xy_x0;
    p->xz_z0 = t_y * sin(theta);
    if (p->xz_x0 > 1.0 || p->xz_z0 > 1.0) return -1;
    const double b_top = (1.0 - p->xz_z0) / sin(theta);
    const double b_x = (1.0 - p->xz_x0) / (cos(phi) * cos(theta));
    if (b_top < b_x) {
        p->xz_len = b_top;
        p->yz_active = 0;
        p->xy_top = FO_XZ_END; p->xz_top = FO_XY_END; p->yz_top = 0;
    } else {
        p->xz_len = b_x;
        p->yz_active = 1;
        p->yz_y0 = (1.0 - p->xz_x0) * tan(phi);
        p->yz_z0 = p->xz_len * sin(theta) + p->xz_z0;
        p->yz_len = (1.0 - p->yz_z0) / sin(theta);
        p->xy_top = FO_YZ_END; p->xz_top = FO_XY_END; p->yz_top = FO_XZ_END;
    }
    return 0;
}

/* entry point of the layer above `below`: equiSources.f90:1507-1522,
 * transportRoutinesModule.f90:167-182 */
static int fo_entry_above(const fo_pattern *below, double phi, double theta, double *x0, double *y0)
{
    switch (below->xy_top) {
    case FO_XY_END:
        *x0 = below->xy_x0 + cos(phi) / tan(theta);
        *y0 = below->xy_y0 + sin(phi) / tan(theta);
        break;
    case FO_XZ_END:
        *x0 = below->xz_x0 + below->xz_len * cos(theta) * cos(phi);
        *y0 = below->xz_len * cos(theta) * sin(phi);
        break;
    case FO_YZ_END:
        *x0 = below->yz_len * cos(theta) * cos(phi);
        *y0 = below->yz_y0 + below->yz_len * cos(theta) * sin(phi);
        break;
    default: return -1;
    }
    if (*x0 > 1.0 || *y0 > 1.0) return -2; /* equiSources.f90:1523-1527 */
    return 0;
}

int fo_layer_patterns(int n, double phi, double theta, fo_pattern *L)
{
    memset(L, 0, (size_t)n * sizeof *L);
    for (int i = 0; i < n; ++i) {
        if (i == 0) { L[i].xy_x0 = 0.5; L[i].xy_y0 = 0.5; }
        else {
            int rc = fo_entry_above(&L[i - 1], phi, theta, &L[i].xy_x0, &L[i].xy_y0);
            if (rc) return rc;
        }
        int rc = fo_set_pattern(&L[i], phi, theta);
        if (rc) return rc;
    }
    return 0;
}

/* -------------------------------------------------------- segment arithmetic */
/* transportRoutinesModule.f90:651-678 and 1036-1054.  Advances *I through one
 * segment and returns the term added to the cell's running mean. */
/* emitting: 1 an emissivity array was given (the reference's eta, :673-678), 2 a source-function array: the device then takes
 * that path for every segment, whatever the local values.  One or the other (both: FO_ERR_ARG at the entry points). */
static double fo_segment(double *I, double kappa, double eta, double src, int emitting, double dpath, int arith,
                         double *noise, double *Inoise)
{
    const double Iin = *I;
    const double tau = kappa * dpath;
    double mean;
    if (arith == FO_ARITH_EXACT) {
        /* extended precision throughout the segment, one rounding of each result (ftte_oracle.h) */
        const long double t = (long double)kappa * (long double)dpath;
        const long double a = expl(-t), one_minus_a = -expm1l(-t);
        const long double g = t > 0.0L ? one_minus_a / t : 1.0L;
        if (emitting == 2) {
            *I = (double)((long double)src + ((long double)Iin - (long double)src) * a);
            return (double)((long double)src + ((long double)Iin - (long double)src) * g);
        }
        if (!emitting) {
            *I = (double)((long double)Iin * a);
            return (double)((long double)Iin * g);
        }
        const long double emit = (tau > (double)1.e-10f) ? one_minus_a / (long double)kappa : (long double)dpath;
        const long double Iout = (long double)Iin * a + (long double)eta * emit / (long double)dpath;
        *I = (double)Iout;
        if (Iout < (long double)Iin) {
            /* (Iin - Iout)/log(Iin/Iout) with the logarithm of a ratio near 1 taken through log1p */
            const long double d = (long double)Iin - Iout;
            return (double)(d / -log1pl(-d / (long double)Iin));
        }
        return (double)(0.5L * ((long double)Iin + Iout));
    }
    if (emitting == 2) {
        /* a source function S (the build's own extension, not in the reference): the formal solution with S constant along the
         * piece, I(t) = S + (Iin - S) exp(-t), and its exact path mean S + (Iin - S)(1 - exp(-tau))/tau (ftte_math.h:
         * ftte_segment_source).  "Reference arithmetic" here is the straightforward evaluation with libm's exp. */
        if (arith == FO_ARITH_DEVICE) mean = ftte_segment_source(&fo_device_consts, fo_device_consts.c[9], I, tau, src);
        else {
            const double absorb = exp(-tau);
            const double g = tau > 0.0 ? (1.0 - absorb) / tau : 1.0;
            *I = src + (Iin - src) * absorb;
            mean = src + (Iin - src) * g;
        }
        if (noise) { /* 1 - absorb loses eps/2 absolute to cancellation: (eps/2)/tau relative in g */
            if (tau > 0.0) *noise += fabs(Iin - src) * (0x1p-53 / tau) + 4 * 0x1p-52 * fabs(mean);
            if (Inoise) *Inoise = *Inoise * exp(-tau) + 4 * 0x1p-52 * fabs(*I);
        }
        return mean;
    }
    if (arith == FO_ARITH_DEVICE) {
        mean = emitting ? ftte_segment_emit(&fo_device_consts, I, tau, eta, 0.0) : ftte_segment(&fo_device_consts, I, tau);
    } else {
        const double absorb = exp(-tau);
        /* (float)1.e-10: the threshold is a default-real literal, :658 */
        const double emit = (tau > (double)1.e-10f) ? (1.0 - absorb) / kappa : dpath;
        double Iout = Iin * absorb + eta * emit / dpath;
        *I = Iout;
        mean = (Iout < Iin) ? (Iin - Iout) / log(Iin / Iout) : 0.5 * (Iin + Iout);
    }
    /* rounding noise of the quotient (Iin-Iout)/log(Iin/Iout): the division Iin/Iout perturbs the logarithm by up to
     * eps/2 absolute, i.e. the mean by a relative (eps/2)/|log(Iin/Iout)| (= (eps/2)/tau without emission);
     * see ftte_oracle.h, fo_diffuse_sweep_uniform: `noise` */
    if (noise) {
        const double Iout = *I;
        if (!emitting) { if (tau > 0.0) *noise += Iin * (0x1p-53 / tau); }
        else {
            /* with emission the reference formula also loses 1-exp(-tau) to cancellation (absolute eps/2 in
             * 1-tmpabs, i.e. |eta|*(eps/2)/tau in Iout), and that error travels down the ray: carry a bound on the
             * error of the intensity itself (Inoise, attenuated like the intensity) and charge it to the means */
            const double a = exp(-tau);
            const double in_n = Inoise ? *Inoise : 0.0;
            double out_n = in_n * a + 4 * 0x1p-52 * fabs(Iout);
            if (tau > 0.0) out_n += (fabs(eta) / tau + fabs(src)) * 0x1p-53;
            if (Inoise) *Inoise = out_n;
            if (Iout < Iin && Iout > 0.0) *noise += fmin(mean, mean * (0x1p-53 / log(Iin / Iout)));
            *noise += in_n + out_n;
        }
    }
    return mean;
}

static double fo_cell_mean(double acc, int nseg, double w, int arith)
{
    if (arith == FO_ARITH_DEVICE) return ftte_cell_mean(acc, nseg, w);
    if (arith == FO_ARITH_EXACT) return (double)((long double)acc / (long double)nseg * (long double)w);
    return acc / (double)(float)nseg * w; /* Jmean/float(imean) * weight, :953 */
}

/* which storage axis (0 = i, 1 = j, 2 = k) the march axis of an izone lands on */
static int fo_march_axis(int izone)
{
    int a1, b1, c1, a2, b2, c2;
    fo_rotate_indices(1, 1, 1, 4, 4, 4, izone, &a1, &b1, &c1);
    fo_rotate_indices(2, 1, 1, 4, 4, 4, izone, &a2, &b2, &c2);
    if (a1 != a2) return 0;
    if (b1 != b2) return 1;
    (void)c1; (void)c2;
    return 2;
}

static int fo_seg_slot(int end_code) /* storage slot of a segment type: xy 0, xz 1, yz 2 */
{
    return end_code == FO_XY_END ? 0 : (end_code == FO_XZ_END ? 1 : 2);
}

/* --------------------------------------------------------------- uniform grid */
int fo_diffuse_sweep_uniform(int n, int nnu, const double *kappa, const double *eta, const double *src, double box,
                             int ndir, const double *phiL, const double *thetaL, const double *w, const double *uvb,
                             double *J, int arith, int order, double *noise)
{
    const int emitting = src != NULL ? 2 : eta != NULL ? 1 : 0;
    if (eta != NULL && src != NULL) return -99; /* one form of emission or the other */
    const size_t ncell = (size_t)n * n * n;
    if (noise) memset(noise, 0, (size_t)nnu * ncell * sizeof *noise);
    const size_t plane = (size_t)n * n;
    const int nacc = (order == FO_ORDER_CLASSED) ? 3 : 1;
    double *acc = calloc((size_t)nacc * nnu * ncell, sizeof *acc);
    /* segment outputs of the layer below and of the current layer: [slot][j][k][nu] */
    double *below = malloc(3 * plane * nnu * sizeof *below);
    double *here = malloc(3 * plane * nnu * sizeof *here);
    /* error bounds of those intensities, carried only for the emitting noise estimate */
    double *below_n = calloc(3 * plane * nnu, sizeof *below_n);
    double *here_n = calloc(3 * plane * nnu, sizeof *here_n);
    fo_pattern *L = malloc((size_t)n * sizeof *L);
    int status = 0;
    const double delta = box / (double)n; /* equiSources.f90:1570 */

    for (int d = 0; d < ndir && !status; ++d) {
        double phi, theta;
        int izone;
        int rc = fo_fold_direction(phiL[d], thetaL[d], &phi, &theta, &izone);
        if (rc) { status = rc - 10 * d; break; }
        rc = fo_layer_patterns(n, phi, theta, L);
        if (rc) { status = -4; break; }
        double *Jd = acc + (size_t)((nacc == 3) ? fo_march_axis(izone) : 0) * nnu * ncell;

        for (int i = 1; i <= n; ++i) {
            const fo_pattern *P = &L[i - 1];
            const fo_pattern *Pb = (i > 1) ? &L[i - 2] : NULL;
            for (int j = 1; j <= n; ++j) {
                for (int k = 1; k <= n; ++k) {
                    int ic, jc, kc;
                    fo_rotate_indices(i, j, k, n, n, n, izone, &ic, &jc, &kc);
                    const size_t cell = ((size_t)(ic - 1) * n + (jc - 1)) * n + (kc - 1);
                    const size_t at = ((size_t)(j - 1) * n + (k - 1)) * nnu;
                    for (int g = 0; g < nnu; ++g) {
                        const double kap = kappa[(size_t)g * ncell + cell];
                        const double em = eta ? eta[(size_t)g * ncell + cell] : 0.0;
                        const double sf = src ? src[(size_t)g * ncell + cell] : 0.0;
                        double sum = 0.0, I, nz = 0.0;
                        double *nzp = noise ? &nz : NULL;
                        int nseg = 0;
                        /* xy segment <- cell (i-1,j,k), transportRoutinesModule.f90:594-611 */
                        double In;
                        I = Pb ? below[fo_seg_slot(Pb->xy_top) * plane * nnu + at + g] : uvb[g];
                        In = Pb ? below_n[fo_seg_slot(Pb->xy_top) * plane * nnu + at + g] : 0.0;
                        sum += fo_segment(&I, kap, em, sf, emitting, delta * P->xy_len, arith, nzp, &In);
                        here[0 * plane * nnu + at + g] = I;
                        here_n[0 * plane * nnu + at + g] = In;
                        ++nseg;
                        if (P->xz_active) { /* <- cell (i,j-1,k), :708-772 */
                            I = (j > 1) ? here[fo_seg_slot(P->xz_top) * plane * nnu + at - (size_t)n * nnu + g] : uvb[g];
                            In = (j > 1) ? here_n[fo_seg_slot(P->xz_top) * plane * nnu + at - (size_t)n * nnu + g] : 0.0;
                            sum += fo_segment(&I, kap, em, sf, emitting, delta * P->xz_len, arith, nzp, &In);
                            here[1 * plane * nnu + at + g] = I;
                            here_n[1 * plane * nnu + at + g] = In;
                            ++nseg;
                        }
                        if (P->yz_active) { /* <- cell (i,j,k-1), :830-894 */
                            I = (k > 1) ? here[fo_seg_slot(P->yz_top) * plane * nnu + at - (size_t)nnu + g] : uvb[g];
                            In = (k > 1) ? here_n[fo_seg_slot(P->yz_top) * plane * nnu + at - (size_t)nnu + g] : 0.0;
                            sum += fo_segment(&I, kap, em, sf, emitting, delta * P->yz_len, arith, nzp, &In);
                            here[2 * plane * nnu + at + g] = I;
                            here_n[2 * plane * nnu + at + g] = In;
                            ++nseg;
                        }
                        Jd[(size_t)g * ncell + cell] += fo_cell_mean(sum, nseg, w[d], arith);
                        if (noise) noise[(size_t)g * ncell + cell] += nz / nseg * w[d];
                    }
                }
            }
            double *t = below; below = here; here = t;
            t = below_n; below_n = here_n; here_n = t;
        }
    }

    if (!status) {
        const size_t tot = (size_t)nnu * ncell;
        if (nacc == 1) memcpy(J, acc, tot * sizeof *J);
        else for (size_t x = 0; x < tot; ++x) J[x] = (acc[x] + acc[tot + x]) + acc[2 * tot + x];
    }
    free(acc); free(below); free(here); free(below_n); free(here_n); free(L);
    return status;
}

/* --------------------------------------------------------------------- AMR tree
 * A node is a cell of the fully threaded tree (definitionsModule.f90:163-180);
 * children are indexed by their *storage* position (a,b,c) in {1,2}^3 as
 * 4(a-1)+2(b-1)+(c-1), which is also the depth-first leaf order of the cell
 * array (equiSources.f90:4044-4079). */
typedef struct {
    int refined, level, parent;
    int child[8];
    int64_t leaf;  /* cell-array index if !refined */
    int pat;       /* index into the pattern pool of the current direction */
    int nb[3];     /* upstream node for the xy, xz, yz segment, -1 = domain boundary */
} fo_node;

typedef struct {
    fo_pattern p;
    int sub[2]; /* patterns of the two sub-layers of a refined cell with this pattern, -1 = not built */
} fo_pnode;

typedef struct {
    fo_node *node; int nnode, capnode;
    fo_pnode *pn; int npn, cappn;
    const int32_t *level; int64_t ncell, cursor;
    int n, izone, nnu, arith;
    int cs[2][2][2]; /* storage child index of sweep child (i,j,k) */
    double phi, theta;
    const double *kappa, *uvb, *eta, *src;
    double *Iout;    /* [nnode][3][nnu] */
    double *Jd, *noise;
    double w;
    int err;
} fo_tree;

static int fo_new_node(fo_tree *T)
{
    if (T->nnode == T->capnode) {
        T->capnode = T->capnode ? 2 * T->capnode : 1024;
        T->node = realloc(T->node, (size_t)T->capnode * sizeof *T->node);
    }
    return T->nnode++;
}

static int fo_new_pnode(fo_tree *T)
{
    if (T->npn == T->cappn) {
        T->cappn = T->cappn ? 2 * T->cappn : 1024;
        T->pn = realloc(T->pn, (size_t)T->cappn * sizeof *T->pn);
    }
    T->pn[T->npn].sub[0] = T->pn[T->npn].sub[1] = -1;
    return T->npn++;
}

/* readCellArray.f90:154-187 */
static void fo_grow(fo_tree *T, int me, int level)
{
    if (T->err) return;
    if (T->cursor >= T->ncell) { T->err = -20; return; }
    const int lv = T->level[T->cursor];
    T->node[me].level = level;
    if (lv == level) {
        T->node[me].refined = 0;
        T->node[me].leaf = T->cursor++;
    } else if (lv > level) {
        T->node[me].refined = 1;
        for (int c = 0; c < 8; ++c) {
            int ch = fo_new_node(T);
            T->node[me].child[c] = ch;
            T->node[ch].parent = me;
            fo_grow(T, ch, level + 1);
        }
    } else T->err = -21;
}

/* transportRoutinesModule.f90:121-218: the two sub-layer patterns under a refined cell */
static int fo_sub_pattern(fo_tree *T, int parent, int sub)
{
    if (T->pn[parent].sub[0] < 0) {
        int lo = fo_new_pnode(T), hi = fo_new_pnode(T);
        const fo_pattern *pp = &T->pn[parent].p;
        fo_pattern *pl = &T->pn[lo].p, *ph = &T->pn[hi].p;
        memset(pl, 0, sizeof *pl); memset(ph, 0, sizeof *ph);
        pl->xy_x0 = (pp->xy_x0 < 0.5) ? 2.0 * pp->xy_x0 : 2.0 * pp->xy_x0 - 1.0; /* :151-155 */
        pl->xy_y0 = (pp->xy_y0 < 0.5) ? 2.0 * pp->xy_y0 : 2.0 * pp->xy_y0 - 1.0; /* :156-160 */
        if (fo_set_pattern(pl, T->phi, T->theta)) T->err = -22;
        if (fo_entry_above(pl, T->phi, T->theta, &ph->xy_x0, &ph->xy_y0)) T->err = -23; /* :167-186 */
        if (fo_set_pattern(ph, T->phi, T->theta)) T->err = -22;
        T->pn[parent].sub[0] = lo; T->pn[parent].sub[1] = hi;
    }
    return T->pn[parent].sub[sub];
}

static void fo_attach(fo_tree *T, int me, int pat) /* :200-211 */
{
    T->node[me].pat = pat;
    if (!T->node[me].refined) return;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k)
                fo_attach(T, T->node[me].child[T->cs[i][j][k]], fo_sub_pattern(T, pat, i));
}

/* get??Neighbour, transportRoutinesModule.f90:455-558.  face: 0 xy (coords x,y),
 * 1 xz (x,z), 2 yz (y,z).  Descends into `c` choosing the child that holds the
 * entry point (a,b); comparison `.le. 0.5` as in the reference. */
static int fo_descend(const fo_tree *T, int c, int face, double a, double b)
{
    while (T->node[c].refined) {
        const int ha = (a <= 0.5) ? 0 : 1, hb = (b <= 0.5) ? 0 : 1;
        int i, j, k;
        /* sweep frame: i <-> z, j <-> y, k <-> x */
        if (face == 0) { i = 1; k = ha; j = hb; }      /* upper sub-layer of the cell below; x->k, y->j */
        else if (face == 1) { j = 1; k = ha; i = hb; } /* far-y half; x->k, z->i */
        else { k = 1; j = ha; i = hb; }                /* far-x half; y->j, z->i */
        c = T->node[c].child[T->cs[i][j][k]];
        a = 2.0 * a - ha; b = 2.0 * b - hb;
    }
    return c;
}

/* findNeighbours, transportRoutinesModule.f90:264-418.  seq holds the sweep
 * indices (i,j,k) of the cell at every level from 0 (base, 1..n) to its own. */
static void fo_link(fo_tree *T, int leaf, const int *seq, int level)
{
    const fo_pattern *P = &T->pn[T->node[leaf].pat].p;
    for (int face = 0; face < 3; ++face) {
        T->node[leaf].nb[face] = -1;
        double a, b;
        if (face == 0) { a = P->xy_x0; b = P->xy_y0; }
        else if (face == 1) { if (!P->xz_active) continue; a = P->xz_x0; b = P->xz_z0; }
        else { if (!P->yz_active) continue; a = P->yz_y0; b = P->yz_z0; }
        int c = leaf;
        for (int lv = level; lv >= 0; --lv) {
            const int i = seq[3 * lv], j = seq[3 * lv + 1], k = seq[3 * lv + 2];
            c = T->node[c].parent;
            /* the index that must exceed 1 for a sibling to exist on the upstream side */
            const int along = (face == 0) ? i : (face == 1) ? j : k;
            if (along > 1) {
                int sib;
                const int si = i - (face == 0), sj = j - (face == 1), sk = k - (face == 2);
                if (lv == 0) {
                    int ic, jc, kc;
                    fo_rotate_indices(si, sj, sk, T->n, T->n, T->n, T->izone, &ic, &jc, &kc);
                    sib = (int)(((size_t)(ic - 1) * T->n + (jc - 1)) * T->n + (kc - 1)); /* base cells are nodes 0..n^3-1 */
                } else sib = T->node[c].child[T->cs[si - 1][sj - 1][sk - 1]];
                T->node[leaf].nb[face] = fo_descend(T, sib, face, a, b);
                break;
            }
            /* no sibling at this level: express the entry point in the parent's units (:306-317 etc.) */
            if (face == 0) { b = b / 2.0 + (j == 1 ? 0.0 : 0.5); a = a / 2.0 + (k == 1 ? 0.0 : 0.5); }
            else if (face == 1) { b = b / 2.0 + (i == 1 ? 0.0 : 0.5); a = a / 2.0 + (k == 1 ? 0.0 : 0.5); }
            else { b = b / 2.0 + (i == 1 ? 0.0 : 0.5); a = a / 2.0 + (j == 1 ? 0.0 : 0.5); }
        }
    }
}

static void fo_link_all(fo_tree *T, int me, int *seq, int level) /* :421-453 */
{
    if (!T->node[me].refined) { fo_link(T, me, seq, level); return; }
    for (int i = 1; i <= 2; ++i)
        for (int j = 1; j <= 2; ++j)
            for (int k = 1; k <= 2; ++k) {
                seq[3 * (level + 1)] = i; seq[3 * (level + 1) + 1] = j; seq[3 * (level + 1) + 2] = k;
                fo_link_all(T, T->node[me].child[T->cs[i - 1][j - 1][k - 1]], seq, level + 1);
            }
}

/* incoming intensity of one segment, transportRoutinesModule.f90:594-649 (and the
 * xz / yz copies of that block) */
static double fo_incoming(fo_tree *T, int me, int face, int g)
{
    const int up = T->node[me].nb[face];
    if (up < 0) return T->uvb[g];
    const fo_pattern *U = &T->pn[T->node[up].pat].p;
    const double *Iu = T->Iout + ((size_t)up * 3) * T->nnu;
    const int top = (face == 0) ? U->xy_top : (face == 1) ? U->xz_top : U->yz_top;
    if (top != 0) return Iu[(size_t)fo_seg_slot(top) * T->nnu + g];
    /* the upstream cell has no segment ending on that face: legal only for a coarser neighbour */
    if (T->node[me].level <= T->node[up].level) { T->err = -24; return 0.0; }
    if (U->xz_active) return 0.5 * (Iu[(size_t)1 * T->nnu + g] + Iu[g]);
    if (U->yz_active) return 0.5 * (Iu[(size_t)2 * T->nnu + g] + Iu[g]);
    return Iu[g];
}

static void fo_transport(fo_tree *T, int me, double cell_size) /* :560-963 */
{
    if (T->node[me].refined) {
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k)
                    fo_transport(T, T->node[me].child[T->cs[i][j][k]], cell_size / 2.0);
        return;
    }
    const fo_pattern *P = &T->pn[T->node[me].pat].p;
    const int64_t cell = T->node[me].leaf;
    double *Io = T->Iout + ((size_t)me * 3) * T->nnu;
    for (int g = 0; g < T->nnu; ++g) {
        const double kap = T->kappa[(size_t)g * T->ncell + cell];
        const double em = T->eta ? T->eta[(size_t)g * T->ncell + cell] : 0.0;
        const double sf = T->src ? T->src[(size_t)g * T->ncell + cell] : 0.0;
        const int emitting = T->src != NULL ? 2 : T->eta != NULL ? 1 : 0;
        double sum = 0.0, I, nz = 0.0;
        double *nzp = T->noise ? &nz : NULL;
        int nseg = 0;
        I = fo_incoming(T, me, 0, g);
        sum += fo_segment(&I, kap, em, sf, emitting, cell_size * P->xy_len, T->arith, nzp, NULL);
        Io[g] = I; ++nseg;
        if (P->xz_active) {
            I = fo_incoming(T, me, 1, g);
            sum += fo_segment(&I, kap, em, sf, emitting, cell_size * P->xz_len, T->arith, nzp, NULL);
            Io[(size_t)1 * T->nnu + g] = I; ++nseg;
        }
        if (P->yz_active) {
            I = fo_incoming(T, me, 2, g);
            sum += fo_segment(&I, kap, em, sf, emitting, cell_size * P->yz_len, T->arith, nzp, NULL);
            Io[(size_t)2 * T->nnu + g] = I; ++nseg;
        }
        T->Jd[(size_t)g * T->ncell + cell] += fo_cell_mean(sum, nseg, T->w, T->arith);
        if (T->noise) T->noise[(size_t)g * T->ncell + cell] += nz / nseg * T->w;
    }
}

int fo_diffuse_sweep_tree(int n, int64_t ncell, const int32_t *level, int nnu, const double *kappa, const double *eta,
                          const double *src, double box, int ndir, const double *phiL, const double *thetaL,
                          const double *w, const double *uvb, double *J, int arith, int order, double *noise)
{
    fo_tree T;
    memset(&T, 0, sizeof T);
    if (eta != NULL && src != NULL) return -99; /* one form of emission or the other */
    T.eta = eta; T.src = src;
    T.noise = noise;
    if (noise) memset(noise, 0, (size_t)nnu * ncell * sizeof *noise);
    T.level = level; T.ncell = ncell; T.n = n; T.nnu = nnu; T.arith = arith; T.kappa = kappa; T.uvb = uvb;

    const int nbase = n * n * n;
    for (int b = 0; b < nbase; ++b) fo_new_node(&T); /* base cells first, in storage order */
    for (int b = 0; b < nbase; ++b) { T.node[b].parent = -1; fo_grow(&T, b, 0); }
    if (!T.err && T.cursor != ncell) T.err = -25;
    if (T.err) { free(T.node); return T.err; }

    const int nacc = (order == FO_ORDER_CLASSED) ? 3 : 1;
    double *acc = calloc((size_t)nacc * nnu * ncell, sizeof *acc);
    T.Iout = calloc((size_t)T.nnode * 3 * nnu, sizeof *T.Iout);
    fo_pattern *L = malloc((size_t)n * sizeof *L);
    int seq[3 * 40];

    for (int d = 0; d < ndir && !T.err; ++d) {
        int rc = fo_fold_direction(phiL[d], thetaL[d], &T.phi, &T.theta, &T.izone);
        if (rc) { T.err = rc - 10 * d; break; }
        if (fo_layer_patterns(n, T.phi, T.theta, L)) { T.err = -4; break; }
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    int a, b, c;
                    fo_rotate_indices(i + 1, j + 1, k + 1, 2, 2, 2, T.izone, &a, &b, &c); /* equiSources.f90:1485-1491 */
                    T.cs[i][j][k] = 4 * (a - 1) + 2 * (b - 1) + (c - 1);
                }
        T.npn = 0;
        for (int i = 0; i < n; ++i) { int p = fo_new_pnode(&T); T.pn[p].p = L[i]; }
        T.Jd = acc + (size_t)((nacc == 3) ? fo_march_axis(T.izone) : 0) * nnu * ncell;
        T.w = w[d];

        /* patterns, then links, then transport: three passes over the base grid in
         * sweep order (equiSources.f90:1495-1553, 1557-1566, 1572-1796) */
        for (int pass = 0; pass < 3 && !T.err; ++pass)
            for (int i = 1; i <= n; ++i)
                for (int j = 1; j <= n; ++j)
                    for (int k = 1; k <= n; ++k) {
                        int ic, jc, kc;
                        fo_rotate_indices(i, j, k, n, n, n, T.izone, &ic, &jc, &kc);
                        const int b = (int)(((size_t)(ic - 1) * n + (jc - 1)) * n + (kc - 1));
                        if (pass == 0) fo_attach(&T, b, i - 1);
                        else if (pass == 1) { seq[0] = i; seq[1] = j; seq[2] = k; fo_link_all(&T, b, seq, 0); }
                        else fo_transport(&T, b, box / (double)n);
                    }
    }

    if (!T.err) {
        const size_t tot = (size_t)nnu * ncell;
        if (nacc == 1) memcpy(J, acc, tot * sizeof *J);
        else for (size_t x = 0; x < tot; ++x) J[x] = (acc[x] + acc[tot + x]) + acc[2 * tot + x];
    }
    free(acc); free(T.Iout); free(L); free(T.node); free(T.pn);
    return T.err;
}

/* ---------------------------------------------------------------- opacities */
void fo_compute_opacities(int64_t ncell, int nnu, const double *HI, const double *HeI, const double *HeII,
                          const double *beta, double *kappa)
{
    /* equiSources.f90:4977-4980: kappa_g = HI*b24_g + HeI*b26_g + HeII*b25_g, left to right */
    for (int g = 0; g < nnu; ++g)
        for (int64_t c = 0; c < ncell; ++c)
            kappa[(size_t)g * ncell + c] = HI[c] * beta[0 * nnu + g] + HeI[c] * beta[1 * nnu + g] + HeII[c] * beta[2 * nnu + g];
}

/* ----------------------------------------------------- device arithmetic, exposed for tests */
void fo_device_attenuation(int64_t count, const double *tau, double *e, double *g)
{
    for (int64_t i = 0; i < count; ++i) ftte_attenuation(&fo_device_consts, tau[i], &e[i], &g[i]);
}

void fo_device_log(int64_t count, const double *x, double *out)
{
    for (int64_t i = 0; i < count; ++i) out[i] = ftte_log1p(&fo_device_consts, x[i] - 1.0);
}

/* ftte_segment_source element-wise: I[i] is Iin on entry and Iout on return, mean[i] the exact path mean */
void fo_device_segment_source(int64_t count, double *I, const double *tau, const double *src, double *mean)
{
    for (int64_t i = 0; i < count; ++i) mean[i] = ftte_segment_source(&fo_device_consts, fo_device_consts.c[9], &I[i], tau[i], src[i]);
}

/* ftte_segment_emit element-wise: I[i] is Iin on entry and Iout on return, mean[i] the cell's share */
void fo_device_segment_emit(int64_t count, double *I, const double *tau, const double *eta, const double *src, double *mean)
{
    for (int64_t i = 0; i < count; ++i) mean[i] = ftte_segment_emit(&fo_device_consts, &I[i], tau[i], eta[i], src[i]);
}

void fo_device_cell_mean(int64_t count, const double *acc, int nseg, double w, double *out)
{
    for (int64_t i = 0; i < count; ++i) out[i] = ftte_cell_mean(acc[i], nseg, w);
}
