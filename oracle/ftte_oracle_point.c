/* ftte_oracle_point.c -- TEST INFRASTRUCTURE ONLY (see ftte_oracle.h).
 *
 * Plain-C restatement of the reference's point-source path:
 *   P3  stellarBetaTable (stellarBetaTable.f90), dustCrossSection (dustModule.f90:30-73),
 *       stellarPopulation (stellarPopulationModule.f90:7-50)
 *   P2  getRatesHydrogenHelium (equiSources.f90:4157-4311)
 *   P1  startNewLongRay / drawSegment / find??Neighbour / zoom??Neighbour / absoluteCoordinates /
 *       localizeSplitContinuationCell (equiSources.f90:2412-2595, 2647-2960, 3011-3385), rmax (:304-309), and the
 *       per-source loop (:1256-1329) without its escape-fraction printout.
 * Operation order and float32-widened literals follow the reference so that the vectors produced by the
 * reference's own compiled code (tests/golden/point*.npz) are reproduced.
 */
#include "ftte_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define F(x) ((double)(x##f)) /* a default-real literal of the reference, widened */

static const double kHydrogen = (double)13.598f, kHeI = (double)24.587f, kHeII = (double)54.418f;
static double c_light(void) { return (double)2.99792458e10f; }
static double ev_to_erg(void) { return 1.60217646e-12; }
static double ev_to_hz(void) { return 1.60217646e-12 / (double)6.6260693e-27f; }

/* dustModule.f90:30-73, SMC branch (the only one the reference calls, stellarBetaTable.f90:36) */
double fo_dust_cross_section(double lambda_um, const double *a_smc /* [7][5] row-major */)
{
    double sigma = 0.0;
    for (int i = 0; i < 7; ++i) {
        const double *a = a_smc + 5 * i;
        const double x = lambda_um / a[0];
        sigma = sigma + a[1] / (pow(x, a[3]) + pow(x, -a[4]) + a[2]);
    }
    return F(1.1) * sigma * (double)0.9210340372f;
}

/* stellarPopulationModule.f90:7-50.  spec: [5][37][1221] (metal, spectrum, wavelength), wavelength in cm ascending.
 * iSpectrum, iMetal 1-based as in the reference. */
static double fo_stellar_population(const double *spec, const double *wavelength, int iSpectrum, double cS, int iMetal,
                                    double cM, double freq)
{
#define SL(m, s, w) spec[(((size_t)(m) - 1) * 37 + ((s) - 1)) * 1221 + ((w) - 1)]
    const double lam = c_light() / (freq * ev_to_hz());
    int iw = 1;
    while (lam > wavelength[iw]) ++iw; /* wavelength(iw+1), 1-based */
    double cw = (lam - wavelength[iw - 1]) / (wavelength[iw] - wavelength[iw - 1]);
    cw = fmin(fmax(0.0, cw), 1.0);
    const double sp1 = cS * ((1.0 - cw) * SL(iMetal, iSpectrum + 1, iw) + cw * 1.0 * SL(iMetal, iSpectrum + 1, iw + 1)) +
                       (1.0 - cS) * ((1.0 - cw) * SL(iMetal, iSpectrum, iw) + cw * SL(iMetal, iSpectrum, iw + 1));
    const double sp2 = cS * ((1.0 - cw) * SL(iMetal + 1, iSpectrum + 1, iw) + cw * 1.0 * SL(iMetal + 1, iSpectrum + 1, iw + 1)) +
                       (1.0 - cS) * ((1.0 - cw) * SL(iMetal + 1, iSpectrum, iw) + cw * SL(iMetal + 1, iSpectrum, iw + 1));
    double sp = (1.0 - cM) * sp1 + cM * sp2;
    const double nu_hz = freq * ev_to_hz();
    sp = pow(10.0, sp) / F(1.e-8) * c_light() / (nu_hz * nu_hz);
    return sp;
#undef SL
}

static double fo_pow4(double x) { return x * x * x * x; } /* (a)**4 as the reference's compiler expands it: checked bitwise against the golden tables */

/* stellarBetaTable.f90.  tables: [6][11^4], order reactionRate1..3, energyRate1..3, flat index
 * ((idust*11 + i3)*11 + i2)*11 + i1 (the Fortran array order).  output_sigma: [4][300] (24, 25, 26, dust) or NULL. */
void fo_stellar_beta_table(const double *a_smc, const double *wavelength, const double *spec, int iSpectrum, double cS,
                           int iMetal, double cM, double *tables, double *total_integral, double *output_sigma)
{
    enum { NF = 400, ND = 11, NT = 11 * 11 * 11 * 11 };
    const double pi = fo_pi();
    static double nu[NF], s24[NF], s25[NF], s26[NF], sd[NF];
    const double freqdel = F(0.02);
    for (int i = 0; i < NF; ++i) {
        nu[i] = pow(10.0, (double)i * freqdel);
        const double lambda = c_light() / (nu[i] * ev_to_hz()) * F(1.e8);
        sd[i] = fo_dust_cross_section(lambda / F(1.e4), a_smc) * F(1.e-22);
        if (nu[i] > kHydrogen) {
            const double dum = sqrt(nu[i] / kHydrogen - 1);
            s24[i] = F(6.3e-18) * fo_pow4(kHydrogen / nu[i]) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        } else s24[i] = 0.0;
        if (nu[i] > kHeII) {
            const double dum = sqrt(nu[i] / kHeII - 1);
            s25[i] = F(1.58e-18) * fo_pow4(kHeII / nu[i]) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
        } else s25[i] = 0.0;
        if (nu[i] > kHeI)
            s26[i] = F(7.42e-18) * (F(1.66) * pow(nu[i] / kHeI, (double)(-2.05f)) - F(0.66) * pow(nu[i] / kHeI, (double)(-3.05f)));
        else s26[i] = 0.0;
    }
    if (output_sigma) { /* :119-152 */
        const double lower = kHydrogen, upper = 10.0 * kHydrogen;
        for (int ie = 1; ie <= 300; ++ie) {
            const double freq = lower * exp((double)((float)(ie - 1) / (float)299) * (log(upper) - log(lower)));
            const double lambda = c_light() / (freq * ev_to_hz()) * F(1.e8);
            output_sigma[3 * 300 + ie - 1] = fo_dust_cross_section(lambda / F(1.e4), a_smc) * F(1.e-22);
            double v;
            if (freq > kHydrogen) {
                const double dum = sqrt(freq / kHydrogen - 1);
                v = F(6.3e-18) * fo_pow4(kHydrogen / freq) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
            } else v = (freq == kHydrogen) ? F(6.3e-18) : 0.0;
            output_sigma[0 * 300 + ie - 1] = v;
            if (freq > kHeII) {
                const double dum = sqrt(freq / kHeII - 1);
                v = F(1.58e-18) * fo_pow4(kHeII / freq) * exp(4.0 - 4.0 * atan(dum) / dum) / (1 - exp(-2.0 * pi / dum));
            } else v = 0.0;
            output_sigma[1 * 300 + ie - 1] = v;
            if (freq > kHeI)
                v = F(7.42e-18) * (F(1.66) * pow(freq / kHeI, (double)(-2.05f)) - F(0.66) * pow(freq / kHeI, (double)(-3.05f)));
            else v = 0.0;
            output_sigma[2 * 300 + ie - 1] = v;
        }
    }

    memset(tables, 0, sizeof(double) * 6 * NT);
    double total = 0.0;
    const double thr[3] = {kHydrogen, kHeI, kHeII};
    for (int i = 1; i < NF; ++i) { /* do i = 2, nfreq */
        const double freq = nu[i], delta_nu = nu[i] - nu[i - 1];
        const double lum = fo_stellar_population(spec, wavelength, iSpectrum, cS, iMetal, cM, freq);
        const double dtmp = lum / (freq * ev_to_erg()) * delta_nu * ev_to_hz();
        if (freq >= kHydrogen) total = total + dtmp;
        for (int i1 = 0; i1 < ND; ++i1)
            for (int i2 = 0; i2 < ND; ++i2)
                for (int i3 = 0; i3 < ND; ++i3)
                    for (int idd = 0; idd < ND; ++idd) {
                        double t1 = (double)((float)i1 / (float)10) * 10.0;
                        double t2 = (double)((float)i2 / (float)10) * 10.0;
                        double t3 = (double)((float)i3 / (float)10) * 10.0;
                        double td = (double)((float)idd / (float)10) * 10.0;
                        t1 = s24[i] / F(6.3e-18) * t1;
                        t2 = s26[i] / F(7.42e-18) * t2;
                        t3 = s25[i] / F(1.58e-18) * t3;
                        td = sd[i] / F(5.4116737e-22) * td;
                        const size_t at = (((size_t)idd * ND + i3) * ND + i2) * ND + i1;
                        for (int r = 0; r < 3; ++r)
                            if (freq >= thr[r]) {
                                const double a = dtmp * exp(-(t1 + t2 + t3 + td));
                                tables[(size_t)r * NT + at] = tables[(size_t)r * NT + at] + a;
                                tables[(size_t)(3 + r) * NT + at] = tables[(size_t)(3 + r) * NT + at] + (freq - thr[r]) * ev_to_erg() * a;
                            }
                    }
    }
    *total_integral = total;
}

/* getRatesHydrogenHelium, equiSources.f90:4157-4311.  reaction 1..3; dust = dustApproximation (0: noDust). */
void fo_get_rates(const double *tables, int dust, int reaction, double tau1, double tau2, double tau3, double tau_dust,
                  double *number_rate, double *heating_rate)
{
    enum { ND = 11, NT = 11 * 11 * 11 * 11 };
    if (tau1 > 10.0 || tau2 > 10.0 || tau3 > 10.0 || tau_dust > 10.0) { *number_rate = 0.0; *heating_rate = 0.0; return; }
    const int i1 = (int)(tau1 / 10.0 * 10.0), i2 = (int)(tau2 / 10.0 * 10.0), i3 = (int)(tau3 / 10.0 * 10.0);
    const double c1 = tau1 * 10.0 / 10.0 - (double)i1, c2 = tau2 * 10.0 / 10.0 - (double)i2, c3 = tau3 * 10.0 / 10.0 - (double)i3;
    int idd = 0;
    double cd = 0.0;
    if (dust != 0) { idd = (int)(tau_dust / 10.0 * 10.0); cd = tau_dust * 10.0 / 10.0 - (double)idd; }
    for (int which = 0; which < 2; ++which) {
        const double *R = tables + (size_t)((which ? 3 : 0) + reaction - 1) * NT;
#define T(a, b, c, d) log(R[((((size_t)(d)) * ND + (c)) * ND + (b)) * ND + (a)])
        double v[2];
        for (int dd = 0; dd < 2; ++dd) {
            const int q = idd + dd;
            v[dd] = c1 * ((1. - c3) * (1. - c2) * T(i1 + 1, i2, i3, q) + c3 * (1. - c2) * T(i1 + 1, i2, i3 + 1, q) +
                          c2 * (1. - c3) * T(i1 + 1, i2 + 1, i3, q) + c3 * c2 * T(i1 + 1, i2 + 1, i3 + 1, q)) +
                    (1. - c1) * ((1. - c3) * (1. - c2) * T(i1, i2, i3, q) + c3 * (1. - c2) * T(i1, i2, i3 + 1, q) +
                                 c2 * (1. - c3) * T(i1, i2 + 1, i3, q) + c3 * c2 * T(i1, i2 + 1, i3 + 1, q));
        }
#undef T
        const double out = exp((1. - cd) * v[0] + cd * v[1]);
        if (which) *heating_rate = out; else *number_rate = out;
    }
}

/* equiSources.f90:304-309: single-precision expression, then halved */
void fo_rmax(double *rmax /* [30] */)
{
    for (int ir = 1; ir <= 30; ++ir) {
        const float v = sqrtf(3.f) * (sqrtf(0.5f * powf(4.f, (float)(ir - 1)) - 1.f / 12.f) + 0.5f);
        rmax[ir - 1] = (double)v / 2.0;
    }
}

/* ----------------------------------------------------------------------------------------------- the tracer */
typedef struct { int refined, level, parent, child0; int64_t leaf; } pnode;

typedef struct {
    pnode *node; int nnode, cap;
    const int32_t *levels; int64_t ncell, cursor;
    int n, dust, err;
    const double *HI, *HeI, *HeII, *rho, *abun2, *tables;
    double box, rmax[30];
    double *rates; /* [6][ncell]: krate24, 25, 26, crate24, 25, 26 */
    int highest_pixel_level;
    /* escape bookkeeping of the star being traced (module variables of localDefinitions, equiSources.f90:9-12, and of
     * definitions, definitionsModule.f90:290-294): esc = {ndotRemaining[7], ndotBoundary[7], ndotDust, ndotSpectrum[300]} or
     * NULL; out_sigma = outputSigma24, 25, 26, Dust [4][300] or NULL (then ndotSpectrum stays zero) */
    double *esc;
    const double *out_sigma;
    const double *pix; /* optional: (phi, theta) of all pixels of levels 1, 2, ... concatenated, instead of fo_pix2ang_nest */
    int pix_levels;
    /* what the reference passes through module globals */
    int nb_node; double nb_a, nb_b; int nb_seq[40]; int nb_level;
} ptree;

static int pt_new(ptree *T)
{
    if (T->nnode + 8 > T->cap) { T->cap = T->cap ? 2 * T->cap : 4096; T->node = realloc(T->node, (size_t)T->cap * sizeof *T->node); }
    return T->nnode++;
}

static void pt_grow(ptree *T, int me, int level)
{
    if (T->err) return;
    if (T->cursor >= T->ncell) { T->err = -20; return; }
    const int lv = T->levels[T->cursor];
    T->node[me].level = level;
    T->node[me].child0 = -1;
    if (lv == level) { T->node[me].refined = 0; T->node[me].leaf = T->cursor++; }
    else if (lv > level) {
        T->node[me].refined = 1;
        int c0 = -1;
        for (int c = 0; c < 8; ++c) { int ch = pt_new(T); if (c == 0) c0 = ch; T->node[ch].parent = me; }
        T->node[me].child0 = c0;
        for (int c = 0; c < 8; ++c) pt_grow(T, c0 + c, level + 1);
    } else T->err = -21;
}

/* pixel centre of NESTED pixel ipix of level L: from the caller's table when one was given (tests pin the tracer's logic
 * with the reference's own angles: its trigonometry differs from libm in the last bit of one pixel in several hundred) */
static int pt_pixel(const ptree *T, int level, int64_t ipix, double *phi, double *theta)
{
    if (T->pix && level <= T->pix_levels) {
        int64_t off = 0;
        for (int l = 1; l < level; ++l) off += 12 * ((int64_t)1 << (2 * (l - 1)));
        *phi = T->pix[2 * (off + ipix)]; *theta = T->pix[2 * (off + ipix) + 1];
        return 0;
    }
    return fo_pix2ang_nest(1 << (level - 1), ipix, phi, theta);
}

static int pt_child(const ptree *T, int parent, int i, int j, int k) { return T->node[parent].child0 + 4 * (i - 1) + 2 * (j - 1) + (k - 1); }
static int pt_base(const ptree *T, int i, int j, int k) { return ((i - 1) * T->n + (j - 1)) * T->n + (k - 1); }

/* zoom??Neighbour, equiSources.f90:2827-2960.  axis: the axis the ray crosses (0 x / yz face, 1 y / xz face, 2 z / xy
 * face); (a,b) are the two in-face coordinates in axis order (for axis 2: x,y; axis 0: y,z; axis 1: x,z). */
static void pt_zoom(ptree *T, int c, int level, int *seq, double a, double b, int axis, int side)
{
    while (T->node[c].refined) {
        int ia, ib;
        double na, nb;
        if (a < 0.5) { na = 2. * a; ia = 1; } else { na = 2. * a - 1.; ia = 2; }
        if (b < 0.5) { nb = 2. * b; ib = 1; } else { nb = 2. * b - 1.; ib = 2; }
        const int ic = side == 0 ? 2 : 1; /* entering from above (side 0): the far child; from below: the near one */
        int i, j, k;
        if (axis == 2) { i = ia; j = ib; k = ic; }
        else if (axis == 0) { i = ic; j = ia; k = ib; }
        else { i = ia; j = ic; k = ib; }
        ++level;
        seq[3 * level] = i; seq[3 * level + 1] = j; seq[3 * level + 2] = k;
        c = pt_child(T, c, i, j, k);
        a = na; b = nb;
    }
    T->nb_node = c; T->nb_a = a; T->nb_b = b; T->nb_level = level;
    memcpy(T->nb_seq, seq, sizeof(int) * (3 * level + 3));
}

/* find??Neighbour, equiSources.f90:2647-2825.  Returns 1 when the domain boundary is reached. */
static int pt_find(ptree *T, int cell, int level, const int *seq_in, double a, double b, int axis, int side)
{
    int seq[40];
    memcpy(seq, seq_in, sizeof(int) * (3 * level + 3));
    const int pos = axis == 2 ? 2 : (axis == 0 ? 0 : 1); /* which of (i,j,k) changes */
    const int pa = axis == 2 ? 0 : (axis == 0 ? 1 : 0), pb = axis == 2 ? 1 : 2; /* (i,j,k) slots of the in-face coordinates */
    while (level > 0) {
        const int along = seq[3 * level + pos];
        if ((side == 0 && along == 1) || (side == 1 && along == 2)) {
            a = seq[3 * level + pa] == 1 ? 0.5 * a : 0.5 * a + 0.5;
            b = seq[3 * level + pb] == 1 ? 0.5 * b : 0.5 * b + 0.5;
            cell = T->node[cell].parent;
            --level;
        } else {
            seq[3 * level + pos] = side == 0 ? 1 : 2;
            const int sib = pt_child(T, T->node[cell].parent, seq[3 * level], seq[3 * level + 1], seq[3 * level + 2]);
            pt_zoom(T, sib, level, seq, a, b, axis, side);
            return 0;
        }
    }
    if ((side == 0 && seq[pos] == 1) || (side == 1 && seq[pos] == T->n)) return 1;
    seq[pos] += side == 0 ? -1 : 1;
    pt_zoom(T, pt_base(T, seq[0], seq[1], seq[2]), 0, seq, a, b, axis, side);
    return 0;
}

enum { PT_PROCEED = 1, PT_SPLIT = 2, PT_BOUNDARY = 3 };

typedef struct { double phi, theta; int level; } ppixel;

/* drawSegment, equiSources.f90:2412-2595 */
static void pt_draw(ptree *T, int cell, double *pt, const ppixel *px, int level, const int *seq, double *radius, int *strategy,
                    double *length)
{
    const double prox = cos(px->phi) * cos(px->theta), proy = sin(px->phi) * cos(px->theta), proz = sin(px->theta);
    const double t1 = proz > 0. ? (1. - pt[2]) / proz : -pt[2] / proz;
    const double t2 = prox > 0. ? (1. - pt[0]) / prox : -pt[0] / prox;
    const double t3 = proy > 0. ? (1. - pt[1]) / proy : -pt[1] / proy;
    int axis;
    double tmp;
    if (t1 < fmin(t2, t3)) { axis = 2; tmp = t1; }
    else if (t2 < fmin(t1, t3)) { axis = 0; tmp = t2; }
    else { axis = 1; tmp = t3; }
    const double scale = (double)(float)(1 << level);
    const double rm = T->rmax[px->level - 1];
    if (*radius * scale + tmp < rm || px->level == 6) {
        *strategy = PT_PROCEED;
        *length = tmp;
        *radius = *radius + tmp / scale;
        const double ex = pt[0] + tmp * prox, ey = pt[1] + tmp * proy, ez = pt[2] + tmp * proz;
        int side, hit;
        if (axis == 2) { side = proz < 0. ? 0 : 1; hit = pt_find(T, cell, level, seq, ex, ey, 2, side); }
        else if (axis == 0) { side = prox < 0. ? 0 : 1; hit = pt_find(T, cell, level, seq, ey, ez, 0, side); }
        else { side = proy < 0. ? 0 : 1; hit = pt_find(T, cell, level, seq, ex, ez, 1, side); }
        if (hit) { *strategy = PT_BOUNDARY; return; }
        const double face = side == 0 ? 1. : 0.;
        if (axis == 2) { pt[2] = face; pt[0] = T->nb_a; pt[1] = T->nb_b; }
        else if (axis == 0) { pt[0] = face; pt[1] = T->nb_a; pt[2] = T->nb_b; }
        else { pt[1] = face; pt[0] = T->nb_a; pt[2] = T->nb_b; }
        if (pt[0] < 0. || pt[0] > 1. || pt[1] < 0. || pt[1] > 1. || pt[2] < 0. || pt[2] > 1.) T->err = -30; /* checkPoint */
    } else if (*radius * scale >= rm) {
        *strategy = PT_SPLIT;
        *length = 0.;
    } else {
        *strategy = PT_SPLIT;
        tmp = rm - *radius * scale;
        *length = tmp;
        *radius = *radius + tmp / scale;
        pt[0] = pt[0] + tmp * prox; pt[1] = pt[1] + tmp * proy; pt[2] = pt[2] + tmp * proz;
    }
}

/* startNewLongRay, equiSources.f90:3120-3385 */
static void pt_ray(ptree *T, int start_cell, const double *start_pt, const ppixel *px, int64_t iray_start, int level,
                   const int *start_seq, double start_radius, double ndot, double d1, double d2, double d3, double dd)
{
    if (T->err) return;
    int cell = start_cell, seq[40], strategy = PT_PROCEED, lvl = level;
    double pt[3] = {start_pt[0], start_pt[1], start_pt[2]}, radius = start_radius;
    memcpy(seq, start_seq, sizeof(int) * (3 * level + 3));
    const int64_t nc = T->ncell;
    while (strategy == PT_PROCEED && !T->err) {
        double len;
        const double old_radius = radius;
        pt_draw(T, cell, pt, px, lvl, seq, &radius, &strategy, &len);
        const double cell_size = T->box / ((double)((float)(1 << lvl) * (float)T->n));
        const double L = cell_size * len;
        const int64_t c = T->node[cell].leaf;
        const double tau1 = L * T->HI[c] * F(6.3e-18), tau2 = L * T->HeI[c] * F(7.42e-18), tau3 = L * T->HeII[c] * F(1.58e-18);
        double taud = 0.;
        if (T->dust == 1) taud = L * T->HI[c] * F(5.4116737e-22) * T->abun2[c] / F(0.2);
        else if (T->dust == 2) taud = L * F(0.76) * T->rho[c] / F(1.6726231e-24) * F(5.4116737e-22) * T->abun2[c] / F(0.2);
        if (T->esc) { /* :3198-3233: what is left of the ray where it crosses the output radii, and what left through the box faces */
            static const float out_radius[7] = {0.1f, 0.3f, 1.f, 3.f, 10.f, 30.f, 100.f}; /* [kpc], equiSources.f90:10 */
            const double kpc = F(1.e3) * F(3.08568025e18);
            for (int ir = 0; ir < 7; ++ir) {
                const double tmp = (double)out_radius[ir] * kpc;
                const double tmp1 = old_radius * T->box / (double)(float)T->n, tmp2 = radius * T->box / (double)(float)T->n;
                if (tmp >= tmp1 && tmp <= tmp2) {
                    const double ratio = (tmp - tmp1) / (tmp2 - tmp1);
                    T->esc[ir] = T->esc[ir] + ndot * exp(-(ratio * (tau1 + taud) + d1 + dd));
                    if (ir == 6) {
                        const double o1 = ratio * tau1 + d1, o2 = ratio * tau2 + d2, o3 = ratio * tau3 + d3, od = ratio * taud + dd;
                        T->esc[14] = T->esc[14] + ndot * exp(-od);
                        if (T->out_sigma)
                            for (int ie = 0; ie < 300; ++ie) {
                                const double e1 = T->out_sigma[ie] / F(6.30e-18) * o1, e2 = T->out_sigma[600 + ie] / F(7.42e-18) * o2,
                                             e3 = T->out_sigma[300 + ie] / F(1.58e-18) * o3, e4 = T->out_sigma[900 + ie] / F(5.4116737e-22) * od;
                                T->esc[15 + ie] = T->esc[15 + ie] + ndot * exp(-(e1 + e2 + e3 + e4));
                            }
                    }
                }
            }
            if (strategy == PT_BOUNDARY) {
                const double tmp = radius * T->box / ((double)(float)T->n * kpc);
                for (int ir = 0; ir < 7; ++ir)
                    if ((double)out_radius[ir] > tmp) T->esc[7 + ir] = T->esc[7 + ir] + ndot;
            }
        }
        if (fmin(fmin(d1 + tau1, d2 + tau2), fmin(d3 + tau3, dd + taud)) > 100.) strategy = PT_BOUNDARY;
        double a, b, ea, eb;
        fo_get_rates(T->tables, T->dust, 1, d1, d2, d3, dd, &a, &ea);
        fo_get_rates(T->tables, T->dust, 1, d1 + tau1, d2, d3, dd, &b, &eb);
        T->rates[0 * nc + c] = T->rates[0 * nc + c] + ndot * (a - b);
        T->rates[3 * nc + c] = T->rates[3 * nc + c] + ndot * (ea - eb);
        fo_get_rates(T->tables, T->dust, 2, d1, d2, d3, dd, &a, &ea);
        fo_get_rates(T->tables, T->dust, 2, d1, d2 + tau2, d3, dd, &b, &eb);
        T->rates[2 * nc + c] = T->rates[2 * nc + c] + ndot * (a - b); /* krate26 */
        T->rates[5 * nc + c] = T->rates[5 * nc + c] + ndot * (ea - eb);
        fo_get_rates(T->tables, T->dust, 3, d1, d2, d3, dd, &a, &ea);
        fo_get_rates(T->tables, T->dust, 3, d1, d2, d3 + tau3, dd, &b, &eb);
        T->rates[1 * nc + c] = T->rates[1 * nc + c] + ndot * (a - b); /* krate25 */
        T->rates[4 * nc + c] = T->rates[4 * nc + c] + ndot * (ea - eb);
        d1 = d1 + tau1; d2 = d2 + tau2; d3 = d3 + tau3; dd = dd + taud;
        if (strategy == PT_PROCEED) {
            cell = T->nb_node; lvl = T->nb_level;
            memcpy(seq, T->nb_seq, sizeof(int) * (3 * lvl + 3));
        }
    }
    if (strategy != PT_SPLIT || T->err) return;

    for (int iray = 1; iray <= 4; ++iray) {
        ppixel child;
        child.level = px->level + 1;
        if (pt_pixel(T, child.level, 4 * iray_start + iray - 5, &child.phi, &child.theta)) { T->err = -31; return; }
        if (child.level > T->highest_pixel_level) T->highest_pixel_level = child.level;
        /* absoluteCoordinates, :3011-3047 */
        double p[3] = {pt[0], pt[1], pt[2]};
        for (int l = lvl; l > 0; --l)
            for (int q = 0; q < 3; ++q) p[q] = seq[3 * l + q] == 1 ? 0.5 * p[q] : 0.5 * p[q] + 0.5;
        const double fn = (double)(float)T->n;
        double xb = ((double)(float)(seq[0] - 1) + p[0]) / fn, yb = ((double)(float)(seq[1] - 1) + p[1]) / fn,
               zb = ((double)(float)(seq[2] - 1) + p[2]) / fn;
        xb = xb + radius / fn * (cos(child.phi) * cos(child.theta) - cos(px->phi) * cos(px->theta));
        yb = yb + radius / fn * (sin(child.phi) * cos(child.theta) - sin(px->phi) * cos(px->theta));
        zb = zb + radius / fn * (sin(child.theta) - sin(px->theta));
        if (xb < 0. || xb > 1. || yb < 0. || yb > 1. || zb < 0. || zb > 1.) {
            strategy = PT_BOUNDARY; /* and stays so: :3336-3345 */
            if (T->esc) {
                static const float out_radius[7] = {0.1f, 0.3f, 1.f, 3.f, 10.f, 30.f, 100.f};
                const double kpc = F(1.e3) * F(3.08568025e18);
                const double tmp = radius * T->box / ((double)(float)T->n * kpc);
                for (int ir = 0; ir < 7; ++ir)
                    if ((double)out_radius[ir] > tmp) T->esc[7 + ir] = T->esc[7 + ir] + ndot / 4.;
            }
        }
        if (strategy == PT_BOUNDARY) continue;
        /* localizeSplitContinuationCell, :3049-3118 */
        int cseq[40], cl = 0;
        int i = (int)(xb * T->n) + 1, j = (int)(yb * T->n) + 1, k = (int)(zb * T->n) + 1;
        if (i < 1 || i > T->n || j < 1 || j > T->n || k < 1 || k > T->n) { T->err = -32; return; }
        cseq[0] = i; cseq[1] = j; cseq[2] = k;
        int cc = pt_base(T, i, j, k);
        double q[3] = {xb * fn - (double)(float)(i - 1), yb * fn - (double)(float)(j - 1), zb * fn - (double)(float)(k - 1)};
        while (T->node[cc].refined) {
            int h[3];
            for (int m = 0; m < 3; ++m) { h[m] = q[m] < 0.5 ? 1 : 2; q[m] = h[m] == 1 ? 2. * q[m] : 2. * q[m] - 1.; }
            ++cl;
            cseq[3 * cl] = h[0]; cseq[3 * cl + 1] = h[1]; cseq[3 * cl + 2] = h[2];
            cc = pt_child(T, cc, h[0], h[1], h[2]);
        }
        if (q[0] < 0. || q[0] > 1. || q[1] < 0. || q[1] > 1. || q[2] < 0. || q[2] > 1.) { T->err = -30; return; }
        pt_ray(T, cc, q, &child, 4 * iray_start + iray - 4, cl, cseq, radius, ndot / 4.0, d1, d2, d3, dd);
    }
}

/* The per-source loop, equiSources.f90:1256-1329.  src_leaf: 0-based cell-array indices; rates: [6][ncell] in the order
 * krate24, krate25, krate26, crate24, crate25, crate26, overwritten (setZeroRates, :4128). */
int fo_point_sources(int n, int64_t ncell, const int32_t *level, const double *HI, const double *HeI, const double *HeII,
                     const double *rho, const double *abun2, double box, int dust, int nsrc, const int64_t *src_leaf,
                     const double *src_ndot, const double *tables, double *rates, int *highest_pixel_level,
                     const double *pix, int pix_levels)
{
    return fo_point_sources_escape(n, ncell, level, HI, HeI, HeII, rho, abun2, box, dust, nsrc, src_leaf, src_ndot, tables, rates,
                                   highest_pixel_level, pix, pix_levels, NULL, NULL, NULL);
}

/* The same with the escape bookkeeping: escape [nsrc][315] = per star ndotRemaining[7], ndotBoundary[7], ndotDust,
 * ndotSpectrum[300] (needs out_sigma [4][300] = outputSigma24, 25, 26, Dust; NULL leaves the spectrum zero), and
 * fraction [nsrc][7] as the main program forms it for its `src:` line, equiSources.f90:1342-1348.  Either may be NULL. */
int fo_point_sources_escape(int n, int64_t ncell, const int32_t *level, const double *HI, const double *HeI, const double *HeII,
                            const double *rho, const double *abun2, double box, int dust, int nsrc, const int64_t *src_leaf,
                            const double *src_ndot, const double *tables, double *rates, int *highest_pixel_level,
                            const double *pix, int pix_levels, const double *out_sigma, double *escape, double *fraction)
{
    ptree T;
    memset(&T, 0, sizeof T);
    T.pix = pix; T.pix_levels = pix_levels;
    T.levels = level; T.ncell = ncell; T.n = n; T.dust = dust; T.HI = HI; T.HeI = HeI; T.HeII = HeII; T.rho = rho;
    T.abun2 = abun2; T.tables = tables; T.box = box; T.rates = rates;
    T.out_sigma = out_sigma;
    double esc_one[315];
    fo_rmax(T.rmax);
    const int nbase = n * n * n;
    T.cap = nbase + 8; T.node = malloc((size_t)T.cap * sizeof *T.node); T.nnode = nbase;
    for (int b = 0; b < nbase; ++b) { T.node[b].parent = -1; pt_grow(&T, b, 0); }
    if (!T.err && T.cursor != ncell) T.err = -25;
    if (T.err) { free(T.node); return T.err; }
    memset(rates, 0, sizeof(double) * 6 * (size_t)ncell);

    for (int s = 0; s < nsrc && !T.err; ++s) {
        /* host leaf and its call sequence */
        int host = -1;
        for (int q = 0; q < T.nnode; ++q) if (!T.node[q].refined && T.node[q].leaf == src_leaf[s]) { host = q; break; }
        if (host < 0) { T.err = -33; break; }
        const int lvl = T.node[host].level;
        int seq[40], c = host;
        for (int l = lvl; l > 0; --l) {
            const int idx = c - T.node[T.node[c].parent].child0;
            seq[3 * l] = idx / 4 + 1; seq[3 * l + 1] = (idx / 2) % 2 + 1; seq[3 * l + 2] = idx % 2 + 1;
            c = T.node[c].parent;
        }
        seq[0] = c / (n * n) + 1; seq[1] = (c / n) % n + 1; seq[2] = c % n + 1;
        const double centre[3] = {0.5, 0.5, 0.5};
        T.esc = (escape || fraction) ? esc_one : NULL;
        memset(esc_one, 0, sizeof esc_one); /* :1267-1270 */
        for (int iray = 1; iray <= 12; ++iray) {
            ppixel px;
            px.level = 1;
            if (pt_pixel(&T, 1, iray - 1, &px.phi, &px.theta)) { T.err = -31; break; }
            pt_ray(&T, host, centre, &px, iray, lvl, seq, 0.0, src_ndot[s] / 12.0, 0., 0., 0., 0.);
        }
        if (escape) memcpy(escape + 315 * (size_t)s, esc_one, sizeof esc_one);
        if (fraction)
            for (int ir = 0; ir < 7; ++ir) /* :1342-1348 */
                fraction[7 * s + ir] = esc_one[7 + ir] < 1. ? esc_one[ir] / (src_ndot[s] - esc_one[7 + ir]) : 0.;
    }
    if (highest_pixel_level) *highest_pixel_level = T.highest_pixel_level;
    free(T.node);
    return T.err;
}
