/* ftte_math.h -- the segment arithmetic of the device sweep, written once.
 *
 * One ray segment of optical depth tau = kappa*dpath attenuates the incoming
 * intensity and contributes its path-mean intensity to the cell:
 *
 *     Iout = Iin * exp(-tau)                 transportRoutinesModule.f90:651-678 (eta == 0)
 *     mean = (Iin - Iout)/log(Iin/Iout)      transportRoutinesModule.f90:1036-1054
 *
 * With zero emissivity log(Iin/Iout) == tau identically, so the log-mean is
 * Iin * g(tau), g(tau) = (1 - exp(-tau))/tau.  The device evaluates g without a
 * logarithm (and, below tau = ln2/2, without the cancellation the reference's
 * quotient suffers: the reference's own rounding noise is ~eps/tau, see
 * tests/test_parity_gpu.py for the tolerance this implies).
 *
 * Everything here is plain IEEE binary64 with explicit fused multiply-adds, no
 * library call whose rounding is unspecified: the same source compiled by gcc
 * for the host (-ffp-contract=off -mfma) and by hipcc for gfx950
 * (-ffp-contract=off) produces identical bits, which is what lets the parity
 * tests compare the GPU against a CPU evaluation bit for bit.
 *
 * Coefficients: tools/fit_exp_poly.py (degree 9, |r| <= ln2/2: approximation
 * error 4.1e-17 in g, 1.6e-17 in exp, before rounding); see ftte_consts below.
 */
#ifndef FTTE_MATH_H
#define FTTE_MATH_H

#if defined(__HIPCC__)
#define FTTE_HD __host__ __device__ __forceinline__
#define FTTE_FMA(a, b, c) __builtin_fma((a), (b), (c))
#define FTTE_RINT(a) __builtin_rint(a)
#define FTTE_LDEXP(a, n) __builtin_ldexp((a), (n))
#define FTTE_FMAX(a, b) __builtin_fmax((a), (b))
#define FTTE_FREXP(a, pe) __builtin_frexp((a), (pe))
#if defined(__HIP_DEVICE_COMPILE__)
/* a/b correctly rounded, for normal-range operands: v_rcp_f64 seed, two Newton steps, quotient, one residual
 * correction -- the instruction sequence hipcc itself emits for an IEEE fp64 division, minus the
 * v_div_scale / v_div_fixup range handling that 0.29 <= a <= 1, 0.34 <= b < 1e300 never needs (lanes outside
 * that range discard the result).  Same bits as the host's `/`: checked by the bitwise parity tests. */
__device__ __forceinline__ double ftte_div(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double t = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, t, y);
    t = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, t, y);
    const double q = a * y;
    const double r = __builtin_fma(-b, q, a);
    return __builtin_fma(r, y, q);
}
#define FTTE_DIV(a, b) ftte_div((a), (b))
/* evaluate x here, in every lane: without this hipcc turns the select between the two forms of g into a
 * divergent branch around the division and keeps every row's operands alive across it (2x the registers) */
#define FTTE_KEEP(x) asm volatile("" : "+v"(x))
/* true if the condition holds in any lane of the wavefront: lets a whole wave skip work none of its lanes needs
 * (the result of every lane is the same as if each had evaluated the condition for itself) */
#define FTTE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0ul)
#else
#define FTTE_DIV(a, b) ((a) / (b))
#define FTTE_KEEP(x) ((void)0)
#define FTTE_ANY(c) (c)
#endif
#else
#include <math.h>
#define FTTE_HD static inline
#define FTTE_FMA(a, b, c) fma((a), (b), (c))
#define FTTE_RINT(a) rint(a)
#define FTTE_LDEXP(a, n) ldexp((a), (n))
#define FTTE_FMAX(a, b) fmax((a), (b))
#define FTTE_FREXP(a, pe) frexp((a), (pe))
#define FTTE_DIV(a, b) ((a) / (b))
#define FTTE_KEEP(x) ((void)0)
#define FTTE_ANY(c) (c)
#endif

/* The constants of the attenuation pair, passed as a table: on the device the table arrives through the kernel
 * argument segment, so the coefficients sit in scalar registers and every Horner step is one v_fma_f64 with a scalar
 * addend (as literals hipcc parks them in vector registers and pays a v_mov_b64 per step). */
typedef struct {
    double log2e;   /* 1/ln2 */
    double ln2_hi;  /* ln2, low 21 bits clear: n*ln2_hi exact for |n| < 2^21 */
    double ln2_lo;  /* ln2 - ln2_hi */
    double x_floor; /* exp(x_floor) == 0 in binary64; keeps n inside int range */
    double c[10];   /* (exp(r)-1-r)/r^2 ~ c0 + c1 r + ... + c9 r^9, tools/fit_exp_poly.py */
    double sqrt_half; /* sqrt(1/2) */
    double lg[7];   /* (atanh(s)/s - 1)/s^2 ~ lg0 + lg1 z + ... + lg6 z^6, z = s^2, tools/fit_log_poly.py */
    double s_max;   /* (sqrt(2)-1)/(sqrt(2)+1): below it the log-mean of two intensities needs no logarithm */
    double mn[7];   /* (s/atanh(s) - 1)/s^2 ~ mn0 + mn1 z + ... + mn6 z^6, tools/fit_mean_poly.py */
    double thin_max; /* the largest tau with rint(fl(-tau * log2e)) == 0, i.e. n = 0 in the range reduction: ln2/2 rounded down to
                      * 0x1.62e42fefa39efp-2 (found by bisection, checked ulp by ulp on both sides: tests/test_device_math.py) */
} ftte_consts;

#define FTTE_CONSTS_INIT                                                                                              \
    {                                                                                                                 \
        0x1.71547652b82fep+0, 0x1.62e42fee00000p-1, 0x1.a39ef35793c76p-33, -1000.0,                                   \
        {                                                                                                             \
            0x1.0000000000001p-1, 0x1.5555555555556p-3, 0x1.5555555553d63p-5, 0x1.11111111109b3p-7,                   \
                0x1.6c16c1788bd90p-10, 0x1.a01a01a7c41d5p-13, 0x1.a019b90d2ae7ap-16, 0x1.71de0dae63bb3p-19,           \
                0x1.289185613a3d6p-22, 0x1.af38a9b0ec855p-26                                                          \
        },                                                                                                            \
            0x1.6a09e667f3bcdp-1,                                                                                     \
        {                                                                                                             \
            0x1.5555555555558p-2, 0x1.9999999995273p-3, 0x1.2492492dfd879p-3, 0x1.c71c62d5e43dfp-4,                   \
                0x1.7462b91bd6c45p-4, 0x1.39fdcce1da1aep-4, 0x1.2b5f6919aa514p-4                                      \
        },                                                                                                            \
            0x1.5f619980c4336p-3,                                                                                     \
        {                                                                                                             \
            -0x1.5555555555556p-2, -0x1.6c16c16c1511dp-4, -0x1.7d6d2c2f3c50dp-5, -0x1.eeb2c916bdc4dp-6,               \
                -0x1.6522b1e23e2e9p-6, -0x1.123931a9bb85bp-6, -0x1.e26cad68d7a18p-7                                   \
        },                                                                                                            \
            0x1.62e42fefa39efp-2                                                                                      \
    }

/* e = exp(-tau), g = (1-exp(-tau))/tau  (g(0) = 1).  tau >= 0 expected; a
 * negative tau (unphysical opacity) still evaluates, up to overflow. */
FTTE_HD int ftte_attenuation_lead(const ftte_consts *K, double lead, double tau, double *e_out, double *g_out);
FTTE_HD void ftte_attenuation(const ftte_consts *K, double tau, double *e_out, double *g_out)
{
    (void)ftte_attenuation_lead(K, K->c[9], tau, e_out, g_out);
}

/* The range reduction.  n = rint(-tau log2e) is 0 exactly for |tau| <= thin_max (the product is monotonic in tau, the bound is the
 * last tau whose rounded product rounds to 0): a wavefront of such segments needs no reduction at all, and the test of the
 * wavefront is one comparison of tau itself.  The polynomial's argument is carried with the opposite sign, t = -r = tau + n ln2,
 * so that for n = 0 it is tau as it stands (negating an operand of a fused multiply-add is free, a negated copy is not); the two
 * reduction steps round to the same magnitudes either way. */
FTTE_HD int ftte_all_thin(const ftte_consts *K, double tau) { return !FTTE_ANY(!(__builtin_fabs(tau) <= K->thin_max)); }

FTTE_HD double ftte_reduce(const ftte_consts *K, double tau)
{
    const double nf = FTTE_RINT(-tau * K->log2e);
    double t = FTTE_FMA(nf, K->ln2_hi, tau);
    t = FTTE_FMA(nf, K->ln2_lo, t);
    FTTE_KEEP(t); /* a branch of the wavefront: nothing of this is hoisted in front of the test */
    return t;
}

/* g = expm1(r)/r and e = exp(r) for r = -t, |t| <= ln2/2 */
FTTE_HD void ftte_exp_reduced(const ftte_consts *K, double lead, double t, double *e_out, double *g_out)
{
    double q = lead;
    q = FTTE_FMA(q, -t, K->c[8]);
    q = FTTE_FMA(q, -t, K->c[7]);
    q = FTTE_FMA(q, -t, K->c[6]);
    q = FTTE_FMA(q, -t, K->c[5]);
    q = FTTE_FMA(q, -t, K->c[4]);
    q = FTTE_FMA(q, -t, K->c[3]);
    q = FTTE_FMA(q, -t, K->c[2]);
    q = FTTE_FMA(q, -t, K->c[1]);
    q = FTTE_FMA(q, -t, K->c[0]);
    const double g = FTTE_FMA(-t, q, 1.0);
    *g_out = g;
    *e_out = FTTE_FMA(-t, g, 1.0);
}

/* What the lanes outside the thin range need on top (every lane of a wavefront with such a lane passes through here; the thin ones
 * keep their values): the scaling by 2^n, the floor at exp(-1000) = 0, g by division. */
FTTE_HD void ftte_thick_part(const ftte_consts *K, double tau, double *e_io, double *g_io)
{
    /* below x_floor exp(x) is 0 in binary64 (and t, q of such a lane are whatever the reduction of a huge x gives: not used);
     * n is kept inside the range of an int for it */
    const double nf = FTTE_RINT(-tau * K->log2e);
    const int n = (int)FTTE_FMAX(nf, -2048.0);
    const double scaled = FTTE_LDEXP(*e_io, n); /* n == 0: e itself */
    const double e = (-tau >= K->x_floor) ? scaled : 0.0;
    double gd = FTTE_DIV(1.0 - e, tau); /* n != 0: tau >= ln2/2, 1-e >= 0.29, no cancellation */
    FTTE_KEEP(gd);
    *g_io = (nf == 0.0) ? *g_io : gd;
    *e_io = e;
}

/* `lead`: the polynomial's leading coefficient, K->c[9], handed in by the caller.  A device kernel that keeps it in a vector
 * register for its whole run saves the move every first Horner step otherwise starts with (an instruction reads one scalar
 * operand: c9 * r + c8 with both coefficients in scalar registers needs one of them copied first).
 *
 * Returns whether any lane of the wavefront (on the host: this evaluation) left the thin range |tau| <= thin_max, where n = 0,
 * exp(-tau) = exp(r) needs no scaling and g is the polynomial itself.  Everything the other lanes need -- the range reduction,
 * the scaling by 2^n, the floor at exp(-1000) = 0, the division -- is done behind that one test, so a wavefront of thin segments
 * (five in six of the benchmark's) executes the test, eleven fused multiply-adds, and nothing else.  The values are
 * those of the straightforward form x = max(-tau, x_floor), e = ldexp(exp(r), n), g = n ? (1 - e)/tau : expm1(r)/r. */
FTTE_HD int ftte_attenuation_lead(const ftte_consts *K, double lead, double tau, double *e_out, double *g_out)
{
    if (ftte_all_thin(K, tau)) {
        ftte_exp_reduced(K, lead, tau, e_out, g_out);
        return 0;
    }
    double e, g;
    ftte_exp_reduced(K, lead, ftte_reduce(K, tau), &e, &g);
    ftte_thick_part(K, tau, &e, &g);
    *e_out = e;
    *g_out = g;
    return 1;
}

/* One segment: advances the ray intensity and returns the path-mean intensity
 * the cell receives from it.  Iout == 0 (underflow) yields a zero mean, as the
 * reference's (Iin-0)/log(Iin/0) does.  A thin segment (n = 0: exp(-tau) > 0.7) takes a nonzero Iin to a nonzero Iout --
 * the product with the smallest subnormal still rounds to it --, and with Iin = 0 the mean is 0 as it stands: the fix-up is
 * needed only where some lane is thick.  Two complete paths behind the one test of the wavefront: the thin one is the test,
 * eleven fused multiply-adds on tau itself and the two products, and nothing else. */
FTTE_HD double ftte_segment_lead(const ftte_consts *K, double lead, double *I, double tau)
{
    double e, g;
    const double Iin = *I;
    if (__builtin_expect(ftte_all_thin(K, tau), 1)) {
        ftte_exp_reduced(K, lead, tau, &e, &g);
        *I = Iin * e;
        return Iin * g;
    }
    ftte_exp_reduced(K, lead, ftte_reduce(K, tau), &e, &g);
    ftte_thick_part(K, tau, &e, &g);
    const double Iout = Iin * e;
    double mean = Iin * g;
    mean = (Iout == 0.0) ? 0.0 : mean;
    FTTE_KEEP(mean); /* (a branch, not a pair of selects every lane pays for) */
    *I = Iout;
    return mean;
}

FTTE_HD double ftte_segment(const ftte_consts *K, double *I, double tau)
{
    return ftte_segment_lead(K, K->c[9], I, tau);
}

/* One segment through a cell with a SOURCE FUNCTION S (the build's own extension for source iterations: the reference has no
 * enabled emission and no scattering loop).  With S constant along the piece the formal solution is I(t) = S + (Iin - S) exp(-t):
 *     Iout = S + (Iin - S) exp(-tau),        mean = (1/tau) int_0^tau I dt = S + (Iin - S) g(tau),   g = (1 - exp(-tau))/tau,
 * the exact path mean -- what the reference's log-mean (Iin - Iout)/log(Iin/Iout) is for S = 0 (there log(Iin/Iout) = tau), and what
 * it only approximates once a source term makes the profile something else than one exponential.  For S = 0 the two fused
 * multiply-adds return Iin e and Iin g as they stand: the same bits as ftte_segment_lead.  In radiative equilibrium (Iin = S) nothing
 * changes, exactly.  Three instructions on top of the attenuation pair; the same wavefront test as there. */
FTTE_HD double ftte_segment_source(const ftte_consts *K, double lead, double *I, double tau, double S)
{
    double e, g;
    const double dI = *I - S;
    if (__builtin_expect(ftte_all_thin(K, tau), 1)) {
        ftte_exp_reduced(K, lead, tau, &e, &g);
        *I = FTTE_FMA(dI, e, S);
        return FTTE_FMA(dI, g, S);
    }
    ftte_exp_reduced(K, lead, ftte_reduce(K, tau), &e, &g);
    ftte_thick_part(K, tau, &e, &g);
    const double Iout = FTTE_FMA(dI, e, S);
    double mean = FTTE_FMA(dI, g, S);
    mean = (Iout == 0.0) ? 0.0 : mean; /* (S = 0 and complete extinction: as ftte_segment_lead) */
    FTTE_KEEP(mean);
    *I = Iout;
    return mean;
}

/* log(1 + t) for t >= 0, accurate also for tiny t (where forming 1 + t first would lose t's low bits):
 * 1 + t = 2^k m, m in [sqrt(1/2), sqrt(2)); for t < sqrt(2) - 1, k = 0 and f = m - 1 = t exactly.
 * log m = 2 atanh(s), s = f/(2+f), |s| <= 0.1716, degree-6 polynomial in s^2 (approximation error 4.7e-18).
 * Same bits on host and device: frexp, one correctly rounded division, explicit FMAs. */
FTTE_HD double ftte_log1p(const ftte_consts *K, double t)
{
    int k;
    double m = FTTE_FREXP(1.0 + t, &k); /* [1/2, 1) */
    if (m < K->sqrt_half) { m = m + m; k -= 1; }
    const int small = t < 0x1.a827999fcef32p-2; /* sqrt(2) - 1 */
    const double f = small ? t : m - 1.0;
    const double kf = small ? 0.0 : (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = K->lg[6];
    p = FTTE_FMA(p, z, K->lg[5]);
    p = FTTE_FMA(p, z, K->lg[4]);
    p = FTTE_FMA(p, z, K->lg[3]);
    p = FTTE_FMA(p, z, K->lg[2]);
    p = FTTE_FMA(p, z, K->lg[1]);
    p = FTTE_FMA(p, z, K->lg[0]);
    const double s2 = s + s;
    const double logm = FTTE_FMA(s2, z * p, s2);
    return FTTE_FMA(kf, K->ln2_hi, FTTE_FMA(kf, K->ln2_lo, logm));
}

/* One segment with the reference's emissivity.  The reference's (never enabled) emission term, transportRoutinesModule.f90:673-678:
 *     Iout = Iin*exp(-tau) + eta * ((tau > 1e-10) ? (1-exp(-tau))/kappa : dpath) / dpath  =  Iin*e + eta*g(tau)
 * (note the reference's extra 1/dpath: its eta is not an emissivity per unit length).  `src` adds S*(1-exp(-tau)) = (src*tau)*g to
 * Iout in the same form (kept for cross-checks on the host; the kernels' source-function mode is ftte_segment_source above, with
 * the exact path mean).  With emission log(Iin/Iout) != tau, so the cell's share is the reference's log-mean itself
 * (transportRoutinesModule.f90:1044-1048), evaluated from the difference Iin-Iout (below): the same number, without
 * the reference's loss of the difference when the quotient Iin/Iout is rounded (fatal near Iout = Iin, i.e. wherever
 * the radiation field is close to the source function). */
FTTE_HD double ftte_segment_emit(const ftte_consts *K, double *I, double tau, double eta, double src)
{
    double e, g;
    ftte_attenuation(K, tau, &e, &g);
    const double Iin = *I;
    const double emis = FTTE_FMA(src, tau, eta);
    const double Iout = FTTE_FMA(emis, g, Iin * e);
    *I = Iout;
    const double diff = Iin - Iout, sum = Iin + Iout;
    const double rising = 0.5 * sum;
    /* (Iin-Iout)/log(Iin/Iout) = A s/atanh(s) with A = (Iin+Iout)/2, s = (Iin-Iout)/(Iin+Iout).  While Iin/Iout < sqrt(2)
     * (s < s_max: wherever the field is near the source function, and in every thin segment) s/atanh(s) = 1 + z h(z),
     * z = s^2, is a polynomial: one division, no logarithm. */
    /* Deep inside an opaque, source-free region the intensities fall through the bottom of the normal range: below 2^-900 both
     * operands are first scaled by 2^200 (exact, and the quotient is the same number), so that the device's division -- a
     * reciprocal refined by Newton steps, which needs normal operands and a representable 1/sum -- and the host's `/` still
     * round the same quotient the same way instead of the device producing NaN from 1/subnormal = inf. */
    const double lift = (sum < 0x1p-900) ? 0x1p+200 : 1.0;
    const double s = FTTE_DIV(diff * lift, sum * lift);
    const double z = s * s;
    double h = K->mn[6];
    h = FTTE_FMA(h, z, K->mn[5]);
    h = FTTE_FMA(h, z, K->mn[4]);
    h = FTTE_FMA(h, z, K->mn[3]);
    h = FTTE_FMA(h, z, K->mn[2]);
    h = FTTE_FMA(h, z, K->mn[1]);
    h = FTTE_FMA(h, z, K->mn[0]);
    double falling = FTTE_FMA(rising * z, h, rising);
    if (FTTE_ANY(s >= K->s_max)) { /* a steep drop somewhere in the wavefront: the general form for those lanes */
        double steep = diff / ftte_log1p(K, diff / Iout); /* Iout == 0: selected away below (IEEE divisions: the compiler's own) */
        FTTE_KEEP(steep);
        falling = (s >= K->s_max) ? steep : falling;
    }
    return (Iout < Iin) ? ((Iout == 0.0) ? 0.0 : falling) : rising;
}

/* (acc/nseg)*w with acc/nseg correctly rounded for nseg in {1,2,3} without a
 * division: q = acc*(1/3) refined by one residual step is the correctly
 * rounded quotient (Markstein), as an IEEE division would give.
 * transportRoutinesModule.f90:953-955. */
FTTE_HD double ftte_cell_mean(double acc, int nseg, double w)
{
    double q;
    if (nseg == 3) {
        const double third = 0x1.5555555555555p-2;
        q = acc * third;
        const double res = FTTE_FMA(-3.0, q, acc);
        q = FTTE_FMA(res, third, q);
    } else if (nseg == 2) {
        q = acc * 0.5;
    } else {
        q = acc;
    }
    return q * w;
}

#endif /* FTTE_MATH_H */
