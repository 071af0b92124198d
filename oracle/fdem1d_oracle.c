/*
 * oracle/fdem1d_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU restatement of GeoBIPy's 1-D layered-earth FDEM forward
 * solve, its analytic Jacobian, and the Gaussian data misfit / log-likelihood.
 * It is the checker for the HIP path (tests/, __graft_entry__.smoke()) and the
 * "port" CPU baseline timed by bench.py.  Nothing under geobipy_amd/ may call it.
 *
 * Parity pin: validated in the build container against (i) the imported
 * reference (tests/golden/make_golden.py, fixtures in the .npz fixtures in tests/golden) and
 * (ii) the reference's own known-answer files tests/data_checks/resolve_*_clean.csv
 * (copied to tests/golden/) with the reference's np.allclose criterion and the
 * tighter |d| <= 1e-7 ppm + 1e-9|ref| bound.
 *
 * Every function names the reference lines it follows (paths relative to
 * /root/reference/geobipy/src/classes/):
 *   FD  = forwardmodelling/Electromagnetic/FD/fdem1d_numba.py
 *   DP  = data/datapoint/DataPoint.py
 *   MVN = statistics/MvNormalDistribution.py
 *
 * The operation order of the reference is kept (sequential abscissa sums,
 * reciprocal-then-multiply for Yn, Smith complex division as numpy/numba do),
 * so differences to the interpreted reference stay at the 1e-12 ppm level.
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NC0 120 /* FD:18 */
#define NC1 140 /* FD:19 */

typedef struct { double re, im; } cx;

static const double PI = 3.14159265358979323846;

/* FD:13-17 */
static double mu0(void) { return 4.e-7 * PI; }
static double eps0(void) { const double c = 299792458.0; return 1.0 / (mu0() * pow(c, 2.0)); }

static inline cx cx_(double re, double im) { cx z = {re, im}; return z; }
static inline cx cadd(cx a, cx b) { return cx_(a.re + b.re, a.im + b.im); }
static inline cx csub(cx a, cx b) { return cx_(a.re - b.re, a.im - b.im); }
static inline cx cmul(cx a, cx b) { return cx_(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline cx cscale(cx a, double s) { return cx_(a.re * s, a.im * s); }
/* Smith's algorithm: what numpy's complex128 true_divide loop and numba's
 * complex division both use. */
static inline cx cdiv(cx a, cx b)
{
    double abr = fabs(b.re), abi = fabs(b.im);
    if (abr >= abi) {
        if (abr == 0.0 && abi == 0.0) return cx_(a.re / abr, a.im / abr);
        double rat = b.im / b.re;
        double scl = 1.0 / (b.re + b.im * rat);
        return cx_((a.re + a.im * rat) * scl, (a.im - a.re * rat) * scl);
    } else {
        double rat = b.re / b.im;
        double scl = 1.0 / (b.im + b.re * rat);
        return cx_((a.re * rat + a.im) * scl, (a.im * rat - a.re) * scl);
    }
}
static inline cx csqrt_(cx a) { double complex r = csqrt(CMPLX(a.re, a.im)); return cx_(creal(r), cimag(r)); }
static inline cx cexp_(cx a) { double complex r = cexp(CMPLX(a.re, a.im)); return cx_(creal(r), cimag(r)); }
static inline cx cneg(cx a) { return cx_(-a.re, -a.im); }
static inline cx radd(double r, cx a) { return cx_(r + a.re, a.im); }   /* real + complex */
static inline cx rsub(double r, cx a) { return cx_(r - a.re, -a.im); }  /* real - complex */

/* FD:441-448 cTanh: overflow-safe complex tanh */
static cx cTanh(cx z)
{
    if (z.re > 0.0) {
        cx t = cexp_(cscale(z, -2.0));
        return cdiv(rsub(1.0, t), radd(1.0, t));
    } else {
        cx t = cexp_(cscale(z, 2.0));
        return cdiv(cx_(t.re - 1.0, t.im), cx_(t.re + 1.0, t.im));
    }
}

/* Work arrays for one abscissa set: un, Y, Yn each [(L+1), F, C] (FD:161-163). */
typedef struct {
    int nL1, nF, nC;
    cx *un, *Y, *Yn;
} coef_t;

#define IDX(w, k, i, jc) (((size_t)(k) * (w)->nF + (i)) * (w)->nC + (jc))

static int coef_alloc(coef_t *w, int nL1, int nF, int nC)
{
    size_t n = (size_t)nL1 * nF * nC;
    w->nL1 = nL1; w->nF = nF; w->nC = nC;
    w->un = (cx *)malloc(n * sizeof(cx));
    w->Y = (cx *)malloc(n * sizeof(cx));
    w->Yn = (cx *)malloc(n * sizeof(cx));
    return (w->un && w->Y && w->Yn) ? 0 : -1;
}
static void coef_free(coef_t *w) { free(w->un); free(w->Y); free(w->Yn); }

/* FD:157-191 initCoefficients.  par/kappa/perm have the air layer prepended. */
static void initCoefficients(coef_t *w, int nLayers, const double *frequencies,
                             const double *lamda2 /* [F, C] */, const double *par,
                             const double *kappa, const double *perm)
{
    const int nL1 = nLayers + 1, nF = w->nF, nC = w->nC;
    for (int k = 0; k < nL1; ++k) {
        double p = par[k];
        double mu = mu0() * (1.0 + kappa[k]);
        double eps = eps0() * (1.0 + perm[k]);
        for (int i = 0; i < nF; ++i) {
            double omega = 2.0 * PI * frequencies[i];  /* FD:166-168 */
            cx yn = cx_(p, omega * eps);                /* FD:176 */
            cx zn = cx_(0.0, omega * mu);               /* FD:177 */
            cx ynzn = cmul(yn, zn);                     /* FD:178 */
            cx zn1 = cdiv(cx_(1.0, 0.0), zn);           /* FD:179 */
            for (int jc = 0; jc < nC; ++jc) {
                cx tmp = csqrt_(radd(lamda2[(size_t)i * nC + jc], ynzn)); /* FD:182 */
                /* python: ynzn + lamda2 keeps imag of ynzn */
                w->un[IDX(w, k, i, jc)] = tmp;
                w->Yn[IDX(w, k, i, jc)] = cmul(tmp, zn1); /* FD:184 */
            }
        }
    }
    for (int i = 0; i < nF; ++i)
        for (int jc = 0; jc < nC; ++jc)
            w->Y[IDX(w, nL1 - 1, i, jc)] = w->Yn[IDX(w, nL1 - 1, i, jc)]; /* FD:187-189 */
}

/* FD:194-219 M1_0: admittance recursion, returns rTE[F,C], u0[F,C] */
static void M1_0(coef_t *w, int nLayers, const double *thk, cx *rTE, cx *u0)
{
    const int nF = w->nF, nC = w->nC;
    for (int k = nLayers - 1; k > 0; --k) {
        double t = thk[k];
        for (int i = 0; i < nF; ++i)
            for (int jc = 0; jc < nC; ++jc) {
                cx Yn_ = w->Yn[IDX(w, k, i, jc)];
                cx Y_ = w->Y[IDX(w, k + 1, i, jc)];
                cx z = cscale(w->un[IDX(w, k, i, jc)], t);
                cx a0 = cTanh(z);
                /* FD:210 Yn_ * (Y_ + (Yn_ * a0)) / (Yn_ + (Y_ * a0)) : (a*b)/c */
                cx num = cmul(Yn_, cadd(Y_, cmul(Yn_, a0)));
                cx den = cadd(Yn_, cmul(Y_, a0));
                w->Y[IDX(w, k, i, jc)] = cdiv(num, den);
            }
    }
    for (int i = 0; i < nF; ++i)
        for (int jc = 0; jc < nC; ++jc) {
            u0[(size_t)i * nC + jc] = w->un[IDX(w, 0, i, jc)];
            cx Yn_ = w->Yn[IDX(w, 0, i, jc)];
            cx Y_ = w->Y[IDX(w, 1, i, jc)];
            rTE[(size_t)i * nC + jc] = cdiv(csub(Yn_, Y_), cadd(Yn_, Y_)); /* FD:217 */
        }
}

/* numpy complex ** 2.0 / ** 3.0 -> repeated multiplication (npy_cpow integer path) */
static inline cx cpow2(cx a) { return cmul(a, a); }
static inline cx cpow3(cx a) { return cmul(a, cmul(a, a)); }

/* Test switch (NOT in the reference): 0 = the reference's M1_1 expression (FD:269-274), 1 = the true derivative of the forward
 * recursion -- the reference's "(Y_2 - Yn_2) * tanuh + 2.0 * Yn_2" becomes "(Y_2 + Yn_2) * tanuh" (DESIGN.md 3.4).  Used by the CPU
 * replays of device chains that run in exact-Jacobian mode (tests/config5_replay.py); checked against central differences of
 * oracle_fdem1dfwd in tests/test_oracle_golden.py. */
static int g_exact_jacobian = 0;
void oracle_set_exact_jacobian(int exact) { g_exact_jacobian = exact != 0; }

/* FD:222-303 M1_1: recursion + per-layer sensitivities sens[L,F,C] */
static void M1_1(coef_t *w, int nLayers, const double *frequencies, const double *thk,
                 const double *par, const double *kappa, cx *u0, cx *sens)
{
    const int nF = w->nF, nC = w->nC;
    const size_t FC = (size_t)nF * nC;
    cx *accumulate = (cx *)malloc((size_t)(nLayers - 1) * FC * sizeof(cx));

    for (int k = nLayers - 1; k > 0; --k) {
        int k1 = k + 1, k2 = k - 1;
        double p = par[k], t = thk[k];
        double mu = mu0() * (1.0 + kappa[k]);
        for (int i = 0; i < nF; ++i) {
            double omega = 2.0 * PI * frequencies[i];
            cx oTmp = cx_(0.0, omega * mu * t); /* FD:245 */
            for (int jc = 0; jc < nC; ++jc) {
                cx Yn_ = w->Yn[IDX(w, k, i, jc)];
                cx Yn_2 = cpow2(Yn_), Yn_3 = cpow3(Yn_);
                cx Y_ = w->Y[IDX(w, k1, i, jc)];
                cx Y_2 = cpow2(Y_);
                cx un_ = w->un[IDX(w, k, i, jc)];
                cx z = cscale(un_, t);
                cx tanuh = cTanh(z);
                cx tanuh2 = cpow2(tanuh);
                cx num = cadd(Y_, cmul(Yn_, tanuh));
                cx den = cadd(Yn_, cmul(Y_, tanuh));
                w->Y[IDX(w, k, i, jc)] = cdiv(cmul(Yn_, num), den); /* FD:265 */
                /* FD:267 (Yn_2 * (1 - tanuh2)) * den**-2.0 ; npy_cpow(-2) = 1/(den*den) */
                cx den2 = cpow2(den);
                cx inv_den2 = cdiv(cx_(1.0, 0.0), den2);
                accumulate[(size_t)k2 * FC + (size_t)i * nC + jc] = cmul(cmul(Yn_2, rsub(1.0, tanuh2)), inv_den2);
                /* FD:269-274 */
                cx kappaFactor = cmul(oTmp, csub(cmul(Y_2, Yn_), Yn_3));
                cx lead = cdiv(cx_(p, 0.0), cmul(cscale(un_, 2.0), den2));
                cx t1 = cmul(cmul(cscale(Yn_, 2.0), Y_), tanuh2);
                cx t2 = csub(cmul(kappaFactor, tanuh2), kappaFactor);
                cx t3 = cmul(csub(Y_2, Yn_2), tanuh);
                cx t4 = cscale(Yn_2, 2.0);
                if (g_exact_jacobian) { t3 = cmul(cadd(Y_2, Yn_2), tanuh); t4 = cx_(0.0, 0.0); }
                sens[(size_t)k2 * FC + (size_t)i * nC + jc] = cmul(lead, cadd(cadd(cadd(t1, t2), t3), t4));
            }
        }
    }
    {
        double p = par[nLayers];
        for (int i = 0; i < nF; ++i)
            for (int jc = 0; jc < nC; ++jc) /* FD:277-279 */
                sens[(size_t)(nLayers - 1) * FC + (size_t)i * nC + jc] =
                    cdiv(cx_(p, 0.0), cscale(w->un[IDX(w, nLayers, i, jc)], 2.0));
    }
    for (int k = 1; k < nLayers - 1; ++k) /* FD:281-285 prefix product */
        for (size_t q = 0; q < FC; ++q)
            accumulate[(size_t)k * FC + q] = cmul(accumulate[(size_t)k * FC + q], accumulate[(size_t)(k - 1) * FC + q]);

    for (int i = 0; i < nF; ++i)
        for (int jc = 0; jc < nC; ++jc) { /* FD:287-296 */
            size_t q = (size_t)i * nC + jc;
            u0[q] = w->un[IDX(w, 0, i, jc)];
            cx a0 = w->Yn[IDX(w, 0, i, jc)];
            cx a1 = w->Y[IDX(w, 1, i, jc)];
            cx a2 = cdiv(cx_(1.0, 0.0), cadd(a0, a1));
            cx a22 = cpow2(a2);
            cx s0 = cmul(cscale(a0, -2.0), a22);
            sens[q] = cmul(sens[q], s0); /* FD:298-300 */
            for (int k = 1; k < nLayers; ++k) /* FD:302-306 */
                sens[(size_t)k * FC + q] = cmul(sens[(size_t)k * FC + q], cmul(s0, accumulate[(size_t)(k - 1) * FC + q]));
        }
    free(accumulate);
}

/* FD:130-154 calcFdemSensitivity1D */
static void calcFdemSensitivity1D(coef_t *w, int nLayers, const double *frequencies,
                                  const double *lamda2, const double *par, const double *kappa,
                                  const double *perm, const double *thk, cx *u0, cx *sens)
{
    const int nF = w->nF, nC = w->nC;
    initCoefficients(w, nLayers, frequencies, lamda2, par, kappa, perm);
    if (nLayers == 1) {
        double p = par[1];
        for (int i = 0; i < nF; ++i)
            for (int jc = 0; jc < nC; ++jc) {
                size_t q = (size_t)i * nC + jc;
                cx s = cdiv(cx_(p, 0.0), cscale(w->un[IDX(w, 1, i, jc)], 2.0)); /* FD:140-142 */
                u0[q] = w->un[IDX(w, 0, i, jc)];
                cx a0 = w->Yn[IDX(w, 0, i, jc)];
                cx a1 = w->Y[IDX(w, 1, i, jc)];
                cx a2 = cdiv(cx_(1.0, 0.0), cadd(a0, a1));
                /* FD:150  -2.0 * a0 * sens * a2**2.0 */
                sens[q] = cmul(cmul(cscale(a0, -2.0), s), cpow2(a2));
            }
    } else {
        M1_1(w, nLayers, frequencies, thk, par, kappa, u0, sens);
    }
}

/* FD:410-438 Hzz */
static void Hzz(double tHeight, double rHeight, double moments, double separation,
                const cx *rTE, const cx *u0, const double *w0, const double *lamda0, cx *H, cx *H0)
{
    double hSum = rHeight + tHeight, hDiff = rHeight - tHeight;
    double a2 = moments / (4.0 * PI * separation);
    cx h = cx_(0, 0), h0 = cx_(0, 0);
    for (int jc = 0; jc < NC0; ++jc) {
        double w0_ = a2 * w0[jc];
        cx u0_ = u0[jc];
        double J0_ = lamda0[jc];
        cx a0 = cexp_(cscale(cneg(u0_), hSum));
        cx a1 = cdiv(cx_(pow(J0_, 3.0), 0.0), u0_);
        cx k = cmul(cadd(a0, cmul(rTE[jc], cexp_(cscale(u0_, hDiff)))), a1);
        h = cadd(h, cscale(k, w0_));
        k = cmul(a0, a1);
        h0 = cadd(h0, cscale(k, w0_));
    }
    *H = h; *H0 = h0;
}

/* FD:306-355 Hxx */
static void Hxx(double tHeight, double rHeight, double moments, double rx, double separation,
                const cx *rTEj0, const double *w0, const double *lamda0, const double *lamda02,
                const cx *rTEj1, const double *w1, const double *lamda1, cx *H, cx *H0)
{
    double hSum = rHeight + tHeight, hDiff = rHeight - tHeight;
    double r = 1.0 / separation;
    double c0 = -(moments / (4.0 * PI)) * r;
    double d0 = c0 * pow(rx * r, 2.0);
    double d1 = c0 * (r - ((2.0 * pow(rx, 2.0)) * pow(r, 3.0)));
    cx h = cx_(0, 0), h0 = cx_(0, 0);
    for (int jc = 0; jc < NC1; ++jc) {
        if (jc < NC0) {
            double w0_ = d0 * w0[jc];
            double J0_ = lamda0[jc];
            double a1 = lamda02[jc];
            double a0 = exp(-J0_ * hSum);
            cx k = cscale(rsub(a0, cscale(rTEj0[jc], exp(J0_ * hDiff))), a1);
            h = cadd(h, cscale(k, w0_));
            double k1 = a0 * a1;
            h0 = cx_(h0.re + k1 * w0_, h0.im);
        }
        double w1_ = d1 * w1[jc];
        double J1_ = lamda1[jc];
        double b0 = exp(-J1_ * hSum);
        cx k2 = cscale(rsub(b0, cscale(rTEj1[jc], exp(J1_ * hDiff))), J1_);
        h = cadd(h, cscale(k2, w1_));
        double k3 = b0 * J1_;
        h0 = cx_(h0.re + k3 * w1_, h0.im);
    }
    *H = h; *H0 = h0;
}

/* FD:358-381 Hxz (tid 3) and FD:384-408 Hzx (tid 7; exponent uses u1) */
static void Hxz(double tHeight, double rHeight, double moments, double rx, double separation,
                const cx *rTE1, const double *w1, const double *lamda1, const double *lamda12, cx *H, cx *H0)
{
    double hSum = rHeight + tHeight, hDiff = rHeight - tHeight;
    double d1 = (rx * moments) / (4.0 * PI * separation);
    cx h = cx_(0, 0), h0 = cx_(0, 0);
    for (int jc = 0; jc < NC1; ++jc) {
        double w1_ = d1 * w1[jc], J1_ = lamda1[jc], a1 = lamda12[jc];
        double b0 = exp(-J1_ * hSum);
        cx k = cscale(rsub(b0, cscale(rTE1[jc], exp(J1_ * hDiff))), a1);
        h = cadd(h, cscale(k, w1_));
        double kk = b0 * a1;
        h0 = cx_(h0.re + kk * w1_, h0.im);
    }
    *H = h; *H0 = h0;
}
static void Hzx(double tHeight, double rHeight, double moments, double rx, double separation,
                const cx *rTE1, const cx *u1, const double *w1, const double *lamda12, cx *H, cx *H0)
{
    double hSum = rHeight + tHeight, hDiff = rHeight - tHeight;
    double d1 = (rx * moments) / (4.0 * PI * separation);
    cx h = cx_(0, 0), h0 = cx_(0, 0);
    for (int jc = 0; jc < NC1; ++jc) {
        double w1_ = d1 * w1[jc], a1 = lamda12[jc];
        cx u1_ = u1[jc];
        cx b0 = cexp_(cscale(cneg(u1_), hSum));
        cx k = cscale(csub(b0, cmul(rTE1[jc], cexp_(cscale(u1_, hDiff)))), a1);
        h = cadd(h, cscale(k, w1_));
        k = cscale(b0, a1);
        h0 = cadd(h0, cscale(k, w1_));
    }
    *H = h; *H0 = h0;
}

static int uses_j0(const int32_t *tid, int nF)
{
    for (int i = 0; i < nF; ++i) { /* FD:47-50 */
        int t = tid[i];
        if (t == 1 || t == 2 || t == 4 || t == 5 || t == 9) return 1;
    }
    return 0;
}

static int hankel(int id, int i, const double *tHeight, const double *rHeight, const double *moments,
                  const double *rx, const double *separation, const double *w0, const double *lamda0,
                  const double *lamda02, const double *w1, const double *lamda1, const double *lamda12,
                  const cx *r0, const cx *u0j0, const cx *r1, const cx *u0j1, cx *H, cx *H0)
{
    switch (id) { /* FD:57-66 */
    case 1: Hxx(tHeight[i], rHeight[i], moments[i], rx[i], separation[i], r0 + (size_t)i * NC0, w0,
                lamda0 + (size_t)i * NC0, lamda02 + (size_t)i * NC0, r1 + (size_t)i * NC1, w1,
                lamda1 + (size_t)i * NC1, H, H0); return 0;
    case 3: Hxz(tHeight[i], rHeight[i], moments[i], rx[i], separation[i], r1 + (size_t)i * NC1, w1,
                lamda1 + (size_t)i * NC1, lamda12 + (size_t)i * NC1, H, H0); return 0;
    case 7: Hzx(tHeight[i], rHeight[i], moments[i], rx[i], separation[i], r1 + (size_t)i * NC1,
                u0j1 + (size_t)i * NC1, w1, lamda12 + (size_t)i * NC1, H, H0); return 0;
    case 9: Hzz(tHeight[i], rHeight[i], moments[i], separation[i], r0 + (size_t)i * NC0,
                u0j0 + (size_t)i * NC0, w0, lamda0 + (size_t)i * NC0, H, H0); return 0;
    default: return -1; /* reference leaves H uninitialised; we refuse */
    }
}

static void prepend_air(int nLayers, const double *src, double *dst)
{ dst[0] = 0.0; memcpy(dst + 1, src, (size_t)nLayers * sizeof(double)); }

/* FD:24-68 nbFdem1dfwd.  out = complex128[F] as interleaved (re, im). returns 0 / -1 (bad tid) */
int oracle_fdem1dfwd(int nF, int nLayers, const int32_t *tid, const double *frequencies,
                     const double *tHeight, const double *rHeight, const double *moments,
                     const double *rx, const double *separation, const double *w0,
                     const double *lamda0, const double *lamda02, const double *w1,
                     const double *lamda1, const double *lamda12, const double *scale,
                     const double *conductivity, const double *susceptibility,
                     const double *permeability, const double *thickness, double *out)
{
    const int nL1 = nLayers + 1;
    double *par = (double *)malloc(4 * (size_t)nL1 * sizeof(double));
    double *kappa = par + nL1, *perm = kappa + nL1, *thk = perm + nL1;
    prepend_air(nLayers, conductivity, par);
    prepend_air(nLayers, susceptibility, kappa);
    prepend_air(nLayers, permeability, perm);
    prepend_air(nLayers, thickness, thk);

    int useJ0 = uses_j0(tid, nF), rc = 0;
    coef_t c0 = {0}, c1 = {0};
    cx *r0 = NULL, *u0 = NULL, *r1, *u1;
    if (useJ0) {
        coef_alloc(&c0, nL1, nF, NC0);
        r0 = (cx *)malloc(2 * (size_t)nF * NC0 * sizeof(cx)); u0 = r0 + (size_t)nF * NC0;
        initCoefficients(&c0, nLayers, frequencies, lamda02, par, kappa, perm);
        M1_0(&c0, nLayers, thk, r0, u0);
    }
    coef_alloc(&c1, nL1, nF, NC1);
    r1 = (cx *)malloc(2 * (size_t)nF * NC1 * sizeof(cx)); u1 = r1 + (size_t)nF * NC1;
    initCoefficients(&c1, nLayers, frequencies, lamda12, par, kappa, perm);
    M1_0(&c1, nLayers, thk, r1, u1);

    for (int i = 0; i < nF; ++i) {
        cx H, H0;
        if (hankel(tid[i], i, tHeight, rHeight, moments, rx, separation, w0, lamda0, lamda02, w1,
                   lamda1, lamda12, r0, u0, r1, u1, &H, &H0)) { rc = -1; out[2 * i] = out[2 * i + 1] = NAN; continue; }
        cx q = cdiv(csub(H, H0), H0);         /* FD:68  1e6 * scale * ((H - H0) / H0) */
        double s = 1.e6 * scale[i];
        out[2 * i] = s * q.re; out[2 * i + 1] = s * q.im;
    }
    if (useJ0) { coef_free(&c0); free(r0); }
    coef_free(&c1); free(r1); free(par);
    return rc;
}

/* FD:71-121 nbFdem1dsen.  J = complex128[F, L] interleaved. */
int oracle_fdem1dsen(int nF, int nLayers, const int32_t *tid, const double *frequencies,
                     const double *tHeight, const double *rHeight, const double *moments,
                     const double *rx, const double *separation, const double *w0,
                     const double *lamda0, const double *lamda02, const double *w1,
                     const double *lamda1, const double *lamda12, const double *scale,
                     const double *conductivity, const double *susceptibility,
                     const double *permeability, const double *thickness, double *J)
{
    const int nL1 = nLayers + 1;
    double *par = (double *)malloc(4 * (size_t)nL1 * sizeof(double));
    double *kappa = par + nL1, *perm = kappa + nL1, *thk = perm + nL1;
    prepend_air(nLayers, conductivity, par);
    prepend_air(nLayers, susceptibility, kappa);
    prepend_air(nLayers, permeability, perm);
    prepend_air(nLayers, thickness, thk);

    int useJ0 = uses_j0(tid, nF), rc = 0;
    coef_t c0 = {0}, c1 = {0};
    cx *u0 = NULL, *s0 = NULL, *u1, *s1;
    if (useJ0) {
        coef_alloc(&c0, nL1, nF, NC0);
        u0 = (cx *)malloc((size_t)(1 + nLayers) * nF * NC0 * sizeof(cx)); s0 = u0 + (size_t)nF * NC0;
        calcFdemSensitivity1D(&c0, nLayers, frequencies, lamda02, par, kappa, perm, thk, u0, s0);
    }
    coef_alloc(&c1, nL1, nF, NC1);
    u1 = (cx *)malloc((size_t)(1 + nLayers) * nF * NC1 * sizeof(cx)); s1 = u1 + (size_t)nF * NC1;
    calcFdemSensitivity1D(&c1, nLayers, frequencies, lamda12, par, kappa, perm, thk, u1, s1);

    for (int k = 0; k < nLayers; ++k)
        for (int i = 0; i < nF; ++i) {
            cx dH, dH0;
            const cx *sk0 = s0 ? s0 + (size_t)k * nF * NC0 : NULL;
            const cx *sk1 = s1 + (size_t)k * nF * NC1;
            double *o = J + 2 * ((size_t)i * nLayers + k);
            if (hankel(tid[i], i, tHeight, rHeight, moments, rx, separation, w0, lamda0, lamda02, w1,
                       lamda1, lamda12, sk0, u0, sk1, u1, &dH, &dH0)) { rc = -1; o[0] = o[1] = NAN; continue; }
            /* FD:117-119  1e6 * scale[i] * (dH - dH0) / dH0 : ((s * d) / dH0) */
            cx q = cdiv(cscale(csub(dH, dH0), 1.e6 * scale[i]), dH0);
            o[0] = q.re; o[1] = q.im;
        }
    if (useJ0) { coef_free(&c0); free(u0); }
    coef_free(&c1); free(u1); free(par);
    return rc;
}

/*
 * DP:268-282 std, EmDataPoint.py:44-56 active, DP:200-214 deltaD, DP:502-525 data_misfit,
 * DP:491-500 + MVN:201-216 likelihood(log=True) with the diagonal covariance the
 * datapoint keeps (variance written at DP:279-280).  Closed form of MVN:209-216 for a
 * diagonal matrix: -(Na/2) ln(2 pi) - 0.5 * sum ln var_i - 0.5 * sum r_i^2 / var_i.
 * pred/obs: [N]; returns chi2, logL, n_active.
 */
void oracle_gauss_loglike(int N, const double *pred, const double *obs, double rel, double add,
                          double *std_out, double *chi2, double *logL, int *n_active)
{
    double s2 = 0.0, logdet = 0.0;
    int na = 0;
    for (int i = 0; i < N; ++i) {
        double variance = pow(rel * obs[i], 2.0) + pow(add, 2.0); /* DP:274 */
        double sd = sqrt(variance);
        if (std_out) std_out[i] = sd;
        int active = (obs[i] > 0.0) && !isnan(obs[i]);           /* EmDataPoint.py:54-56 */
        if (!active) continue;
        double r = (pred[i] - obs[i]) * (1.0 / sd);               /* DP:523-524 */
        s2 += r * r;
        logdet += log(variance);
        ++na;
    }
    *chi2 = s2;
    *logL = -(0.5 * na) * log(2.0 * PI) - 0.5 * logdet - 0.5 * s2;
    if (n_active) *n_active = na;
}

/* ------------------------------------------------------------------------- *
 * Batched driver used by tests and by bench.py's cpu_baseline ("port") leg:
 * B soundings that share one system (tables passed once), SoA-free simple
 * layout sigma[B, Lmax], thk[B, Lmax] (last finite thickness ignored: the
 * reference passes widths with last = inf and never reads it), nlayers[B].
 * Heights follow FD/fdem1d.py:31-32:  tH = z + Tx.z ; rH = -tH + Rx.z.
 * pred[B, 2F] = [Re(out_0..F-1), Im(out_0..F-1)] (FdemDataPoint.py:544-545).
 * OpenMP over soundings when compiled with -fopenmp.
 * ------------------------------------------------------------------------- */
int oracle_fdem_forward_loglike_batch(
    int B, int nF, int Lmax, const int32_t *tid, const double *frequencies, const double *tx_z,
    const double *rx_z, const double *tx_moment, const double *scale, const double *rx_off,
    const double *separation, const double *w0, const double *lamda0, const double *lamda02,
    const double *w1, const double *lamda1, const double *lamda12, const int32_t *nlayers,
    const double *sigma, const double *thk, const double *height, const double *obs,
    const double *rel, const double *add, double *pred, double *chi2, double *logL, int nthreads)
{
    int rc = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 16) reduction(| : rc)
#endif
    for (int b = 0; b < B; ++b) {
        double tH[64], rH[64], outc[128], zero[64];
        int L = nlayers[b];
        double *z0 = L <= 64 ? zero : (double *)calloc((size_t)L, sizeof(double));
        if (L <= 64) memset(zero, 0, sizeof(zero));
        for (int i = 0; i < nF; ++i) { tH[i] = height[b] + tx_z[i]; rH[i] = -tH[i] + rx_z[i]; }
        int r = oracle_fdem1dfwd(nF, L, tid, frequencies, tH, rH, tx_moment, rx_off, separation, w0, lamda0,
                                 lamda02, w1, lamda1, lamda12, scale, sigma + (size_t)b * Lmax, z0, z0,
                                 thk + (size_t)b * Lmax, outc);
        rc |= (r != 0);
        double *p = pred + (size_t)b * 2 * nF;
        for (int i = 0; i < nF; ++i) { p[i] = outc[2 * i]; p[nF + i] = outc[2 * i + 1]; }
        if (obs)
            oracle_gauss_loglike(2 * nF, p, obs + (size_t)b * 2 * nF, rel[b], add[b], NULL, chi2 + b, logL + b, NULL);
        if (L > 64) free(z0);
    }
    return rc ? -1 : 0;
}
