// gbp_math.h -- fp64 complex arithmetic and transcendental kernels for gfx950 (CDNA4).
//
// CDNA4 has no fp64 exp / sincos / sqrt at full precision in hardware: v_rsq_f64 / v_rcp_f64
// deliver ~2^-23 relative accuracy seeds at quarter rate, everything else is v_fma_f64 (4 cycles
// per wave64).  The forward solve is bound by exactly these sequences, so they are
// written here by hand with the minimum number of fp64 issues:
//   * sqrt_rsqrt : one v_rsq_f64 seed + two coupled Goldschmidt steps + one residual correction,
//                  returning BOTH sqrt(x) and 1/(2 sqrt(x)) (the complex sqrt needs both).
//   * exp_neg    : Cody-Waite reduction by ln2 + degree-13 Horner + v_ldexp_f64.
//   * sincos_cw  : 3-term FMA Cody-Waite reduction by pi/2 (exact to |x| ~ 1e6, far beyond the
//                  |arg| < ~1500 the recursion can produce before exp underflows) + the classic
//                  minimax kernels on [-pi/4, pi/4].
// All functions are also compilable by a host C++ compiler (GBP_HD expands to `inline`) so that
// tests can check their accuracy against libm without a GPU; the product only ever runs them on
// the device.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define GBP_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define GBP_HD inline
#endif

namespace gbp {

struct cplx {
    double re, im;
};

GBP_HD cplx mk(double re, double im) { cplx z; z.re = re; z.im = im; return z; }
GBP_HD cplx operator+(cplx a, cplx b) { return mk(a.re + b.re, a.im + b.im); }
GBP_HD cplx operator-(cplx a, cplx b) { return mk(a.re - b.re, a.im - b.im); }
GBP_HD cplx operator*(cplx a, cplx b)
{
    return mk(__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re));
}
GBP_HD cplx operator*(cplx a, double s) { return mk(a.re * s, a.im * s); }
GBP_HD cplx conj(cplx a) { return mk(a.re, -a.im); }

GBP_HD double rsq_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(x);  // v_rsq_f64
#else
    return 1.0 / std::sqrt(x);
#endif
}
GBP_HD double rcp_seed(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);  // v_rcp_f64
#else
    return 1.0 / x;
#endif
}
GBP_HD double ldexp_i(double x, int e)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ldexp(x, e);  // v_ldexp_f64
#else
    return std::ldexp(x, e);
#endif
}
GBP_HD int frexp_exp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_frexp_exp(x);  // v_frexp_exp_i32_f64
#else
    int e;
    (void)std::frexp(x, &e);
    return e;
#endif
}

// g = sqrt(x), h = 1/(2 sqrt(x)) for normal positive x (x == 0 is NOT handled: callers on the
// hot path always have x = |un^2| > 0; the host-side table builder uses std::sqrt).
GBP_HD void sqrt_rsqrt(double x, double& g, double& h)
{
    double y = rsq_seed(x);
    g = x * y;
    h = 0.5 * y;
    double r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    r = __builtin_fma(-g, h, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
}

GBP_HD double rcp(double x)
{
    double y = rcp_seed(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    return y;
}

// principal sqrt(a + i b) for b >= 0 (b = omega mu0 sigma > 0 in every earth layer).
GBP_HD cplx csqrt_upper(double a, double b)
{
    double m, hm;
    sqrt_rsqrt(__builtin_fma(a, a, b * b), m, hm);
    double s = 0.5 * (m + __builtin_fabs(a));
    double g, h;
    sqrt_rsqrt(s, g, h);
    double o = b * h;  // b / (2 sqrt(s))
    return (a >= 0.0) ? mk(g, o) : mk(o, g);
}

// exp(x) for x <= 0 (any x < -745.2 flushes to 0; x > 0 small is still accurate up to ~700).
GBP_HD double exp_neg(double x)
{
    const double L2E = 1.4426950408889634074;
    const double LN2HI = 6.93147180369123816490e-01;
    const double LN2LO = 1.90821492927058770002e-10;
    double xx = x < -746.0 ? -746.0 : x;
    double kf = __builtin_rint(xx * L2E);
    double r = __builtin_fma(-kf, LN2HI, xx);
    r = __builtin_fma(-kf, LN2LO, r);
    double p = 1.6059043836821613e-10;                   // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);        // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);       // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);       // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);      // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);        // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);       // 1/7!
    p = __builtin_fma(p, r, 1.3888888888888889e-03);      // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);       // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);      // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);      // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    double v = ldexp_i(p, (int)kf);
    return x < -746.0 ? 0.0 : v;
}

// sin and cos of x, |x| < ~1e6.
GBP_HD void sincos_cw(double x, double& s, double& c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double P1 = 1.5707963267948966;
    const double P2 = 6.123233995736766e-17;
    const double P3 = -1.4973849048591698e-33;
    double kf = __builtin_rint(x * TWO_OVER_PI);
    double r = __builtin_fma(-kf, P1, x);
    r = __builtin_fma(-kf, P2, r);
    r = __builtin_fma(-kf, P3, r);
    double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = __builtin_fma(ps, z, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    double sr = __builtin_fma(ps * z, r, r);
    double pc = -1.13596475577881948265e-11;
    pc = __builtin_fma(pc, z, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    int k = (int)kf;
    double so = (k & 1) ? cr : sr;
    double co = (k & 1) ? sr : cr;
    s = (k & 2) ? -so : so;
    c = ((k + 1) & 2) ? -co : co;
}

// exp(x + i t) for x <= 0
GBP_HD cplx cexp_neg(double x, double t)
{
    double e = exp_neg(x);
    double s, c;
    sincos_cw(t, s, c);
    return mk(e * c, e * s);
}

// a / b, no overflow guard: callers keep |b| in a safe range
GBP_HD cplx cdiv(cplx a, cplx b)
{
    double inv = rcp(__builtin_fma(b.re, b.re, b.im * b.im));
    cplx n = a * conj(b);
    return mk(n.re * inv, n.im * inv);
}

}  // namespace gbp
