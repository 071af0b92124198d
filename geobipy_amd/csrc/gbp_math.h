// gbp_math.h -- fp64 complex arithmetic and transcendental kernels for gfx950 (CDNA4).
//
// CDNA4 has no fp64 exp / sincos / sqrt at full precision in hardware: v_rsq_f64 / v_rcp_f64
// deliver ~2^-23 relative accuracy seeds at quarter rate, everything else is v_fma_f64 (4 cycles
// per wave64).  The forward solve is bound by exactly these sequences, so they are
// written here by hand with the minimum number of fp64 issues:
//   * sqrt_rsqrt : one v_rsq_f64 seed + ONE third-order (Halley) step, returning BOTH sqrt(x) and
//                  1/sqrt(x) (the complex sqrt needs both).
//   * exp_neg    : reduction by ln2/64, 64-entry 2^(j/64) table in LDS, degree-5 Taylor, v_ldexp_f64.
//   * sincos_tab : reduction by pi/32 (2-term FMA Cody-Waite, exact far beyond the |arg| < ~1500 the
//                  recursion can produce before exp underflows), 64-entry sin/cos table in LDS,
//                  degree-7/8 Taylor kernels and the angle-addition formulas -- no quadrant selects.
//   Polynomial coefficients live in SGPRs (struct MathK) so that each Horner step is ONE v_fma_f64.
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

#include "gbp_math_tables.h"

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

// v_rsq_f64 / v_rcp_f64 deliver seeds with a measured relative error of 5.2e-8 / 4.5e-8 on gfx950
// (tests/test_gpu_math.py).  One THIRD-order step takes them to full fp64:
//   rsqrt: e = 1 - x y^2,  y' = y (1 + e/2 + 3 e^2/8)   error ~ (5/16) e^3 = 4e-23   (5 issues)
//   rcp  : e = 1 - x y,    y' = y (1 + e + e^2)          error ~ e^3       = 9e-23   (3 issues)
// instead of two second-order (Newton / Goldschmidt) steps (7 and 4 issues).

// sqrt(x) for normal positive x, ~1 ulp.
GBP_HD double sqrt_fast(double x)
{
    double y = rsq_seed(x);
    double t = x * y;
    double e = __builtin_fma(-t, y, 1.0);
    double q = e * __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(t, q, t);
}

// g = sqrt(x), y = 1/sqrt(x) for normal positive x (x == 0 is NOT handled: callers on the hot path
// always have x > 0; the host-side table builder uses std::sqrt).
GBP_HD void sqrt_rsqrt(double x, double& g, double& yo)
{
    double y = rsq_seed(x);
    double t = x * y;
    double e = __builtin_fma(-t, y, 1.0);
    double q = e * __builtin_fma(0.375, e, 0.5);
    g = __builtin_fma(t, q, t);
    yo = __builtin_fma(y, q, y);
}

GBP_HD double rcp(double x)
{
    double y = rcp_seed(x);
    double e = __builtin_fma(-x, y, 1.0);
    double p = __builtin_fma(e, e, e);
    return __builtin_fma(y, p, y);
}

// principal sqrt(a + i b) for b > 0 (b = omega mu0 sigma in an earth layer), given the per-layer
// constants b2 = b^2 and bc = b / sqrt(2) (wave-uniform, precomputed once per layer and frequency):
//   m = |a + i b|,  s2 = m + |a| = 2 s,  sqrt(s) = sqrt(s2)/sqrt(2),  b/(2 sqrt(s)) = bc / sqrt(s2)
// DIRECT = true is for callers that know a >= 0, or a < 0 with b^2 >= 16 a^2 (a = lambda^2 - omega^2 mu0 eps0 is negative
// only below the displacement-current term, where |a| is far smaller than b in any conducting layer): then m + a keeps
// at least 3/4 of m, re = sqrt((m + a)/2) loses less than half a bit, and the branch swap -- two 64-bit selects per layer
// of the recursion -- is not needed.
template <bool DIRECT = false>
GBP_HD cplx csqrt_upper2(double a, double b2, double bc)
{
    const double RSQRT2 = 0.70710678118654752440;
    double m = sqrt_fast(__builtin_fma(a, a, b2));
    double g2, y2;
    if (DIRECT) {
        sqrt_rsqrt(m + a, g2, y2);
        return mk(g2 * RSQRT2, bc * y2);
    }
    sqrt_rsqrt(m + __builtin_fabs(a), g2, y2);
    double g = g2 * RSQRT2;
    double o = bc * y2;
    return (a >= 0.0) ? mk(g, o) : mk(o, g);
}
GBP_HD cplx csqrt_upper(double a, double b) { return csqrt_upper2(a, b * b, b * 0.70710678118654752440); }

// Scalar constants of the transcendental kernels.  On the device they are loaded once per wave from
// __constant__ memory with scalar loads and stay in SGPRs, so every Horner step is a single
// v_fma_f64 (vgpr, vgpr, sgpr) -- literal fp64 constants would cost one v_mov_b64 per step.
struct MathK {
    double inv_ln2_64, ln2_64_hi, ln2_64_lo, e5, e4, e3;             // exp:    64/ln2, ln2/64 (hi, lo), 1/120, 1/24, 1/6
    double inv_pi_32, pi_32_hi, pi_32_lo, s3, s2, s1, c4, c3, c2;    // sincos: 32/pi, pi/32 (hi, lo), Taylor coefficients
};
struct alignas(16) SinCos {
    double s, c;
};
// constants + lookup tables (LDS on the device: EXP2_64 is conflict-free for ds_read_b64 because its 64
// entries cover the 64 banks exactly once; SINCOS_64 is one ds_read_b128 per lane)
struct MathCtx {
    MathK k;
    // e4, s2, c3 again, but pinned in VGPRs: the first Horner step of each polynomial has TWO constant
    // operands (c_n * r + c_{n-1}) and a VALU instruction may read only one SGPR pair, so the compiler
    // would otherwise re-materialise one of them with a v_mov_b64 in every layer iteration.
    double e4_v, s2_v, c3_v;
    const double* exp2_64;   // 2^(j/64), j = 0..63
    const SinCos* sincos_64; // sin, cos of 2 pi j / 64
};

#define GBP_MATHK_INIT                                                                                    \
    {                                                                                                     \
        INV_LN2_64, LN2_64_HI, LN2_64_LO, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, INV_PI_32, PI_32_HI,          \
            PI_32_LO, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0, 1.0 / 40320.0, -1.0 / 720.0, 1.0 / 24.0     \
    }

// exp(x) for x <= 0 (flushes to 0 below ~-745; accurate for small positive x too).
//   x = k ln2/64 + r, |r| <= ln2/128:  exp(x) = 2^(k>>6) * 2^((k&63)/64) * (1 + r + ... + r^5/120)
// truncation r^6/720 < 3.6e-17; 16 VALU issues + one LDS read.
// round-to-nearest integer of x * inv via the 1.5 * 2^52 trick: t = fma(x, inv, MAGIC) holds the integer in
// its low mantissa bits (so the int32 is the low dword of t, no v_cvt) and kf = t - MAGIC; 2 issues instead of
// mul + v_rndne_f64 + v_cvt_i32_f64.  Valid for |x * inv| < 2^31.
GBP_HD double round_mul(double x, double inv, int& k)
{
    const double MAGIC = 6755399441055744.0;  // 1.5 * 2^52
    const double t = __builtin_fma(x, inv, MAGIC);
    k = (int)(unsigned)__builtin_bit_cast(unsigned long long, t);
    return t - MAGIC;
}

GBP_HD double exp_neg(const MathCtx& M, double x)
{
    double xx = __builtin_fmax(x, -800.0);
    int k;
    double kf = round_mul(xx, M.k.inv_ln2_64, k);
    double r = __builtin_fma(-kf, M.k.ln2_64_hi, xx);
    r = __builtin_fma(-kf, M.k.ln2_64_lo, r);
    double p = __builtin_fma(M.k.e5, r, M.e4_v);
    p = __builtin_fma(p, r, M.k.e3);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    double t = M.exp2_64[k & 63];
    return ldexp_i(t * p, k >> 6);
}

// sin and cos of x for |x| < ~2e8 (beyond that the caller's exp factor has long underflowed).
//   x = k pi/32 + r, |r| <= pi/64: angle addition with the tabulated sin/cos of k pi/32 and degree-7/8
//   Taylor kernels (truncation < 5e-18); no quadrant logic.  21 VALU issues + one LDS read.
GBP_HD void sincos_tab(const MathCtx& M, double x, double& s, double& c)
{
    int k;
    double kf = round_mul(x, M.k.inv_pi_32, k);
    double r = __builtin_fma(-kf, M.k.pi_32_hi, x);
    r = __builtin_fma(-kf, M.k.pi_32_lo, r);
    SinCos t = M.sincos_64[k & 63];
    double z = r * r;
    double ps = __builtin_fma(M.k.s3, z, M.s2_v);
    ps = __builtin_fma(ps, z, M.k.s1);
    double sr = __builtin_fma(ps * z, r, r);
    double pc = __builtin_fma(M.k.c4, z, M.c3_v);
    pc = __builtin_fma(pc, z, M.k.c2);
    double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
    s = __builtin_fma(t.s, cr, t.c * sr);
    c = __builtin_fma(t.c, cr, -(t.s * sr));
}

// exp(x + i t) for x <= 0
GBP_HD cplx cexp_neg(const MathCtx& M, double x, double t)
{
    double e = exp_neg(M, x);
    double s, c;
    sincos_tab(M, t, s, c);
    return mk(e * c, e * s);
}

// ------------------------------------------------------------------------------------------
// ln x and sin / cos for the SAMPLER's per-chain stages (gbp_rjmcmc.h; round 6).  Those stages are dependency chains of a single wave
// -- what they cost is the number of dependent instructions -- and 40 % of the instructions of an accept + proposal launch were the
// library's log (98 VALU instructions), cos and sincos (~150 - 250: its argument reduction serves any double) behind the generator's
// Box-Muller step, the error-level walks and the priors.  The arguments here are tame: positive normal numbers for the logarithm, an
// angle in [0, 2 pi] for the circular functions.  No tables (the packed stages have no LDS to spare), ~1 ulp.
//
// ln x, x > 0 finite (0 -> -inf, +inf -> +inf, NaN / negative -> NaN): x = 2^e m with m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1),
//   ln m = 2 atanh s = 2 s + 2 s^3 (1/3 + s^2 / 5 + ... + s^20 / 23),  |s| <= 0.1716: the first neglected term is 3e-19 of 2 s;
// s gets one correction step after the reciprocal (its error is the result's), e ln 2 is added in two parts.  ~34 VALU issues.
GBP_HD double frexp_mant(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_frexp_mant(x);  // v_frexp_mant_f64: [0.5, 1)
#else
    int e;
    return std::frexp(x, &e);
#endif
}
GBP_HD double log_pos(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;     // ln 2 = HI + LO, HI with 21 trailing zero bits
    double m = frexp_mant(x);
    int e = frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e -= low ? 1 : 0;
    const double f = m - 1.0;                      // exact
    const double d = 2.0 + f;                      // exact (m + 1 <= 2.42 carries every bit of m)
    const double inv = rcp(d);
    double s = f * inv;
    s = __builtin_fma(__builtin_fma(-s, d, f), inv, s);
    const double z = s * s;
    double p = __builtin_fma(z, 1.0 / 23.0, 1.0 / 21.0);
    p = __builtin_fma(p, z, 1.0 / 19.0);
    p = __builtin_fma(p, z, 1.0 / 17.0);
    p = __builtin_fma(p, z, 1.0 / 15.0);
    p = __builtin_fma(p, z, 1.0 / 13.0);
    p = __builtin_fma(p, z, 1.0 / 11.0);
    p = __builtin_fma(p, z, 1.0 / 9.0);
    p = __builtin_fma(p, z, 1.0 / 7.0);
    p = __builtin_fma(p, z, 1.0 / 5.0);
    p = __builtin_fma(p, z, 1.0 / 3.0);
    const double ef = (double)e;
    // e LN2_HI is exact for |e| <= 2^21 / ...: 1 074 needs 11 bits, HI has 21 spare
    double r = __builtin_fma(ef, LN2_LO, (s + s) * (z * p));
    r = r + (s + s);
    r = __builtin_fma(ef, LN2_HI, r);
    const double inf = __builtin_huge_val();
    r = x == 0.0 ? -inf : r;
    r = x == inf ? inf : r;
    return x < 0.0 ? __builtin_nan("") : r;        // (NaN in -> NaN out through the arithmetic)
}

// sin a and cos a for a in [0, 2 pi] (any |a| < ~1e5 in fact): a = k pi/2 + r, |r| <= pi/4 (two-term Cody-Waite with fma), the
// published minimax kernels of degree 13 / 14 on r (fdlibm's k_sin / k_cos coefficients), quadrant by selects.  ~40 VALU issues for both.
GBP_HD void sincos_quadrant(double a, double& sn, double& cs)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01, PIO2_HI = 1.57079632679489655800e+00, PIO2_LO = 6.12323399573676603587e-17;
    int k;
    const double kf = round_mul(a, TWO_OVER_PI, k);
    double r = __builtin_fma(-kf, PIO2_HI, a);
    r = __builtin_fma(-kf, PIO2_LO, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(ps, z, 2.75573137070700676789e-06);
    ps = __builtin_fma(ps, z, -1.98412698298579493134e-04);
    ps = __builtin_fma(ps, z, 8.33333333332248946124e-03);
    ps = __builtin_fma(ps, z, -1.66666666666666324348e-01);
    const double s = __builtin_fma(ps * z, r, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(pc, z, -2.75573143513906633035e-07);
    pc = __builtin_fma(pc, z, 2.48015872894767294178e-05);
    pc = __builtin_fma(pc, z, -1.38888888888741095749e-03);
    pc = __builtin_fma(pc, z, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * (z * pc));      // (k_cos's compensated form of 1 - z/2 + z^2 P)
    const bool swap = (k & 1) != 0;
    const double s0 = swap ? c : s, c0 = swap ? s : c;
    sn = (k & 2) ? -s0 : s0;
    cs = ((k + 1) & 2) ? -c0 : c0;
}

// a / b, no overflow guard: callers keep |b| in a safe range
GBP_HD cplx cdiv(cplx a, cplx b)
{
    double inv = rcp(__builtin_fma(b.re, b.re, b.im * b.im));
    cplx n = a * conj(b);
    return mk(n.re * inv, n.im * inv);
}

// 1 / b: the bits of cdiv(mk(1, 0), b) for finite b with b.re != 0 (there the products with the zero imaginary part of the numerator
// only add signed zeros), without the four issues spent on them
GBP_HD cplx crcp(cplx b)
{
    double inv = rcp(__builtin_fma(b.re, b.re, b.im * b.im));
    return mk(b.re * inv, -b.im * inv);
}

}  // namespace gbp
