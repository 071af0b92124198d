// gbp_fdem_point.h -- one abscissa point of the 1-D layered-earth FDEM kernel.
//
// Replaces, for a single (frequency, abscissa) pair, the reference's
//   initCoefficients   forwardmodelling/Electromagnetic/FD/fdem1d_numba.py:157-191
//   M1_0 + cTanh       fdem1d_numba.py:194-219, 441-448
// and the integrand of Hzz/Hxx/Hxz/Hzx (fdem1d_numba.py:306-438).
//
// Re-derivation (not a transcription).  The reference recurses the surface admittance
//   Y_k = Yn_k (Y_{k+1} + Yn_k tanh(u_k t_k)) / (Yn_k + Y_{k+1} tanh(u_k t_k)),  Yn_k = u_k / (i w mu0)
// with one complex tanh (exp + division) and one complex division per layer.  With kappa == 0 for
// every layer (the wrapper passes zeros, FD/fdem1d.py:36-37) the factor 1/(i w mu0) is common to all
// Yn_k and cancels: with Yh = Y * (i w mu0) the recursion is homogeneous in u_k.  Writing
// Yh_{k+1} = N / D (projective form) and e = exp(-2 u_k t_k),
//   A = u_k D + N,   B = u_k D - N,
//   N' = u_k (A - e B),   D' = A + e B,
// which is exactly Y_k = u_k (1 - rho e)/(1 + rho e), rho = B / A, with NO division and no tanh:
// 3 complex multiplies + 4 complex adds per layer.  The reflection coefficient needs the only
// division:  rTE = (u_0 D - N) / (u_0 D + N).
//
// The Hankel integrand is then rTE * exp(ue * hDiff) * coef, where the free-space part of the
// reference's H (a0 * a1 * w) is dropped analytically: the reference forms H and H0 separately and
// subtracts (FD:68); we accumulate H - H0 directly, which removes the cancellation against the
// ~60x larger alternating filter terms of H0 (hard part 1).  H0 itself depends only on
// the acquisition system and is precomputed on the host once per system.
#pragma once
#include "gbp_math.h"

namespace gbp {

// Per-frequency constants (device copy lives in the system handle).
struct Channel {
    double wmu;         // omega * mu0                      (zn = i * wmu, FD:177)
    double w2me;        // (omega eps0) * (omega mu0)       (-Re(yn zn), FD:176-178)
    double hd0;         // rx_z - 2 tx_z : hDiff = rHeight - tHeight = hd0 - 2 * altitude (FD/fdem1d.py:31-32)
    double g_re, g_im;  // 1e6 * scale / H0                 (FD:68)
    int off, npts;      // slice of the point tables
    int real_exp;       // 1: exponent uses lambda (Hxx, Hxz), 0: u0 (Hzz, Hzx) -- folded into the ue table
    int tid;
};

#define GBP_PT_FIELDS 7  // a | u0.re | u0.im | coef.re | coef.im | ue.re | ue.im

// One abscissa point as the kernels read it (SoA in memory, see gbp_fdem_tables.h)
struct Point {
    double a;
    cplx u0, coef, ue;
};
GBP_HD Point load_point(const double* __restrict__ pts, int npts_total, int j)
{
    Point p;
    p.a = pts[j];
    p.u0 = mk(pts[(size_t)npts_total + j], pts[2 * (size_t)npts_total + j]);
    p.coef = mk(pts[3 * (size_t)npts_total + j], pts[4 * (size_t)npts_total + j]);
    p.ue = mk(pts[5 * (size_t)npts_total + j], pts[6 * (size_t)npts_total + j]);
    return p;
}

// The point of the same filter abscissa for a receiver at horizontal distance rho' = rho / s when the tables were built for rho (raw
// Hankel handles of DIPOLE sources, csrc/gbp_tdem.h: lam = base / rho, coef = lam^2 w / (4 pi rho) or lam w / (4 pi rho^2)): lam -> s lam,
// coef -> s^3 coef.  A sampled receiver position of the time-domain sampler is evaluated with its chain's table set and this scalar
// instead of tables of its own (gbp_td_moves.scale; wave-uniform).
GBP_HD void scale_point(Point& p, double s)
{
    const double s2 = s * s;
    p.a *= s2;
    p.u0.re *= s; p.u0.im *= s;
    p.ue.re *= s; p.ue.im *= s;
    const double s3 = s2 * s;
    p.coef.re *= s3; p.coef.im *= s3;
}

// Per-layer, per-frequency constants of one sounding (wave-uniform; the kernel keeps them in LDS and
// every lane reads them by broadcast, so no VALU issue is spent on uniform arithmetic in the layer loop).
struct alignas(16) LayerK {
    double b2;  // (omega mu0 sigma_k)^2
    double bc;  // omega mu0 sigma_k / sqrt(2)
};

// rTE numerator / denominator for one point.  a = lambda^2 - w2me; lay[k], t2[k] = -2 thk[k] for the
// sounding's L layers (t2[L-1] is never read -- the reference passes inf there).
template <bool DIRECT = false>   // see csqrt_upper2
GBP_HD void rte_num_den(const MathCtx& M, double a, int L, const LayerK* __restrict__ lay,
                        const double* __restrict__ t2, cplx u0, cplx& num, cplx& den)
{
    cplx N = csqrt_upper2<DIRECT>(a, lay[L - 1].b2, lay[L - 1].bc);  // basement: Yh_L = u_L
    cplx D = mk(1.0, 0.0);
    for (int k = L - 2; k >= 0; --k) {
        const LayerK lk = lay[k];
        const double tk = t2[k];
        cplx u = csqrt_upper2<DIRECT>(a, lk.b2, lk.bc);
        cplx e = cexp_neg(M, tk * u.re, tk * u.im);
        cplx uD = u * D;
        cplx A = uD + N, B = uD - N;
        cplx eB = e * B;
        N = u * (A - eB);
        D = A + eB;
        if ((k & 7) == 7) {  // keep |N|, |D| away from the fp64 range limits for deep models
            int s = -frexp_exp(__builtin_fmax(__builtin_fabs(D.re), __builtin_fabs(D.im)));
            N = mk(ldexp_i(N.re, s), ldexp_i(N.im, s));
            D = mk(ldexp_i(D.re, s), ldexp_i(D.im, s));
        }
    }
    cplx uD = u0 * D;
    num = uD - N;
    den = uD + N;
}

// One term of H - H0: rTE * exp(ue * hD) * coef
// `real_ue` (wave-uniform, decided by the caller): ue.im == 0 for every lane of the pass -- the abscissae above the free-space wavenumber,
// i.e. every point of an abscissa window at survey altitudes, and every point of a raw (time-domain) handle.  cexp_neg(x, 0) is
// (exp(x), +0) exactly (sincos_tab(0) = (0, 1) with no rounding) and a complex product with (e, +0) rounds like the two real products,
// so the short path returns the same bits for 25 VALU issues less per pass.
GBP_HD cplx hankel_term(const MathCtx& M, cplx num, cplx den, cplx ue, double hD, cplx coef, bool real_ue = false)
{
    // |den| is at most 8 layers of growth away from the last renormalisation (<= ~1e30, >= ~1e-50),
    // so |den|^2 is safely inside the fp64 range
    if (real_ue) return cdiv(num * (coef * exp_neg(M, ue.re * hD)), den);
    const cplx E = cexp_neg(M, ue.re * hD, ue.im * hD);
    return cdiv(num * (E * coef), den);
}

// ------------------------------------------------------------------------------------------
// Jacobian: d rTE / d ln(sigma_m) for every layer m of one abscissa point, times Q = E * coef
// (replaces calcFdemSensitivity1D + M1_1, fdem1d_numba.py:130-154, 222-303).
//
// Chain rule through the admittance recursion, in the kappa = 0 "hat" units of this file:
//   d rTE / d Yh_1       = -2 u_0 / (u_0 + Yh_1)^2                               (FD:292-295 "s0")
//   d Yh_k / d Yh_{k+1}  = u_k^2 (1 - T^2) / den^2 = 4 u_k^2 e / Dd^2            (FD:267 "accumulate")
//   sigma_k d u_k / d sigma_k = i b_k / (2 u_k),   b_k = omega mu0 sigma_k
//   d Yh_k / d u_k       = bracket / Dd^2,  Dd = u(1+e) + Yh'(1-e),  e = exp(-2 u t), with
//     exact     : (Y'^2 + u^2)(1 - e^2) + 2 u Y' (1-e)^2 - 4 t u e (Y'^2 - u^2)
//     reference : (Y'^2 - u^2)(1 - e^2) + 2 u^2 (1+e)^2 + 2 u Y' (1-e)^2 - 4 t u e (Y'^2 - u^2)
// The reference's bracket (FD:269-274: "(Y_2 - Yn_2) * tanuh + 2.0 * Yn_2") is NOT the derivative of its
// own forward recursion for layers above the half-space -- the correct term is (Y_2 + Yn_2) * tanuh; a
// finite-difference check of the reference shows O(1) relative errors there.
// Parity means reproducing the reference, so EXACT = false is the default; EXACT = true gives the true
// derivative.  The basement layer (d Yh_L / d u_L = 1) agrees in both.
//
// Bottom-up sweep with suffix propagation: D[m] holds d Yh_{k} / d ln sigma_m for all m >= k and is
// multiplied by "accumulate" as the sweep moves up (the reference stores accumulate[] and sens[] for all
// (layer, frequency, abscissa) and does a prefix product afterwards).  D lives in LDS on the device:
// element m of this lane is D[m * stride].
// Returns this point's term of the forward sum, rTE * Q = Q (u0 - Yh_1) / (u0 + Yh_1), which the sweep has at hand
// (fm_dlogc: prediction and Jacobian of the same model from one pass).
template <bool EXACT>
GBP_HD cplx sens_point(const MathCtx& M, double a, int L, const LayerK* __restrict__ lay,
                       const double* __restrict__ t2, cplx u0, cplx Q, cplx* D, int stride)
{
    const double RSQRT2 = 0.70710678118654752440;
    // i b / (2 u) = (b/2) (u.im + i u.re) / |u|^2
    auto ihb_over_u = [&](double bc, cplx u) {
        const double f = (bc * RSQRT2) * rcp(__builtin_fma(u.re, u.re, u.im * u.im));
        return mk(f * u.im, f * u.re);
    };
    cplx Y = csqrt_upper2(a, lay[L - 1].b2, lay[L - 1].bc);
    cplx fwd = mk(0.0, 0.0);
    D[(L - 1) * stride] = ihb_over_u(lay[L - 1].bc, Y);
    for (int k = L - 2; k >= 0; --k) {
        const LayerK lk = lay[k];
        const double tk = t2[k];  // -2 t_k
        const cplx u = csqrt_upper2(a, lk.b2, lk.bc);
        const cplx e = cexp_neg(M, tk * u.re, tk * u.im);
        // with S = u + Y', De = u - Y':  Dd = S + e De,  Nn = S - e De,  Yh = u Nn / Dd
        //   exact bracket = (1 - e)(S^2 + e De^2) - 2 tk u e S De        (= dYh/du * Dd^2)
        //   reference     = exact + 4 u^2 e (1 + e)                      (FD:269-274, see above)
        //   accumulate    = 4 u^2 e / Dd^2
        const cplx S = u + Y, De = u - Y;
        const cplx eDe = e * De;
        const cplx Dd = S + eDe, Nn = S - eDe;
        const cplx inv = crcp(Dd);
        const cplx inv2 = inv * inv;
        const cplx ue = u * e;
        const cplx P1 = mk(1.0 - e.re, -e.im) * (S * S + eDe * De);
        const cplx P2 = (ue * (S * De)) * (2.0 * tk);
        const cplx acc = ((u * ue) * inv2) * 4.0;
        cplx dY = (P1 - P2) * inv2;
        if (!EXACT) dY = dY + acc * mk(1.0 + e.re, e.im);
        cplx W = ihb_over_u(lk.bc, u) * dY;
        Y = (u * Nn) * inv;
        cplx fac = acc;
        if (k == 0) {  // top layer: fold d rTE / d Yh_1 and the Hankel factor Q into this last sweep
            const cplx i0 = crcp(u0 + Y);
            const cplx QQ = Q * ((u0 * (i0 * i0)) * -2.0);
            fac = acc * QQ;
            W = W * QQ;
            fwd = Q * ((u0 - Y) * i0);
        }
#pragma unroll 4
        for (int m = k + 1; m < L; ++m) D[m * stride] = D[m * stride] * fac;
        D[k * stride] = W;
    }
    if (L == 1) {  // half-space only: no layer loop ran
        const cplx i0 = crcp(u0 + Y);
        D[0] = D[0] * (Q * ((u0 * (i0 * i0)) * -2.0));
        fwd = Q * ((u0 - Y) * i0);
    }
    return fwd;
}

}  // namespace gbp
