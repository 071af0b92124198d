// tests/host_emul/fdem_host_emul.cpp -- TEST-ONLY host build of the device math headers.
//
// Compiles geobipy_amd/csrc/gbp_math.h + gbp_fdem_point.h + gbp_fdem_tables.h with g++ so that the
// numerical scheme of the HIP kernels (projective admittance recursion, hand-written fp64
// sqrt/exp/sincos, H - H0 accumulated directly) can be checked against the oracle in the CPU test
// tier (`-m "not gpu"`).  It is NOT part of the product: geobipy_amd never loads it, and the GPU
// tests check the real kernels through the C ABI.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../geobipy_amd/csrc/gbp_fdem_tables.h"

namespace {
alignas(16) const double H_EXP2[64] = GBP_EXP2_64_LIST;
alignas(16) const double H_SINCOS[128] = GBP_SINCOS_64_LIST;
gbp::MathCtx host_ctx()
{
    using namespace gbp;
    MathCtx M;
    const MathK k = GBP_MATHK_INIT;
    M.k = k;
    M.e4_v = k.e4;
    M.s2_v = k.s2;
    M.c3_v = k.c3;
    M.exp2_64 = H_EXP2;
    M.sincos_64 = reinterpret_cast<const SinCos*>(H_SINCOS);
    return M;
}
}  // namespace

extern "C" {

// forward for B soundings, same argument meaning as gbp_fdem_system_create + gbp_fdem_forward (host pointers)
static double g_window_eps = 0.0, g_window_alt = 0.0;
static int g_last_npts = 0;
// test knob: the next emul_fdem_forward calls use the accuracy-budgeted abscissa window (eps <= 0: all abscissae)
void emul_set_window(double eps_ppm, double min_altitude) { g_window_eps = eps_ppm; g_window_alt = min_altitude; }
int emul_last_npoints() { return g_last_npts; }

int emul_fdem_forward(int nF, const int32_t* tid, const double* frequencies, const double* tx_z,
                      const double* rx_z, const double* tx_moment, const double* scale, const double* rx_off,
                      const double* separation, const double* w0, const double* lamda0, const double* w1,
                      const double* lamda1, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                      const double* thk, const double* height, double* pred)
{
    gbp::SystemTables t;
    const char* msg = "";
    int rc = gbp::build_system_tables(nF, tid, frequencies, tx_z, rx_z, tx_moment, scale, rx_off, separation, w0,
                                      lamda0, w1, lamda1, &t, &msg);
    if (rc != 0) return rc;
    gbp::window_system_tables(&t, g_window_eps, g_window_alt);
    g_last_npts = t.npts;
    const gbp::MathCtx M = host_ctx();
    const double* pts = t.soa.data();
    for (int b = 0; b < B; ++b) {
        const int L = nlayers[b];
        const double* sig = sigma + (size_t)b * Lmax;
        const double* th = thk + (size_t)b * Lmax;
        std::vector<gbp::LayerK> lay(L);
        std::vector<double> t2(L, 0.0);
        for (int k = 0; k < L - 1; ++k) t2[k] = -2.0 * th[k];
        for (int f = 0; f < nF; ++f) {
            const gbp::Channel& ch = t.chan[f];
            for (int k = 0; k < L; ++k) {
                const double bb = ch.wmu * sig[k];
                lay[k].b2 = bb * bb;
                lay[k].bc = bb * 0.70710678118654752440;
            }
            const double hD = ch.hd0 - 2.0 * height[b];
            double are = 0.0, aim = 0.0;
            for (int j = ch.off; j < ch.off + ch.npts; ++j) {
                const gbp::Point pt = gbp::load_point(pts, t.npts, j);
                gbp::cplx num, den;
                gbp::rte_num_den(M, pt.a, L, lay.data(), t2.data(), pt.u0, num, den);
                gbp::cplx term = gbp::hankel_term(M, num, den, pt.ue, hD, pt.coef);
                are += term.re;
                aim += term.im;
            }
            pred[(size_t)b * 2 * nF + f] = ch.g_re * are - ch.g_im * aim;
            pred[(size_t)b * 2 * nF + nF + f] = ch.g_re * aim + ch.g_im * are;
        }
    }
    return 0;
}

// Jacobian for B soundings: J[B, 2F, Lmax] (d pred / d ln sigma), exact = 0 reproduces the reference formula
int emul_fdem_sens(int nF, const int32_t* tid, const double* frequencies, const double* tx_z, const double* rx_z,
                   const double* tx_moment, const double* scale, const double* rx_off, const double* separation,
                   const double* w0, const double* lamda0, const double* w1, const double* lamda1, int B, int Lmax,
                   const int32_t* nlayers, const double* sigma, const double* thk, const double* height, int exact,
                   double* J)
{
    gbp::SystemTables t;
    const char* msg = "";
    int rc = gbp::build_system_tables(nF, tid, frequencies, tx_z, rx_z, tx_moment, scale, rx_off, separation, w0,
                                      lamda0, w1, lamda1, &t, &msg);
    if (rc != 0) return rc;
    const gbp::MathCtx M = host_ctx();
    const double* pts = t.soa.data();
    for (int b = 0; b < B; ++b) {
        const int L = nlayers[b];
        const double* sig = sigma + (size_t)b * Lmax;
        const double* th = thk + (size_t)b * Lmax;
        std::vector<gbp::LayerK> lay(L);
        std::vector<double> t2(L, 0.0);
        std::vector<gbp::cplx> D(L), acc(L);
        for (int k = 0; k < L - 1; ++k) t2[k] = -2.0 * th[k];
        for (int f = 0; f < nF; ++f) {
            const gbp::Channel& ch = t.chan[f];
            for (int k = 0; k < L; ++k) {
                const double bb = ch.wmu * sig[k];
                lay[k].b2 = bb * bb;
                lay[k].bc = bb * 0.70710678118654752440;
                acc[k] = gbp::mk(0.0, 0.0);
            }
            const double hD = ch.hd0 - 2.0 * height[b];
            for (int j = ch.off; j < ch.off + ch.npts; ++j) {
                const gbp::Point pt = gbp::load_point(pts, t.npts, j);
                const gbp::cplx E = gbp::cexp_neg(M, pt.ue.re * hD, pt.ue.im * hD);
                if (exact) gbp::sens_point<true>(M, pt.a, L, lay.data(), t2.data(), pt.u0, E * pt.coef, D.data(), 1);
                else gbp::sens_point<false>(M, pt.a, L, lay.data(), t2.data(), pt.u0, E * pt.coef, D.data(), 1);
                for (int m = 0; m < L; ++m) acc[m] = acc[m] + D[m];
            }
            for (int m = 0; m < Lmax; ++m) {
                double re = 0.0, im = 0.0;
                if (m < L) { re = ch.g_re * acc[m].re - ch.g_im * acc[m].im; im = ch.g_re * acc[m].im + ch.g_im * acc[m].re; }
                J[((size_t)b * 2 * nF + f) * Lmax + m] = re;
                J[((size_t)b * 2 * nF + nF + f) * Lmax + m] = im;
            }
        }
    }
    return 0;
}

// accuracy probes for the scalar kernels
void emul_exp_neg(int n, const double* x, double* y) { const gbp::MathCtx M = host_ctx(); for (int i = 0; i < n; ++i) y[i] = gbp::exp_neg(M, x[i]); }
void emul_sincos(int n, const double* x, double* s, double* c) { const gbp::MathCtx M = host_ctx(); for (int i = 0; i < n; ++i) gbp::sincos_tab(M, x[i], s[i], c[i]); }
void emul_csqrt(int n, const double* a, const double* b, double* re, double* im)
{
    for (int i = 0; i < n; ++i) { gbp::cplx z = gbp::csqrt_upper(a[i], b[i]); re[i] = z.re; im[i] = z.im; }
}
void emul_sqrt_rsqrt(int n, const double* x, double* g, double* y) { for (int i = 0; i < n; ++i) gbp::sqrt_rsqrt(x[i], g[i], y[i]); }
void emul_rcp(int n, const double* x, double* y) { for (int i = 0; i < n; ++i) y[i] = gbp::rcp(x[i]); }
}

// the sampler's logarithm and circular functions (gbp_math.h: log_pos, sincos_quadrant), element-wise, for the CPU-tier accuracy test
extern "C" void emul_log_pos(int n, const double* x, double* out)
{
    for (int i = 0; i < n; ++i) out[i] = gbp::log_pos(x[i]);
}
extern "C" void emul_sincos_quadrant(int n, const double* x, double* sn, double* cs)
{
    for (int i = 0; i < n; ++i) gbp::sincos_quadrant(x[i], sn[i], cs[i]);
}
