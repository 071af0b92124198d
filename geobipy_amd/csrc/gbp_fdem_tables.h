// gbp_fdem_tables.h -- host-side (CPU, once per acquisition system) construction of the per-abscissa
// tables the kernels read.  Pure C++ (no HIP) so that tests can build the same tables without a GPU.
//
// For every frequency f and every abscissa j the kernel needs
//   a              lambda_j^2 - w^2 mu0 eps0      (system/FdemSystem.py:67-109; FD:176-182)
//   u0             un[0] = sqrt(lambda^2 - w^2 mu0 eps0): the air layer of initCoefficients
//                  (forwardmodelling/Electromagnetic/FD/fdem1d_numba.py:172-185 with sigma = 0)
//   ue             exponent of the integrand: u0 (Hzz, Hzx) or lambda (Hxx, Hxz)
//   coef           everything of the Hankel integrand that does not depend on the earth model or the
//                  altitude: zz  +lambda^3/u0 * w0 * m/(4 pi r)            (FD:410-438)
//                            xx  -lambda0^2 d0 w0 (J0 part), -lambda1 d1 w1 (J1 part)  (FD:306-355)
//                            xz/zx  -lambda1^2 d1 w1                        (FD:358-408)
// and per frequency the free-space field H0 (FD:68 denominator), accumulated sequentially in the
// reference's order, folded into g = 1e6 * scale / H0.
#pragma once
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>

#include "../../include/geobipy_amd.h"
#include "gbp_fdem_point.h"

namespace gbp {

struct SystemTables {
    int nF = 0, npts = 0, max_pts = 0;
    std::vector<Channel> chan;
    std::vector<double> h0;   // (re, im) per frequency
    std::vector<double> soa;  // GBP_PT_FIELDS arrays of [npts]: a | u0re | u0im | cre | cim | uere | ueim
};

// returns GBP_OK or an error code; `msg` receives a static description on failure
inline int build_system_tables(int nF, const int32_t* tid, const double* frequencies, const double* tx_z,
                               const double* rx_z, const double* tx_moment, const double* scale,
                               const double* rx_off, const double* separation, const double* w0,
                               const double* lamda0, const double* w1, const double* lamda1, SystemTables* s,
                               const char** msg)
{
    typedef std::complex<double> zc;
    if (nF < 1 || nF > GBP_MAX_FREQ) { *msg = "nF must be in [1, 128]"; return GBP_ERR_INVALID_ARG; }
    if (!tid || !frequencies || !tx_z || !rx_z || !tx_moment || !scale || !rx_off || !separation || !w0 ||
        !lamda0 || !w1 || !lamda1) { *msg = "NULL system array"; return GBP_ERR_INVALID_ARG; }
    for (int f = 0; f < nF; ++f) {
        if (tid[f] != 1 && tid[f] != 3 && tid[f] != 7 && tid[f] != 9) {
            *msg = "tensor id outside {1,3,7,9}"; return GBP_ERR_UNSUPPORTED_TID;
        }
        if (!(frequencies[f] > 0.0) || !std::isfinite(frequencies[f]) || !(separation[f] > 0.0) ||
            !std::isfinite(separation[f])) {
            *msg = "frequency and separation must be finite and > 0"; return GBP_ERR_BAD_SYSTEM;
        }
    }
    s->nF = nF;
    s->chan.assign(nF, Channel());
    s->h0.assign(2 * (size_t)nF, 0.0);
    s->max_pts = 0;
    std::vector<double> pa, u0r, u0i, cre, cim, uer, uei;

    const double pi = 3.14159265358979323846;
    const double mu0 = 4.e-7 * pi;                        // fdem1d_numba.py:15
    const double c = 299792458.0;
    const double eps0 = 1.0 / (mu0 * std::pow(c, 2.0));   // fdem1d_numba.py:17


    for (int f = 0; f < nF; ++f) {
        Channel& ch = s->chan[f];
        const double omega = 2.0 * pi * frequencies[f];   // fdem1d_numba.py:166-168
        ch.wmu = omega * mu0;
        ch.w2me = (omega * eps0) * (omega * mu0);
        ch.hd0 = rx_z[f] - 2.0 * tx_z[f];
        ch.off = (int)pa.size();
        ch.tid = tid[f];
        const double tH = tx_z[f], rH = -tH + rx_z[f];     // fdem1d.py:31-32 at altitude 0
        const double hS = rH + tH;                         // independent of the altitude
        const double m = tx_moment[f], r = separation[f], rx = rx_off[f];
        const double* l0 = lamda0 + (size_t)f * GBP_NC0;
        const double* l1 = lamda1 + (size_t)f * GBP_NC1;
        zc H0(0.0, 0.0);
        // a = lambda^2 - w2me: real part of un^2 before the i*wmu*sigma term (FD:178-182); ue: the
        // exponent of the integrand, u0 for Hzz/Hzx and lambda for Hxx/Hxz
        auto push = [&](double l, zc u, zc cf) {
            pa.push_back(l * l - ch.w2me); u0r.push_back(u.real()); u0i.push_back(u.imag());
            cre.push_back(cf.real()); cim.push_back(cf.imag());
            uer.push_back(ch.real_exp ? l : u.real()); uei.push_back(ch.real_exp ? 0.0 : u.imag());
        };
        auto air_u = [&](double l) { return std::sqrt(zc(l * l - ch.w2me, 0.0)); };  // un[0], FD:182 with sigma = 0
        if (tid[f] == 9) {                                 // Hzz, fdem1d_numba.py:410-438
            ch.real_exp = 0;
            const double a2 = m / (4.0 * pi * r);
            for (int j = 0; j < GBP_NC0; ++j) {
                const double w_ = a2 * w0[j];
                const zc u = air_u(l0[j]);
                const zc a1 = zc(std::pow(l0[j], 3.0), 0.0) / u;
                H0 += (std::exp(-u * hS) * a1) * w_;
                push(l0[j], u, a1 * w_);
            }
        } else if (tid[f] == 1) {                          // Hxx, fdem1d_numba.py:306-355
            ch.real_exp = 1;
            const double ri = 1.0 / r;
            const double c0 = -(m / (4.0 * pi)) * ri;
            const double d0 = c0 * std::pow(rx * ri, 2.0);
            const double d1 = c0 * (ri - ((2.0 * std::pow(rx, 2.0)) * std::pow(ri, 3.0)));
            double h0 = 0.0;
            for (int j = 0; j < GBP_NC1; ++j) {            // free-space sum in the reference's interleaved order
                if (j < GBP_NC0) h0 += (std::exp(-l0[j] * hS) * (l0[j] * l0[j])) * (d0 * w0[j]);
                h0 += (std::exp(-l1[j] * hS) * l1[j]) * (d1 * w1[j]);
            }
            H0 = zc(h0, 0.0);
            for (int j = 0; j < GBP_NC0; ++j) push(l0[j], air_u(l0[j]), zc(-(l0[j] * l0[j]) * (d0 * w0[j]), 0.0));
            for (int j = 0; j < GBP_NC1; ++j) push(l1[j], air_u(l1[j]), zc(-l1[j] * (d1 * w1[j]), 0.0));
        } else {                                           // Hxz (3) / Hzx (7), fdem1d_numba.py:358-408
            ch.real_exp = (tid[f] == 3) ? 1 : 0;
            const double d1 = (rx * m) / (4.0 * pi * r);
            for (int j = 0; j < GBP_NC1; ++j) {
                const double w_ = d1 * w1[j];
                const double a1 = l1[j] * l1[j];
                const zc u = air_u(l1[j]);
                if (tid[f] == 3) H0 += zc((std::exp(-l1[j] * hS) * a1) * w_, 0.0);
                else             H0 += (std::exp(-u * hS) * a1) * w_;
                push(l1[j], u, zc(-a1 * w_, 0.0));
            }
        }
        ch.npts = (int)pa.size() - ch.off;
        if (ch.npts > s->max_pts) s->max_pts = ch.npts;
        const zc g = zc(1.e6 * scale[f], 0.0) / H0;        // fdem1d_numba.py:68
        ch.g_re = g.real();
        ch.g_im = g.imag();
        s->h0[2 * f] = H0.real();
        s->h0[2 * f + 1] = H0.imag();
    }
    s->npts = (int)pa.size();
    s->soa.clear();
    s->soa.reserve(GBP_PT_FIELDS * pa.size());
    for (const std::vector<double>* v : {&pa, &u0r, &u0i, &cre, &cim, &uer, &uei}) s->soa.insert(s->soa.end(), v->begin(), v->end());
    return GBP_OK;
}

// Accuracy-budgeted abscissa window.  |rTE| <= 1 for a passive layered earth (checked numerically over
// 2000 random models in the design notes), so abscissa j of frequency f contributes at most
//   T_j = |g_f| * |coef_j| * exp(Re(ue_j) * hDiff)   [ppm],   hDiff <= hd0_f - 2 * min_altitude
// to the output.  Dropping a prefix (small lambda: coef ~ lambda^2 w -> 0) and a suffix (large lambda:
// exp(-2 z lambda) -> 0) of each frequency's abscissae whose bounds sum to <= eps_ppm/2 each changes no output
// by more than eps_ppm.  At eps = 1e-12 ppm (1e-5 of the parity tolerance, 5000x below the reference's own
// arithmetic noise) roughly half of the 120 abscissae go.  At least 64 points per frequency are kept so that
// a 64-lane pass never spans more than two frequencies.  eps_ppm <= 0 leaves the tables untouched.
// relative = true: eps is a fraction of |g_f sum_j coef_j exp(ue_j hD)|, the value the sum takes for rTE = 1 (the image-source
// field: the inductive limit, the largest a frequency's output gets) at min_altitude -- for tables whose outputs are not ppm
// (the time-domain nodal spectra, gbp_hankel_system_create_raw).
inline void window_system_tables(SystemTables* s, double eps_ppm, double min_altitude, bool relative = false)
{
    if (!(eps_ppm > 0.0)) return;
    const int P = s->npts;
    std::vector<std::vector<double>> cols(GBP_PT_FIELDS);
    std::vector<Channel> chan = s->chan;
    int newP = 0, max_pts = 0;
    for (int f = 0; f < s->nF; ++f) {
        const Channel& ch = s->chan[f];
        const double hD = ch.hd0 - 2.0 * min_altitude;
        const double gabs = std::hypot(ch.g_re, ch.g_im);
        std::vector<double> T(ch.npts);
        for (int j = 0; j < ch.npts; ++j) {
            const int q = ch.off + j;
            const double cabs = std::hypot(s->soa[3 * (size_t)P + q], s->soa[4 * (size_t)P + q]);
            T[j] = gabs * cabs * std::exp(s->soa[5 * (size_t)P + q] * std::min(hD, 0.0));
        }
        double budget = 0.5 * eps_ppm;
        if (relative) {
            double sr = 0.0, si = 0.0, tsum = 0.0;
            for (int j = 0; j < ch.npts; ++j) {
                const int q = ch.off + j;
                const double d = std::exp(s->soa[5 * (size_t)P + q] * std::min(hD, 0.0));
                sr += s->soa[3 * (size_t)P + q] * d; si += s->soa[4 * (size_t)P + q] * d; tsum += T[j];
            }
            const double scale = gabs * std::hypot(sr, si);
            budget *= scale > 0.0 ? scale : tsum;
        }
        int lo = 0, hi = ch.npts;
        double acc = 0.0;
        while (lo < hi && acc + T[lo] <= budget) acc += T[lo++];
        acc = 0.0;
        while (hi > lo && acc + T[hi - 1] <= budget) acc += T[--hi];
        while (hi - lo < 64 && (lo > 0 || hi < ch.npts)) {   // keep >= 64 points: re-admit the larger neighbour
            if (lo > 0 && (hi >= ch.npts || T[lo - 1] >= T[hi])) --lo; else ++hi;
        }
        chan[f].off = newP;
        chan[f].npts = hi - lo;
        newP += hi - lo;
        if (hi - lo > max_pts) max_pts = hi - lo;
        for (int k = 0; k < GBP_PT_FIELDS; ++k)
            cols[k].insert(cols[k].end(), s->soa.begin() + (size_t)k * P + ch.off + lo,
                           s->soa.begin() + (size_t)k * P + ch.off + hi);
    }
    s->chan = chan;
    s->npts = newP;
    s->max_pts = max_pts;
    s->soa.clear();
    for (int k = 0; k < GBP_PT_FIELDS; ++k) s->soa.insert(s->soa.end(), cols[k].begin(), cols[k].end());
}

}  // namespace gbp
