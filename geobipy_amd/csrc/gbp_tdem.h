// gbp_tdem.h -- C-level time-domain system (include/geobipy_amd.h, "Time-domain systems"): the counterpart of
// gatdaem1d.TDAEMSystem(stmfile) / .forwardmodel(Geometry, Earth) as the reference calls it
// (forwardmodelling/Electromagnetic/TD/tdem1d.py:89-96, system/TdemSystem_GAAEM.py:8-35, system/Loop_pair.py:63-77).
// Host code only; the device work is gbp_fdem_forward (spline-node spectra on a raw Hankel handle) + k_td_apply (windows).
// Included at the end of gbp_fdem.hip.  Same pipeline and conventions as geobipy_amd/tdem.py (which the parity tests hold
// against the reference's CSV known answers): nodes at BaseFrequency * 10^(i / FrequenciesPerDecade), natural cubic spline
// of Re / Im in log10 f, periodic bipolar waveform by FFT over one period, first-order low-pass sections, area-under-curve
// or boxcar (+- 1e-7 s) windows, all folded into one matrix W[2 * n_nodes, n_windows] per system.
#pragma once
#include <mutex>

#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <utility>

namespace td {

typedef std::complex<double> zc;
constexpr double PI = 3.14159265358979323846264338327950288;
constexpr double MU0 = 4.0e-7 * PI;

// ---- FFT of any length (Bluestein on a radix-2 kernel); forward transform, unnormalised ---------------------------------
inline void fft_pow2(std::vector<zc>& a, bool inverse)
{
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = 2.0 * PI / (double)len * (inverse ? 1.0 : -1.0);
        std::vector<zc> w(len / 2);
        for (size_t k = 0; k < len / 2; ++k) w[k] = zc(std::cos(ang * (double)k), std::sin(ang * (double)k));
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const zc u = a[i + k], v = a[i + k + len / 2] * w[k];
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

inline std::vector<zc> fft_any(const std::vector<zc>& x)
{
    const size_t n = x.size();
    if ((n & (n - 1)) == 0) { std::vector<zc> a = x; fft_pow2(a, false); return a; }
    size_t m = 1;
    while (m < 2 * n - 1) m <<= 1;
    std::vector<zc> chirp(n), a(m, zc(0, 0)), b(m, zc(0, 0));
    for (size_t k = 0; k < n; ++k) {
        const unsigned long long k2 = ((unsigned long long)k * k) % (2ULL * n);        // k^2 mod 2n keeps the angle small
        const double ang = -PI * (double)k2 / (double)n;
        chirp[k] = zc(std::cos(ang), std::sin(ang));
    }
    for (size_t k = 0; k < n; ++k) a[k] = x[k] * chirp[k];
    b[0] = std::conj(chirp[0]);
    for (size_t k = 1; k < n; ++k) b[k] = b[m - k] = std::conj(chirp[k]);
    fft_pow2(a, false); fft_pow2(b, false);
    for (size_t k = 0; k < m; ++k) a[k] *= b[k];
    fft_pow2(a, true);
    std::vector<zc> out(n);
    for (size_t k = 0; k < n; ++k) out[k] = a[k] * chirp[k] / (double)m;
    return out;
}

inline double interp1(const std::vector<double>& xp, const std::vector<double>& fp, double x)      // numpy.interp
{
    if (x <= xp.front()) return fp.front();
    if (x >= xp.back()) return fp.back();
    size_t hi = std::upper_bound(xp.begin(), xp.end(), x) - xp.begin();
    const size_t lo = hi - 1;
    return fp[lo] + (fp[hi] - fp[lo]) * (x - xp[lo]) / (xp[hi] - xp[lo]);
}

// second derivatives of the natural cubic spline through (x, y)
inline std::vector<double> spline_m(const std::vector<double>& x, const std::vector<double>& y)
{
    const int n = (int)x.size();
    std::vector<double> m(n, 0.0), cp(n, 0.0), dp(n, 0.0);
    for (int i = 1; i < n - 1; ++i) {
        const double h0 = x[i] - x[i - 1], h1 = x[i + 1] - x[i];
        const double a = h0, b = 2.0 * (h0 + h1), c = h1, d = 6.0 * ((y[i + 1] - y[i]) / h1 - (y[i] - y[i - 1]) / h0);
        const double den = b - a * cp[i - 1];
        cp[i] = c / den;
        dp[i] = (d - a * dp[i - 1]) / den;
    }
    for (int i = n - 2; i >= 1; --i) m[i] = dp[i] - cp[i] * m[i + 1];
    return m;
}

inline double spline_eval(const std::vector<double>& x, const std::vector<double>& y, const std::vector<double>& m, double t)
{
    int hi = (int)(std::upper_bound(x.begin(), x.end(), t) - x.begin());
    hi = std::min(std::max(hi, 1), (int)x.size() - 1);
    const int lo = hi - 1;
    const double h = x[hi] - x[lo], A = (x[hi] - t) / h, B = (t - x[lo]) / h;
    return A * y[lo] + B * y[hi] + ((A * A * A - A) * m[lo] + (B * B * B - B) * m[hi]) * h * h / 6.0;
}

struct Stm {
    std::map<std::string, std::string> kv;
    std::vector<double> wt, wc, w_lo, w_hi;
    double num(const char* k, double dflt) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? dflt : std::atof(it->second.c_str());
    }
    std::string str(const char* k, const char* dflt) const
    {
        auto it = kv.find(k);
        return it == kv.end() ? std::string(dflt) : it->second;
    }
};

inline std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline bool parse_stm(const char* text, Stm* out)
{
    std::string all(text), line;
    int mode = 0;
    size_t pos = 0;
    while (pos <= all.size()) {
        size_t e = all.find('\n', pos);
        if (e == std::string::npos) e = all.size();
        line = all.substr(pos, e - pos);
        pos = e + 1;
        const size_t cm = line.find("//");
        if (cm != std::string::npos) line = line.substr(0, cm);
        line = trim(line);
        if (line.empty()) continue;
        if (line.find("WaveFormCurrent Begin") != std::string::npos) mode = 1;
        else if (line.find("WindowTimes Begin") != std::string::npos) mode = 2;
        else if ((line.size() >= 4 && line.compare(line.size() - 4, 4, " End") == 0) || line == "End") mode = 0;
        else if (mode != 0) {
            double a = 0, b = 0;
            if (std::sscanf(line.c_str(), "%lf %lf", &a, &b) != 2) return false;
            if (mode == 1) { out->wt.push_back(a); out->wc.push_back(b); } else { out->w_lo.push_back(a); out->w_hi.push_back(b); }
        } else {
            const size_t eq = line.find('=');
            if (eq != std::string::npos) out->kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
        }
    }
    return !out->wt.empty() && !out->w_lo.empty() && out->kv.count("BaseFrequency") && out->kv.count("WaveformDigitisingFrequency");
}

inline std::vector<double> numbers(const std::string& s)
{
    std::vector<double> v;
    const char* p = s.c_str();
    char* end = nullptr;
    for (double x = std::strtod(p, &end); p != end; x = std::strtod(p, &end)) { v.push_back(x); p = end; }
    return v;
}

}  // namespace td

// Geometry (geobipy_amd/tdem_geometry.py is the same statement in numpy; tests hold the two together): GA-AEM's frame x = flight
// direction, y = left, z = up; roll / pitch / yaw in degrees = right-handed rotations about x / y / z, body -> earth matrix
// R = Rz(yaw) Ry(pitch) Rx(roll).  In the frame whose x' axis points from the transmitter to the receiver the secondary field is a
// combination of five Hankel integrals per spline node -- the basis integrals B0L, B1L (vertical moment: the system's loop), B0, B1,
// BA (horizontal moment: a dipole) -- which are the "frequencies" of the raw Hankel handle; a row's geometry is the small real
// matrix that mixes their nodal spectra into those of the output components (gbp_td_mix).
#define GBP_TD_NBASIS 5
struct gbp_tdem_system {
    td::Stm stm;
    double f0 = 0, fs = 0, moment = 1, loop_radius = 0;
    double scale[3] = {0, 0, 0};                 // X / Y / ZOutputScaling
    int comp[3] = {0, 0, 0};                     // the output components in channel order: 0 = x, 1 = y, 2 = z
    bool dbdt = true, area = false;
    int n_samples = 0, n_windows = 0, n_components = 0, n_nodes = 0;
    std::vector<double> nodes, centres, W;       // W[2 n_nodes][n_windows] of ONE component (components share it)
    std::vector<double> w0, w1;                  // Hankel filter weights (J0: 120, J1: 140)
    std::vector<double> Wb;                      // block matrix for k_td_apply: [2 nc n_nodes][nc n_windows]
    double* d_Wb = nullptr;
    // One raw Hankel handle per table layout = (set of basis integrals, receiver on / off the transmitter's axis), holding one table
    // set per (horizontal distance, dz) seen so far (gbp_hankel_system_add_set); rows of a call pick theirs by index.
    struct Layout {
        gbp_fdem_system* h = nullptr;
        std::map<std::pair<double, double>, int> set_of;
        int basis[GBP_TD_NBASIS], n_basis = 0;
        int32_t *d_src = nullptr, *d_col = nullptr;      // index maps of gbp_td_mix for this layout
    };
    std::map<int, Layout> layouts;               // key: basis mask | on-axis << 8
    // Staging + device scratch of ONE call in flight (the entries are re-entrant -- any number of host threads may call
    // gbp_tdem_forward / gbp_tdem_fm_dlogc on one handle, each on a stream of its own): grown as needed and kept, so a warm handle
    // allocates nothing per call.  A call leases a workspace: the one it used last on the same stream (stream order makes the reuse
    // safe), else one whose last call has completed (`done`), else a new one.
    struct Work {
        hipStream_t q = nullptr;
        bool busy = false;
        hipEvent_t done = nullptr;
        std::vector<int32_t> h_set;
        std::vector<double> h_height, h_weights;
        double *d_height = nullptr, *d_nodal = nullptr, *d_weights = nullptr, *d_jnodal = nullptr;
        int32_t* d_set = nullptr;
        size_t cap_rows = 0, cap_set = 0, cap_nodal = 0, cap_weights = 0, cap_jnodal = 0;
    };
    std::vector<std::unique_ptr<Work>> works;
    // `mu` guards what calls share: the workspace pool, the lazily uploaded operator, and the layouts -- a call looks its table sets
    // up, grows them if its geometry is new (device tables are re-allocated then: hipFree waits for the kernels that still read the
    // old ones, all of which were enqueued under this lock) and enqueues its launches with the pointers it read, all under the lock;
    // the GPU work of concurrent callers overlaps, their host-side enqueue (tens of microseconds per call) takes turns.
    std::mutex mu;
    double hankel_eps = 1.0e-12;                 // per-sounding abscissa windows (gbp_tdem_system_set_hankel_eps); 0: all abscissae
};

namespace td {

constexpr int MAX_TABLE_SETS = 4096;             // table sets one layout keeps; a call that needs more is refused (bin the offsets)
constexpr double ON_AXIS_RHO = 1.0e-2;           // tdem_geometry.ON_AXIS_RHO

inline gbp_status build_operator(gbp_tdem_system* s, const char** msg)
{
    const Stm& m = s->stm;
    s->f0 = m.num("BaseFrequency", 0.0);
    s->fs = m.num("WaveformDigitisingFrequency", 0.0);
    if (!(s->f0 > 0.0) || !(s->fs > 2.0 * s->f0)) { *msg = "BaseFrequency / WaveformDigitisingFrequency missing or inconsistent"; return GBP_ERR_BAD_SYSTEM; }
    s->moment = m.num("NumberOfTurns", 1.0) * m.num("PeakCurrent", 1.0) * m.num("LoopArea", 1.0);
    s->loop_radius = m.num("ModellingLoopRadius", 0.0);
    s->scale[0] = m.num("XOutputScaling", 0.0);
    s->scale[1] = m.num("YOutputScaling", 0.0);
    s->scale[2] = m.num("ZOutputScaling", 0.0);
    s->n_components = 0;
    for (int k = 0; k < 3; ++k)
        if (s->scale[k] != 0.0) s->comp[s->n_components++] = k;
    if (s->n_components == 0) { *msg = "no output component has a non-zero scaling"; return GBP_ERR_BAD_SYSTEM; }
    const std::string ot = m.str("OutputType", "dB/dt");
    s->dbdt = ot.size() >= 2 && (ot[0] == 'd' || ot[0] == 'D') && (ot[1] == 'b' || ot[1] == 'B');
    const std::string ws = m.str("WindowWeightingScheme", "Boxcar");
    s->area = ws.size() >= 4 && (ws[0] == 'A' || ws[0] == 'a');
    const int N = s->n_samples = (int)std::llround(s->fs / s->f0);
    const int nw = s->n_windows = (int)m.w_lo.size();
    s->centres.resize(nw);
    for (int w = 0; w < nw; ++w) {
        s->centres[w] = 0.5 * (m.w_lo[w] + m.w_hi[w]);
        if (w > 0 && !(s->centres[w] > s->centres[w - 1])) { *msg = "Receiver window times must monotonically increase"; return GBP_ERR_BAD_SYSTEM; }
    }
    // spline nodes: f0 * 10^(i / fpd), one below the base frequency, up to the first at or above Nyquist
    const double fpd = m.num("FrequenciesPerDecade", 5.0);
    const int n_up = (int)std::ceil(std::log10(0.5 * s->fs / s->f0) * fpd - 1e-9) + 1;
    s->nodes.clear();
    for (int i = -1; i < n_up; ++i) s->nodes.push_back(s->f0 * std::pow(10.0, (double)i / fpd));
    const int n = s->n_nodes = (int)s->nodes.size();
    if (n * s->n_components > GBP_MAX_FREQ) { *msg = "too many spline nodes x components (limit 128)"; return GBP_ERR_BAD_SYSTEM; }
    // digitised current over one period (a table spanning half a period continues with opposite polarity)
    const double dt = 1.0 / s->fs, T = 1.0 / s->f0, t0 = m.wt.front();
    std::vector<zc> cur(N, zc(0, 0));
    if (std::fabs((m.wt.back() - m.wt.front()) - 0.5 * T) <= 2.0 / s->fs) {
        const int half = N / 2;
        for (int i = 0; i < half; ++i) {
            const double c = interp1(m.wt, m.wc, t0 + (double)i * dt);
            cur[i] = c; cur[half + i] = -c;
        }
    } else {
        for (int i = 0; i < N; ++i) cur[i] = interp1(m.wt, m.wc, t0 + (double)i * dt);
    }
    const std::vector<zc> I = fft_any(cur);
    const int nh = N / 2 + 1;
    // g_k = I_k * mu0 * moment * (-i 2 pi f_k for dB/dt) * prod (1 / (1 + i f_k / fc))^order ; g_0 = 0
    const std::vector<double> fc = numbers(m.str("CutOffFrequency", "")), ord = numbers(m.str("Order", ""));
    std::vector<zc> g(nh);
    for (int k = 0; k < nh; ++k) {
        const double f = (double)k * s->f0;
        zc fac(MU0 * s->moment, 0.0);
        if (s->dbdt) fac *= zc(0.0, -2.0 * PI * f);
        for (size_t q = 0; q < fc.size() && q < ord.size(); ++q) {
            const zc sec = 1.0 / zc(1.0, f / fc[q]);
            for (int r = 0; r < (int)ord[q]; ++r) fac *= sec;
        }
        g[k] = k == 0 ? zc(0, 0) : I[k] * fac;
    }
    // natural cubic spline basis of the nodes at the harmonics (log10 f, clipped to the node range)
    std::vector<double> x(n);
    for (int j = 0; j < n; ++j) x[j] = std::log10(s->nodes[j]);
    std::vector<double> S((size_t)nh * n, 0.0);
    for (int j = 0; j < n; ++j) {
        std::vector<double> e(n, 0.0);
        e[j] = 1.0;
        const std::vector<double> mm = spline_m(x, e);
        for (int k = 1; k < nh; ++k) {
            const double f = std::min(std::max((double)k * s->f0, s->nodes.front()), s->nodes.back());
            S[(size_t)k * n + j] = spline_eval(x, e, mm, std::log10(f));
        }
    }
    // windows: B_w,k = conj(fft(A_w)[k]);  sum_t A_w[t] x_t = (1/N) sum_k c_k Re(X_k B_w,k),  c = 1, 2, ..., 2, (1 if N even)
    s->W.assign((size_t)2 * n * nw, 0.0);
    for (int w = 0; w < nw; ++w) {
        const double a = m.w_lo[w], b = m.w_hi[w];
        std::vector<zc> A(N, zc(0, 0));
        if (s->area) {
            const int Q = 257;
            for (int q = 0; q < Q; ++q) {
                const double tq = a + (b - a) * (double)q / (double)(Q - 1);
                double wq = (b - a) / (double)(Q - 1);
                if (q == 0 || q == Q - 1) wq *= 0.5;
                const double p = (tq - t0) / dt;
                int i0 = (int)std::floor(p);
                i0 = std::min(std::max(i0, 0), N - 2);
                const double fr = p - (double)i0;
                A[i0] += wq * (1.0 - fr) / (b - a);
                A[i0 + 1] += wq * fr / (b - a);
            }
        } else {
            int cnt = 0;
            for (int i = 0; i < N; ++i) {
                const double t = t0 + (double)i * dt;
                if (t >= a - 1.0e-7 && t <= b + 1.0e-7) ++cnt;
            }
            if (cnt == 0) { *msg = "a Boxcar window holds no sample"; return GBP_ERR_BAD_SYSTEM; }
            for (int i = 0; i < N; ++i) {
                const double t = t0 + (double)i * dt;
                if (t >= a - 1.0e-7 && t <= b + 1.0e-7) A[i] = 1.0 / (double)cnt;
            }
        }
        const std::vector<zc> FA = fft_any(A);
        for (int k = 1; k < nh; ++k) {
            const double ck = (N % 2 == 0 && k == nh - 1) ? 1.0 : 2.0;
            const zc gb = g[k] * std::conj(FA[k]) * (ck / (double)N);
            const double* Sk = &S[(size_t)k * n];
            for (int j = 0; j < n; ++j) {
                s->W[(size_t)j * nw + w] += Sk[j] * gb.real();
                s->W[(size_t)(n + j) * nw + w] -= Sk[j] * gb.imag();
            }
        }
    }
    // block matrix over the components: nodal layout [Re(comp 0 nodes), Re(comp 1 nodes), Im(comp 0 ...), Im(comp 1 ...)]
    const int nc = s->n_components;
    s->Wb.assign((size_t)2 * nc * n * nc * nw, 0.0);
    const size_t ld = (size_t)nc * nw;
    for (int c = 0; c < nc; ++c)
        for (int j = 0; j < n; ++j)
            for (int w = 0; w < nw; ++w) {
                s->Wb[((size_t)c * n + j) * ld + (size_t)c * nw + w] = s->W[(size_t)j * nw + w];
                s->Wb[((size_t)nc * n + (size_t)c * n + j) * ld + (size_t)c * nw + w] = s->W[(size_t)(n + j) * nw + w];
            }
    return GBP_OK;
}

// body -> earth rotation, degrees (tdem_geometry.rotation)
inline void rotation(double roll, double pitch, double yaw, double R[3][3])
{
    const double d = PI / 180.0;
    const double cr = std::cos(roll * d), sr = std::sin(roll * d), cp = std::cos(pitch * d), sp = std::sin(pitch * d);
    const double cy = std::cos(yaw * d), sy = std::sin(yaw * d);
    R[0][0] = cy * cp; R[0][1] = cy * sp * sr - sy * cr; R[0][2] = cy * sp * cr + sy * sr;
    R[1][0] = sy * cp; R[1][1] = sy * sp * sr + cy * cr; R[1][2] = sy * sp * cr - cy * sr;
    R[2][0] = -sp;     R[2][1] = cp * sr;                R[2][2] = cp * cr;
}

// w[3][5]: field along the receiver's axis k = sum_i w[k][i] * basis integral i, before output sign and scaling
// (tdem_geometry.basis_weights: the two must agree)
inline void basis_weights(const double* gm, bool loop, double w[3][GBP_TD_NBASIS])
{
    const double rho = std::hypot(gm[4], gm[5]);
    const bool on = rho == 0.0;
    const double c = on ? 1.0 : gm[4] / rho, s = on ? 0.0 : gm[5] / rho;
    double Rt[3][3], Rr[3][3];
    rotation(gm[1], gm[2], gm[3], Rt);
    rotation(gm[7], gm[8], gm[9], Rr);
    const double Rz[3][3] = {{c, -s, 0.0}, {s, c, 0.0}, {0.0, 0.0, 1.0}};
    double u[3], V[3][3];
    for (int i = 0; i < 3; ++i) {          // u = Rz^T (R_tx z^),  V = R_rx^T Rz
        u[i] = 0.0;
        for (int j = 0; j < 3; ++j) u[i] += Rz[j][i] * Rt[j][2];
        for (int k = 0; k < 3; ++k) {
            V[i][k] = 0.0;
            for (int j = 0; j < 3; ++j) V[i][k] += Rr[j][i] * Rz[j][k];
        }
    }
    for (int k = 0; k < 3; ++k) {
        w[k][0] = V[k][2] * u[2];
        w[k][1] = V[k][0] * u[2];
        w[k][2] = V[k][0] * u[0];
        w[k][3] = -V[k][2] * u[0];
        w[k][4] = -V[k][0] * u[0] + V[k][1] * u[1];
        if (!loop) { w[k][0] += w[k][2]; w[k][1] += w[k][3]; w[k][2] = w[k][3] = 0.0; }
        if (on) { w[k][2] += 0.5 * w[k][4]; w[k][1] = w[k][3] = w[k][4] = 0.0; }
    }
}

// raw Hankel tables of one table set (rho, dz) for the basis integrals of a layout (TdemSystem.hankel_tables, all abscissae)
struct RawTables {
    std::vector<int32_t> npts;
    std::vector<double> wmu, hd0, g, tables;
};

inline gbp_status build_tables(const gbp_tdem_system* s, double rho, double dz, const int* basis, int n_basis, RawTables* out)
{
    const double a = s->loop_radius, k4 = 1.0 / (4.0 * PI);
    const bool on_axis = rho == 0.0;
    if (on_axis && !(a > 0.0)) return fail(GBP_ERR_BAD_SYSTEM, "a receiver on the transmitter axis needs a finite ModellingLoopRadius%s");
    std::vector<double> cols[GBP_PT_FIELDS];
    auto base0 = [](int j) { return std::pow(10.0, -8.3885 + 0.0904226468670 * (double)j); };        // FdemSystem.py:67-83
    auto base1 = [](int j) { return std::pow(10.0, -7.91001919 + 0.087967143957 * (double)j); };      // :85-101
    auto srcz = [&](double lam) { return a > 0.0 ? lam * std::cyl_bessel_j(1.0, lam * a) / (2.0 * PI * a) : lam * lam * k4; };
    for (int q = 0; q < n_basis; ++q) {
        const int i = basis[q];
        bool use_j1;
        double div;
        if (on_axis) {
            if (i == 0) { use_j1 = true; div = a; }                 // B0L: J0(0) = 1, the loop's own J1(lam a) is the filter kernel
            else if (i == 2) { use_j1 = false; div = ON_AXIS_RHO; } // B0 (and BA -> B0 / 2) a hair off the axis
            else return fail(GBP_ERR_INVALID_ARG, "basis integral vanishes on the axis%s");
        } else {
            use_j1 = i == 1 || i == 3 || i == 4;
            div = rho;
        }
        const int np = use_j1 ? GBP_NC1 : GBP_NC0;
        std::vector<double> lam(np), coef(np);
        for (int j = 0; j < np; ++j) {
            lam[j] = (use_j1 ? base1(j) : base0(j)) / div;
            const double w = (use_j1 ? s->w1[j] : s->w0[j]) / div;
            double src;
            if (on_axis) src = i == 0 ? lam[j] / (2.0 * PI * a) : lam[j] * lam[j] * k4;
            else if (i <= 1) src = srcz(lam[j]);
            else if (i <= 3) src = lam[j] * lam[j] * k4;
            else src = lam[j] * k4 / rho;
            coef[j] = src * w;
        }
        for (int nd = 0; nd < s->n_nodes; ++nd) {
            out->npts.push_back(np);
            out->wmu.push_back(2.0 * PI * s->nodes[nd] * MU0);
            out->hd0.push_back(-dz);
            out->g.push_back(1.0); out->g.push_back(0.0);
            for (int j = 0; j < np; ++j) {
                cols[0].push_back(lam[j] * lam[j]); cols[1].push_back(lam[j]); cols[2].push_back(0.0); cols[3].push_back(coef[j]);
                cols[4].push_back(0.0); cols[5].push_back(lam[j]); cols[6].push_back(0.0);
            }
        }
    }
    for (int f = 0; f < GBP_PT_FIELDS; ++f) out->tables.insert(out->tables.end(), cols[f].begin(), cols[f].end());
    return GBP_OK;
}

template <class T>
inline gbp_status grow(T** p, size_t* cap, size_t need)
{
    if (need <= *cap) return GBP_OK;
    if (*p) { (void)hipDeviceSynchronize(); (void)hipFree(*p); *p = nullptr; *cap = 0; }   // (work of earlier calls may still read it)
    const size_t n = need + need / 4;
    const hipError_t e = hipMalloc((void**)p, sizeof(T) * n);
    if (e != hipSuccess) return fail(GBP_ERR_HIP, "scratch allocation failed: %s", hipGetErrorString(e));
    *cap = n;
    return GBP_OK;
}

// windows (and, with J, their derivatives with respect to ln sigma) of B soundings: the body of gbp_tdem_forward / _fm_dlogc
inline gbp_status run(gbp_tdem_system* s, int B, const double* geometry, int Lmax, const int32_t* nlayers, const double* sigma,
                      const double* thk, double* out, double* J, void* stream)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (B < 0 || Lmax < 1) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0 and Lmax >= 1%s");
    if (B == 0) return GBP_OK;
    if (!geometry || !nlayers || !sigma || !thk || !out) return fail(GBP_ERR_INVALID_ARG, "NULL pointer%s");
    for (int b = 0; b < B; ++b) {
        const double* gm = geometry + (size_t)b * 10;
        for (int i = 0; i < 10; ++i)
            if (!std::isfinite(gm[i])) return fail(GBP_ERR_INVALID_ARG, "geometry must be finite%s");
        if (!(gm[0] >= 0.0)) return fail(GBP_ERR_INVALID_ARG, "transmitter height must be >= 0%s");
    }
    const hipStream_t q = (hipStream_t)stream;
    const bool loop = s->loop_radius > 0.0;
    const int nc = s->n_components, n = s->n_nodes, N = nc * s->n_windows, n_out = 2 * nc * n;
    // this call's workspace (see gbp_tdem_system::Work); given back -- with an event on the caller's stream -- on every way out
    struct Lease {
        gbp_tdem_system* s; gbp_tdem_system::Work* w; hipStream_t q;
        ~Lease()
        {
            if (!w) return;
            if (w->done) (void)hipEventRecord(w->done, q);
            std::lock_guard<std::mutex> lk(s->mu);
            w->busy = false;
        }
    } lease{s, nullptr, q};
    try {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            if (s->d_Wb == nullptr) {
                hipError_t e = hipMalloc((void**)&s->d_Wb, sizeof(double) * s->Wb.size());
                if (e == hipSuccess) e = hipMemcpy(s->d_Wb, s->Wb.data(), sizeof(double) * s->Wb.size(), hipMemcpyHostToDevice);
                if (e != hipSuccess) { if (s->d_Wb) (void)hipFree(s->d_Wb); s->d_Wb = nullptr; return fail(GBP_ERR_HIP, "window operator upload failed: %s", hipGetErrorString(e)); }
            }
            for (auto& w : s->works)
                if (!w->busy && w->q == q) { lease.w = w.get(); break; }
            if (!lease.w)
                for (auto& w : s->works)
                    if (!w->busy && (w->done == nullptr || hipEventQuery(w->done) == hipSuccess)) { lease.w = w.get(); break; }
            (void)hipGetLastError();                          // (hipEventQuery's hipErrorNotReady is not an error of this call)
            if (!lease.w) {
                s->works.emplace_back(new gbp_tdem_system::Work());
                lease.w = s->works.back().get();
                if (hipEventCreateWithFlags(&lease.w->done, hipEventDisableTiming) != hipSuccess) lease.w->done = nullptr;
            }
            lease.w->busy = true;
            lease.w->q = q;
        }
        gbp_tdem_system::Work* const ws = lease.w;
        // A workspace leased again on the SAME stream may still be the source of the previous call's pageable hipMemcpyAsync's (the
        // last run of a call is not synchronised): wait for that call's end before its host staging vectors are resized / overwritten.
        // (only hipErrorNotReady is cleared: a real asynchronous failure -- of the wait, or left by the previous call's kernels -- ends this call)
        if (ws->done != nullptr) {
            hipError_t qe = hipEventQuery(ws->done);
            if (qe == hipErrorNotReady) { (void)hipGetLastError(); qe = hipEventSynchronize(ws->done); }
            if (qe != hipSuccess) return fail(GBP_ERR_HIP, "the previous call on this workspace failed: %s", hipGetErrorString(qe));
        }
        ws->h_height.resize(B);
        ws->h_set.resize(B);
        for (int b = 0; b < B; ++b) ws->h_height[b] = geometry[(size_t)b * 10];
        gbp_status st;
        if ((st = grow(&ws->d_height, &ws->cap_rows, (size_t)B)) != GBP_OK) return st;
        if ((st = grow(&ws->d_set, &ws->cap_set, (size_t)B)) != GBP_OK) return st;
        GBP_HIP(hipMemcpyAsync(ws->d_height, ws->h_height.data(), sizeof(double) * (size_t)B, hipMemcpyHostToDevice, q));
        auto on_axis = [&](int b) { return geometry[(size_t)b * 10 + 4] == 0.0 && geometry[(size_t)b * 10 + 5] == 0.0; };
        // Jacobians go through the window stage in chunks of rows so that the nodal Jacobian scratch stays bounded
        const int chunk_rows = J != nullptr ? 2048 : B;
        for (int b0 = 0; b0 < B;) {
            const bool ax = on_axis(b0);
            int b1 = b0 + 1;
            while (b1 < B && on_axis(b1) == ax && b1 - b0 < chunk_rows) ++b1;
            const int nr = b1 - b0;
            // the basis integrals some row of this run needs, and every row's weights for all five
            std::vector<double> wall((size_t)nr * nc * GBP_TD_NBASIS);
            bool used[GBP_TD_NBASIS] = {true, false, false, false, false};
            for (int b = b0; b < b1; ++b) {
                double w[3][GBP_TD_NBASIS];
                basis_weights(geometry + (size_t)b * 10, loop, w);
                for (int c = 0; c < nc; ++c) {
                    const int k = s->comp[c];
                    const double f = (k == 2 ? 1.0 : -1.0) * s->scale[k];      // the reference negates GA-AEM's z (TdemDataPoint.py:1013-1015)
                    for (int i = 0; i < GBP_TD_NBASIS; ++i) {
                        const double v = f * w[k][i];
                        wall[((size_t)(b - b0) * nc + c) * GBP_TD_NBASIS + i] = v;
                        if (v != 0.0) used[i] = true;
                    }
                }
            }
            int mask = 0;
            for (int i = 0; i < GBP_TD_NBASIS; ++i) if (used[i]) mask |= 1 << i;
            std::unique_lock<std::mutex> shared(s->mu);       // layouts, their table sets and this run's launches (released at the end of the turn)
            gbp_tdem_system::Layout& ly = s->layouts[mask | ((int)ax << 8)];
            if (ly.n_basis == 0)
                for (int i = 0; i < GBP_TD_NBASIS; ++i) if (used[i]) ly.basis[ly.n_basis++] = i;
            const int nb = ly.n_basis, nF_in = nb * n, n_in = 2 * nF_in, n_w = nc * nb;
            if (nF_in > GBP_MAX_FREQ) return fail(GBP_ERR_INVALID_ARG, "spline nodes x basis integrals of this geometry exceed the limit of 128 frequencies%s");
            // table sets: one per distinct (rho, dz); consecutive rows with the same key share one look-up
            bool grown = false;
            double lo = ws->h_height[b0], hi = lo, last_rho = -1.0, last_dz = 0.0;
            int last_set = -1;
            for (int b = b0; b < b1; ++b) {
                const double* gm = geometry + (size_t)b * 10;
                lo = std::min(lo, ws->h_height[b]); hi = std::max(hi, ws->h_height[b]);
                const double rho = std::hypot(gm[4], gm[5]);
                if (last_set < 0 || rho != last_rho || gm[6] != last_dz) {
                    const auto key = std::make_pair(rho, gm[6]);
                    auto it = ly.set_of.find(key);
                    if (it == ly.set_of.end()) {
                        if ((int)ly.set_of.size() >= MAX_TABLE_SETS)
                            return fail(GBP_ERR_INVALID_ARG, "more than 4096 distinct (horizontal distance, dz) receiver offsets on one handle: bin the offsets (e.g. to 0.1 m)%s");
                        RawTables t;
                        st = build_tables(s, rho, gm[6], ly.basis, nb, &t);
                        if (st != GBP_OK) return st;
                        if (ly.h == nullptr)
                            st = gbp_hankel_system_create_raw((int)t.npts.size(), t.npts.data(), t.wmu.data(), t.hd0.data(), t.g.data(), t.tables.data(), &ly.h);
                        else
                            st = gbp_hankel_system_add_set(ly.h, t.hd0.data(), t.tables.data());
                        if (st != GBP_OK) return st;
                        it = ly.set_of.emplace(key, (int)ly.set_of.size()).first;
                        grown = true;
                    }
                    last_rho = rho; last_dz = gm[6]; last_set = it->second;
                }
                ws->h_set[b] = last_set;
            }
            gbp_fdem_system* h = ly.h;
            const bool sets = ly.set_of.size() > 1;
            if (s->hankel_eps > 0.0) {        // per-sounding abscissa windows: 1 m altitude bins covering this run (kept, and widened, across calls)
                int first = (int)std::floor(lo), last = std::min((int)std::floor(hi), first + 1023);
                if (grown || h->n_bins == 0 || first < h->bin0 || last >= h->bin0 + h->n_bins) {
                    if (h->n_bins > 0) { first = std::min(first, h->bin0); last = std::min(std::max(last, h->bin0 + h->n_bins - 1), first + 1023); }
                    st = gbp_hankel_system_add_bins(h, s->hankel_eps, 1, first, last - first + 1);
                }
            } else if (sets && (grown || h->d_bins == nullptr)) {
                st = gbp_hankel_system_add_bins(h, 0.0, 1, 0, 0);                 // descriptors of the sets' full tables
            }
            if (st != GBP_OK) return st;
            // index maps of the mix (per layout, once)
            if (ly.d_src == nullptr) {
                std::vector<int32_t> src((size_t)n_out * nb, -1), col((size_t)n_out * nb, 0);
                for (int c = 0; c < nc; ++c)
                    for (int t = 0; t < nb; ++t)
                        for (int j = 0; j < n; ++j) {
                            const int m = c * n + j;
                            src[(size_t)m * nb + t] = t * n + j;
                            src[(size_t)(nc * n + m) * nb + t] = nF_in + t * n + j;
                            col[(size_t)m * nb + t] = col[(size_t)(nc * n + m) * nb + t] = c * nb + t;
                        }
                hipError_t e = hipMalloc((void**)&ly.d_src, sizeof(int32_t) * src.size());
                if (e == hipSuccess) e = hipMalloc((void**)&ly.d_col, sizeof(int32_t) * col.size());
                if (e == hipSuccess) e = hipMemcpy(ly.d_src, src.data(), sizeof(int32_t) * src.size(), hipMemcpyHostToDevice);
                if (e == hipSuccess) e = hipMemcpy(ly.d_col, col.data(), sizeof(int32_t) * col.size(), hipMemcpyHostToDevice);
                if (e != hipSuccess) {
                    if (ly.d_src) (void)hipFree(ly.d_src);
                    if (ly.d_col) (void)hipFree(ly.d_col);
                    ly.d_src = ly.d_col = nullptr;
                    return fail(GBP_ERR_HIP, "mix tables upload failed: %s", hipGetErrorString(e));
                }
            }
            // this run's weights (the layout's basis integrals only), set indices, scratch
            ws->h_weights.resize((size_t)nr * n_w);
            for (int r = 0; r < nr; ++r)
                for (int c = 0; c < nc; ++c)
                    for (int t = 0; t < nb; ++t)
                        ws->h_weights[((size_t)r * nc + c) * nb + t] = wall[((size_t)r * nc + c) * GBP_TD_NBASIS + ly.basis[t]];
            if ((st = grow(&ws->d_weights, &ws->cap_weights, (size_t)nr * n_w)) != GBP_OK) return st;
            if ((st = grow(&ws->d_nodal, &ws->cap_nodal, (size_t)nr * n_in)) != GBP_OK) return st;
            if (J != nullptr && (st = grow(&ws->d_jnodal, &ws->cap_jnodal, (size_t)nr * n_in * Lmax)) != GBP_OK) return st;
            // (the staging vector is reused by the next run: the copy must have left the host before that)
            GBP_HIP(hipMemcpyAsync(ws->d_weights, ws->h_weights.data(), sizeof(double) * (size_t)nr * n_w, hipMemcpyHostToDevice, q));
            if (sets) GBP_HIP(hipMemcpyAsync(ws->d_set + b0, ws->h_set.data() + b0, sizeof(int32_t) * (size_t)nr, hipMemcpyHostToDevice, q));
            const int32_t* rows = sets ? ws->d_set + b0 : nullptr;
            const size_t ro = (size_t)b0 * Lmax;
            if (J == nullptr)
                st = gbp_fdem_forward_rows_ex(h, nr, Lmax, nlayers + b0, sigma + ro, thk + ro, ws->d_height + b0, ws->d_nodal, rows, 0, stream);
            else
                st = gbp_fdem_fm_dlogc_rows_ex(h, nr, Lmax, nlayers + b0, sigma + ro, thk + ro, ws->d_height + b0, ws->d_nodal, ws->d_jnodal, Lmax, 1,
                                               rows, 0, stream);
            if (st != GBP_OK) return st;
            gbp_td_mix mix;
            mix.n_in = n_in; mix.terms = nb; mix.n_weights = n_w; mix.src = ly.d_src; mix.col = ly.d_col; mix.weights = ws->d_weights;
            mix.offset = nullptr;
            st = gbp_td_apply_mix(nr, Lmax, n_out, N, nlayers + b0, s->d_Wb, ws->d_nodal, J ? ws->d_jnodal : nullptr, out + (size_t)b0 * N,
                                  J ? J + (size_t)b0 * N * Lmax : nullptr, &mix, stream);
            if (st != GBP_OK) return st;
            shared.unlock();
            // (the weight staging vector and the nodal scratch are reused by the next run of this call: this run must have left them)
            if (b1 < B) GBP_HIP(hipStreamSynchronize(q));
            b0 = b1;
        }
    } catch (const std::bad_alloc&) {
        return fail(GBP_ERR_INVALID_ARG, "out of host memory%s");
    }
    return GBP_OK;
}

}  // namespace td

extern "C" {

gbp_status gbp_tdem_system_create(const char* stm_text, const double* w0, const double* w1, gbp_tdem_system** out)
{
    if (!out) return fail(GBP_ERR_INVALID_ARG, "out is NULL%s");
    *out = nullptr;
    if (!stm_text || !w0 || !w1) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    gbp_tdem_system* s = new (std::nothrow) gbp_tdem_system();
    if (!s) return fail(GBP_ERR_INVALID_ARG, "out of host memory%s");
    if (!td::parse_stm(stm_text, &s->stm)) { delete s; return fail(GBP_ERR_BAD_SYSTEM, "not a time-domain .stm text (waveform, windows, BaseFrequency, WaveformDigitisingFrequency)%s"); }
    s->w0.assign(w0, w0 + GBP_NC0);
    s->w1.assign(w1, w1 + GBP_NC1);
    const char* msg = "";
    const gbp_status st = td::build_operator(s, &msg);
    if (st != GBP_OK) { delete s; return fail(st, "%s", msg); }
    *out = s;                                   // (host tables only: the operator goes to the device with the first forward call)
    return GBP_OK;
}

void gbp_tdem_system_destroy(gbp_tdem_system* s)
{
    if (!s) return;
    for (auto& kv : s->layouts) {
        gbp_fdem_system_destroy(kv.second.h);
        if (kv.second.d_src) (void)hipFree(kv.second.d_src);
        if (kv.second.d_col) (void)hipFree(kv.second.d_col);
    }
    if (s->d_Wb) (void)hipFree(s->d_Wb);
    for (auto& w : s->works) {
        for (void* p : {(void*)w->d_height, (void*)w->d_nodal, (void*)w->d_weights, (void*)w->d_jnodal, (void*)w->d_set})
            if (p) (void)hipFree(p);
        if (w->done) (void)hipEventDestroy(w->done);
    }
    delete s;
}

gbp_status gbp_tdem_system_set_hankel_eps(gbp_tdem_system* s, double eps)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (!(eps >= 0.0)) return fail(GBP_ERR_INVALID_ARG, "eps must be >= 0%s");
    std::lock_guard<std::mutex> lk(s->mu);
    if (eps != s->hankel_eps)                 // the cached tables were windowed for the old budget
        for (auto& kv : s->layouts)
            if (kv.second.h) gbp_hankel_system_clear_bins(kv.second.h);
    s->hankel_eps = eps;
    return GBP_OK;
}

gbp_status gbp_tdem_system_info(const gbp_tdem_system* s, int* n_windows, int* n_components, int* n_nodes, double* loop_radius)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (n_windows) *n_windows = s->n_windows;
    if (n_components) *n_components = s->n_components;
    if (n_nodes) *n_nodes = s->n_nodes;
    if (loop_radius) *loop_radius = s->loop_radius;
    return GBP_OK;
}

gbp_status gbp_tdem_system_tables(const gbp_tdem_system* s, double* window_centres, double* node_frequencies, double* W)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (window_centres) std::memcpy(window_centres, s->centres.data(), sizeof(double) * s->centres.size());
    if (node_frequencies) std::memcpy(node_frequencies, s->nodes.data(), sizeof(double) * s->nodes.size());
    if (W) std::memcpy(W, s->W.data(), sizeof(double) * s->W.size());
    return GBP_OK;
}

gbp_status gbp_tdem_forward(gbp_tdem_system* s, int B, const double* geometry, int Lmax, const int32_t* nlayers, const double* sigma,
                            const double* thk, double* out, void* stream)
{
    return td::run(s, B, geometry, Lmax, nlayers, sigma, thk, out, nullptr, stream);
}

gbp_status gbp_tdem_fm_dlogc(gbp_tdem_system* s, int B, const double* geometry, int Lmax, const int32_t* nlayers, const double* sigma,
                             const double* thk, double* out, double* J, void* stream)
{
    if (!J) return fail(GBP_ERR_INVALID_ARG, "J is NULL%s");
    return td::run(s, B, geometry, Lmax, nlayers, sigma, thk, out, J, stream);
}

}  // extern "C"
