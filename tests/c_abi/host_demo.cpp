// A torch-free, Python-free host of libgeobipy_amd.so: everything goes through include/geobipy_amd.h and the HIP runtime.
// Reads a flat binary of inputs (written by tests/test_c_abi_host.py), builds the system handle, evaluates the fused
// forward + chi^2 + logL, then allocates the sampler's buffers itself, initialises every chain at its half-space and runs
// gbp_rj_run; writes predictions and the end state of the chains to a flat binary.  Test infrastructure: it shows that the
// drop-in boundary needs nothing but C types.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/geobipy_amd.h"

#define CHECK(x)                                                                         \
    do {                                                                                 \
        if ((x) != 0) { std::fprintf(stderr, "%s failed: %s\n", #x, gbp_last_error()); return 2; } \
    } while (0)
#define HIP(x)                                                                           \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 3; } \
    } while (0)

static std::vector<double> rd(FILE* f, size_t n) { std::vector<double> v(n); if (std::fread(v.data(), 8, n, f) != n) std::exit(4); return v; }
template <class T> static T* dev(size_t n) { void* p = nullptr; if (hipMalloc(&p, n * sizeof(T)) != hipSuccess || hipMemset(p, 0, n * sizeof(T)) != hipSuccess) std::exit(5); return (T*)p; }
template <class T> static T* up(const std::vector<T>& v) { T* p = dev<T>(v.size()); if (hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::exit(6); return p; }
template <class T> static std::vector<T> down(const T* p, size_t n) { std::vector<T> v(n); if (hipMemcpy(v.data(), p, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::exit(7); return v; }

int main(int argc, char** argv)
{
    if (argc < 3) return 1;
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return 1;
    std::vector<double> hd = rd(f, 8);                       // nF, B, K, n_iterations, seed, -, -, -
    const int nF = (int)hd[0], B = (int)hd[1], K = (int)hd[2], n_it = (int)hd[3], N = 2 * nF;
    const uint64_t seed = (uint64_t)hd[4];
    std::vector<double> tid_d = rd(f, nF), freq = rd(f, nF), tx_z = rd(f, nF), rx_z = rd(f, nF), tx_m = rd(f, nF), scale = rd(f, nF),
                        rx_off = rd(f, nF), sep = rd(f, nF), w0 = rd(f, GBP_NC0), lam0 = rd(f, (size_t)nF * GBP_NC0), w1 = rd(f, GBP_NC1),
                        lam1 = rd(f, (size_t)nF * GBP_NC1);
    std::vector<double> height = rd(f, B), data = rd(f, (size_t)B * N), sigma0 = rd(f, B), opt = rd(f, 20);
    std::fclose(f);
    std::vector<int32_t> tid(nF);
    for (int i = 0; i < nF; ++i) tid[i] = (int32_t)tid_d[i];

    gbp_fdem_system* sys = nullptr;
    // per-sounding abscissa windows, 1 m altitude bins from 20 m up: the handle geobipy_amd's Python host side uses by default
    CHECK(gbp_fdem_system_create_binned(nF, tid.data(), freq.data(), tx_z.data(), rx_z.data(), tx_m.data(), scale.data(), rx_off.data(),
                                        sep.data(), w0.data(), lam0.data(), w1.data(), lam1.data(), 1.0e-10, 20, 64, &sys));
    // chain state: half-space models
    std::vector<int32_t> k1(B, 1);
    std::vector<double> sig((size_t)B * K, 1.0), edges((size_t)B * K, INFINITY), rel(B, opt[0]), add(B, opt[1]), lmp(B);
    for (int b = 0; b < B; ++b) { sig[(size_t)b * K] = sigma0[b]; lmp[b] = std::log(sigma0[b]); }
    gbp_rj_chains c;
    std::memset(&c, 0, sizeof(c));
    c.B = B;
    c.data = up(data); c.height = up(height); c.log_mean_prior = up(lmp);
    c.k = up(k1); c.edges = up(edges); c.sigma = up(sig); c.rel = up(rel); c.add = up(add);
    c.pred = dev<double>((size_t)B * N); c.J = dev<double>((size_t)B * N * K);
    c.prior = dev<double>(B); c.like = dev<double>(B); c.misfit = dev<double>(B);
    c.action = dev<int32_t>(B); c.k_r = dev<int32_t>(B); c.nl_a = dev<int32_t>(3 * (size_t)B); c.nl_c = dev<int32_t>(3 * (size_t)B);
    c.nl_b = dev<int32_t>(B);
    c.edges_r = dev<double>((size_t)B * K); c.sigma_r = dev<double>((size_t)B * K); c.thk_r = dev<double>((size_t)B * K);
    c.rel_p = dev<double>(B); c.add_p = dev<double>(B); c.pred_r = dev<double>((size_t)B * N); c.J_r = dev<double>((size_t)B * N * K);
    c.chol = dev<double>((size_t)B * K * K); c.log_prop = dev<double>((size_t)B * K); c.sigma_p = dev<double>((size_t)B * K);
    c.pred_p = dev<double>((size_t)B * N); c.misfit_p = dev<double>(B); c.like_p = dev<double>(B); c.J_p = dev<double>((size_t)B * N * K);
    c.log_ratio = dev<double>(B); c.n_accepted = dev<int64_t>(B); c.k_hist = dev<int32_t>((size_t)B * (K + 1));
    c.best_posterior = dev<double>(B); c.best_k = dev<int32_t>(B); c.best_edges = dev<double>((size_t)B * K);
    c.best_sigma = dev<double>((size_t)B * K);

    // initial prediction / misfit / likelihood / Jacobian through the same entries the sampler uses
    double* thk0 = dev<double>((size_t)B * K);
    CHECK(gbp_fdem_forward_loglike_ex(sys, B, K, c.k, c.sigma, thk0, c.height, c.data, c.rel, c.add, c.pred, c.misfit, c.like, 4, nullptr));
    CHECK(gbp_fdem_sensitivity_ex(sys, B, K, c.k, c.sigma, thk0, c.height, c.J, 1, 1, nullptr));
    HIP(hipDeviceSynchronize());
    std::vector<double> pred0 = down(c.pred, (size_t)B * N), chi0 = down(c.misfit, B), like0 = down(c.like, B);

    gbp_rj_options o;
    std::memset(&o, 0, sizeof(o));
    o.max_layers = K; o.n_channels = N; o.solve_gradient = 1; o.solve_relative_error = 1; o.solve_additive_error = 1; o.exact_jacobian = 1;
    o.n_error_bins = 99; o.forward_waves = 4; o.n_rel_groups = 1; o.n_add_groups = 1;
    o.min_edge = opt[2]; o.max_edge = opt[3]; o.min_width = opt[4];
    o.p_birth = opt[5]; o.p_death = opt[6]; o.p_perturb = opt[7]; o.p_none = opt[8];
    o.value_precision = opt[9]; o.gradient_precision = opt[10]; o.alpha = opt[11];
    o.rel_min[0] = opt[12]; o.rel_max[0] = opt[13]; o.rel_sd[0] = opt[14]; o.add_min[0] = opt[15]; o.add_max[0] = opt[16]; o.add_sd[0] = opt[17];
    o.seed = seed;
    // prior of a half-space: uniform on k, zero gradient, log-uniform error levels (Model.probability, DataPoint.probability)
    const double LOG_2PI = 1.8378770664093454835606594728112;
    const double prior0 = -std::log((double)K - 1.0) - 0.5 * LOG_2PI + 0.5 * std::log(o.gradient_precision)
                          - std::log(std::log(opt[13]) - std::log(opt[12])) - std::log(std::log(opt[16]) - std::log(opt[15]));
    std::vector<double> pr(B, prior0), best(B);
    for (int b = 0; b < B; ++b) best[b] = prior0 + like0[b];
    HIP(hipMemcpy(c.prior, pr.data(), B * 8, hipMemcpyHostToDevice));
    HIP(hipMemcpy(c.best_posterior, best.data(), B * 8, hipMemcpyHostToDevice));

    CHECK(gbp_rj_run(sys, &o, &c, 0, n_it, 1, nullptr));
    HIP(hipDeviceSynchronize());
    std::vector<int32_t> k = down(c.k, B), kh = down(c.k_hist, (size_t)B * (K + 1));
    std::vector<int64_t> acc = down(c.n_accepted, B);
    std::vector<double> sg = down(c.sigma, (size_t)B * K), mis = down(c.misfit, B);

    // time-domain boundary (optional 3rd argument: a .stm file): gbp_tdem_system_create on its TEXT, then B soundings of a
    // 3-layer earth through gbp_tdem_forward(handle, B, geometry[B, 10], ...) -- the GA-AEM call of TD/tdem1d.py:89-96
    std::vector<double> td_out;
    if (argc > 3) {
        FILE* sf = std::fopen(argv[3], "rb");
        if (!sf) return 1;
        std::string text;
        char buf[4096];
        for (size_t n; (n = std::fread(buf, 1, sizeof(buf), sf)) > 0;) text.append(buf, n);
        std::fclose(sf);
        gbp_tdem_system* ts = nullptr;
        CHECK(gbp_tdem_system_create(text.c_str(), w0.data(), w1.data(), &ts));
        int nw = 0, nc = 0, nn = 0;
        double radius = 0.0;
        CHECK(gbp_tdem_system_info(ts, &nw, &nc, &nn, &radius));
        std::vector<double> geom((size_t)B * 10, 0.0), tsig((size_t)B * 3), tthk((size_t)B * 3, 0.0);
        std::vector<int32_t> tnl(B, 3);
        for (int b = 0; b < B; ++b) {
            geom[(size_t)b * 10] = height[b]; geom[(size_t)b * 10 + 4] = -13.0; geom[(size_t)b * 10 + 6] = 2.0;
            tsig[(size_t)b * 3] = 0.01; tsig[(size_t)b * 3 + 1] = 0.1; tsig[(size_t)b * 3 + 2] = sigma0[b];
            tthk[(size_t)b * 3] = 20.0 + b; tthk[(size_t)b * 3 + 1] = 40.0;
        }
        int32_t* d_nl = up(tnl);
        double *d_sg = up(tsig), *d_th = up(tthk), *d_o = dev<double>((size_t)B * nc * nw);
        CHECK(gbp_tdem_forward(ts, B, geom.data(), 3, d_nl, d_sg, d_th, d_o, nullptr));
        HIP(hipDeviceSynchronize());
        td_out = down(d_o, (size_t)B * nc * nw);
        gbp_tdem_system_destroy(ts);
        std::printf("time domain: %d windows x %d component(s), %d spline nodes, loop radius %.3f m\n", nw, nc, nn, radius);
    }

    FILE* g = std::fopen(argv[2], "wb");
    if (!g) return 1;
    std::fwrite(pred0.data(), 8, pred0.size(), g); std::fwrite(chi0.data(), 8, B, g); std::fwrite(like0.data(), 8, B, g);
    std::vector<double> kd(k.begin(), k.end()), ad(acc.begin(), acc.end()), khd(kh.begin(), kh.end());
    std::fwrite(kd.data(), 8, kd.size(), g); std::fwrite(ad.data(), 8, ad.size(), g); std::fwrite(khd.data(), 8, khd.size(), g);
    std::fwrite(sg.data(), 8, sg.size(), g); std::fwrite(mis.data(), 8, B, g);
    if (!td_out.empty()) std::fwrite(td_out.data(), 8, td_out.size(), g);
    std::fclose(g);
    gbp_fdem_system_destroy(sys);
    std::printf("%s: %d chains x %d iterations through the C ABI\n", gbp_version(), B, n_it);
    return 0;
}
