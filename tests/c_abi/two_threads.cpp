// Two host threads share ONE system handle (SURVEY 8b "Threading": re-entrant, no hidden state): each has its own stream, its own
// batch and its own per-row table-set indices -- an ARGUMENT of gbp_fdem_forward_rows_ex / gbp_fdem_fm_dlogc_rows_ex since round 3
// (the handle used to carry the row map between a set_rows call and the launch).  Every call of either thread must equal the
// single-threaded result bit for bit.  Build: hipcc -O2 -std=c++17 two_threads.cpp -lgeobipy_amd -lpthread
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/geobipy_amd.h"

#define CHECK(call)                                                                                 \
    do {                                                                                            \
        gbp_status st_ = (call);                                                                    \
        if (st_ != GBP_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, st_, gbp_last_error()); std::exit(2); } \
    } while (0)
#define HIP(call)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(3); } \
    } while (0)

template <class T> T* up(const std::vector<T>& v)
{
    T* d = nullptr;
    HIP(hipMalloc((void**)&d, sizeof(T) * v.size()));
    HIP(hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return d;
}

struct Batch {
    int B, L;
    int32_t *nl, *set;
    double *sigma, *thk, *height, *pred, *J;
    std::vector<double> ref_pred, ref_J;
    hipStream_t q;
};

int main()
{
    int ndev = 0;
    if (gbp_device_count(&ndev) != GBP_OK || ndev < 1) { std::fprintf(stderr, "no device\n"); return 1; }
    // a raw Hankel handle: 6 "frequencies" x 96 log-spaced abscissae, and two further table sets (other offsets)
    const int nF = 6, np_ = 96, P = nF * np_, n_sets = 3;
    std::vector<int32_t> npts(nF, np_);
    std::vector<double> wmu(nF), hd0(nF), g(2 * nF), tables((size_t)7 * P);
    auto fill = [&](double r, double dz, std::vector<double>& t, std::vector<double>& h) {
        for (int f = 0; f < nF; ++f) {
            h[f] = -dz;
            for (int j = 0; j < np_; ++j) {
                const double lam = std::pow(10.0, -4.0 + 4.5 * j / (np_ - 1)) / r;
                const int q = f * np_ + j;
                t[0 * P + q] = lam * lam; t[1 * P + q] = lam; t[2 * P + q] = 0.0;
                t[3 * P + q] = lam * lam * std::cos(0.3 * j) / r; t[4 * P + q] = 0.0; t[5 * P + q] = lam; t[6 * P + q] = 0.0;
            }
        }
    };
    for (int f = 0; f < nF; ++f) { wmu[f] = 2.0 * M_PI * 100.0 * std::pow(4.0, f) * 4e-7 * M_PI; g[2 * f] = 1.0; g[2 * f + 1] = 0.0; }
    fill(13.0, 2.0, tables, hd0);
    gbp_fdem_system* sys = nullptr;
    CHECK(gbp_hankel_system_create_raw(nF, npts.data(), wmu.data(), hd0.data(), g.data(), tables.data(), &sys));
    for (int s = 1; s < n_sets; ++s) {
        std::vector<double> t2((size_t)7 * P), h2(nF);
        fill(13.0 + 4.0 * s, 2.0 + 0.5 * s, t2, h2);
        CHECK(gbp_hankel_system_add_set(sys, h2.data(), t2.data()));
    }
    CHECK(gbp_hankel_system_add_bins(sys, 1e-12, 1, 20, 30));       // per-sounding abscissa windows too: the whole descriptor path

    Batch bt[2];
    for (int t = 0; t < 2; ++t) {
        Batch& b = bt[t];
        b.B = 700 + 300 * t; b.L = 5;
        std::vector<int32_t> nl(b.B), set(b.B);
        std::vector<double> sg((size_t)b.B * b.L), th((size_t)b.B * b.L), h(b.B);
        for (int i = 0; i < b.B; ++i) {
            nl[i] = 1 + (i * 7 + t) % b.L;
            set[i] = (i * 7 + i / 3 + t) % n_sets;
            h[i] = 22.0 + (i * 13 % 170) * 0.1 + t;
            for (int k = 0; k < b.L; ++k) {
                sg[(size_t)i * b.L + k] = std::pow(10.0, -3.0 + 3.0 * ((i * 31 + k * 17 + t * 5) % 97) / 96.0);
                th[(size_t)i * b.L + k] = 2.0 + ((i * 11 + k * 29) % 40);
            }
        }
        b.nl = up(nl); b.set = up(set); b.sigma = up(sg); b.thk = up(th); b.height = up(h);
        HIP(hipMalloc((void**)&b.pred, sizeof(double) * (size_t)b.B * 2 * nF));
        HIP(hipMalloc((void**)&b.J, sizeof(double) * (size_t)b.B * 2 * nF * b.L));
        HIP(hipStreamCreate(&b.q));
        // single-threaded reference
        CHECK(gbp_fdem_forward_rows_ex(sys, b.B, b.L, b.nl, b.sigma, b.thk, b.height, b.pred, b.set, 2, b.q));
        CHECK(gbp_fdem_fm_dlogc_rows_ex(sys, b.B, b.L, b.nl, b.sigma, b.thk, b.height, nullptr, b.J, b.L, 1, b.set, 0, b.q));
        HIP(hipStreamSynchronize(b.q));
        b.ref_pred.resize((size_t)b.B * 2 * nF); b.ref_J.resize((size_t)b.B * 2 * nF * b.L);
        HIP(hipMemcpy(b.ref_pred.data(), b.pred, sizeof(double) * b.ref_pred.size(), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(b.ref_J.data(), b.J, sizeof(double) * b.ref_J.size(), hipMemcpyDeviceToHost));
    }
    // the two batches really use different sets per row
    std::atomic<int> bad(0);
    const int rounds = 200;
    auto work = [&](int t) {
        Batch& b = bt[t];
        HIP(hipSetDevice(0));
        std::vector<double> p(b.ref_pred.size()), J(b.ref_J.size());
        for (int r = 0; r < rounds; ++r) {
            HIP(hipMemsetAsync(b.pred, 0xFF, sizeof(double) * p.size(), b.q));
            CHECK(gbp_fdem_forward_rows_ex(sys, b.B, b.L, b.nl, b.sigma, b.thk, b.height, b.pred, b.set, 2, b.q));
            if (r % 4 == 0) CHECK(gbp_fdem_fm_dlogc_rows_ex(sys, b.B, b.L, b.nl, b.sigma, b.thk, b.height, nullptr, b.J, b.L, 1, b.set, 0, b.q));
            HIP(hipStreamSynchronize(b.q));
            HIP(hipMemcpy(p.data(), b.pred, sizeof(double) * p.size(), hipMemcpyDeviceToHost));
            if (std::memcmp(p.data(), b.ref_pred.data(), sizeof(double) * p.size()) != 0) ++bad;
            if (r % 4 == 0) {
                HIP(hipMemcpy(J.data(), b.J, sizeof(double) * J.size(), hipMemcpyDeviceToHost));
                if (std::memcmp(J.data(), b.ref_J.data(), sizeof(double) * J.size()) != 0) ++bad;
            }
        }
    };
    std::thread t0(work, 0), t1(work, 1);
    t0.join(); t1.join();
    // and the rows DO depend on their set: set 0 for every row gives other numbers
    Batch& b = bt[0];
    CHECK(gbp_fdem_forward_rows_ex(sys, b.B, b.L, b.nl, b.sigma, b.thk, b.height, b.pred, nullptr, 2, b.q));
    HIP(hipStreamSynchronize(b.q));
    std::vector<double> p0(b.ref_pred.size());
    HIP(hipMemcpy(p0.data(), b.pred, sizeof(double) * p0.size(), hipMemcpyDeviceToHost));
    int differ = 0, finite = 1;
    for (size_t i = 0; i < p0.size(); ++i) { differ += p0[i] != b.ref_pred[i]; finite &= std::isfinite(b.ref_pred[i]) ? 1 : 0; }
    gbp_fdem_system_destroy(sys);
    std::printf("two threads x %d rounds on one handle: %d mismatching results; rows whose set matters: %d of %zu values; finite %d\n",
                rounds, bad.load(), differ, p0.size(), finite);
    return (bad.load() == 0 && differ > (int)p0.size() / 3 && finite) ? 0 : 4;
}
