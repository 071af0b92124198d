// Two host threads share ONE gbp_tdem_system handle (SURVEY 8b "Threading": re-entrant) -- each with its own stream and its own batch
// of soundings whose receiver offsets and attitudes change from row to row, calling gbp_tdem_forward / gbp_tdem_fm_dlogc concurrently.
// Thread 1 meets NEW (horizontal distance, dz) pairs in later rounds, so the handle grows its table sets (device tables are
// re-allocated) while thread 0's launches are in flight.  Every result of either thread must equal, bit for bit, what a fresh handle
// returns for the same call single-threaded.   usage: two_threads_tdem <system.stm> <weights.bin: 120 + 140 doubles>
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
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
    int B, L, N;
    int32_t* nl;
    double *sigma, *thk, *out, *J;
    hipStream_t q;
    std::vector<std::vector<double>> geom;        // one geometry block per variant (variant v: offsets seen from round v * 10 on)
    std::vector<std::vector<double>> ref_out, ref_fwd, ref_J;   // windows from fm_dlogc / from forward (other kernels: other last bits), Jacobian
};

static std::vector<double> geometry(int B, int t, int variant)
{
    std::vector<double> g((size_t)B * 10, 0.0);
    for (int i = 0; i < B; ++i) {
        double* r = &g[(size_t)i * 10];
        const int o = (i * 7 + t) % 3 + 3 * variant;                 // three offsets per variant, other ones per thread
        r[0] = 28.0 + (i * 13 % 90) * 0.1 + t;                       // transmitter height
        r[1] = 0.0; r[2] = (i % 5) * 0.7 - 1.0; r[3] = 0.0;         // tx roll, pitch, yaw
        r[4] = -12.0 - 0.75 * o - 0.25 * t; r[5] = (o % 2) * 0.5; r[6] = 2.0 + 0.125 * o;     // dx, dy, dz
        r[7] = (i % 3) * 0.5; r[8] = (i % 4) * 0.6; r[9] = 0.0;     // rx roll, pitch, yaw
    }
    return g;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: two_threads_tdem <system.stm> <weights.bin>\n"); return 1; }
    int ndev = 0;
    if (gbp_device_count(&ndev) != GBP_OK || ndev < 1) { std::fprintf(stderr, "no device\n"); return 1; }
    std::ifstream sf(argv[1]);
    std::stringstream ss; ss << sf.rdbuf();
    const std::string text = ss.str();
    std::vector<double> w(GBP_NC0 + GBP_NC1);
    { std::ifstream wf(argv[2], std::ios::binary); wf.read(reinterpret_cast<char*>(w.data()), sizeof(double) * w.size()); if (!wf) { std::fprintf(stderr, "weights file too short\n"); return 1; } }
    auto make = [&]() { gbp_tdem_system* s = nullptr; CHECK(gbp_tdem_system_create(text.c_str(), w.data(), w.data() + GBP_NC0, &s)); return s; };
    gbp_tdem_system* shared = make();
    int nw = 0, nc = 0, nn = 0; double radius = 0.0;
    CHECK(gbp_tdem_system_info(shared, &nw, &nc, &nn, &radius));
    const int n_variants = 3, rounds = 30;
    Batch bt[2];
    for (int t = 0; t < 2; ++t) {
        Batch& b = bt[t];
        b.B = 300 + 150 * t; b.L = 4; b.N = nw * nc;
        std::vector<int32_t> nl(b.B);
        std::vector<double> sg((size_t)b.B * b.L), th((size_t)b.B * b.L);
        for (int i = 0; i < b.B; ++i) {
            nl[i] = 1 + (i * 5 + t) % b.L;
            for (int k = 0; k < b.L; ++k) {
                sg[(size_t)i * b.L + k] = std::pow(10.0, -3.0 + 3.0 * ((i * 31 + k * 17 + t * 5) % 97) / 96.0);
                th[(size_t)i * b.L + k] = 3.0 + ((i * 11 + k * 29) % 40);
            }
        }
        b.nl = up(nl); b.sigma = up(sg); b.thk = up(th);
        HIP(hipMalloc((void**)&b.out, sizeof(double) * (size_t)b.B * b.N));
        HIP(hipMalloc((void**)&b.J, sizeof(double) * (size_t)b.B * b.N * b.L));
        HIP(hipStreamCreate(&b.q));
        // single-threaded references, each variant on a FRESH handle (results must not depend on what a handle has seen before)
        for (int v = 0; v < n_variants; ++v) {
            b.geom.push_back(geometry(b.B, t, v));
            gbp_tdem_system* fresh = make();
            CHECK(gbp_tdem_fm_dlogc(fresh, b.B, b.geom[v].data(), b.L, b.nl, b.sigma, b.thk, b.out, b.J, b.q));
            HIP(hipStreamSynchronize(b.q));
            b.ref_out.emplace_back((size_t)b.B * b.N); b.ref_J.emplace_back((size_t)b.B * b.N * b.L);
            HIP(hipMemcpy(b.ref_out[v].data(), b.out, sizeof(double) * b.ref_out[v].size(), hipMemcpyDeviceToHost));
            HIP(hipMemcpy(b.ref_J[v].data(), b.J, sizeof(double) * b.ref_J[v].size(), hipMemcpyDeviceToHost));
            CHECK(gbp_tdem_forward(fresh, b.B, b.geom[v].data(), b.L, b.nl, b.sigma, b.thk, b.out, b.q));
            HIP(hipStreamSynchronize(b.q));
            b.ref_fwd.emplace_back((size_t)b.B * b.N);
            HIP(hipMemcpy(b.ref_fwd[v].data(), b.out, sizeof(double) * b.ref_fwd[v].size(), hipMemcpyDeviceToHost));
            gbp_tdem_system_destroy(fresh);
        }
    }
    std::atomic<int> bad(0), calls(0);
    auto work = [&](int t) {
        Batch& b = bt[t];
        HIP(hipSetDevice(0));
        std::vector<double> o((size_t)b.B * b.N), J((size_t)b.B * b.N * b.L);
        for (int r = 0; r < rounds; ++r) {
            const int v = t == 1 ? std::min(n_variants - 1, r / 10) : r % n_variants;   // thread 1: new offsets from rounds 10 and 20 on; thread 0: all the time
            HIP(hipMemsetAsync(b.out, 0xFF, sizeof(double) * o.size(), b.q));
            const bool with_j = r % 3 == 0;
            if (with_j) CHECK(gbp_tdem_fm_dlogc(shared, b.B, b.geom[v].data(), b.L, b.nl, b.sigma, b.thk, b.out, b.J, b.q));
            else CHECK(gbp_tdem_forward(shared, b.B, b.geom[v].data(), b.L, b.nl, b.sigma, b.thk, b.out, b.q));
            HIP(hipStreamSynchronize(b.q));
            HIP(hipMemcpy(o.data(), b.out, sizeof(double) * o.size(), hipMemcpyDeviceToHost));
            if (std::memcmp(o.data(), (with_j ? b.ref_out[v] : b.ref_fwd[v]).data(), sizeof(double) * o.size()) != 0) ++bad;
            if (with_j) {
                HIP(hipMemcpy(J.data(), b.J, sizeof(double) * J.size(), hipMemcpyDeviceToHost));
                if (std::memcmp(J.data(), b.ref_J[v].data(), sizeof(double) * J.size()) != 0) ++bad;
            }
            ++calls;
        }
    };
    std::thread t0(work, 0), t1(work, 1);
    t0.join(); t1.join();
    int finite = 1;
    for (double v : bt[0].ref_out[0]) finite &= std::isfinite(v) ? 1 : 0;
    gbp_tdem_system_destroy(shared);
    std::printf("two threads x %d calls on one time-domain handle (%d windows x %d components, %d nodes): %d mismatching results; finite %d\n",
                calls.load(), nw, nc, nn, bad.load(), finite);
    return (bad.load() == 0 && finite) ? 0 : 4;
}
