// Device-resident rjMCMC step (include/geobipy_amd.h, section "Device-resident rjMCMC step").
// Included at the end of gbp_fdem.hip (same translation unit: shares fail() / GBP_HIP and the kernel launchers).
//
// Three kernels hold the host logic of one iteration of the reference's Inference1D.accept_reject
// (inversion/Inference1D.py:537-631); between them the forward / Jacobian kernels above are entered with
// per-chain layer counts of 0 for the chains that do not need them (a workgroup whose sounding has 0 layers exits).
//   k_rj_propose  one wave (small blocks) or one thread (large blocks) per chain: structural move, value remapping,
//                 error proposals
//   k_rj_newton   one wave per chain:  Gauss-Newton precision, its Cholesky factor in LDS, mean and sample
//   k_rj_accept   one wave per chain:  priors, reversible-jump proposal ratio, Metropolis test, state update, posteriors
//   k_rj_newton8 / k_rj_accept8: the same for models of <= 8 layers, 8 lanes per chain; their launch may carry the deeper chains'
//                 workgroups too (one launch per stage, one_stage_launch)
#pragma once

// Register budget of the two per-chain physics kernels (k_rj_physics, k_rj_persistent): they are launched with at most 4 waves
// per workgroup, so they are declared __launch_bounds__(256) and amdgpu_waves_per_eu picks the VGPR cap: 4 waves per SIMD = 128
// VGPRs, 3 = 168, 2 = 256 (gfx950: 512 VGPRs per SIMD lane).
// Measured (same box, scripts/bench_rj_modes.py through scripts/ab builds; chain-iterations/s, reference-Jacobian mode):
//     waves per SIMD (VGPR cap, spills physics / persistent)   4 (128; 25-27 / 9)   3 (168; 0 / 7)   2 (256; 0 / 0)
//     k_rj_physics,    8 192 Resolve chains, lock-step                37.6 M             35.7 M          28.5 M
//     k_rj_persistent, 1 024 Resolve chains                           16.9 M             18.3 M          19.2 M
//     k_rj_persistent, 1 024 ten-frequency chains                     15.5 M             16.6 M          17.4 M
// The lock-step physics kernel is throughput-bound: the fourth wave per SIMD buys more than the 25 spilled registers cost (they sat
// in the Jacobian pass of models above 8 layers; since round 5 that pass sums two row groups per evaluation instead of eight and the
// kernel has 118 VGPRs, none spilled, no scratch -- GBP_RJ_DEEP_NG below; the table is the measurement that chose four waves).  The persistent kernel is a latency chain per workgroup with at most 4 workgroups of 2 waves resident per
// CU (LDS): it never uses more than 2 waves per SIMD, so the whole register file is free -- no spills, +13 %.
#ifndef GBP_RJ_PHYSICS_WAVES_PER_EU
#define GBP_RJ_PHYSICS_WAVES_PER_EU 4
#endif
// Layers of a model whose Jacobian working set (one complex per lane and layer) the lock-step physics kernel keeps in LDS; deeper models
// work in their chain's global block (sens_body's deep variant: same sums, same bits).  8: 16.6 KB per two-wave workgroup, eight
// workgroups per CU; 6: 12.5 KB, ten per CU -- what a fifth wave per SIMD (GBP_RJ_PHYSICS_WAVES_PER_EU 5) needs.
#ifndef GBP_RJ_PHYSICS_LDS_LAYERS
#define GBP_RJ_PHYSICS_LDS_LAYERS 8
#endif
#ifndef GBP_RJ_PERSISTENT_WAVES_PER_EU
#define GBP_RJ_PERSISTENT_WAVES_PER_EU 2
#endif
#include <exception>
#include <string>
#include <thread>

#ifndef GBP_RJ_PHYSICS_MAX_WAVES
#define GBP_RJ_PHYSICS_MAX_WAVES 4
#endif
#define GBP_RJ_PHYSICS_BOUNDS __launch_bounds__(64 * GBP_RJ_PHYSICS_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(GBP_RJ_PHYSICS_WAVES_PER_EU, GBP_RJ_PHYSICS_WAVES_PER_EU)))
#define GBP_RJ_PERSISTENT_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GBP_RJ_PERSISTENT_WAVES_PER_EU, GBP_RJ_PERSISTENT_WAVES_PER_EU)))

namespace rj {

// The caller's options plus the logarithms of the option constants the stages need at every iteration (prior bounds of the
// error levels, depth range, prior normalisations): computed once per launch on the host (extend(), std::log) instead of ~20
// library calls per chain and iteration.
struct RjOpt : gbp_rj_options {
    double log_min_edge, log_max_edge, log_layers_m1, log_value_precision, log_gradient_precision;
    double log_rel_min[4], log_rel_max[4], log_add_min[4], log_add_max[4];
    double nlog_rel_span[4], nlog_add_span[4];               // -log(log(max) - log(min)): the log-uniform prior density
    double nlog_height_span;                                 // -log(2 height_half_width): the uniform prior density of the height
};

inline RjOpt extend(const gbp_rj_options& o)
{
    static thread_local gbp_rj_options last;                  // (every launch of a run passes the same options)
    static thread_local RjOpt cached;
    static thread_local bool have = false;
    if (have && std::memcmp(&last, &o, sizeof(o)) == 0) return cached;
    RjOpt x;
    static_cast<gbp_rj_options&>(x) = o;
    x.log_min_edge = std::log(o.min_edge); x.log_max_edge = std::log(o.max_edge);
    x.log_layers_m1 = std::log((double)o.max_layers - 1.0);
    x.log_value_precision = std::log(o.value_precision); x.log_gradient_precision = std::log(o.gradient_precision);
    for (int g = 0; g < 4; ++g) {
        x.log_rel_min[g] = std::log(o.rel_min[g]); x.log_rel_max[g] = std::log(o.rel_max[g]);
        x.log_add_min[g] = std::log(o.add_min[g]); x.log_add_max[g] = std::log(o.add_max[g]);
        x.nlog_rel_span[g] = -std::log(x.log_rel_max[g] - x.log_rel_min[g]);
        x.nlog_add_span[g] = -std::log(x.log_add_max[g] - x.log_add_min[g]);
    }
    x.nlog_height_span = o.solve_height ? -std::log(2.0 * o.height_half_width) : 0.0;
    std::memcpy(&last, &o, sizeof(o)); cached = x; have = true;
    return x;
}


// ---------------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11), counter = (chain, iteration, stream, draw), key = seed
// ---------------------------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

__host__ __device__ inline double u53(uint32_t a, uint32_t b)      // [0, 1) with 53 random bits
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

constexpr double TWO_PI = 6.283185307179586476925286766559;

// Ordering point between the lanes of ONE wave that exchange data through LDS / global memory (the per-chain stages below are
// executed by a single wave, inside single-wave workgroups as well as inside the two-wave persistent kernel, so a workgroup
// barrier would be both unnecessary and -- in the latter -- a deadlock).  A wave's memory operations execute in program order;
// this only keeps the compiler from moving them across.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The transcendental functions and the generator are CALLED, not inlined: inlined at their ~60 call sites they made the
// proposal stage 13 000 instructions (27 000 in the persistent kernel's copy) and the accept stage 11 000 -- ~400 KB of code
// per iteration of the persistent kernel against a 64 KB instruction cache, for stages that run as one wave per SIMD and
// cannot hide a fetch miss.  Same library routines, same bits.
#define GBP_RJ_CALL __attribute__((noinline))
// The packed per-chain stages are single-wave workgroups on a GPU they leave almost empty: occupancy is worth nothing to them, a register
// spilled to scratch is a trip to memory on their dependency chain -- they may take the whole register file (spills go to AGPRs).
#define GBP_RJ_LATENCY_KERNEL __attribute__((amdgpu_waves_per_eu(1, 1)))
// ... and their waves share SIMDs with the physics waves of the other sub-blocks (a stand-alone k_rj_step8 wave lives 29 us, inside an
// iteration 44): s_setprio 3 puts the few hundred waves that ARE the dependency chain ahead of the thousands that fill the machine.
// Same-box A/B (scripts/ab_rj.py): 8 192 chains 45.6 -> 46.8 M chain-it/s, 4 096: 32.8 -> 33.3; the persistent kernel raises its serial
// stages (wave 0 of a chain) the same way: 1 024 chains 19.2 -> 19.6 M, 512: + 4.5 %.  (A priority by model depth for the persistent
// kernel's physics waves -- the launch ends with its slowest chain -- measured nothing.)
#ifndef GBP_RJ_LATENCY_PRIO
#define GBP_RJ_LATENCY_PRIO 3
#endif
#ifndef GBP_RJ_SHARES_UP_TO
#define GBP_RJ_SHARES_UP_TO 700
#endif
// Round 6: the physics launches take their chains deepest model first (k_rj_order_by_layers, once per call and sub-block): a launch lasts as
// long as its slowest workgroup, a workgroup's life grows with its chain's layer count (19.6 us + 3.6 us per layer for a Jacobian pass at
// 8 192 chains), and layer counts change by one per accepted birth / death -- the order of the call's first iteration stays a good one.
// Which workgroup evaluates which chain changes no bit.  GBP_RJ_SPLIT_DEEP = n > 0: the deepest 1 / n of a launch's chains also get
// a second workgroup, which takes half of the frequencies of a Jacobian pass of 4 or more layers (share / n_shares of sens_body).
#ifndef GBP_RJ_ORDERED_PHYSICS
#define GBP_RJ_ORDERED_PHYSICS 1
#endif
#ifndef GBP_RJ_REORDER_EVERY
#define GBP_RJ_REORDER_EVERY 256     // iterations between two sorts of a call
#endif
#ifndef GBP_RJ_SPLIT_DEEP
#define GBP_RJ_SPLIT_DEEP 0
#endif
#ifndef GBP_RJ_SPLIT_MIN_LAYERS
#define GBP_RJ_SPLIT_MIN_LAYERS 4
#endif
// Row groups (of 8 layers) the Jacobian pass of a model of MORE than 8 layers sums per evaluation (sens_body<EXACT, NG>; the pass is
// repeated for the next 8 NG layers).  8 -- one evaluation for any model -- keeps 16 complex accumulators and costs the physics kernel 25
// spilled registers and 104 B of scratch; 2 covers 16 layers per evaluation (two evaluations for 17 - 32 layers) with none: 118 VGPRs,
// 0 B.  The rows are the same sums in the same order whatever the grouping: same bits (tests/test_rjmcmc_gpu.py, the deep cases).
#ifndef GBP_RJ_DEEP_NG
#define GBP_RJ_DEEP_NG 2
#endif
#ifndef GBP_RJ_PERSISTENT_PRIO
#define GBP_RJ_PERSISTENT_PRIO 3
#endif

#define GBP_RJ_SERIAL_PRIO(P) do { if (GBP_RJ_PERSISTENT_PRIO > 0) __builtin_amdgcn_s_setprio(P); } while (0)
#define GBP_RJ_RAISE_PRIO() do { if (GBP_RJ_LATENCY_PRIO > 0) __builtin_amdgcn_s_setprio(GBP_RJ_LATENCY_PRIO); } while (0)
// channels whose Jacobian column the packed stages' one-trip variants hold in registers (N <= this: Resolve 12, ten frequencies 20)
#ifndef GBP_RJ_COLUMN_ROWS
#define GBP_RJ_COLUMN_ROWS 24
#endif
// The one-trip variants of the packed Newton / accept stages (everything requested in one batch behind the move, logarithms and the
// generator inlined, interface widths from the group's registers, counters as atomic adds).  Round 6: the ACCEPT variant is ON -- with its
// Cholesky / Jacobian-column requests issued by the groups whose move needs them instead of by every group of a wave, 8 192 chains 52.4 ->
// 53.5 M chain-it/s, 4 096: 38.0 -> 39.0, 2 048: 24.3 -> 25.1, HBM traffic 157 -> 142 MB per iteration (docs/notes_r6.md); the Newton variant
// on top adds nothing (25.0 / 38.7 / 52.9) and stays off.  Round 5's measurement, when both were measurement variants only:
// same-box A/B (scripts/ab_rj.py, M chain-it/s, off | Newton | accept | both): 8 192 chains 44.25 | 44.26 | 44.32 | 44.56, 4 096: 30.62 |
// 30.53 | 30.38 | 30.31, 2 048: 21.01 | 21.18 | 21.06 | 21.33 -- + 1 % at best -- against + 30 MB of HBM traffic per iteration of 8 192
// chains (203 vs 173 MB: the batch requests what the plain code skipped) and kernels no shorter under the profiler (Newton 25.5 vs 21 us,
// accept + proposal 46.9 vs 44.2 us).  docs/notes_r5.md.
#ifndef GBP_RJ_ONE_TRIP_NEWTON
#define GBP_RJ_ONE_TRIP_NEWTON 0
#endif
#ifndef GBP_RJ_ONE_TRIP_ACCEPT
#define GBP_RJ_ONE_TRIP_ACCEPT 1
#endif
// Round 6: the logarithm and the circular functions of these stages are gbp_math.h's log_pos / sincos_quadrant -- 34 and ~40 VALU issues
// where the library's log took 98 and its cos / sincos behind Box-Muller 150 - 250 (argument reductions for any double; here the arguments
// are positive normal numbers and an angle in [0, 2 pi]), ~1 ulp either way.  These calls were 40 % of the instructions of an accept +
// proposal launch, and the stages are dependency chains of one wave.  GBP_RJ_LIBM_MATH restores the library routines (A/B builds).
#ifdef GBP_RJ_LIBM_MATH
__device__ GBP_RJ_CALL double rj_log(double x) { return log(x); }
#else
__device__ GBP_RJ_CALL double rj_log(double x) { return gbp::log_pos(x); }
#endif
__device__ GBP_RJ_CALL double rj_exp(double x) { return exp(x); }
// (the one-trip stages inline their logarithms -- a call would wait for the loads in flight: the same routine, the same bits)
#ifdef GBP_RJ_LIBM_MATH
__device__ __forceinline__ double rj_log_inl(double x) { return log(x); }
#else
__device__ __forceinline__ double rj_log_inl(double x) { return gbp::log_pos(x); }
#endif
__device__ GBP_RJ_CALL U4 philox_call(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t stream, uint32_t n)
{
    return philox(seed, chain, iter, stream, n);
}
struct Pair { double a, b; };
#ifdef GBP_RJ_LIBM_MATH
__device__ GBP_RJ_CALL double box_muller_cos(double u1, double u2) { return sqrt(-2.0 * log(1.0 - u1)) * cos(TWO_PI * u2); }
__device__ GBP_RJ_CALL Pair box_muller_pair(double u1, double u2)     // (returned in registers: two output pointers of a call are two objects in scratch)
{
    const double rad = sqrt(-2.0 * log(1.0 - u1)), ang = TWO_PI * u2;
    Pair z;
    z.a = rad * cos(ang); z.b = rad * sin(ang);
    return z;
}
__device__ __forceinline__ Pair box_muller_pair_inl(double u1, double u2) { return box_muller_pair(u1, u2); }
#else
__device__ __forceinline__ Pair box_muller_pair_inl(double u1, double u2)
{
    const double rad = sqrt(-2.0 * gbp::log_pos(1.0 - u1)), ang = TWO_PI * u2;       // (the angle rounded as the host emulation rounds it)
    double sn, cs;
    gbp::sincos_quadrant(ang, sn, cs);
    Pair z;
    z.a = rad * cs; z.b = rad * sn;
    return z;
}
__device__ GBP_RJ_CALL Pair box_muller_pair(double u1, double u2)     // (returned in registers: two output pointers of a call are two objects in scratch)
{
    return box_muller_pair_inl(u1, u2);
}
__device__ __forceinline__ double box_muller_cos(double u1, double u2) { return box_muller_pair(u1, u2).a; }
#endif

struct Rng {                                                    // sequential draws of one (chain, iteration, stream)
    uint64_t seed; uint32_t chain, iter, stream, n; double buf; bool have;
    __device__ Rng(uint64_t s, uint32_t c, uint32_t i, uint32_t st) : seed(s), chain(c), iter(i), stream(st), n(0), buf(0.0), have(false) {}
    __device__ double uniform()
    {
        if (have) { have = false; return buf; }
        const U4 r = philox_call(seed, chain, iter, stream, n++);
        buf = u53(r.z, r.w); have = true;
        return u53(r.x, r.y);
    }
    __device__ double normal()
    {
        const double u1 = uniform(), u2 = uniform();
        return box_muller_cos(u1, u2);
    }
};

__device__ inline void normal_pair(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t stream, uint32_t j, double& z0, double& z1)
{
    const U4 r = philox_call(seed, chain, iter, stream, j);
    const Pair z = box_muller_pair(u53(r.x, r.y), u53(r.z, r.w));
    z0 = z.a; z1 = z.b;
}

enum { NONE = 0, INSERT = 1, DELETE = 2, PERTURB = 3 };

// Key of chain b's random streams: its global index in the survey, so that the chains do not depend on how the survey is
// sharded or on the rows of a block being re-packed (chain_id), see RjOpt.first_chain.
__device__ inline uint32_t chain_key(const RjOpt& o, const gbp_rj_chains& c, int b)
{
    return (uint32_t)((c.chain_id != nullptr && b < c.B) ? (uint64_t)c.chain_id[b] : o.first_chain + (uint64_t)b);
}


constexpr double INF = __builtin_huge_val();
constexpr double LOG_2PI = 1.8378770664093454835606594728112;

// Error levels of one chain: up to 4 relative and 4 additive groups (TDEM: one relative level per system x component, one
// additive level per system, DataPoint.py:268-282 / TdemDataPoint.py:361-365; FDEM with one system: one of each).
struct Levels { double rel[4], add[4]; };

__device__ inline Levels load_levels(const RjOpt& o, const double* rel, const double* add, size_t b)
{
    Levels e;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        e.rel[g] = g < o.n_rel_groups ? rel[b * o.n_rel_groups + g] : 1.0;
        e.add[g] = g < o.n_add_groups ? add[b * o.n_add_groups + g] : 1.0;
    }
    return e;
}

// a ? x : y, entry by entry.  (The conditional operator on two Levels OBJECTS yields an lvalue -- a pointer chosen by a select -- and
// both objects then live in scratch, 144 B per lane in the accept stages: every read of an error level a trip to memory.)
__device__ inline Levels select_levels(bool a, const Levels& x, const Levels& y)
{
    Levels r;
#pragma unroll
    for (int g = 0; g < 4; ++g) { r.rel[g] = a ? x.rel[g] : y.rel[g]; r.add[g] = a ? x.add[g] : y.add[g]; }
    return r;
}

// entry g of a Levels array.  By value: selects on four loaded VALUES; handed a pointer, the compiler turns the selects into one load at
// a selected address -- a dynamically indexed array, i.e. scratch, on every channel of the accept stages' misfit loop.
__device__ inline double pick4(const double (&v)[4], int g)
{
    const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    return g == 0 ? v0 : (g == 1 ? v1 : (g == 2 ? v2 : v3));
}

// variance of channel n with datum d: (rel_g d)^2 + (add_g' add_scale_n)^2
__device__ inline double variance_at(const gbp_rj_chains& c, const Levels& e, double d, int n)
{
    const double rd = pick4(e.rel, c.rel_group != nullptr ? c.rel_group[n] : 0) * d;
    double an = pick4(e.add, c.add_group != nullptr ? c.add_group[n] : 0);
    if (c.add_scale != nullptr) an *= c.add_scale[n];
    return rd * rd + an * an;
}

// Joint proposal of the G levels of one kind (StatArray.propose with a multivariate log-normal proposal of diagonal
// covariance, statistics/StatArray.py:578-638): all are redrawn while any is outside its prior; the current values are
// kept at the 10th redraw.
__device__ inline void propose_levels(Rng& r, const double* cur, int G, const double* sd, const double* llo, const double* lhi, double* out)
{
    double x[4];
    bool ok;
    auto draw = [&]() {
        ok = true;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < G) {
                x[g] = rj_log(cur[g]) + sd[g] * r.normal();
                ok = ok && x[g] >= llo[g] && x[g] <= lhi[g];
            }
    };
    draw();
    int tries = 0;
    while (!ok) {
        draw();
        if (++tries == 10) {
#pragma unroll
            for (int g = 0; g < 4; ++g) out[g] = cur[g];
            return;
        }
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g] = g < G ? rj_exp(x[g]) : cur[g];
}

__device__ inline int bucket_of(int k) { return k <= 8 ? 0 : 1; }

// The structural move of one chain (RectilinearMesh1D.perturb :1018-1118).  edge(j): interface j of the current model;
// below(depth): number of interfaces shallower than depth.
template <class EdgeAt, class CountBelow>
__device__ inline void choose_move(const RjOpt& o, Rng& r, int k, bool idle, EdgeAt edge, CountBelow below, int& action,
                                   int& idx, double& val)
{
    const int K = o.max_layers;
    const double lo = o.log_min_edge, hi = o.log_max_edge, mw = o.min_width;
    action = NONE; idx = 0; val = 0.0;
    bool done = idle;
    for (int round = 0; round < 8 && !done; ++round) {          // redraw the event when the tries run out
        const double pb = (k == K) ? 0.0 : o.p_birth, pd = (k == 1) ? 0.0 : o.p_death, pp = (k == 1) ? 0.0 : o.p_perturb;
        const double u = r.uniform() * (pb + pd + pp + o.p_none);
        if (u < pb) {                                           // birth (:1061-1081); the reference's 10th try always fails
            for (int t = 0; t < 9; ++t) {
                const double depth = rj_exp(lo + r.uniform() * (hi - lo));
                const int pos = below(depth);
                const double prev = pos > 0 ? edge(pos - 1) : 0.0, next = pos < k - 1 ? edge(pos) : INF;
                if (depth - prev > mw && next - depth > mw) { action = INSERT; idx = pos + 1; val = depth; done = true; break; }
            }
        } else if (u < pb + pd) {                               // death (:1083-1087)
            const int i = (int)floor(r.uniform() * (double)(k - 1));
            idx = min(i, k - 2) + 1; action = DELETE; done = true;
        } else if (u < pb + pd + pp) {                          // perturb (:1089-1118)
            for (int t = 0; t < 9; ++t) {
                const int i = min((int)floor(1.0 + r.uniform() * (double)(k - 1)), k - 1);
                const double n = r.normal();
                const double dz = (n > 0.0 ? 1.0 : (n < 0.0 ? -1.0 : 0.0)) * mw * r.uniform();
                const int ii = i - 1;
                const double ne = edge(ii) + dz;
                const double prev = ii > 0 ? edge(ii - 1) : 0.0, next = ii < k - 2 ? edge(ii + 1) : INF;
                const double first = ii == 0 ? ne : edge(0), last = ii == k - 2 ? ne : edge(k - 2);
                if (ne - prev > mw && next - ne > mw && first > o.min_edge && last < o.max_edge) {
                    action = PERTURB; idx = i; val = dz; done = true; break;
                }
            }
        } else {
            done = true;
        }
    }
}

// Remapping rule (Model.perturb_structure: insert copies the layer above, delete averages the merged pair) for entry j,
// given the entry itself and its neighbours in the current model.
__device__ inline void remap_entry(int action, int idx, double val, int kr, int j, double e_j, double e_up, double e_dn, double s_j,
                                   double s_up, double s_dn, double& ev, double& sv)
{
    ev = INF; sv = 1.0;
    if (action == INSERT) {
        if (j < kr - 1) ev = j < idx - 1 ? e_j : (j == idx - 1 ? val : e_up);
        if (j < kr) sv = j < idx ? s_j : s_up;
    } else if (action == DELETE) {
        if (j < kr - 1) ev = j < idx - 1 ? e_j : e_dn;
        if (j < kr) sv = j < idx - 1 ? s_j : (j == idx - 1 ? 0.5 * (s_j + s_dn) : s_dn);
    } else {
        if (j < kr - 1) ev = e_j + ((action == PERTURB && j == idx - 1) ? val : 0.0);
        if (j < kr) sv = s_j;
    }
}

__device__ inline void write_move(const RjOpt& o, const gbp_rj_chains& c, Rng& r, int b, int action, int kr, const Levels* now = nullptr,
                                  bool have_now = false)
{   // per-chain scalars: layer counts for the kernels that follow (row 0: all, rows 1-2: by bucket), error proposals
    // (`now`, have_now: the chain's current error levels when the caller holds them -- the fused accept + proposal launch --, else read from
    //  memory.  A flag beside a pointer that is always valid, not a null pointer: a pointer chosen by a select keeps the caller's object in scratch.)
    const int bk = bucket_of(kr);
    const bool jump = action == INSERT || action == DELETE;
    for (int i = 0; i < 3; ++i) {
        const bool mine = i == 0 || bk == i - 1;
        c.nl_a[(size_t)i * c.B + b] = (action != NONE && mine) ? kr : 0;
        c.nl_c[(size_t)i * c.B + b] = (jump && mine) ? kr : 0;
    }
    c.nl_b[b] = jump ? 0 : kr;          // jump proposals get their prediction from the Jacobian pass instead
    c.action[b] = action;
    c.k_r[b] = kr;
    // the data point's height (Point.perturb, reached first through DataPoint.perturb's super().perturb(): pointcloud/Point.py
    // :614-621): random walk redrawn while outside the uniform prior, the current height kept at the 10th redraw
    if (o.solve_height) {
        const double cur_h = c.height[b], lo_h = c.height0[b] - o.height_half_width, hi_h = c.height0[b] + o.height_half_width;
        double x = cur_h + o.height_scale * r.normal();
        int tries = 0;
        while (!(x >= lo_h && x <= hi_h)) {
            x = cur_h + o.height_scale * r.normal();
            if (++tries == 10) { x = cur_h; break; }
        }
        c.height_p[b] = x;
    }
    // error levels (DataPoint.perturb: relative then additive)
    const Levels cur = have_now ? *now : load_levels(o, c.rel, c.add, (size_t)b);
    Levels out = cur;
    if (o.solve_relative_error) propose_levels(r, cur.rel, o.n_rel_groups, o.rel_sd, o.log_rel_min, o.log_rel_max, out.rel);
    if (o.solve_additive_error && !o.additive_independent)
        propose_levels(r, cur.add, o.n_add_groups, o.add_sd, o.log_add_min, o.log_add_max, out.add);
    if (o.solve_additive_error && o.additive_independent) {      // Tempest's multipliers: one draw about a fixed centre (gbp_rj_options)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            if (g < o.n_add_groups) out.add[g] = rj_exp(rj_log(o.add_centre[g]) + o.add_sd[g] * r.normal());
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < o.n_rel_groups) c.rel_p[(size_t)b * o.n_rel_groups + g] = out.rel[g];
        if (g < o.n_add_groups) c.add_p[(size_t)b * o.n_add_groups + g] = out.add[g];
    }
}

// Small blocks of soundings: one wave per chain (4 chains per workgroup).  Lane j holds interface j and layer j, every
// lane runs the same draws (wave-uniform control flow), neighbour look-ups are cross-lane reads, rows are written coalesced.
// (k, e_row, s_row: the chain's current model -- c.k[b] and its rows of c.edges / c.sigma, or, right behind an accept stage in the same
//  kernel, the layer count and the rows that stage has just made current: the fused step never reads back what it has just written)
__device__ __forceinline__ int propose_wave_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int b, int lane, int k,
                                                 const double* e_row, const double* s_row)
{   // returns the proposal's layer count
    const int K = o.max_layers;
    const double ej = lane < k - 1 ? e_row[lane] : INF;
    const double sj = lane < k ? s_row[lane] : 1.0;
    Rng r(o.seed, chain_key(o, c, b), iter, 0);
    int action, idx;
    double val;
    choose_move(o, r, k, o.schedule == 1 && c.status[b] != 0, [&](int j) { return __shfl(ej, j, 64); },
                [&](double depth) { return (int)__popcll(__ballot(ej < depth)); }, action, idx, val);
    const int kr = k + (action == INSERT) - (action == DELETE);
    const int up = max(lane - 1, 0), dn = min(lane + 1, 63);
    double ev, sv;
    remap_entry(action, idx, val, kr, lane, ej, __shfl(ej, up, 64), __shfl(ej, dn, 64), sj, __shfl(sj, up, 64), __shfl(sj, dn, 64), ev, sv);
    const double ev_up = __shfl(ev, up, 64);
    if (lane < K) {
        c.edges_r[(size_t)b * K + lane] = ev;
        c.sigma_r[(size_t)b * K + lane] = sv;
        c.thk_r[(size_t)b * K + lane] = lane < kr - 1 ? ev - (lane > 0 ? ev_up : 0.0) : 0.0;
    }
    Rng r0 = r;                                                  // every lane continues the same stream; lane 0 writes
    if (lane == 0) write_move(o, c, r0, b, action, kr);
    return kr;
}
__device__ __forceinline__ void propose_wave_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int b, int lane)
{
    const int K = o.max_layers;
    (void)propose_wave_body(o, c, iter, b, lane, c.k[b], c.edges + (size_t)b * K, c.sigma + (size_t)b * K);
}

// The same proposal by the 8 lanes that hold a chain in the packed stages (models of at most 7 layers: a birth then still fits the group;
// an 8-layer model takes the thread-per-chain rows on the group's first lane).  Lane j holds interface j and layer j; every lane of the
// group runs the same draws; neighbour look-ups are reads within the group.  Same draws, same remapped rows, same records as the other
// proposal kernels (tests/test_rjmcmc_gpu.py holds the drivers that use them against each other bit for bit).
// (have: the chain's entries i of the current rows and its error levels are in registers -- e_reg, s_reg, now)
__device__ __forceinline__ int propose8_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int lane, int b, int k,
                                             const double* e_row, const double* s_row, bool have = false, double e_reg = 0.0,
                                             double s_reg = 0.0, const Levels* now = nullptr, int status = -1)
{   // returns the proposal's layer count (to every lane of the group)
    const int i = lane & 7, base = lane & ~7, K = o.max_layers;
    const bool idle = o.schedule == 1 && (status >= 0 ? status : c.status[b]) != 0;
    if (k > 7 || K < 8) {                                        // (group-uniform)
        int kr_out = 0;
        if (i == 0) {
            Rng r(o.seed, chain_key(o, c, b), iter, 0);
            int action, idx;
            double val;
            choose_move(o, r, k, idle, [&](int j) { return e_row[j]; },
                        [&](double depth) { int pos = 0; while (pos < k - 1 && e_row[pos] < depth) ++pos; return pos; }, action, idx, val);
            const int kr = k + (action == INSERT) - (action == DELETE);
            double above = 0.0;
            // (a rolling window over the rows, as propose_rows: behind an accepted move e_row IS the chain's row of edges_r, which this
            //  loop rewrites -- entry j is written after entry j + 1 has been read, and the entry above comes from the window)
            double e_j = 0 < k - 1 ? e_row[0] : INF, s_j = 0 < k ? s_row[0] : 1.0, e_up = e_j, s_up = s_j;
            for (int j = 0; j < K; ++j) {
                const double e_dn = j + 1 < k - 1 ? e_row[j + 1] : INF, s_dn = j + 1 < k ? s_row[j + 1] : 1.0;
                double ev, sv;
                remap_entry(action, idx, val, kr, j, e_j, e_up, j + 1 < K ? e_dn : e_j, s_j, s_up, j + 1 < K ? s_dn : s_j, ev, sv);
                c.edges_r[(size_t)b * K + j] = ev;
                c.sigma_r[(size_t)b * K + j] = sv;
                c.thk_r[(size_t)b * K + j] = j < kr - 1 ? ev - above : 0.0;
                above = ev;
                e_up = e_j; s_up = s_j; e_j = e_dn; s_j = s_dn;
            }
            write_move(o, c, r, b, action, kr, now, have);
            kr_out = kr;
        }
        return __shfl(kr_out, base, 64);
    }
    const double ej = i < k - 1 ? (have ? e_reg : e_row[i]) : INF;
    const double sj = i < k ? (have ? s_reg : s_row[i]) : 1.0;
    Rng r(o.seed, chain_key(o, c, b), iter, 0);
    int action, idx;
    double val;
    choose_move(o, r, k, idle, [&](int j) { return __shfl(ej, base + j, 64); },
                [&](double depth) { return (int)__popcll((__ballot(ej < depth) >> base) & 0xFFull); }, action, idx, val);
    const int kr = k + (action == INSERT) - (action == DELETE);
    const int up = max(i - 1, 0), dn = min(i + 1, 7);
    const double e_up = __shfl(ej, base + up, 64), s_up = __shfl(sj, base + up, 64);
    const double e_dn_ = __shfl(ej, base + dn, 64), s_dn_ = __shfl(sj, base + dn, 64);
    const double e_dn = i == 7 ? INF : e_dn_, s_dn = i == 7 ? 1.0 : s_dn_;      // (entry 8 of a model of at most 7 layers)
    double ev, sv;
    remap_entry(action, idx, val, kr, i, ej, e_up, e_dn, sj, s_up, s_dn, ev, sv);
    const double ev_up = __shfl(ev, base + up, 64);
    c.edges_r[(size_t)b * K + i] = ev;
    c.sigma_r[(size_t)b * K + i] = sv;
    c.thk_r[(size_t)b * K + i] = i < kr - 1 ? ev - (i > 0 ? ev_up : 0.0) : 0.0;
    for (int j = i + 8; j < K; j += 8) {                         // (beyond the group: kr <= 8 layers -- the rows' empty entries)
        c.edges_r[(size_t)b * K + j] = INF;
        c.sigma_r[(size_t)b * K + j] = 1.0;
        c.thk_r[(size_t)b * K + j] = 0.0;
    }
    Rng r0 = r;
    if (i == 0) write_move(o, c, r0, b, action, kr, now, have);
    return kr;
}

__global__ __launch_bounds__(256) void k_rj_propose_wave(RjOpt o, gbp_rj_chains c, uint32_t iter)
{
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= c.B) return;
    propose_wave_body(o, c, iter, b, threadIdx.x & 63);
}

// Large blocks: one thread per chain (the wave version would spend 64 lanes on every scalar decision).
// e, s: the chain's current rows; er, sr, tr: where the remapped rows go (er / sr may be e / s themselves: entry j is written
// after entry j + 1 has been read, and choose_move is done with e before the first write).  PAIRS: rows in global memory are
// written two entries (16 bytes) at a time when they are 16-byte aligned (K even).
template <bool PAIRS>
__device__ __forceinline__ void propose_rows(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int b, const double* e,
                                             const double* s, double* er, double* sr, double* tr)
{
    const int K = o.max_layers;
    const int k = c.k[b];
    Rng r(o.seed, chain_key(o, c, b), iter, 0);
    int action, idx;
    double val;
    choose_move(o, r, k, o.schedule == 1 && c.status[b] != 0, [&](int j) { return e[j]; },
                [&](double depth) { int pos = 0; while (pos < k - 1 && e[pos] < depth) ++pos; return pos; }, action, idx, val);
    const int kr = k + (action == INSERT) - (action == DELETE);
    double above = 0.0;
    double e_j = 0 < k - 1 ? e[0] : INF, s_j = 0 < k ? s[0] : 1.0, e_up = e_j, s_up = s_j;   // rolling window over the rows
    const bool pairs = PAIRS && (K & 1) == 0;
    double ev0 = 0.0, sv0 = 0.0, tv0 = 0.0;
    for (int j = 0; j < K; ++j) {
        const double e_dn = j + 1 < k - 1 ? e[j + 1] : INF, s_dn = j + 1 < k ? s[j + 1] : 1.0;
        double ev, sv;
        remap_entry(action, idx, val, kr, j, e_j, e_up, j + 1 < K ? e_dn : e_j, s_j, s_up, j + 1 < K ? s_dn : s_j, ev, sv);
        const double tv = j < kr - 1 ? ev - above : 0.0;
        if (!pairs) {
            er[j] = ev; sr[j] = sv; tr[j] = tv;
        } else if (j & 1) {
            *reinterpret_cast<double2*>(er + j - 1) = make_double2(ev0, ev);
            *reinterpret_cast<double2*>(sr + j - 1) = make_double2(sv0, sv);
            *reinterpret_cast<double2*>(tr + j - 1) = make_double2(tv0, tv);
        } else {
            ev0 = ev; sv0 = sv; tv0 = tv;
        }
        above = ev;
        e_up = e_j; s_up = s_j; e_j = e_dn; s_j = s_dn;
    }
    write_move(o, c, r, b, action, kr);
}

__device__ __forceinline__ void propose_thread_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int b)
{
    const int K = o.max_layers;
    propose_rows<true>(o, c, iter, b, c.edges + (size_t)b * K, c.sigma + (size_t)b * K, c.edges_r + (size_t)b * K,
                       c.sigma_r + (size_t)b * K, c.thk_r + (size_t)b * K);
}

__global__ __launch_bounds__(128) void k_rj_propose_thread(RjOpt o, gbp_rj_chains c, uint32_t iter)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    propose_thread_body(o, c, iter, b);
}

// The same, with the rows of the workgroup's 64 chains staged through LDS: a thread's row is 8 K bytes from its neighbour's,
// so in the kernel above every load of the serial loop over the row is a separate trip to memory (the loop is latency bound)
// and every store instruction writes 64 partial cache lines (2.7 x write amplification measured, profiles/r2).  Here the 64
// rows, contiguous in memory, come in and go out as whole cache lines and the serial loop runs on LDS (row stride K | 1
// doubles: the 64 lanes of a column access fall on distinct bank pairs).  Same draws, same values.
#define GBP_RJ_PROPOSE_ROWS 64
#define GBP_RJ_PROPOSE_THREADS 256
__global__ __launch_bounds__(GBP_RJ_PROPOSE_THREADS) void k_rj_propose_staged(RjOpt o, gbp_rj_chains c, uint32_t iter)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int K = o.max_layers, KS = K | 1, t = threadIdx.x;
    const int b0 = blockIdx.x * GBP_RJ_PROPOSE_ROWS, nb = min(GBP_RJ_PROPOSE_ROWS, c.B - b0);
    double* se = reinterpret_cast<double*>(sh_dyn);
    double* ss = se + GBP_RJ_PROPOSE_ROWS * KS;
    double* st = ss + GBP_RJ_PROPOSE_ROWS * KS;
    const size_t g0 = (size_t)b0 * K;                              // (b0 is a multiple of 64: 16-byte aligned for any K)
    const int n_el = nb * K, n_pair = (n_el + 1) >> 1;
    // copy in: all four waves, 16 bytes per lane and array, four loads in flight per lane before the first LDS write
    for (int p0 = t; p0 < n_pair; p0 += 4 * GBP_RJ_PROPOSE_THREADS) {
        double2 ve[4], vs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * GBP_RJ_PROPOSE_THREADS;
            if (p < n_pair) {
                if (2 * p + 1 < n_el) {
                    ve[u] = *reinterpret_cast<const double2*>(c.edges + g0 + 2 * p);
                    vs[u] = *reinterpret_cast<const double2*>(c.sigma + g0 + 2 * p);
                } else {
                    ve[u] = make_double2(c.edges[g0 + 2 * p], 0.0);
                    vs[u] = make_double2(c.sigma[g0 + 2 * p], 0.0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + u * GBP_RJ_PROPOSE_THREADS;
            if (p < n_pair) {
                const int i0 = 2 * p, r0 = i0 / K, c0 = i0 - r0 * K;
                se[r0 * KS + c0] = ve[u].x; ss[r0 * KS + c0] = vs[u].x;
                if (i0 + 1 < n_el) {
                    const int r1 = c0 + 1 < K ? r0 : r0 + 1, c1 = c0 + 1 < K ? c0 + 1 : 0;
                    se[r1 * KS + c1] = ve[u].y; ss[r1 * KS + c1] = vs[u].y;
                }
            }
        }
    }
    __syncthreads();
    if (t < nb) propose_rows<false>(o, c, iter, b0 + t, se + t * KS, ss + t * KS, se + t * KS, ss + t * KS, st + t * KS);
#ifdef GBP_RJ_PROPOSE_DELAY_TICKS
    {   // (sensitivity builds only: is an iteration bound by the dependency chain of a sub-block or by the throughput of the physics launches?)
        const long long d0 = (long long)wall_clock64();
        while ((long long)wall_clock64() - d0 < GBP_RJ_PROPOSE_DELAY_TICKS) {}
    }
#endif
    __syncthreads();
    for (int p = t; p < n_pair; p += GBP_RJ_PROPOSE_THREADS) {
        const int i0 = 2 * p, r0 = i0 / K, c0 = i0 - r0 * K;
        const int r1 = c0 + 1 < K ? r0 : r0 + 1, c1 = c0 + 1 < K ? c0 + 1 : 0;
        if (i0 + 1 < n_el) {
            *reinterpret_cast<double2*>(c.edges_r + g0 + i0) = make_double2(se[r0 * KS + c0], se[r1 * KS + c1]);
            *reinterpret_cast<double2*>(c.sigma_r + g0 + i0) = make_double2(ss[r0 * KS + c0], ss[r1 * KS + c1]);
            *reinterpret_cast<double2*>(c.thk_r + g0 + i0) = make_double2(st[r0 * KS + c0], st[r1 * KS + c1]);
        } else {
            c.edges_r[g0 + i0] = se[r0 * KS + c0];
            c.sigma_r[g0 + i0] = ss[r0 * KS + c0];
            c.thk_r[g0 + i0] = st[r0 * KS + c0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// shared per-chain algebra (one wave = one chain, matrices in LDS with row stride K + 1)
// ---------------------------------------------------------------------------------------------------------------
// t2[j] = gradient_precision * Wz[j, j]^2 of RectilinearMesh1D.gradient_operator (mesh/RectilinearMesh1D.py:747-786)
__device__ inline double width_x(const double* e, int k, int j)
{
    if (j < k - 1) return e[j] - (j > 0 ? e[j - 1] : 0.0);
    if (k == 2) return e[0];
    return (e[k - 2] - (k > 2 ? e[k - 3] : 0.0)) + e[k - 2];
}

__device__ inline void prior_t2(const RjOpt& o, const double* e, int k, int lane, double* t2)
{
    if (lane < k - 1) {
        const double c2c = 0.5 * (width_x(e, k, lane) + width_x(e, k, lane + 1)) * (double)(k - 1);
        t2[lane] = o.solve_gradient ? o.gradient_precision / (c2c * c2c) : 0.0;
    }
}

// (Wm'Wm v)_i for the tridiagonal prior operator (Model.prior_derivative, model/Model.py:421-430)
__device__ inline double prior_apply(const RjOpt& o, const double* t2, int k, int i, const double* v)
{
    if (k == 1) return (o.value_precision + (o.solve_gradient ? o.gradient_precision : 0.0)) * v[0];
    const double up = i > 0 ? t2[i - 1] : 0.0, dn = i < k - 1 ? t2[i] : 0.0;
    double y = (o.value_precision + up + dn) * v[i];
    if (i > 0) y -= up * v[i - 1];
    if (i < k - 1) y -= dn * v[i + 1];
    return y;
}

__device__ inline double prior_entry(const RjOpt& o, const double* t2, int k, int i, int j)   // j <= i
{
    if (k == 1) return o.value_precision + (o.solve_gradient ? o.gradient_precision : 0.0);
    if (i == j) return o.value_precision + (i > 0 ? t2[i - 1] : 0.0) + (i < k - 1 ? t2[i] : 0.0);
    return j == i - 1 ? -t2[j] : 0.0;
}

// data weights with the error levels (rel, add): P = active / std^2, PR = P * (pred - data)  (DataPoint.py:268-282, 340-349)
// (variance_at: relative level of the channel's group, additive level of its group times add_scale[n] -- TdemDataPoint.std)
__device__ inline void data_weights(const gbp_rj_chains& c, const double* data, const double* pred, const Levels& e, int N, int lane,
                                    int stride, double* P, double* PR)
{
    for (int n = lane; n < N; n += stride) {
        const double d = data[n];
        const bool act = d > 0.0;
        const double w = act ? 1.0 / variance_at(c, e, d, n) : 0.0;
        P[n] = w;
        PR[n] = act ? w * (pred[n] - d) : 0.0;
    }
}

// The packed (8 lanes per chain) stages are latency chains of one wave: what they cost is the number of DEPENDENT trips to memory,
// not bytes or flops (profiles/r3/summary_rjmcmc_8192.json: 0.04 - 0.05 VALU issue utilisation, ~25 cycles per instruction).  A loop
// `for n: x = row[n]; use(x)` is one trip per turn -- the compiler cannot hoist a load over the stores (or the branches) of the turn
// before.  The helpers below issue the loads of several turns back to back and keep the arithmetic, and its order, as it was.
//
// TRIPS = true selects them (the lock-step launches of small and medium blocks, whose iteration is this latency chain); the
// persistent kernel (rows in LDS: nothing to wait for, the extra selects only cost) and the large blocks (throughput bound: the extra
// registers cost occupancy) take the plain loops -- measured in gbp_rj_newton below.
//
// f(n, M[n][col]) for the rows n = 0 .. N - 1 of a row-major [N][K] matrix in ascending order; `on` == false: f(n, 0.0), nothing read.
template <bool TRIPS, class F>
__device__ __forceinline__ void for_column(const double* M, int K, int N, int col, bool on, F f)
{
    if constexpr (!TRIPS) {
        for (int n = 0; n < N; ++n) f(n, on ? M[(size_t)n * K + col] : 0.0);
    } else {
        const int cc = on ? col : 0;                         // (a column every lane may read: the loads are unconditional)
        double cur[8], nxt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) cur[u] = M[(size_t)min(u, N - 1) * K + cc];
        for (int n0 = 0; n0 < N; n0 += 8) {                  // the next eight rows are in flight while these are used
            if (n0 + 8 < N) {
#pragma unroll
                for (int u = 0; u < 8; ++u) nxt[u] = M[(size_t)min(n0 + 8 + u, N - 1) * K + cc];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n0 + u < N) f(n0 + u, on ? cur[u] : 0.0);   // (wave-uniform bound)
#pragma unroll
            for (int u = 0; u < 8; ++u) cur[u] = nxt[u];
        }
    }
}

// f(n, data[n], pred[n], variance of channel n) for the channels n = i, i + 8, ... of one lane of a group of 8 (variance_at);
// TRIPS: four channels per trip to memory.
template <bool TRIPS, class F>
__device__ __forceinline__ void for_channels8(const gbp_rj_chains& c, const double* data, const double* pred, const Levels& e, int N, int i, F f)
{
    if constexpr (!TRIPS) {
        for (int n = i; n < N; n += 8) {
            const double d = data[n];
            f(n, d, d > 0.0 ? pred[n] : 0.0, d > 0.0 ? variance_at(c, e, d, n) : 1.0);
        }
    } else {
        for (int n0 = i; n0 < N; n0 += 32) {
            double d[4], p[4], as[4];
            int rg[4], ag[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = min(n0 + 8 * u, N - 1);
                d[u] = data[n]; p[u] = pred[n];
                rg[u] = c.rel_group != nullptr ? c.rel_group[n] : 0;
                ag[u] = c.add_group != nullptr ? c.add_group[n] : 0;
                as[u] = c.add_scale != nullptr ? c.add_scale[n] : 1.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int n = n0 + 8 * u;
                if (n < N) {
                    const double rd = pick4(e.rel, rg[u]) * d[u];
                    double an = pick4(e.add, ag[u]);
                    if (c.add_scale != nullptr) an *= as[u];
                    f(n, d[u], p[u], rd * rd + an * an);
                }
            }
        }
    }
}

// data_weights for a group of 8 lanes
template <bool TRIPS>
__device__ __forceinline__ void data_weights8(const gbp_rj_chains& c, const double* data, const double* pred, const Levels& e, int N, int i,
                                              double* P, double* PR)
{
    for_channels8<TRIPS>(c, data, pred, e, N, i, [&](int n, double d, double p, double var) {
        const bool act = d > 0.0;
        const double w = act ? 1.0 / var : 0.0;
        P[n] = w;
        PR[n] = act ? w * (p - d) : 0.0;
    });
}

// copy of the entries j = i, i + 8, ... < n of one or two rows (s2 == nullptr: one); TRIPS: the loads of four turns go out together,
// then their stores (turn by turn the compiler may not move a load above a store that could alias it: one trip per turn)
template <bool TRIPS>
__device__ __forceinline__ void copy_strided8(double* d1, const double* s1, double* d2, const double* s2, int n, int i)
{
    if constexpr (!TRIPS) {
        for (int j = i; j < n; j += 8) { d1[j] = s1[j]; if (s2 != nullptr) d2[j] = s2[j]; }
    } else {
        for (int j0 = i; j0 < n; j0 += 32) {
            double v1[4], v2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int j = min(j0 + 8 * u, n - 1); v1[u] = s1[j]; v2[u] = s2 != nullptr ? s2[j] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + 8 * u;
                if (j < n) { d1[j] = v1[u]; if (s2 != nullptr) d2[j] = v2[u]; }
            }
        }
    }
}

// solve C C' x = g in place (g -> x), C lower in A (row stride KS); lane-parallel column sweeps
__device__ inline void chol_solve(const double* A, int KS, int k, int lane, double* g, bool forward, bool backward)
{
    if (forward)
        for (int j = 0; j < k; ++j) {
            if (lane == j) g[j] = g[j] / A[j * KS + j];
            wave_sync();
            if (lane > j && lane < k) g[lane] -= A[lane * KS + j] * g[j];
            wave_sync();
        }
    if (backward)
        for (int j = k - 1; j >= 0; --j) {
            if (lane == j) g[j] = g[j] / A[j * KS + j];
            wave_sync();
            if (lane < j) g[lane] -= A[j * KS + lane] * g[j];
            wave_sync();
        }
}

__device__ inline double wave_sum(double v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// LDS carve-up of both wave-per-chain kernels
struct Lds {
    double *A, *g, *t2, *v, *w, *P, *PR, *row;
    __device__ Lds(unsigned char* base, int K, int N)
    {
        double* p = reinterpret_cast<double*>(base);
        A = p; p += (size_t)K * (K + 1);
        g = p; p += K; t2 = p; p += K; v = p; p += K; w = p; p += K; row = p; p += K;
        P = p; p += N; PR = p;
    }
    static size_t bytes(int K, int N) { return ((size_t)K * (K + 1) + 5 * (size_t)K + 2 * (size_t)N) * sizeof(double); }
};

__device__ __forceinline__ void newton_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int min_k, int b, int lane,
                                            unsigned char* sh_dyn)
{   // Model.stochastic_newton_perturbation (model/Model.py:368-419): precision = J'PJ + Wm'Wm at the remapped model,
    // mean = ln sigma - alpha * precision^-1 g, sample ~ N(mean, precision^-1) = mean + C^-T z with precision = C C'
    // One wave per chain; chains with at most min_k layers are left to newton8_body.
    const int K = o.max_layers, N = o.n_channels, KS = K + 1;
    Lds s(sh_dyn, K, N);
    const int k = c.k_r[b];
    if (k <= min_k) return;
    const bool changed = c.action[b] != NONE;
    const double* J = (changed ? c.J_r : c.J) + (size_t)b * N * K;
    const double* pred = (changed ? c.pred_r : c.pred) + (size_t)b * N;
    const double* e = c.edges_r + (size_t)b * K;
    const double* sr = c.sigma_r + (size_t)b * K;
    data_weights(c, c.data + (size_t)b * N, pred, load_levels(o, c.rel, c.add, (size_t)b), N, lane, 64, s.P, s.PR);
    prior_t2(o, e, k, lane, s.t2);
    const double lmp = c.log_mean_prior[b];
    const double ls = lane < k ? rj_log(sr[lane]) : 0.0;
    if (lane < k) s.v[lane] = ls - lmp;
    wave_sync();
    // row `lane` of J'PJ in column blocks of 8; J[n, j] is a wave-uniform address (scalar loads), J[n, lane] coalesced
    const int li = min(lane, k - 1);
    double gi = 0.0;
    for (int n = 0; n < N; ++n) gi += J[(size_t)n * K + li] * s.PR[n];
    for (int j0 = 0; j0 < k; j0 += 8) {
        double acc[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int n = 0; n < N; ++n) {
            const double* Jn = J + (size_t)n * K;
            const double jp = Jn[li] * s.P[n];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) acc[jj] += jp * Jn[min(j0 + jj, K - 1)];
        }
        if (lane < k) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int j = j0 + jj;
                if (j <= lane) s.A[lane * KS + j] = acc[jj] + prior_entry(o, s.t2, k, lane, j);
            }
        }
    }
    if (lane < k) s.g[lane] = gi + prior_apply(o, s.t2, k, lane, s.v);
    wave_sync();
    for (int j = 0; j < k; ++j) {                                // Cholesky, lower, in place
        if (lane == j) {
            double d = s.A[j * KS + j];
            for (int m = 0; m < j; ++m) d -= s.A[j * KS + m] * s.A[j * KS + m];
            s.A[j * KS + j] = sqrt(d);
        }
        wave_sync();
        if (lane > j && lane < k) {
            double d = s.A[lane * KS + j];
            for (int m = 0; m < j; ++m) d -= s.A[lane * KS + m] * s.A[j * KS + m];
            s.A[lane * KS + j] = d / s.A[j * KS + j];
        }
        wave_sync();
    }
    double* C = c.chol + (size_t)b * K * K;
    if (lane < k)
        for (int j = 0; j <= lane; ++j) C[(size_t)lane * K + j] = s.A[lane * KS + j];
    chol_solve(s.A, KS, k, lane, s.g, true, true);
    if (lane < 32) {
        double z0, z1;
        normal_pair(o.seed, chain_key(o, c, b), iter, 1, (uint32_t)lane, z0, z1);
        if (2 * lane < K) s.w[2 * lane] = z0;
        if (2 * lane + 1 < K) s.w[2 * lane + 1] = z1;
    }
    wave_sync();
    chol_solve(s.A, KS, k, lane, s.w, false, true);
    if (lane < K) {
        const double lp = lane < k ? (ls - o.alpha * s.g[lane]) + s.w[lane] : 0.0;
        c.log_prop[(size_t)b * K + lane] = lp;
        c.sigma_p[(size_t)b * K + lane] = lane < k ? rj_exp(lp) : 1.0;
    }
}

// The wave-per-chain bodies are for the models of more than 8 layers -- under a percent of the chains of a survey.  One workgroup per
// chain meant a launch of B workgroups that exit at once (5 - 6 us of every iteration at 2 731 chains, profiles/r4); here a wave SCANS 64
// chains (one coalesced read of their layer counts) and runs the body for those that are its own: B / 64 workgroups (round 5).
template <class Body>
__device__ __forceinline__ void for_deep_chains(int group, bool mine, Body body)
{
    const int b0 = group * 64;
    unsigned long long todo = __ballot(mine);
    while (todo != 0ull) {
        const int l = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        body(b0 + l);
        wave_sync();                                             // (the next chain reuses the LDS block)
    }
}

__global__ __launch_bounds__(64) void k_rj_newton(RjOpt o, gbp_rj_chains c, uint32_t iter, int min_k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int b = (int)blockIdx.x * 64 + (int)threadIdx.x;
    for_deep_chains(blockIdx.x, b < c.B && c.k_r[min(b, c.B - 1)] > min_k,
                    [&](int bb) { newton_body(o, c, iter, min_k, bb, threadIdx.x, sh_dyn); });
}

// The same for chains with at most 8 layers -- the common case -- packed 8 lanes per chain, 8 chains per wave: row i of the
// 8 x 8 system lives in the registers of lane i of the chain's group, columns of the Cholesky factor are passed around with
// cross-lane reads (lane j also keeps column j for the transposed solves), so there is no LDS traffic and no barrier in
// the factorisation or the substitutions.  Rows >= k are identity rows.
// Cross-lane reads within a chain's group of 8 lanes as DPP moves (VALU, a few cycles) instead of ds_bpermute (a trip through
// the LDS crossbar, ~150 cycles): the factorisation and the substitutions below are chains of ~60 DEPENDENT cross-lane reads,
// executed by one wave per SIMD in the small-block regime, so their latency is what an iteration costs.  Issued by all 64 lanes.
template <int CTRL>
__device__ __forceinline__ double dpp_mov64(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int J>   // the value lane J (0..7) of the caller's 8-lane group holds
__device__ __forceinline__ double group_bcast_c(double v)
{
    constexpr int q = J & 3;
    const double a = dpp_mov64<q | (q << 2) | (q << 4) | (q << 6)>(v);   // quad_perm [q,q,q,q]: every quad, its own lane q
    const double b = dpp_mov64<0x141>(a);                                // row_half_mirror: the other quad's lane q
    return (((int)threadIdx.x >> 2) & 1) == (J >> 2) ? a : b;
}
__device__ __forceinline__ double group_bcast(double v, int base, int j)   // j: a constant after unrolling
{
    (void)base;
    switch (j) {
        case 0: return group_bcast_c<0>(v);
        case 1: return group_bcast_c<1>(v);
        case 2: return group_bcast_c<2>(v);
        case 3: return group_bcast_c<3>(v);
        case 4: return group_bcast_c<4>(v);
        case 5: return group_bcast_c<5>(v);
        case 6: return group_bcast_c<6>(v);
        default: return group_bcast_c<7>(v);
    }
}
// the neighbour lanes' values (row_shr:1 / row_shl:1 -- within rows of 16 lanes; callers use them inside a group of 8 only:
// lane 0 of a group ignores `up`, lane 7 ignores `dn`)
__device__ __forceinline__ double lane_up(double v) { return dpp_mov64<0x111>(v); }
__device__ __forceinline__ double lane_dn(double v) { return dpp_mov64<0x101>(v); }

// Interface widths of RectilinearMesh1D.gradient_operator (width_x) from the group's registers: lane j holds interface j of the chain
// (+inf beyond the last) -- the same differences in the same order as width_x on the row in memory, without its dependent loads.
__device__ __forceinline__ double prior_t2_group(const RjOpt& o, double e_i, int k, int i, int base)
{
    const double e_up = lane_up(e_i), e_dn = lane_dn(e_i);
    const double e_km2 = __shfl(e_i, base + max(k - 2, 0), 64), e_km3 = __shfl(e_i, base + max(k - 3, 0), 64), e_0 = __shfl(e_i, base, 64);
    double t2 = 0.0;
    if (i < k - 1 && o.solve_gradient) {
        const double w_i = e_i - (i > 0 ? e_up : 0.0);                                   // width_x(e, k, i): i < k - 1
        double w_n;                                                                      // width_x(e, k, i + 1)
        if (i + 1 < k - 1) w_n = e_dn - e_i;
        else if (k == 2) w_n = e_0;
        else w_n = (e_km2 - (k > 2 ? e_km3 : 0.0)) + e_km2;
        const double c2c = 0.5 * (w_i + w_n) * (double)(k - 1);
        t2 = o.gradient_precision / (c2c * c2c);
    }
    return t2;
}

// `b`: the chain of this lane's 8-lane group, or >= c.B for an idle group; sh_dyn: P[8][N] | PR[8][N].
// KM: the algebra runs on the leading KM x KM block -- every chain of the wave has at most KM layers.  Rows and columns >= k are
// identity rows: their Cholesky column is a unit vector, they add 0 x (finite) to every substitution step, so leaving them out
// changes no bit of rows < k; KM = 8 is the full group.
template <int KM, bool TRIPS>
__device__ __forceinline__ void newton8_core(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int lane, int b,
                                             unsigned char* sh_dyn, int b_idle)
{
    const int slot = lane >> 3, i = lane & 7, base = lane & ~7;
    const int K = o.max_layers, N = o.n_channels;
    int k = b < c.B ? c.k_r[b] : 0;
    const bool live = k >= 1 && k <= 8;              // deeper chains: k_rj_newton.  No early exit: idle groups still take
    if (!live) k = 0;                                //   part in the cross-lane reads
    const size_t bb = b < c.B ? (size_t)b : (size_t)b_idle;   // idle groups read a valid row (never used); not a function of k:
    const bool changed = c.action[bb] != NONE;                 //   the loads below do not wait for the one above
    const double* J = (changed ? c.J_r : c.J) + bb * N * K;
    const double* pred = (changed ? c.pred_r : c.pred) + bb * N;
    const double* e = c.edges_r + bb * K;
    double* P = reinterpret_cast<double*>(sh_dyn) + (size_t)slot * 2 * N;
    double* PR = P + N;
    // Round 5: an iteration is the dependency chain of a sub-block's launches (docs/notes_r5.md), and this stage's 21 us were dependent
    // trips to memory -- data and prediction, then the interface row walked through pointers, then sigma, then the Jacobian column in
    // three batches -- with a library call (which waits for every outstanding load) between them.  ONE_TRIP (the lock-step launches, up to
    // GBP_RJ_COLUMN_ROWS channels): everything is requested in one batch behind the move, the normal draws are formed while it is in
    // flight (inlined: a call would wait for the loads first), the widths come from the group's registers.  Same arithmetic, same order.
    constexpr int NJ = GBP_RJ_COLUMN_ROWS;
    const bool one_trip = TRIPS && GBP_RJ_ONE_TRIP_NEWTON && N <= NJ;          // (wave-uniform)
    double t2 = 0.0, lmp, ls = 0.0, z0 = 0.0, z1 = 0.0;
    double Jc[TRIPS ? NJ : 1];
    if (one_trip) {
        const int ic = i < K ? i : 0;
        const Levels lev = load_levels(o, c.rel, c.add, bb);
        const double* data = c.data + bb * N;
        double d_[3], p_[3], as_[3];
        int rg_[3], ag_[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int n = min(i + 8 * u, N - 1);
            d_[u] = data[n]; p_[u] = pred[n];
            rg_[u] = c.rel_group != nullptr ? c.rel_group[n] : 0;
            ag_[u] = c.add_group != nullptr ? c.add_group[n] : 0;
            as_[u] = c.add_scale != nullptr ? c.add_scale[n] : 1.0;
        }
        const double e_i = e[ic];
        const double sr_i = c.sigma_r[bb * K + ic];
        lmp = c.log_mean_prior[bb];
#pragma unroll
        for (int n = 0; n < NJ; ++n) Jc[n] = J[(size_t)min(n, N - 1) * K + ic];
        if (i < 4) {                                                          // (normal_pair, inlined: the same routines, the same bits)
            const U4 r = philox(o.seed, chain_key(o, c, b), iter, 1, (uint32_t)i);
            const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
            const Pair z = box_muller_pair_inl(u1, u2);
            z0 = z.a; z1 = z.b;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {                                         // data_weights8, channels i, i + 8, i + 16
            const int n = i + 8 * u;
            if (n < N) {
                const double rd = pick4(lev.rel, rg_[u]) * d_[u];
                double an = pick4(lev.add, ag_[u]);
                if (c.add_scale != nullptr) an *= as_[u];
                const double var = rd * rd + an * an;
                const bool act = d_[u] > 0.0;
                const double w = act ? 1.0 / var : 0.0;
                P[n] = w;
                PR[n] = act ? w * (p_[u] - d_[u]) : 0.0;
            }
        }
        t2 = prior_t2_group(o, e_i, k, i, base);
        ls = i < k ? rj_log_inl(sr_i) : 0.0;
    } else {
        data_weights8<TRIPS>(c, c.data + bb * N, pred, load_levels(o, c.rel, c.add, bb), N, i, P, PR);
        if (i < k - 1 && o.solve_gradient) {
            const double c2c = 0.5 * (width_x(e, k, i) + width_x(e, k, i + 1)) * (double)(k - 1);
            t2 = o.gradient_precision / (c2c * c2c);
        }
        lmp = c.log_mean_prior[bb];
        ls = i < k ? rj_log(c.sigma_r[bb * K + i]) : 0.0;
    }
    // (cross-lane reads are issued by all lanes -- a lane that sits out of the instruction cannot be read from)
    const double t2_sh = lane_up(t2);
    const double t2_up = i > 0 ? t2_sh : 0.0;
    const double v = i < k ? ls - lmp : 0.0;
    const double v_sh_up = lane_up(v), v_sh_dn = lane_dn(v);
    const double v_up = i > 0 ? v_sh_up : 0.0, v_dn = i < 7 ? v_sh_dn : 0.0;
    wave_sync();
    double arow[KM], acol[KM];
    double g = 0.0;
#pragma unroll
    for (int j = 0; j < KM; ++j) { arow[j] = 0.0; acol[j] = 0.0; }
    auto jtpj = [&](int n, double Ji) {                                        // J'PJ and J'P r
        const double jp = Ji * P[n];
        g += Ji * PR[n];
#pragma unroll
        for (int j = 0; j < KM; ++j) arow[j] += jp * group_bcast(Ji, base, j);
    };
    if (one_trip) {
        const bool on = i < k && i < K;
#pragma unroll
        for (int n = 0; n < NJ; ++n)
            if (n < N) jtpj(n, on ? Jc[n] : 0.0);                              // (wave-uniform bound)
    } else {
        for_column<TRIPS>(J, K, N, i, i < k && i < K, jtpj);
    }
    {   // + Wm'Wm (tridiagonal), Wm'Wm (ln sigma - ln sigma_ref)
        const double single = o.value_precision + (o.solve_gradient ? o.gradient_precision : 0.0);
        const double diag = k == 1 ? single : o.value_precision + t2_up + t2;
        g += diag * v - t2_up * v_up - t2 * v_dn;
#pragma unroll
        for (int j = 0; j < KM; ++j) {
            if (j == i) arow[j] += diag;
            if (j == i - 1) arow[j] -= t2_up;
            if (i >= k) arow[j] = j == i ? 1.0 : 0.0;
            if (j >= k && j != i) arow[j] = 0.0;
        }
    }
#pragma unroll
    for (int j = 0; j < KM; ++j) {                   // Cholesky, right-looking
        const double cjj = sqrt(group_bcast(arow[j], base, j));
        if (i == j) { arow[j] = cjj; acol[j] = cjj; }
        else if (i > j) arow[j] = arow[j] / cjj;
#pragma unroll
        for (int m = j + 1; m < KM; ++m) {
            const double cmj = group_bcast(arow[j], base, m);       // C[m][j]
            if (i == j) acol[m] = cmj;
            if (i >= m) arow[m] -= arow[j] * cmj;
        }
    }
    if (live && i < k) {
        double* C = c.chol + bb * K * K + (size_t)i * K;
#pragma unroll
        for (int j = 0; j < KM; ++j)
            if (j <= i) C[j] = arow[j];
    }
    auto forward = [&](double x) {                   // C y = x
#pragma unroll
        for (int j = 0; j < KM; ++j) {
            const double yj = group_bcast(x / arow[j], base, j);     // lane j holds C[j][j] in arow[j]
            if (i == j) x = yj;
            if (i > j) x -= arow[j] * yj;
        }
        return x;
    };
    auto backward = [&](double x) {                  // C' y = x
#pragma unroll
        for (int j = KM - 1; j >= 0; --j) {
            const double yj = group_bcast(x / acol[j], base, j);     // lane j holds C[j][j] in acol[j]
            if (i == j) x = yj;
            if (i < j) x -= acol[j] * yj;                            // C[j][i]
        }
        return x;
    };
    const double step = backward(forward(g));        // (C C')^-1 g
    if (!one_trip && i < 4) normal_pair(o.seed, chain_key(o, c, b), iter, 1, (uint32_t)i, z0, z1);
    const double za = __shfl(z0, base + (i >> 1), 64), zb = __shfl(z1, base + (i >> 1), 64);
    const double w = backward((i & 1) ? zb : za);    // C^-T z
    if (live && i < K) {                             // (max_layers may be smaller than the group)
        const double lp = i < k ? (ls - o.alpha * step) + w : 0.0;
        c.log_prop[bb * K + i] = lp;
        c.sigma_p[bb * K + i] = i < k ? (one_trip ? exp(lp) : rj_exp(lp)) : 1.0;
        for (int j = i + 8; j < K; j += 8) { c.log_prop[bb * K + j] = 0.0; c.sigma_p[bb * K + j] = 1.0; }
    }
}

// The packed Newton stage, sized by the deepest chain of the wave (wave-uniform): 2, 4 or all 8 columns.  One chain per wave
// (persistent kernel): the chain's own layer count -- a 2-layer model runs a quarter of the cross-lane algebra of the full group.
template <bool TRIPS>
__device__ __forceinline__ void newton8_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int lane, int b,
                                             unsigned char* sh_dyn, int b_idle = 0)
{
    const int kb = b < c.B ? c.k_r[b] : 0;
    const int mine = kb <= 8 ? kb : 0;                            // (deeper chains are not this stage's)
    const unsigned long long deep4 = __ballot(mine > 4), deep2 = __ballot(mine > 2);
    if (deep4 != 0ull) newton8_core<8, TRIPS>(o, c, iter, lane, b, sh_dyn, b_idle);
    else if (deep2 != 0ull) newton8_core<4, TRIPS>(o, c, iter, lane, b, sh_dyn, b_idle);
    else newton8_core<2, TRIPS>(o, c, iter, lane, b, sh_dyn, b_idle);
}

// The stage's workgroups: 0 .. n_packed - 1 hold eight chains of at most 8 layers each, workgroup n_packed + b is chain b's own wave
// if the chain is deeper (it exits at once otherwise).  A launch holds both kinds or the packed ones only (the deep chains then get
// k_rj_newton: a launch of workgroups that mostly exit at once is faster with that kernel's 69 VGPRs than with the 125 of this one).
// Which of the two owns a chain follows from what neither of them writes (k_r and the move), so they need no order between them.
template <bool TRIPS>
__global__ __launch_bounds__(64) GBP_RJ_LATENCY_KERNEL void k_rj_newton8(RjOpt o, gbp_rj_chains c, uint32_t iter, int n_packed)
{
    GBP_RJ_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    if ((int)blockIdx.x >= n_packed) {                 // (the deep chains' scanning workgroups: see k_rj_newton)
        const int g = (int)blockIdx.x - n_packed, b = g * 64 + (int)threadIdx.x;
        for_deep_chains(g, b < c.B && c.k_r[min(b, c.B - 1)] > 8, [&](int bb) { newton_body(o, c, iter, 8, bb, threadIdx.x, sh_dyn); });
        return;
    }
    newton8_body<TRIPS>(o, c, iter, threadIdx.x, blockIdx.x * 8 + (threadIdx.x >> 3), sh_dyn);
}

__device__ inline double log_uniform_prior(double x, double llo, double lhi, double nlog_span)
{
    const double lx = rj_log(x);
    return (lx >= llo && lx <= lhi) ? nlog_span : -INF;
}

template <bool INL>       // INL: log inlined (the one-trip accept stage: a call would wait for its loads in flight); same routine, same bits
__device__ __forceinline__ double levels_log_prior_t(const double* x, int G, const double* llo, const double* lhi, const double* nlog_span)
{
    double p = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < G) {
            const double lx = INL ? rj_log_inl(x[g]) : rj_log(x[g]);
            p += (lx >= llo[g] && lx <= lhi[g]) ? nlog_span[g] : -INF;
        }
    return p;
}

__device__ inline double levels_log_prior(const double* x, int G, const double* llo, const double* lhi, const double* nlog_span)
{
    double p = 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        if (g < G) p += log_uniform_prior(x[g], llo[g], lhi[g], nlog_span[g]);
    return p;
}

// Error-level posteriors (DataPoint.set_posteriors :651-694): n_error_bins cells uniform in log10 between the prior bounds.
__device__ inline void error_hist_add(const RjOpt& o, const gbp_rj_chains& c, size_t b, const Levels& e, bool fast = false)
{   // (fast: logarithms inlined, counters as atomic adds nobody waits for -- the one-trip accept stage)
    if (c.rel_hist == nullptr) return;
    const double inv_ln10 = 0.43429448190325182765, nb = (double)o.n_error_bins;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (g < o.n_rel_groups) {
            const double r0 = o.log_rel_min[g] * inv_ln10, r1 = o.log_rel_max[g] * inv_ln10;
            const int ir = min(max((int)floor(((fast ? rj_log_inl(e.rel[g]) : rj_log(e.rel[g])) * inv_ln10 - r0) / (r1 - r0) * nb), 0), o.n_error_bins - 1);
            if (fast) atomicAdd(c.rel_hist + (b * o.n_rel_groups + g) * o.n_error_bins + ir, 1);
            else c.rel_hist[(b * o.n_rel_groups + g) * o.n_error_bins + ir] += 1;
        }
        if (g < o.n_add_groups) {
            const double a0 = o.log_add_min[g] * inv_ln10, a1 = o.log_add_max[g] * inv_ln10;
            const int ia = min(max((int)floor(((fast ? rj_log_inl(e.add[g]) : rj_log(e.add[g])) * inv_ln10 - a0) / (a1 - a0) * nb), 0), o.n_error_bins - 1);
            if (fast) atomicAdd(c.add_hist + (b * o.n_add_groups + g) * o.n_error_bins + ia, 1);
            else c.add_hist[(b * o.n_add_groups + g) * o.n_error_bins + ia] += 1;
        }
    }
}


// Conductivity-depth hit map (Model.update_parameter_posterior :819-847): `weight` counts of model (ec, sc, kc) added to one
// chain's map hm[n_value_bins][n_depth_bins]; W lanes share the depth cells.  The samplers call it when a chain's model
// changes (with the number of iterations the old model was the current one) instead of once per iteration.
template <int W>
__device__ inline void hitmap_add(const RjOpt& o, int32_t* hm, const double* ec, const double* sc, int kc, double lmp,
                                  int i, int weight)
{
    const double inv_ln10 = 0.43429448190325182765, Wd = o.value_half_width;
    for (int cell = i; cell < o.n_depth_bins; cell += W) {
        const double zc = ((double)cell + 0.5) * o.depth_bin_width;
        int layer = 0;
        while (layer < kc - 1 && ec[layer] <= zc) ++layer;
        const double v = (rj_log(sc[layer]) - lmp) * inv_ln10;
        const int bin = min(max((int)floor((v + Wd) / (2.0 * Wd) * (double)o.n_value_bins), 0), o.n_value_bins - 1);
        hm[(size_t)bin * o.n_depth_bins + cell] += weight;       // depth fastest: a layer's cells are one contiguous run
    }
}

__device__ inline double group_sum8(double v)
{
    // (the pairs of the xor butterfly: lane ^ 1, lane ^ 2, then the other quad -- whose four lanes hold one value)
    v += dpp_mov64<0xB1>(v); v += dpp_mov64<0x4E>(v); v += dpp_mov64<0x141>(v);
    return v;
}

// Bookkeeping on the post-step state of one chain (Inference1D.update :705-790, infer :641-688), shared by the W lanes that
// own the chain (64: one wave per chain; 8: packed).  The post-step model (ec, sc, kc) is passed from where it came from
// (the proposal buffers when accepted, the untouched state otherwise), never read back from what other lanes just wrote;
// all lanes of a chain sit in one wave, so program order is the only ordering needed between them.
// (have_regs, W == 8: entry i of the post-step rows is in e_now / s_now -- the interface histogram then costs no loads, and the counters
//  are atomic adds whose result nobody waits for: the stage is a latency chain, every read-modify-write was a trip to memory)
template <int W>
__device__ inline int bookkeeping(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int accumulate, size_t b, int i,
                                   int kc, const double* ec, const double* sc, double post, double best_prev, double misfit_now,
                                   const Levels& lev, double lmp, int dwell, double height_now, bool accepted, bool have_regs = false,
                                   double e_now = 0.0, double s_now = 0.0)
{
    const double s_dn_reg = (W == 8 && have_regs) ? lane_dn(s_now) : 0.0;      // (issued by every lane of the wave's live groups)
    const int K = o.max_layers, N = o.n_channels;
    const size_t nh = (size_t)o.n_depth_bins * o.n_value_bins;
    bool reset_best = false;
    int finished = 0;
    const int upd = (int)iter + 1 - (c.iteration0 != nullptr ? c.iteration0[b] : 0);   // this update, 1-based from the chain's (re)start
    if (c.trace_misfit != nullptr && i == 0) {                  // Inference1D.update :713, :749 -- every trace_every-th entry of the two arrays
        const int e = o.trace_every;
        if ((upd - 1) % e == 0 && (upd - 1) / e < o.trace_length) c.trace_misfit[b * o.trace_length + (upd - 1) / e] = misfit_now;
        if (upd % e == 0 && upd / e < o.trace_length) c.trace_accept[b * o.trace_length + upd / e] = accepted ? 1 : 0;
    }
    if (o.schedule == 1) {                                       // the reference's per-sounding schedule
        const int it1 = (int)iter + 1 - (c.iteration0 != nullptr ? c.iteration0[b] : 0);   // counted from the chain's (re)start
        int bi = c.burned_in_iteration[b];
        if (bi < 0) {
            double na = 0.0;
            for (int n = i; n < N; n += W) na += c.data[b * N + n] > 0.0 ? 1.0 : 0.0;
            na = W == 64 ? wave_sum(na) : group_sum8(na);
            if (it1 > o.burn_in_min_iterations && misfit_now < na) {        // burned in: posteriors and best model start over
                bi = it1;
                reset_best = true;
                for (int q = i; q < K + 1; q += W) c.k_hist[b * (K + 1) + q] = 0;
                if (c.rel_hist != nullptr) {
                    for (int q = i; q < o.n_rel_groups * o.n_error_bins; q += W) c.rel_hist[b * o.n_rel_groups * o.n_error_bins + q] = 0;
                    for (int q = i; q < o.n_add_groups * o.n_error_bins; q += W) c.add_hist[b * o.n_add_groups * o.n_error_bins + q] = 0;
                }
                if (c.edge_hist != nullptr)
                    for (int q = i; q < o.n_depth_bins; q += W) c.edge_hist[b * o.n_depth_bins + q] = 0;
                if (c.height_hist != nullptr)
                    for (int q = i; q < o.n_error_bins; q += W) c.height_hist[b * o.n_error_bins + q] = 0;
                if (c.hitmap != nullptr) {
                    for (size_t q = i; q < nh; q += W) c.hitmap[b * nh + q] = 0;
                    dwell = 0;
                }
                if (i == 0) c.burned_in_iteration[b] = bi;
            }
        }
        accumulate = 1;                                          // every iteration; the reset above discards the burn-in
        finished = (bi >= 0 && it1 > o.n_markov_chains + bi) ? 1 : ((bi < 0 && it1 >= o.n_markov_chains) ? 2 : 0);
        if (i == 0 && finished) c.status[b] = finished;         // 1 done: n_markov_chains samples collected; 2 failed to burn in
    }
    const bool posteriors_reset = reset_best;
    const bool best_replaced = reset_best || post > best_prev;
    if (best_replaced) {
        for (int j = i; j < K; j += W) { c.best_edges[b * K + j] = ec[j]; c.best_sigma[b * K + j] = sc[j]; }
        if (i == 0) {
            c.best_posterior[b] = post; c.best_k[b] = kc;
            // (the error levels of the highest-posterior state: Inference1D.update :741-745 keeps the data point.  Four unrolled, guarded
            //  stores: a loop bounded by the run's group count indexes `lev` dynamically, and the accept stages' error levels then live in scratch)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (c.best_rel != nullptr && g < o.n_rel_groups) c.best_rel[b * o.n_rel_groups + g] = lev.rel[g];
                if (c.best_add != nullptr && g < o.n_add_groups) c.best_add[b * o.n_add_groups + g] = lev.add[g];
            }
            if (c.best_height != nullptr) c.best_height[b] = height_now;
            if (c.best_iteration != nullptr) c.best_iteration[b] = upd;
        }
    }
    if (accumulate) {
        if (i == 0) {
            if (have_regs) atomicAdd(c.k_hist + b * (K + 1) + kc, 1);
            else c.k_hist[b * (K + 1) + kc] += 1;
            error_hist_add(o, c, b, lev, have_regs);
            if (c.height_hist != nullptr) {                      // Point.set_z_posterior: the cells of the uniform prior
                const double u = (height_now - (c.height0[b] - o.height_half_width)) / (2.0 * o.height_half_width);
                if (u >= 0.0 && u <= 1.0) c.height_hist[b * o.n_error_bins + min((int)floor(u * (double)o.n_error_bins), o.n_error_bins - 1)] += 1;
            }
        }
        if (c.edge_hist != nullptr && i < kc - 1) {              // interfaces across which sigma changes by > 50 %
            const double ratio = have_regs ? s_dn_reg / s_now : sc[i + 1] / sc[i];              //   (RectilinearMesh1D.update_posteriors :1595-1610)
            if (ratio <= 0.5 || ratio >= 1.5) {
                const int bin = min(max((int)floor((have_regs ? e_now : ec[i]) / o.depth_bin_width), 0), o.n_depth_bins - 1);
                atomicAdd(c.edge_hist + b * o.n_depth_bins + bin, 1);
            }
        }
    }
    if (c.hitmap != nullptr) {
        if (accumulate) dwell += 1;
        if (finished && dwell > 0) {                             // the chain stops here: settle its last model
            hitmap_add<W>(o, c.hitmap + b * nh, ec, sc, kc, lmp, i, dwell);
            dwell = 0;
        }
        if (i == 0) c.hit_dwell[b] = dwell;
    }
    return (best_replaced ? 2 : 0) | (posteriors_reset ? 4 : 0) | (accumulate ? 8 : 0) | (finished ? 16 : 0);     // (gbp_rj_chains.step_flags: bits 1-3; bit 4: callers only)
}

// What an accept stage tells the proposal that follows it in the same kernel (k_rj_step8): whether the move was taken and the layer count
// the chain has now -- its rows are then the proposal's (edges_r, sigma_p) or the untouched current ones.
struct StepState {
    bool accepted; int k_now;
    bool have;                 // the one-trip accept stage: entry i of the chain's rows as they are now, its error levels, its status
    double e_now, s_now;
    Levels lev_now;
    int status;
};

__device__ __forceinline__ void accept_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int accumulate, int min_k, int b,
                                            int lane, unsigned char* sh_dyn, StepState* st = nullptr)
{   // one wave per chain; chains whose current and proposed models both have at most min_k layers are left to accept8_body
    const int K = o.max_layers, N = o.n_channels, KS = K + 1;
    Lds s(sh_dyn, K, N);
    const int k = c.k_r[b], action = c.action[b];
    // the layer count before this proposal, from what no accept stage writes (c.k[b] is what the packed stage of the same launch may
    // be updating -- for ITS chains: a 9 -> 8 layer death accepted here must not look like an 8 -> 8 move over there)
    if (max(k, k - (action == INSERT) + (action == DELETE)) <= min_k) return;
    const double* e = c.edges_r + (size_t)b * K;
    const double* lpv = c.log_prop + (size_t)b * K;
    const double* tr = c.thk_r + (size_t)b * K;
    const double lmp = c.log_mean_prior[b];
    // priors of the proposal (Model.probability :533-575: uniform on k, normal on the gradient of ln sigma)
    double prior_p = -o.log_layers_m1;
    if (o.solve_value) {                                         // log-normal prior on the values (solve_parameter)
        double d2 = 0.0;
        if (lane < k) { const double d = lpv[lane] - lmp; d2 = d * d; }
        d2 = wave_sum(d2);
        prior_p += -0.5 * (double)k * LOG_2PI + 0.5 * (double)k * o.log_value_precision - 0.5 * o.value_precision * d2;
    }
    if (o.solve_gradient) {
        double g2 = 0.0;
        if (lane < k - 1) { const double g = (lpv[lane + 1] - lpv[lane]) / rj_log(tr[lane]); g2 = g * g; }
        g2 = wave_sum(g2);
        const double n = (double)max(1, k - 1);
        prior_p += -0.5 * n * LOG_2PI + 0.5 * n * o.log_gradient_precision - 0.5 * o.gradient_precision * g2;
    }
    if (o.value_max > 0.0) {                                     // parameter_limits (Model.probability :555-558)
        const double sp = lane < k ? c.sigma_p[(size_t)b * K + lane] : o.value_min;
        if (__any(!(sp >= o.value_min && sp <= o.value_max))) prior_p = -INF;
    }
    const Levels lev_p = load_levels(o, c.rel_p, c.add_p, (size_t)b);
    if (o.solve_relative_error) prior_p += levels_log_prior(lev_p.rel, o.n_rel_groups, o.log_rel_min, o.log_rel_max, o.nlog_rel_span);
    if (o.solve_additive_error && !o.additive_independent)
        prior_p += levels_log_prior(lev_p.add, o.n_add_groups, o.log_add_min, o.log_add_max, o.nlog_add_span);
    if (o.solve_height) prior_p += o.nlog_height_span;           // Point.probability: the proposal is inside the uniform prior by construction
    prior_p += o.extra_log_prior;
    double dq = 0.0;
    if (action == INSERT || action == DELETE) {                  // Model.proposal_probabilities (model/Model.py:577-659)
        const double* Jp = c.J_p + (size_t)b * N * K;
        const double* C = c.chol + (size_t)b * K * K;
        data_weights(c, c.data + (size_t)b * N, c.pred_p + (size_t)b * N, lev_p, N, lane, 64, s.P, s.PR);
        prior_t2(o, e, k, lane, s.t2);
        if (lane < k) {
            for (int j = 0; j <= lane; ++j) s.A[lane * KS + j] = C[(size_t)lane * K + j];
            s.v[lane] = lpv[lane] - lmp;
        }
        wave_sync();
        if (lane < k) {
            double gi = prior_apply(o, s.t2, k, lane, s.v);
            for (int n = 0; n < N; ++n) gi += Jp[(size_t)n * K + lane] * s.PR[n];
            s.g[lane] = gi;
        }
        wave_sync();
        chol_solve(s.A, KS, k, lane, s.g, true, true);           // H g'
        const double lrem = lane < k ? rj_log(c.sigma_r[(size_t)b * K + lane]) : 0.0;
        bool bad = false;
        if (lane < k) {
            const double mean_r = lpv[lane] + o.alpha * s.g[lane];
            bad = !(fabs(mean_r) < 11356.0);                     // the reference's long-double exp over/underflows there
            s.v[lane] = lrem - mean_r;                           // d1: ln sigma_rem - reverse mean
            s.w[lane] = lpv[lane] - lrem;                        // d2: ln sigma' - ln sigma_rem
        }
        wave_sync();
        double q1 = 0.0, q2 = 0.0;                               // |C' d|^2 = d' precision d
        if (lane < k) {
            double a1 = 0.0, a2 = 0.0;
            for (int i = lane; i < k; ++i) { const double cij = s.A[i * KS + lane]; a1 += cij * s.v[i]; a2 += cij * s.w[i]; }
            q1 = a1 * a1; q2 = a2 * a2;
        }
        q1 = wave_sum(q1); q2 = wave_sum(q2);
        dq = -0.5 * q1 + 0.5 * q2;
        if (__any(bad)) dq = __builtin_nan("");
    }
    double like_p, misfit_p;
    if (action == INSERT || action == DELETE) {                  // their prediction came with the Jacobian (fm_dlogc): chi^2 / logL here
        const double* pp = c.pred_p + (size_t)b * N;             //   (same arithmetic as the fused forward kernel's epilogue)
        const double* ob = c.data + (size_t)b * N;
        double s2 = 0.0, logdet = 0.0, na = 0.0;
        for (int i = lane; i < N; i += 64) {
            const double ov = ob[i];
            if (ov > 0.0) {
                const double var = variance_at(c, lev_p, ov, i);
                const double r = (pp[i] - ov) * (1.0 / sqrt(var));
                s2 += r * r; logdet += rj_log(var); na += 1.0;
            }
        }
        s2 = wave_sum(s2); logdet = wave_sum(logdet); na = wave_sum(na);
        misfit_p = s2;
        like_p = -(0.5 * na) * 1.8378770664093453 - 0.5 * logdet - 0.5 * s2;
    } else {
        like_p = c.like_p[b];
        misfit_p = c.misfit_p[b];
    }
    const int k_prev = c.k[b];                                   // (the carried state, read before it is overwritten)
    const double prior_c = c.prior[b], like_c = c.like[b], misfit_c = c.misfit[b], best_prev = c.best_posterior[b];
    const double log_ratio = (prior_p - prior_c) + (like_p - like_c) + dq;
    const U4 rr = philox_call(o.seed, chain_key(o, c, b), iter, 2, 0);
    const bool frozen = o.schedule == 1 && c.status[b] != 0;     // a chain that is done (or failed) keeps its final state
    const bool accept = !frozen && rj_log(u53(rr.x, rr.y)) < log_ratio;        // NaN and -inf reject
    if (st != nullptr) { st->accepted = accept; st->k_now = accept ? k : k_prev; }
    wave_sync();
    if (lane == 0) c.log_ratio[b] = log_ratio;
    if (frozen) {
        if (lane == 0 && c.step_flags != nullptr) c.step_flags[b] = 0;
        return;
    }
    const Levels lev_c = load_levels(o, c.rel, c.add, (size_t)b);  // (read before the state is overwritten)
    const double height_now = o.solve_height ? (accept ? c.height_p[b] : c.height[b]) : 0.0;
    const size_t nh = (size_t)o.n_depth_bins * o.n_value_bins;
    int dwell = c.hitmap != nullptr ? c.hit_dwell[b] : 0;        // iterations the current model is still owed to the hit map
    if (accept && dwell > 0) {                                   // the model changes: settle the old one first
        hitmap_add<64>(o, c.hitmap + (size_t)b * nh, c.edges + (size_t)b * K, c.sigma + (size_t)b * K, k_prev, lmp, lane, dwell);
        dwell = 0;
        wave_sync();
    }
    if (accept) {
        if (lane < K) {
            c.edges[(size_t)b * K + lane] = e[lane];
            c.sigma[(size_t)b * K + lane] = c.sigma_p[(size_t)b * K + lane];
        }
        for (int n = lane; n < N; n += 64) c.pred[(size_t)b * N + n] = c.pred_p[(size_t)b * N + n];
        if (action != NONE) {
            // (the columns the Jacobian pass wrote: layer count rounded up to 8)
            const double* Js = (action == PERTURB ? c.J_r : c.J_p) + (size_t)b * N * K;
            double* Jd = c.J + (size_t)b * N * K;
            const int kc = min(K, (k + 7) & ~7);
            for (int i = lane; i < N * kc; i += 64) { const int n = i / kc, j = i - n * kc; Jd[(size_t)n * K + j] = Js[(size_t)n * K + j]; }
        }
        if (lane == 0) {
            c.k[b] = k;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < o.n_rel_groups) c.rel[(size_t)b * o.n_rel_groups + g] = lev_p.rel[g];
                if (g < o.n_add_groups) c.add[(size_t)b * o.n_add_groups + g] = lev_p.add[g];
            }
            c.prior[b] = prior_p; c.like[b] = like_p; c.misfit[b] = misfit_p;
            c.n_accepted[b] += 1;
            if (o.solve_height) const_cast<double*>(c.height)[b] = height_now;
        }
    }
    const int bk = bookkeeping<64>(o, c, iter, accumulate, (size_t)b, lane, accept ? k : k_prev, accept ? e : c.edges + (size_t)b * K,
                                   accept ? c.sigma_p + (size_t)b * K : c.sigma + (size_t)b * K, accept ? prior_p + like_p : prior_c + like_c,
                                   best_prev, accept ? misfit_p : misfit_c, select_levels(accept, lev_p, lev_c), lmp, dwell, height_now, accept);
    if (lane == 0 && c.step_flags != nullptr) c.step_flags[b] = (accept ? 1 : 0) | (bk & 15);
}

// (whether a chain is the wave-per-chain stage's: accept_body's own rule -- the layer counts before and after the proposal)
__device__ __forceinline__ bool accept_is_deep(const gbp_rj_chains& c, int b, int min_k)
{
    if (b >= c.B) return false;
    const int k = c.k_r[b], action = c.action[b];
    return max(k, k - (action == INSERT) + (action == DELETE)) > min_k;
}

__global__ __launch_bounds__(64) void k_rj_accept(RjOpt o, gbp_rj_chains c, uint32_t iter, int accumulate, int min_k)
{   // (scanning workgroups: see k_rj_newton)
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    for_deep_chains(blockIdx.x, accept_is_deep(c, (int)blockIdx.x * 64 + (int)threadIdx.x, min_k),
                    [&](int bb) { accept_body(o, c, iter, accumulate, min_k, bb, threadIdx.x, sh_dyn); });
}

// The same for chains whose current and proposed models have at most 8 layers, packed 8 lanes per chain like k_rj_newton8:
// the reverse-move algebra runs on the Cholesky factor held in registers (row i and column i on lane i), sums over a
// chain are 8-lane butterflies, state copies and posterior updates are strided by 8.
// hitmap_add for the packed kernel (kc <= 8, 8 lanes per chain): the interface depths and the value bin of every layer
// are gathered into registers once (cross-lane reads issued by the whole wave), so a depth cell costs a few compares
// instead of a logarithm and a search through global memory.  Called by all 8 lanes of a group; `on`: the group really adds.
__device__ inline void hitmap_add8(const RjOpt& o, int32_t* hm, const double* ec, const double* sc, int kc, double lmp,
                                   int i, int base, int weight, bool on)
{
    const double inv_ln10 = 0.43429448190325182765, Wd = o.value_half_width;
    double my_edge = INF;
    int my_bin = 0;
    if (on && i < kc) {
        const double v = (rj_log(sc[i]) - lmp) * inv_ln10;
        my_bin = min(max((int)floor((v + Wd) / (2.0 * Wd) * (double)o.n_value_bins), 0), o.n_value_bins - 1);
        if (i < kc - 1) my_edge = ec[i];
    }
    double edge[8];
    int bin[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { edge[j] = __shfl(my_edge, base + j, 64); bin[j] = __shfl(my_bin, base + j, 64); }
    if (!on) return;
    for (int cell = i; cell < o.n_depth_bins; cell += 8) {
        const double zc = ((double)cell + 0.5) * o.depth_bin_width;
        int b = bin[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) b = edge[j - 1] <= zc ? bin[j] : b;       // interfaces ascend (+inf beyond the last)
        hm[(size_t)b * o.n_depth_bins + cell] += weight;         // (an atomic add, sent and forgotten, is no faster: measured)
    }
}

// What the one-trip accept stage (accept8_body, round 5) requests in ONE batch behind the move and holds in registers: entry i of the
// proposal's interface row, of the remapped conductivities and of the chain's current rows, row i and column i of the Cholesky factor,
// column i of the Jacobian the move needs (at the proposal for a dimension change -- the reverse-move gradient, and the chain's new
// Jacobian if the move is taken --, at the remapped model for a perturbation).
struct Acc8Pre {
    bool on;
    double e_i, sigma_rem, ce_i, cs_i;
    double cr[8], cc[8];
    double Jc[GBP_RJ_COLUMN_ROWS];
};

// Reverse-move proposal density of the packed accept stage (Model.proposal_probabilities :577-659) on the leading KM x KM block:
// every dimension-changing proposal of the wave has at most KM layers (rows >= k are identity rows, as in newton8_core).
template <int KM, bool TRIPS>
__device__ __forceinline__ double accept8_reverse(const RjOpt& o, const gbp_rj_chains& c, int i, int base, int k, size_t bb, bool jump,
                                                  const double* e, double lpv, double lmp, const double* PR, const Acc8Pre& pre)
{
    const int K = o.max_layers, N = o.n_channels;
    double t2 = 0.0;
    if (pre.on) {
        t2 = prior_t2_group(o, pre.e_i, k, i, base);
    } else if (i < k - 1 && o.solve_gradient) {
        const double c2c = 0.5 * (width_x(e, k, i) + width_x(e, k, i + 1)) * (double)(k - 1);
        t2 = o.gradient_precision / (c2c * c2c);
    }
    const double t2_sh = lane_up(t2);
    const double t2_up = i > 0 ? t2_sh : 0.0;
    const double v = i < k ? lpv - lmp : 0.0;
    const double v_sh_up = lane_up(v), v_sh_dn = lane_dn(v);
    const double v_up = i > 0 ? v_sh_up : 0.0, v_dn = i < 7 ? v_sh_dn : 0.0;
    double arow[KM], acol[KM];
    const bool row = jump && i < k;
    const double sigma_rem = pre.on ? pre.sigma_rem : c.sigma_r[bb * K + (i < K ? i : 0)];
    if (pre.on) {
#pragma unroll
        for (int j = 0; j < KM; ++j) {
            arow[j] = (row && j <= i) ? pre.cr[j] : (j == i ? 1.0 : 0.0);
            acol[j] = (row && j >= i && j < k) ? pre.cc[j] : (j == i ? 1.0 : 0.0);
        }
    } else {
        const double* C = c.chol + bb * K * K;
        if constexpr (TRIPS) {                       // row i and column i of the factor, requested unconditionally (ic, jc: entries
            const int ic = i < K ? i : 0;            //   every lane may read) and selected afterwards
            double cr[KM], cc[KM];
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                const int jc = j < K ? j : 0;
                cr[j] = C[(size_t)ic * K + jc];
                cc[j] = C[(size_t)jc * K + ic];
            }
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                arow[j] = (row && j <= i) ? cr[j] : (j == i ? 1.0 : 0.0);
                acol[j] = (row && j >= i && j < k) ? cc[j] : (j == i ? 1.0 : 0.0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < KM; ++j) {
                arow[j] = (row && j <= i) ? C[(size_t)i * K + j] : (j == i ? 1.0 : 0.0);
                acol[j] = (row && j >= i && j < k) ? C[(size_t)j * K + i] : (j == i ? 1.0 : 0.0);
            }
        }
    }
    double grad = 0.0;
    {
        const double single = o.value_precision + (o.solve_gradient ? o.gradient_precision : 0.0);
        const double diag = k == 1 ? single : o.value_precision + t2_up + t2;
        if (row) grad = diag * v - t2_up * v_up - t2 * v_dn;
        if (pre.on) {
#pragma unroll
            for (int n = 0; n < GBP_RJ_COLUMN_ROWS; ++n)
                if (n < N && row) grad += pre.Jc[n] * PR[n];
        } else {
            for_column<TRIPS>(c.J_p + bb * N * K, K, N, i, row, [&](int n, double Jn) { if (row) grad += Jn * PR[n]; });
        }
    }
#pragma unroll
    for (int j = 0; j < KM; ++j) {                   // C y = grad
        const double yj = group_bcast(grad / arow[j], base, j);
        if (i == j) grad = yj;
        if (i > j) grad -= arow[j] * yj;
    }
#pragma unroll
    for (int j = KM - 1; j >= 0; --j) {              // C' x = y
        const double yj = group_bcast(grad / acol[j], base, j);
        if (i == j) grad = yj;
        if (i < j) grad -= acol[j] * yj;
    }
    const double mean_r = lpv + o.alpha * grad;
    const bool bad = row && !(fabs(mean_r) < 11356.0);
    const double lrem = row ? (pre.on ? rj_log_inl(sigma_rem) : rj_log(sigma_rem)) : 0.0;
    const double d1 = row ? lrem - mean_r : 0.0, d2 = row ? lpv - lrem : 0.0;
    double a1 = 0.0, a2 = 0.0;                       // (C' d)_i = sum_{m >= i} C[m][i] d_m
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        const double d1m = group_bcast(d1, base, m), d2m = group_bcast(d2, base, m);
        if (m >= i) { a1 += acol[m] * d1m; a2 += acol[m] * d2m; }
    }
    const double q1 = group_sum8(row ? a1 * a1 : 0.0), q2 = group_sum8(row ? a2 * a2 : 0.0);
    const unsigned long long badmask = __ballot(bad);
    double dq = jump ? -0.5 * q1 + 0.5 * q2 : 0.0;
    if ((badmask >> base) & 0xFFull) dq = __builtin_nan("");
    return dq;
}

// `b`: the chain of this lane's 8-lane group, or >= c.B for an idle group; sh_dyn: PR[8][N]
template <bool TRIPS>
__device__ __forceinline__ void accept8_body(const RjOpt& o, const gbp_rj_chains& c, uint32_t iter, int accumulate, int lane, int b,
                                             unsigned char* sh_dyn, int b_idle = 0, StepState* st = nullptr)
{
    const int slot = lane >> 3, i = lane & 7, base = lane & ~7;
    const int K = o.max_layers, N = o.n_channels;
    int k = b < c.B ? c.k_r[b] : 0;
    const size_t bb = b < c.B ? (size_t)b : (size_t)b_idle;   // idle groups read a valid row (never used); not a function of k:
    const int action_bb = c.action[bb];                        //   the loads below do not wait for the ones above
    // the layer count before this proposal (= c.k[b] until an accept stage writes it; derived: accept_body)
    const int k_prev = k - (action_bb == INSERT) + (action_bb == DELETE);
    const bool live = k >= 1 && max(k, k_prev) <= 8;         // no early exit: idle groups still take part in cross-lane reads
    if (!live) k = 0;
    const int action = live ? action_bb : NONE;
    const bool jump = action == INSERT || action == DELETE;
    const int status_bb = o.schedule == 1 ? c.status[bb] : 0;
    const bool frozen = o.schedule == 1 && status_bb != 0;
    const double* e = c.edges_r + bb * K;
    const double lmp = c.log_mean_prior[bb];
    // Round 5, ONE TRIP (the lock-step launches, up to GBP_RJ_COLUMN_ROWS channels; docs/notes_r5.md): an iteration is the chain of a
    // sub-block's dependent launches and this stage was ~20 dependent trips to memory with library calls (which wait for every load in
    // flight) between them.  Everything the decision, the state update, the posteriors and the proposal that follows can need is requested
    // HERE in one batch, logarithms and the generator are inlined, counters are atomic adds nobody waits for.  Same arithmetic, same order.
    constexpr int NJ = GBP_RJ_COLUMN_ROWS;
    const bool one_trip = TRIPS && GBP_RJ_ONE_TRIP_ACCEPT && N <= NJ;          // (wave-uniform)
    Acc8Pre pre;
    pre.on = one_trip;
    Levels lev_p1;
    double d_[3], p_[3], as_[3];
    int rg_[3], ag_[3];
    if (one_trip) {
        const int ic1 = i < K ? i : 0;
        lev_p1 = load_levels(o, c.rel_p, c.add_p, bb);
        const double* pp = c.pred_p + bb * N;
        const double* ob = c.data + bb * N;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int n = min(i + 8 * u, N - 1);
            d_[u] = ob[n]; p_[u] = pp[n];
            rg_[u] = c.rel_group != nullptr ? c.rel_group[n] : 0;
            ag_[u] = c.add_group != nullptr ? c.add_group[n] : 0;
            as_[u] = c.add_scale != nullptr ? c.add_scale[n] : 1.0;
        }
        pre.e_i = e[ic1];
        pre.sigma_rem = c.sigma_r[bb * K + ic1];
        pre.ce_i = c.edges[bb * K + ic1];
        pre.cs_i = c.sigma[bb * K + ic1];
        // (round 6: requested by the groups whose move needs them -- the move is known here, a group's eight lanes share it -- instead of
        //  by every group of a wave in which any chain jumps: the Cholesky factor for a third of the chains, a Jacobian column for half)
#pragma unroll
        for (int j = 0; j < 8; ++j) { pre.cr[j] = 0.0; pre.cc[j] = 0.0; }
#pragma unroll
        for (int n = 0; n < NJ; ++n) pre.Jc[n] = 0.0;
#ifdef GBP_RJ_WAVE_PRELOAD
        if (__ballot(jump) != 0ull) {
#else
        if (jump) {
#endif
            const double* C = c.chol + bb * K * K;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int jc = j < K ? j : 0;
                pre.cr[j] = C[(size_t)ic1 * K + jc];
                pre.cc[j] = C[(size_t)jc * K + ic1];
            }
        }
#ifdef GBP_RJ_WAVE_PRELOAD
        if (__ballot(action != NONE) != 0ull) {
#else
        if (action != NONE) {
#endif
            const double* Js = (action == PERTURB ? c.J_r : c.J_p) + bb * N * K;
#pragma unroll
            for (int n = 0; n < NJ; ++n) pre.Jc[n] = Js[(size_t)min(n, N - 1) * K + ic1];
        }
    }
    // everything the decision reads from the chain's rows is requested here, unconditionally (column ic: one every lane may read),
    // and selected where it is used: one trip to memory for all of it
    const int ic = i < K ? i : 0;
    const double lpv_ic = c.log_prop[bb * K + ic], thk_ic = c.thk_r[bb * K + ic], sp_ic = c.sigma_p[bb * K + ic];
    const double misfit_p0 = c.misfit_p[bb], like_p0 = c.like_p[bb];
    const double prior_c = c.prior[bb], like_c = c.like[bb], best_prev = c.best_posterior[bb], misfit_c = c.misfit[bb];
    const Levels lev_c = load_levels(o, c.rel, c.add, bb);
    const double height_c = o.solve_height ? c.height[bb] : 0.0, height_p = o.solve_height ? c.height_p[bb] : 0.0;
    const int dwell0 = c.hitmap != nullptr ? c.hit_dwell[bb] : 0;
    const double lpv = i < k ? lpv_ic : 0.0;
    const double lpv_dn = lane_dn(lpv);
    // priors of the proposal
    double prior_p = -o.log_layers_m1;
    if (o.solve_value) {
        const double d = i < k ? lpv - lmp : 0.0;
        const double d2 = group_sum8(d * d);
        prior_p += -0.5 * (double)k * LOG_2PI + 0.5 * (double)k * o.log_value_precision - 0.5 * o.value_precision * d2;
    }
    if (o.solve_gradient) {
        double g = 0.0;
        if (i < k - 1) g = (lpv_dn - lpv) / (one_trip ? rj_log_inl(thk_ic) : rj_log(thk_ic));
        const double g2 = group_sum8(g * g);
        const double n = (double)max(1, k - 1);
        prior_p += -0.5 * n * LOG_2PI + 0.5 * n * o.log_gradient_precision - 0.5 * o.gradient_precision * g2;
    }
    if (o.value_max > 0.0) {                         // parameter_limits (Model.probability :555-558)
        const double sp = i < k ? sp_ic : o.value_min;
        const unsigned long long out = __ballot(!(sp >= o.value_min && sp <= o.value_max));
        if ((out >> base) & 0xFFull) prior_p = -INF;
    }
    const Levels lev_p = one_trip ? lev_p1 : load_levels(o, c.rel_p, c.add_p, bb);
    if (one_trip) {
        if (o.solve_relative_error) prior_p += levels_log_prior_t<true>(lev_p.rel, o.n_rel_groups, o.log_rel_min, o.log_rel_max, o.nlog_rel_span);
        if (o.solve_additive_error && !o.additive_independent)
            prior_p += levels_log_prior_t<true>(lev_p.add, o.n_add_groups, o.log_add_min, o.log_add_max, o.nlog_add_span);
    } else {
        if (o.solve_relative_error) prior_p += levels_log_prior(lev_p.rel, o.n_rel_groups, o.log_rel_min, o.log_rel_max, o.nlog_rel_span);
        if (o.solve_additive_error && !o.additive_independent)
            prior_p += levels_log_prior(lev_p.add, o.n_add_groups, o.log_add_min, o.log_add_max, o.nlog_add_span);
    }
    if (o.solve_height) prior_p += o.nlog_height_span;           // Point.probability: the proposal is inside the uniform prior by construction
    prior_p += o.extra_log_prior;
    // dimension-changing proposals: data weights at the proposal, chi^2 / logL of the prediction that came with the Jacobian
    double* PR = reinterpret_cast<double*>(sh_dyn) + (size_t)slot * N;
    double s2 = 0.0, logdet = 0.0, na = 0.0;
    if (jump && one_trip) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {                                         // (for_channels8's channels i, i + 8, i + 16 from the registers)
            const int n = i + 8 * u;
            if (n < N) {
                const double rd = pick4(lev_p.rel, rg_[u]) * d_[u];
                double an = pick4(lev_p.add, ag_[u]);
                if (c.add_scale != nullptr) an *= as_[u];
                const double var = rd * rd + an * an;
                const double ov = d_[u], pn = p_[u];
                double pr = 0.0;
                if (ov > 0.0) {
                    const double r = (pn - ov) * (1.0 / sqrt(var));
                    s2 += r * r; logdet += rj_log_inl(var); na += 1.0;
                    pr = (1.0 / var) * (pn - ov);
                }
                PR[n] = pr;
            }
        }
    } else if (jump) {
        const double* pp = c.pred_p + bb * N;
        const double* ob = c.data + bb * N;
        for_channels8<TRIPS>(c, ob, pp, lev_p, N, i, [&](int n, double ov, double pn, double var) {
            double pr = 0.0;
            if (ov > 0.0) {
                const double r = (pn - ov) * (1.0 / sqrt(var));
                s2 += r * r; logdet += rj_log(var); na += 1.0;
                pr = (1.0 / var) * (pn - ov);
            }
            PR[n] = pr;
        });
    }
    s2 = group_sum8(s2); logdet = group_sum8(logdet); na = group_sum8(na);
    wave_sync();
    // reverse-move proposal density (Model.proposal_probabilities :577-659); executed by every group of a wave that holds a
    // dimension-changing proposal (wave-uniform branch: the cross-lane reads inside are issued by all 64 lanes), used by the jumps
    double dq = 0.0;
    {   // (wave-uniform: the cross-lane reads inside are issued by all 64 lanes; sized by the deepest jump of the wave)
        const unsigned long long any = __ballot(jump), deep4 = __ballot(jump && k > 4), deep2 = __ballot(jump && k > 2);
        if (deep4 != 0ull) dq = accept8_reverse<8, TRIPS>(o, c, i, base, k, bb, jump, e, lpv, lmp, PR, pre);
        else if (deep2 != 0ull) dq = accept8_reverse<4, TRIPS>(o, c, i, base, k, bb, jump, e, lpv, lmp, PR, pre);
        else if (any != 0ull) dq = accept8_reverse<2, TRIPS>(o, c, i, base, k, bb, jump, e, lpv, lmp, PR, pre);
    }
    const double misfit_p = jump ? s2 : misfit_p0;
    const double like_p = jump ? -(0.5 * na) * 1.8378770664093453 - 0.5 * logdet - 0.5 * s2 : like_p0;
    const double log_ratio = (prior_p - prior_c) + (like_p - like_c) + dq;
    const U4 rr = one_trip ? philox(o.seed, chain_key(o, c, b), iter, 2, 0) : philox_call(o.seed, chain_key(o, c, b), iter, 2, 0);
    const bool accept = live && !frozen && (one_trip ? rj_log_inl(u53(rr.x, rr.y)) : rj_log(u53(rr.x, rr.y))) < log_ratio;
    if (st != nullptr) {
        st->accepted = accept; st->k_now = accept ? k : k_prev;
        st->have = one_trip;
        if (one_trip) {
            st->e_now = accept ? pre.e_i : pre.ce_i;
            st->s_now = accept ? sp_ic : pre.cs_i;
            st->lev_now = select_levels(accept, lev_p, lev_c);
            st->status = status_bb;
        }
    }
    if (live && i == 0) c.log_ratio[bb] = log_ratio;
    if (live && frozen && i == 0 && c.step_flags != nullptr) c.step_flags[bb] = 0;
    if (!live || frozen) return;                     // (below: cross-lane reads only within a chain's own group)
    const double height_now = accept ? height_p : height_c;
    const size_t nh = (size_t)o.n_depth_bins * o.n_value_bins;
    int dwell = dwell0;
    if (c.hitmap != nullptr) {                       // the model changes: settle the old one in the hit map first
        const bool on = accept && dwell > 0;
        hitmap_add8(o, c.hitmap + bb * nh, c.edges + bb * K, c.sigma + bb * K, k_prev, lmp, i, base, dwell, on);
        if (on) dwell = 0;
    }
    if (accept && one_trip) {
        // (stores only: entry i of the rows is in registers; beyond the group's 8 entries the proposal's rows of a model of at most 8
        //  layers hold the empty entries every proposal / Newton kernel writes there: +inf and 1)
        if (i < K) { c.edges[bb * K + i] = pre.e_i; c.sigma[bb * K + i] = sp_ic; }
        for (int j = i + 8; j < K; j += 8) { c.edges[bb * K + j] = INF; c.sigma[bb * K + j] = 1.0; }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (i + 8 * u < N) c.pred[bb * N + i + 8 * u] = p_[u];
        if (action != NONE && i < K) {
            double* Jd = c.J + bb * N * K;
#pragma unroll
            for (int n = 0; n < NJ; ++n)
                if (n < N) Jd[(size_t)n * K + i] = pre.Jc[n];
        }
    } else if (accept) {
        copy_strided8<TRIPS>(c.edges + bb * K, e, c.sigma + bb * K, c.sigma_p + bb * K, K, i);
        copy_strided8<TRIPS>(c.pred + bb * N, c.pred_p + bb * N, nullptr, nullptr, N, i);
        if (action != NONE) {
            // (columns 0..7 only: both models have at most 8 layers, the Jacobian pass wrote these 8 columns, and nothing reads
            //  a column at or beyond the layer count)
            double* Jd = c.J + bb * N * K;
            for_column<TRIPS>((action == PERTURB ? c.J_r : c.J_p) + bb * N * K, K, N, i, i < K,
                              [&](int n, double v) { if (i < K) Jd[(size_t)n * K + i] = v; });
        }
    }
    if (accept) {
        if (i == 0) {
            c.k[bb] = k;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g < o.n_rel_groups) c.rel[bb * o.n_rel_groups + g] = lev_p.rel[g];
                if (g < o.n_add_groups) c.add[bb * o.n_add_groups + g] = lev_p.add[g];
            }
            c.prior[bb] = prior_p; c.like[bb] = like_p; c.misfit[bb] = misfit_p;
            if (one_trip) atomicAdd(reinterpret_cast<unsigned long long*>(c.n_accepted + bb), 1ull);
            else c.n_accepted[bb] += 1;
            if (o.solve_height) const_cast<double*>(c.height)[bb] = height_now;
        }
    }
    const int bk = bookkeeping<8>(o, c, iter, accumulate, bb, i, accept ? k : k_prev, accept ? e : c.edges + bb * K,
                                  accept ? c.sigma_p + bb * K : c.sigma + bb * K, accept ? prior_p + like_p : prior_c + like_c, best_prev,
                                  accept ? misfit_p : misfit_c, select_levels(accept, lev_p, lev_c), lmp, dwell, height_now, accept, one_trip,
                                  accept ? pre.e_i : pre.ce_i, accept ? sp_ic : pre.cs_i);
    if (st != nullptr && (bk & 16)) st->status = 1;             // (the chain stopped in this very iteration: its next proposal is the idle one)
    if (i == 0 && c.step_flags != nullptr) c.step_flags[bb] = (accept ? 1 : 0) | (bk & 15);
}

template <bool TRIPS>
__global__ __launch_bounds__(64) GBP_RJ_LATENCY_KERNEL void k_rj_accept8(RjOpt o, gbp_rj_chains c, uint32_t iter, int accumulate, int n_packed)
{   // (workgroups as in k_rj_newton8)
    GBP_RJ_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    if ((int)blockIdx.x >= n_packed) {                 // (the deep chains' scanning workgroups)
        const int g = (int)blockIdx.x - n_packed;
        for_deep_chains(g, accept_is_deep(c, g * 64 + (int)threadIdx.x, 8), [&](int bb) { accept_body(o, c, iter, accumulate, 8, bb, threadIdx.x, sh_dyn); });
        return;
    }
    accept8_body<TRIPS>(o, c, iter, accumulate, threadIdx.x, blockIdx.x * 8 + (threadIdx.x >> 3), sh_dyn);
}

// ---------------------------------------------------------------------------------------------------------------
// The accept stage of iteration i and the proposal of iteration i + 1 in ONE launch (round 5).  An iteration of a sub-block is its chain
// of dependent launches -- a build that parks the proposal kernel for 10 / 20 us longer loses exactly 10 / 20 us per iteration at 2 048,
// 4 096 and 8 192 chains alike (scripts/ab_rj.py, docs/notes_r5.md) -- and the proposal launch was 28 us of it: 43 workgroups whose
// first wave walks 64 chains' rows serially.  Here the lanes that have just decided a chain's move propose its next one: the packed
// groups with propose8_body (8 lanes per chain, rows in registers), the deep chains' waves with propose_wave_body.  The proposal
// rewrites k_r and the move -- what tells packed and deep workgroups of an accept stage whose chain is whose -- so ownership inside
// this launch is read from `deep_cur`, flags the PROPOSAL of the iteration wrote (k_rj_propose_flags for the first one of a call), and
// the flags of the next iteration go to `deep_next`.  Same functions, same draws: bit-identical chains.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t step_is_deep(int k_now, int kr) { return max(k_now, kr) > 8 ? 1 : 0; }

template <bool TRIPS>
__global__ __launch_bounds__(64) GBP_RJ_LATENCY_KERNEL void k_rj_step8(RjOpt o, gbp_rj_chains c, uint32_t iter, int accumulate, int n_packed,
                                                 const int32_t* __restrict__ deep_cur, int32_t* __restrict__ deep_next)
{
    GBP_RJ_RAISE_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int K = o.max_layers, lane = threadIdx.x;
    if ((int)blockIdx.x >= n_packed) {                 // the deep chains' scanning workgroups
        const int g = (int)blockIdx.x - n_packed, bl = g * 64 + lane;
        for_deep_chains(g, bl < c.B && deep_cur[min(bl, c.B - 1)] != 0, [&](int bb) {
            StepState st{};
            const int k_before = c.k[bb];
            const bool frozen = o.schedule == 1 && c.status[bb] != 0;
            accept_body(o, c, iter, accumulate, 0, bb, lane, sh_dyn, &st);
            if (frozen) { st.accepted = false; st.k_now = k_before; }
            wave_sync();
            const double* e_row = (st.accepted ? c.edges_r : c.edges) + (size_t)bb * K;
            const double* s_row = (st.accepted ? c.sigma_p : c.sigma) + (size_t)bb * K;
            // (the rows are read into registers before the proposal writes edges_r: propose_wave_body loads first, then stores)
            const int kr = propose_wave_body(o, c, iter + 1, bb, lane, st.k_now, e_row, s_row);
            if (lane == 0) deep_next[bb] = step_is_deep(st.k_now, kr);
        });
        return;
    }
    const int b = (int)blockIdx.x * 8 + (lane >> 3);
    const bool mine = b < c.B && deep_cur[min(b, c.B - 1)] == 0;
    StepState st{};
    const int bq = min(b, c.B - 1);
    const int k_before = c.k[bq];
    const bool frozen = o.schedule == 1 && c.status[bq] != 0;
    // (a chain flagged deep belongs to a scanning workgroup of this launch, which rewrites k_r and the move for iteration + 1 while
    //  this group may still be reading them: a group acts on its chain only when deep_cur says so -- idle groups get chain index c.B)
    accept8_body<TRIPS>(o, c, iter, accumulate, lane, mine ? b : c.B, sh_dyn, 0, &st);
    if (frozen) { st.accepted = false; st.k_now = k_before; }
    wave_sync();
    if (mine) {                                        // (group-uniform; the reads inside stay within the chain's own 8 lanes)
        const double* e_row = (st.accepted ? c.edges_r : c.edges) + (size_t)b * K;
        const double* s_row = (st.accepted ? c.sigma_p : c.sigma) + (size_t)b * K;
        const int kr = propose8_body(o, c, iter + 1, lane, b, st.k_now, e_row, s_row, st.have, st.e_now, st.s_now, &st.lev_now,
                                     st.have ? st.status : -1);
        if ((lane & 7) == 0) deep_next[b] = step_is_deep(st.k_now, kr);
    }
}

// The ownership flags of an iteration whose proposal came from a stand-alone proposal launch (the first of a call)
__global__ __launch_bounds__(256) void k_rj_propose_flags(gbp_rj_chains c, int32_t* __restrict__ deep)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < c.B) deep[b] = accept_is_deep(c, b, 8) ? 1 : 0;
}

// The chains of a launch by descending layer count (counting sort, one workgroup; ties in any order -- which workgroup evaluates which
// chain changes no bit): the order in which k_rj_physics takes them (GBP_RJ_ORDERED_PHYSICS)
__global__ __launch_bounds__(1024) void k_rj_order_by_layers(gbp_rj_chains c, int32_t* __restrict__ order)
{
    __shared__ int cnt[64], pos[64];
    const int tid = threadIdx.x;
    if (tid < 64) cnt[tid] = 0;
    __syncthreads();
    for (int b = tid; b < c.B; b += 1024) atomicAdd(&cnt[63 - min(max(c.k[b], 0), 63)], 1);
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int q = 0; q < 64; ++q) { pos[q] = run; run += cnt[q]; }
    }
    __syncthreads();
    for (int b = tid; b < c.B; b += 1024) order[atomicAdd(&pos[63 - min(max(c.k[b], 0), 63)], 1)] = b;
}

// Settles what the chains' current models are still owed in the hit map (call before reading it).
__global__ __launch_bounds__(64) void k_rj_flush(RjOpt o, gbp_rj_chains c)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int dwell = c.hit_dwell[b];
    if (dwell <= 0) return;
    const int K = o.max_layers;
    hitmap_add<64>(o, c.hitmap + (size_t)b * o.n_depth_bins * o.n_value_bins, c.edges + (size_t)b * K, c.sigma + (size_t)b * K, c.k[b],
                   c.log_mean_prior[b], lane, dwell);
    __syncthreads();
    if (lane == 0) c.hit_dwell[b] = 0;
}

__global__ void k_rj_debug_random(uint64_t seed, uint32_t chain, uint32_t iter, uint32_t stream, int n, double* uni, double* nor)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r(seed, chain, iter, stream);
    for (int i = 0; i < n; ++i) uni[i] = r.uniform();
    for (int j = 0; 2 * j < n; ++j) {
        double z0, z1;
        normal_pair(seed, chain, iter, stream, (uint32_t)j, z0, z1);
        nor[2 * j] = z0;
        if (2 * j + 1 < n) nor[2 * j + 1] = z1;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Time-domain data (TdemDataPoint): the frequency-domain kernels produce the nodal spectrum of a sounding, a constant
// matrix turns it into window values (geobipy_amd/tdem.py: windows = nodal @ W).  One workgroup per sounding; soundings
// with 0 layers are skipped like everywhere else.
// ---------------------------------------------------------------------------------------------------------------
// Geometry mixing (gbp_td_mix, geobipy_amd/tdem_geometry.py): the kernels' nodal spectra are those of the BASIS INTEGRALS of the
// rho-frame; the spectrum of output component m is the per-row real combination  sum_t weights[b, col[m, t]] * in[src[m, t]]
// (transmitter / receiver attitude, azimuth of the offset, output sign and scaling) -- formed while the row is staged in LDS.
template <bool WITH_J>
__global__ __launch_bounds__(64) void k_td_apply(int B, int K, int n_nodal, int N, const int* __restrict__ nl, const double* __restrict__ W,
                                                 const double* __restrict__ nodal, const double* __restrict__ J_nodal,
                                                 double* __restrict__ pred, double* __restrict__ J, gbp_td_mix mix)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];      // nodal[n_nodal] | J_nodal[n_nodal][k]
    const int b = blockIdx.x, lane = threadIdx.x;
    const int k = nl[b];
    if (k <= 0) return;
    double* sn = reinterpret_cast<double*>(sh_dyn);
    double* sj = sn + n_nodal;
    if (mix.n_in <= 0) {
        for (int m = lane; m < n_nodal; m += 64) sn[m] = nodal[(size_t)b * n_nodal + m];
        if (WITH_J)
            for (int q = lane; q < n_nodal * k; q += 64) sj[q] = J_nodal[((size_t)b * n_nodal + q / k) * K + q % k];
    } else {
        const double* w = mix.weights + (size_t)b * mix.n_weights;
        const double* in = nodal + (size_t)b * mix.n_in;
        for (int m = lane; m < n_nodal; m += 64) {
            double acc = 0.0;
            for (int t = 0; t < mix.terms; ++t) {
                const int s = mix.src[m * mix.terms + t];
                if (s >= 0) acc += w[mix.col[m * mix.terms + t]] * in[s];
            }
            sn[m] = acc;
        }
        if (WITH_J)
            for (int q = lane; q < n_nodal * k; q += 64) {
                const int m = q / k, l = q % k;
                double acc = 0.0;
                for (int t = 0; t < mix.terms; ++t) {
                    const int s = mix.src[m * mix.terms + t];
                    if (s >= 0) acc += w[mix.col[m * mix.terms + t]] * J_nodal[((size_t)b * mix.n_in + s) * K + l];
                }
                sj[q] = acc;
            }
    }
    __syncthreads();
    // one output per lane and pass: (gate g, column l) with l = 0 the prediction and l >= 1 column l - 1 of the Jacobian -- N (k + 1)
    // outputs over the 64 lanes (a lane per gate left two thirds of the wave idle at 19 gates and ran (k + 1) n_nodal multiply-adds in
    // sequence); every output is the same sum over the nodal values in the same order
    const int n_out = N * (WITH_J ? k + 1 : 1);
    for (int q = lane; q < n_out; q += 64) {
        const int l = q / N, g = q - l * N;
        const double* src = l == 0 ? sn : sj + (l - 1);
        const int stride = l == 0 ? 1 : k;
        double acc = 0.0;
        for (int m = 0; m < n_nodal; ++m) acc += src[m * stride] * W[(size_t)m * N + g];
        if (l == 0) {
            if (mix.offset != nullptr) acc += mix.offset[(size_t)b * N + g];
            pred[(size_t)b * N + g] = acc;
        } else {
            J[((size_t)b * N + g) * K + (l - 1)] = acc;
        }
    }
    if (WITH_J)
        for (int q = lane; q < N * (K - k); q += 64) {
            const int g = q / (K - k), l = k + q - g * (K - k);
            J[((size_t)b * N + g) * K + l] = 0.0;
        }
}

// The forward-only case of k_td_apply (what gbp_tdem_forward / TdemBatch.forward run: BASELINE config 4) with the window
// operator in LDS: k_td_apply reads W[m, g] from L2 for every sounding -- n_nodal x N doubles = 12 KB each, 200 MB per 16 384 soundings,
// which is what its 28 us were --; here a workgroup of four waves stages W once and walks GBP_TD_PLAIN_ROWS soundings, a wave each at a
// time.  Every output is the same sum over the nodal values in the same order: the same bits.
#define GBP_TD_PLAIN_ROWS 16
__global__ __launch_bounds__(256) void k_td_apply_plain(int B, int n_nodal, int N, const int* __restrict__ nl, const double* __restrict__ W,
                                                         const double* __restrict__ nodal, double* __restrict__ pred, gbp_td_mix mix)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];      // W[n_nodal][N] | nodal[4][n_nodal]
    double* sW = reinterpret_cast<double*>(sh_dyn);
    double* sn = sW + (size_t)n_nodal * N + (size_t)(threadIdx.x >> 6) * n_nodal;
    for (int q = threadIdx.x; q < n_nodal * N; q += 256) sW[q] = W[q];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b_end = min(B, (int)(blockIdx.x + 1) * GBP_TD_PLAIN_ROWS);
    for (int b = blockIdx.x * GBP_TD_PLAIN_ROWS + wave; b < b_end; b += 4) {
        if (nl[b] <= 0) continue;                                                // (wave-uniform)
        if (mix.n_in <= 0) {
            for (int m = lane; m < n_nodal; m += 64) sn[m] = nodal[(size_t)b * n_nodal + m];
        } else {                                                                 // the geometry mixing of the basis integrals, as in k_td_apply
            const double* w = mix.weights + (size_t)b * mix.n_weights;
            const double* in = nodal + (size_t)b * mix.n_in;
            for (int m = lane; m < n_nodal; m += 64) {
                double acc = 0.0;
                for (int t = 0; t < mix.terms; ++t) {
                    const int s_ = mix.src[m * mix.terms + t];
                    if (s_ >= 0) acc += w[mix.col[m * mix.terms + t]] * in[s_];
                }
                sn[m] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int g = lane; g < N; g += 64) {
            double acc = 0.0;
            for (int m = 0; m < n_nodal; ++m) acc += sn[m] * sW[(size_t)m * N + g];
            if (mix.offset != nullptr) acc += mix.offset[(size_t)b * N + g];
            pred[(size_t)b * N + g] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sampled attitude angles of a time-domain loop pair (gbp_td_moves; geobipy_amd/tdem_geometry.py is the host twin of the algebra).
// ---------------------------------------------------------------------------------------------------------------
// Body -> earth rotation Rz(yaw) Ry(pitch) Rx(roll), angles in degrees (GA-AEM's semantics), row-major.
__device__ inline void td_rotation(double roll, double pitch, double yaw, double R[9])
{
    const double d2r = 0.017453292519943295769;
    const double cr = cos(roll * d2r), sr = sin(roll * d2r), cp = cos(pitch * d2r), sp = sin(pitch * d2r), cy = cos(yaw * d2r), sy = sin(yaw * d2r);
    R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
    R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
    R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
}

// Mixing weights [n_blocks, n_basis] and (total-field data) the predicted primary field per window of one GA-AEM tuple g[10]:
// field along receiver axis k = sum_i w[k][i] BASIS_i with V = R_rx' Rz(phi), m' = Rz(-phi) R_tx z^, phi = atan2(dy, dx)
// (tdem_geometry.basis_weights / GeometryMix.primary_field).
__device__ inline void td_weights_of(const gbp_td_moves& mv, const double* g, double* w_out, double* off_out)
{
    double Rt[9], Rr[9];
    td_rotation(g[1], g[2], g[3], Rt);
    td_rotation(g[7], g[8], g[9], Rr);
    const double rho = hypot(g[4], g[5]);
    const bool on = !(rho > 0.0);
    const double cph = on ? 1.0 : g[4] / rho, sph = on ? 0.0 : g[5] / rho;
    const double m[3] = {Rt[2], Rt[5], Rt[8]};                                       // R_tx z^
    const double u[3] = {cph * m[0] + sph * m[1], -sph * m[0] + cph * m[1], m[2]};   // Rz(-phi) m
    double w[3][5];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // V[k][j] = sum_i Rr[i][k] Rz[i][j],  Rz = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
        const double v0 = Rr[0 + k] * cph + Rr[3 + k] * sph, v1 = -Rr[0 + k] * sph + Rr[3 + k] * cph, v2 = Rr[6 + k];
        w[k][0] = v2 * u[2];
        w[k][1] = v0 * u[2];
        w[k][2] = v0 * u[0];
        w[k][3] = -v2 * u[0];
        w[k][4] = -v0 * u[0] + v1 * u[1];
        if (!mv.loop) { w[k][0] += w[k][2]; w[k][1] += w[k][3]; w[k][2] = 0.0; w[k][3] = 0.0; }
        if (on) { w[k][2] += 0.5 * w[k][4]; w[k][1] = 0.0; w[k][3] = 0.0; w[k][4] = 0.0; }
    }
    for (int q = 0; q < mv.n_blocks; ++q) {
        const int k = mv.block_comp[q];
        const double sc = mv.block_scale[q];
        for (int t = 0; t < mv.n_basis; ++t) {
            const int bi = mv.basis[t];
            const double wk = k == 0 ? (bi == 0 ? w[0][0] : bi == 1 ? w[0][1] : bi == 2 ? w[0][2] : bi == 3 ? w[0][3] : w[0][4])
                            : k == 1 ? (bi == 0 ? w[1][0] : bi == 1 ? w[1][1] : bi == 2 ? w[1][2] : bi == 3 ? w[1][3] : w[1][4])
                                     : (bi == 0 ? w[2][0] : bi == 1 ? w[2][1] : bi == 2 ? w[2][2] : bi == 3 ? w[2][3] : w[2][4]);
            w_out[q * mv.n_basis + t] = wk * sc;
        }
    }
    if (off_out != nullptr) {                                    // free-space field of the rotated dipole along the receiver's axes
        const double R[3] = {g[4], g[5], g[6]};
        const double rn2 = R[0] * R[0] + R[1] * R[1] + R[2] * R[2], rn = sqrt(rn2);
        const double mr = m[0] * R[0] + m[1] * R[1] + m[2] * R[2];
        const double f = 1.0 / (4.0 * 3.14159265358979323846 * rn2 * rn);
        const double H[3] = {(3.0 * mr * R[0] / rn2 - m[0]) * f, (3.0 * mr * R[1] / rn2 - m[1]) * f, (3.0 * mr * R[2] / rn2 - m[2]) * f};
        int n0 = 0;
        for (int q = 0; q < mv.n_blocks; ++q) {
            const int k = mv.block_comp[q];
            const double ck = Rr[0 + k] * H[0] + Rr[3 + k] * H[1] + Rr[6 + k] * H[2];        // (R_rx' H)_k
            const double v = mv.block_primary[q] * ck;
            for (int n = 0; n < mv.block_windows[q]; ++n) off_out[n0 + n] = v;
            n0 += mv.block_windows[q];
        }
    }
}

// Proposal stage of the angles (thread per chain): the reference's order within a loop -- pitch, roll, yaw after the positions,
// transmitter before receiver -- is the order of the moves in `mv`; draws from stream 3 of (chain, iteration).
__global__ __launch_bounds__(64) void k_td_moves_propose(RjOpt o, gbp_rj_chains c, gbp_td_moves mv, int n_weights, int N, uint32_t iter)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    double g[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) g[j] = mv.geom[(size_t)b * 10 + j];
    Rng r(o.seed, chain_key(o, c, b), iter, 3);
    for (int q = 0; q < mv.n_moves; ++q) {
        const int e = mv.entry[q];
        double cur = 0.0, c0 = 0.0;
#pragma unroll
        for (int j = 0; j < 10; ++j) if (j == e) { cur = g[j] * mv.sign[q]; c0 = mv.geom0[(size_t)b * 10 + j] * mv.sign[q]; }
        const double lo = c0 - mv.half_width[q], hi = c0 + mv.half_width[q];
        double x = cur + mv.scale[q] * r.normal();
        int tries = 0;
        while (!(x >= lo && x <= hi)) {
            x = cur + mv.scale[q] * r.normal();
            if (++tries == 10) { x = cur; break; }
        }
#pragma unroll
        for (int j = 0; j < 10; ++j) if (j == e) g[j] = x * mv.sign[q];
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) mv.geom_p[(size_t)b * 10 + j] = g[j];
    td_weights_of(mv, g, mv.weights_p + (size_t)b * n_weights, mv.offset_p != nullptr ? mv.offset_p + (size_t)b * N : nullptr);
    if (mv.rho_scale_p != nullptr) {                             // a moved position: the chain's table set at another distance / height
        // (an on-axis table set -- rho_set == 0: a central-loop system whose receiver or transmitter HEIGHT is sampled -- does not
        //  depend on a horizontal distance at all: scale 1, never 0 / 0)
        mv.rho_scale_p[b] = mv.rho_set[b] > 0.0 ? mv.rho_set[b] / hypot(g[4], g[5]) : 1.0;
        c.height_p[b] = g[0] + 0.5 * (g[6] - mv.dz_set[b]);
    }
}

// What the accept stage decided (chains->step_flags) carried over to the angles: state, posteriors, best state.
__global__ __launch_bounds__(64) void k_td_moves_accept(RjOpt o, gbp_rj_chains c, gbp_td_moves mv, int n_weights, int N)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= c.B) return;
    const int flags = c.step_flags[b];
    if (flags & 1) {
        for (int j = 0; j < 10; ++j) mv.geom[(size_t)b * 10 + j] = mv.geom_p[(size_t)b * 10 + j];
        for (int j = 0; j < n_weights; ++j) mv.weights[(size_t)b * n_weights + j] = mv.weights_p[(size_t)b * n_weights + j];
        if (mv.offset != nullptr)
            for (int j = 0; j < N; ++j) mv.offset[(size_t)b * N + j] = mv.offset_p[(size_t)b * N + j];
        if (mv.rho_scale != nullptr) {
            mv.rho_scale[b] = mv.rho_scale_p[b];
            const_cast<double*>(c.height)[b] = c.height_p[b];
        }
    }
    if (mv.hist != nullptr) {
        int32_t* h = mv.hist + (size_t)b * mv.n_moves * 199;
        if (flags & 4)
            for (int j = 0; j < mv.n_moves * 199; ++j) h[j] = 0;
        if (flags & 8)
            for (int q = 0; q < mv.n_moves; ++q) {
                const int e = mv.entry[q];
                const double v = mv.geom[(size_t)b * 10 + e] * mv.sign[q], c0 = mv.geom0[(size_t)b * 10 + e] * mv.sign[q];
                const double uu = (v - (c0 - mv.half_width[q])) / (2.0 * mv.half_width[q]);
                if (uu >= 0.0 && uu <= 1.0) h[q * 199 + min((int)floor(uu * (double)mv.n_bins[q]), mv.n_bins[q] - 1)] += 1;
            }
    }
    if ((flags & 2) && mv.best_geom != nullptr)
        for (int j = 0; j < 10; ++j) mv.best_geom[(size_t)b * 10 + j] = mv.geom[(size_t)b * 10 + j];
}

// chi^2 / logL with the per-channel additive scale, for the soundings with nl > 0 (one wave per sounding)
__global__ __launch_bounds__(64) void k_td_loglike(RjOpt o, gbp_rj_chains c, const int* __restrict__ nl,
                                                   const double* __restrict__ pred, const double* __restrict__ rel,
                                                   const double* __restrict__ add, double* __restrict__ chi2,
                                                   double* __restrict__ logL)
{
    const int b = blockIdx.x, lane = threadIdx.x, N = o.n_channels;
    if (nl[b] <= 0) return;
    const Levels e = load_levels(o, rel, add, (size_t)b);
    const double* obs = c.data;
    double s2 = 0.0, logdet = 0.0, na = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double ov = obs[(size_t)b * N + i];
        if (ov > 0.0) {
            const double var = variance_at(c, e, ov, i);
            const double r = (pred[(size_t)b * N + i] - ov) * (1.0 / sqrt(var));
            s2 += r * r; logdet += rj_log(var); na += 1.0;
        }
    }
    s2 = wave_sum(s2); logdet = wave_sum(logdet); na = wave_sum(na);
    if (lane == 0) {
        chi2[b] = s2;
        logL[b] = -(0.5 * na) * 1.8378770664093453 - 0.5 * logdet - 0.5 * s2;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Persistent per-chain sampler: ONE workgroup owns a chain and runs all of its iterations in one launch -- propose ->
// fm_dlogc at the remapped model -> stochastic-Newton proposal -> fused forward + chi^2 (or fm_dlogc for a dimension change)
// -> accept / bookkeeping -- with workgroup barriers where the lock-step driver (gbp_rj_run_td below) has kernel boundaries.
// Chains are independent, so nothing is exchanged between workgroups and there is no lock-step tail: a block of 1 024 chains
// (BASELINE config 5 split over 8 GPUs) no longer pays ten dependent launches per iteration.  The stages are the SAME device
// functions the lock-step kernels call, on the same arrays (their results do not depend on the number of waves), so the chains
// are bit-identical to gbp_rj_run's (tests/test_rjmcmc_gpu.py::test_persistent_kernel_walks_the_same_chains).
//   workgroup = persistent_waves() waves for the physics stages; the per-chain algebra (8-lane packed or one-wave variants)
//   runs on wave 0.
//   dynamic LDS = max over the stages (persistent_lds_bytes), the math tables are staged once per workgroup lifetime.
// ---------------------------------------------------------------------------------------------------------------
// The stages are separate (non-inlined) functions so that each gets its own register allocation: inlined into one body the
// kernel needed 300 VGPRs and > 500 SGPR spills (occupancy 1).  They receive pointers only -- the two parameter blocks and the
// math tables sit in LDS for the workgroup's lifetime, the polynomial coefficients are re-read from constant memory.
#ifndef GBP_STAGE_ATTR
#define GBP_STAGE_ATTR __attribute__((noinline))
#endif
#ifndef GBP_PHYS_STAGE_ATTR
#define GBP_PHYS_STAGE_ATTR GBP_STAGE_ATTR
#endif
#ifndef GBP_ACCEPT_STAGE_ATTR
#define GBP_ACCEPT_STAGE_ATTR GBP_STAGE_ATTR
#endif
struct PersistentCtx {
    const RjOpt* o;          // LDS copies
    const gbp_rj_chains* c;
    MathLds* math;
    double* sh_out;
    unsigned char* sh_dyn;
    const Channel* chan;              // abscissa window of the chain's current height ...
    const double* pts;
    const Channel* chan_p;            // ... and of the proposal's (the same with a fixed height; a sampled one: set_windows per iteration)
    const double* pts_p;
    unsigned char* deep_scratch;      // global working set of this chain's Jacobian pass for models of > 8 layers, or NULL: LDS
    int npts_total, npts_total_p, F, nw_deep, b;
    double sigma_direct;
    double alt, alt_p;                // height of the current state / of the proposal
    long long t_start;                // clock at the workgroup's start while gbp_rj_debug_stage_ticks is armed (GBP_RJ_LIFE)
};

// LDS block of one chain in the persistent kernel: the doubles of GBP_RJ_D + data[N], then the int32s of GBP_RJ_I
__host__ __device__ inline size_t persistent_chain_doubles(int K, int N)
{
    return (size_t)9 * K + 4 * (size_t)N + 3 * (size_t)N * K + 11;      // (chol stays in global memory: K^2 doubles, read once per jump)
}
__host__ __device__ inline size_t persistent_chain_bytes(int K, int N)
{
    return (persistent_chain_doubles(K, N) * sizeof(double) + 4 * sizeof(int32_t) + 15) & ~(size_t)15;
}

__device__ __forceinline__ gbp::MathCtx math_ctx(MathLds* lds)    // math_setup without the table fill
{
    gbp::MathCtx M;
    M.k = GBP_K;
    M.e4_v = M.k.e4; M.s2_v = M.k.s2; M.c3_v = M.k.c3;
    asm volatile("" : "+v"(M.e4_v), "+v"(M.s2_v), "+v"(M.c3_v));
    M.exp2_64 = lds->exp2_64;
    M.sincos_64 = lds->sincos_64;
    return M;
}

template <bool EXACT>
__device__ GBP_PHYS_STAGE_ATTR void stage_fm_dlogc(const PersistentCtx* x, int at_proposal)
{
    const RjOpt& o = *x->o;
    const gbp_rj_chains& c = *x->c;
    const int b = x->b, K = o.max_layers, N = o.n_channels, L = c.k_r[b];
    const gbp::MathCtx M = math_ctx(x->math);
    const double* sig = (at_proposal ? c.sigma_p : c.sigma_r) + (size_t)b * K;
    double* Jb = (at_proposal ? c.J_p : c.J_r) + (size_t)b * N * K;
    double* pr = (at_proposal ? c.pred_p : c.pred_r) + (size_t)b * N;
    const double* th = c.thk_r + (size_t)b * K;
    // (a sampled height: the remapped model is evaluated at the chain's current height, the proposal at the proposed one -- k_rj_physics)
    const double alt = at_proposal ? x->alt_p : x->alt;
    const Channel* chan = at_proposal ? x->chan_p : x->chan;
    const double* pts = at_proposal ? x->pts_p : x->pts;
    const int npts = at_proposal ? x->npts_total_p : x->npts_total;
    if (L <= 8) sens_body<EXACT, 1>(M, x->sh_dyn, chan, pts, npts, x->F, K, K < 8 ? K : 8, L, sig, th, alt, Jb, pr, (int)(blockDim.x >> 6), min(K, 8));
    else sens_body<EXACT, GBP_RJ_DEEP_NG>(M, x->deep_scratch != nullptr ? x->deep_scratch : x->sh_dyn, chan, pts, npts, x->F, K, K, L,
                             sig, th, alt, Jb, pr, x->nw_deep, min(K, (L + 7) & ~7));
}

__device__ GBP_PHYS_STAGE_ATTR void stage_forward(const PersistentCtx* x)
{
    const RjOpt& o = *x->o;
    const gbp_rj_chains& c = *x->c;
    const int b = x->b, K = o.max_layers, N = o.n_channels;
    const gbp::MathCtx M = math_ctx(x->math);
    forward_body<true>(M, x->sh_out, x->sh_dyn, x->chan_p, x->pts_p, x->npts_total_p, x->F, K, c.k_r[b], c.sigma_p + (size_t)b * K,
                       c.thk_r + (size_t)b * K, x->alt_p, c.data + (size_t)b * N, c.rel_p[b], c.add_p[b], c.pred_p + (size_t)b * N,
                       c.misfit_p + b, c.like_p + b, x->sigma_direct, (int)(blockDim.x >> 6));
}

__device__ GBP_STAGE_ATTR void stage_propose(const PersistentCtx* x, uint32_t iter, int lane)
{   // the cooperative variant (lane j holds interface / layer j: one row load instead of a dependent walk); same draws and
    // the same remapped model as the thread-per-chain kernel the lock-step driver launches
    if (x->o->max_layers <= 64) propose_wave_body(*x->o, *x->c, iter, x->b, lane);
    else if (lane == 0) propose_thread_body(*x->o, *x->c, iter, x->b);
}

__device__ GBP_STAGE_ATTR void stage_newton(const PersistentCtx* x, uint32_t iter, int lane)
{
    const gbp_rj_chains& c = *x->c;
    if (c.k_r[x->b] <= 8) newton8_body<false>(*x->o, c, iter, lane, lane < 8 ? x->b : c.B, x->sh_dyn, x->b);
    else newton_body(*x->o, c, iter, 8, x->b, lane, x->sh_dyn);
}

__device__ GBP_ACCEPT_STAGE_ATTR void stage_accept(const PersistentCtx* x, uint32_t iter, int accumulate, int lane)
{
    const gbp_rj_chains& c = *x->c;
    const int kr = c.k_r[x->b], kp = c.k[x->b];
    if ((kr > kp ? kr : kp) <= 8) accept8_body<false>(*x->o, c, iter, accumulate, lane, lane < 8 ? x->b : c.B, x->sh_dyn, x->b);
    else accept_body(*x->o, c, iter, accumulate, 8, x->b, lane, x->sh_dyn);
}

// (launch bound 1024 although 64 ... 256 threads are launched: it caps the kernel AND the stage functions it calls at 128 VGPRs,
//  the budget the same code has in the lock-step kernels; without it every stage takes ~250 registers: one wave per SIMD)
// Stage clock of chain 0 (s_memtime ticks of the 100 MHz constant clock, accumulated over the launches since the last reset):
// propose | fm_dlogc at the remapped model | newton | forward or fm_dlogc at the proposal | accept | iterations.  Read with
// gbp_rj_debug_stage_ticks; costs one branch per stage in the other workgroups.
__device__ long long GBP_RJ_TICKS[8];
// ... and, while the clock is armed, the life of EVERY workgroup of the launches: [0] sum, [1] count, [2] longest (ticks).  A persistent
// launch ends with its slowest chain (round 6: chain 0's stages sum to 42 us per iteration where the launch takes 53).
__device__ unsigned long long GBP_RJ_LIFE[3];

template <bool EXACT>
__global__ GBP_RJ_PERSISTENT_BOUNDS void k_rj_persistent(RjOpt o_arg, gbp_rj_chains c_arg, const Channel* __restrict__ chan,
                                                       const double* __restrict__ pts, int npts_total, int F, double sigma_direct,
                                                       uint32_t iter0, int n_iter, int accumulate, int nw_deep,
                                                       unsigned char* deep_scratch, size_t deep_bytes, const BinDesc* __restrict__ bins,
                                                       int bin0, int n_bins, const Channel* __restrict__ bin_chan,
                                                       const double* __restrict__ bin_pts)
{
    __shared__ double sh_out[2 * GBP_MAX_FREQ];
    __shared__ MathLds sh_math;
    __shared__ RjOpt sh_o;
    __shared__ gbp_rj_chains sh_c;
    __shared__ PersistentCtx sh_x;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = o_arg.max_layers, N = o_arg.n_channels;
    // The chain's rows of every per-chain array -- carried state AND the scratch the stages hand to each other -- live in LDS
    // for the whole launch: sh_c's pointers are re-based so that `ptr + b * row` lands on the LDS copy (generic pointers:
    // the stage functions are unchanged and do the same arithmetic).  A dependent global load costs ~1 us, an LDS one ~0.1 us,
    // and the per-chain algebra is chains of them.  The stage scratch (sh_dyn) follows the chain block.
    double* ld = reinterpret_cast<double*>(sh_dyn);
    int32_t* li = reinterpret_cast<int32_t*>(ld + persistent_chain_doubles(K, N));
    unsigned char* stage_scratch = sh_dyn + persistent_chain_bytes(K, N);
    if (threadIdx.x == 0) { sh_o = o_arg; sh_c = c_arg; }
    __syncthreads();
#define GBP_RJ_D(X)                                                                                                        \
    X(edges, K, 1) X(sigma, K, 1) X(rel, 1, 1) X(add, 1, 1) X(pred, N, 1) X(J, N * K, 1) X(prior, 1, 1) X(like, 1, 1)      \
    X(misfit, 1, 1) X(best_posterior, 1, 1) X(best_edges, K, 1) X(best_sigma, K, 1) X(edges_r, K, 1) X(sigma_r, K, 1)       \
    X(thk_r, K, 1) X(rel_p, 1, 1) X(add_p, 1, 1) X(pred_r, N, 1) X(J_r, N * K, 1) X(log_prop, K, 1)       \
    X(sigma_p, K, 1) X(pred_p, N, 1) X(misfit_p, 1, 1) X(like_p, 1, 1) X(J_p, N * K, 1) X(log_ratio, 1, 1)
#define GBP_RJ_I(X) X(k, 1, 1) X(best_k, 1, 1) X(action, 1, 1) X(k_r, 1, 1)
    {
        double* p = ld;
        int32_t* q = li;
        const size_t bb = (size_t)b;
#define GBP_IN_D(name, n, carried)                                                                                       \
    {                                                                                                                    \
        const size_t cnt = (size_t)(n);                                                                                  \
        if (carried) for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) p[i] = c_arg.name[bb * cnt + i];             \
        if (threadIdx.x == 0) sh_c.name = p - bb * cnt;                                                                  \
        p += cnt;                                                                                                        \
    }
        GBP_RJ_D(GBP_IN_D)
#undef GBP_IN_D
#define GBP_IN_I(name, n, carried)                                                                                       \
    {                                                                                                                    \
        if (carried && threadIdx.x == 0) q[0] = c_arg.name[bb];                                                          \
        if (threadIdx.x == 0) sh_c.name = q - bb;                                                                        \
        q += 1;                                                                                                          \
    }
        GBP_RJ_I(GBP_IN_I)
#undef GBP_IN_I
        // read-only inputs
        for (size_t i = threadIdx.x; i < (size_t)N; i += blockDim.x) p[i] = c_arg.data[bb * N + i];
        if (threadIdx.x == 0) sh_c.data = p - bb * N;
    }
    // the abscissa window of a height: the bin of the altitude (the system's full tables below the first bin / without bins)
    auto window_of = [&](double alt, const Channel*& wc, const double*& wp, int& wn) {
        wc = chan; wp = pts; wn = npts_total;
        if (bins != nullptr && alt >= (double)bin0) {
            const BinDesc d = bins[min((int)(alt - (double)bin0), n_bins - 1)];
            wc = bin_chan + d.chan_off;
            wp = bin_pts + d.pts_off;
            wn = d.npts_total;
        }
    };
    // (thread 0: the current height's window and -- a sampled height, after the proposal kernel's stage -- the proposal's)
    auto set_windows = [&](bool proposal_too) {
        sh_x.alt = c_arg.height[b];
        window_of(sh_x.alt, sh_x.chan, sh_x.pts, sh_x.npts_total);
        if (proposal_too) {
            sh_x.alt_p = c_arg.height_p[b];
            window_of(sh_x.alt_p, sh_x.chan_p, sh_x.pts_p, sh_x.npts_total_p);
        } else {
            sh_x.alt_p = sh_x.alt; sh_x.chan_p = sh_x.chan; sh_x.pts_p = sh_x.pts; sh_x.npts_total_p = sh_x.npts_total;
        }
    };
    if (threadIdx.x == 0) {
        sh_x.o = &sh_o; sh_x.c = &sh_c; sh_x.math = &sh_math; sh_x.sh_out = sh_out; sh_x.sh_dyn = stage_scratch;
        sh_x.deep_scratch = deep_scratch != nullptr ? deep_scratch + (size_t)b * deep_bytes : nullptr;
        sh_x.F = F; sh_x.nw_deep = nw_deep; sh_x.b = b; sh_x.sigma_direct = sigma_direct;
        set_windows(false);
    }
    const bool moving_height = o_arg.solve_height != 0;
    (void)math_setup(sh_math);                                    // tables -> LDS once per chain; ends with __syncthreads()
    const int32_t* status = c_arg.status;
    const int32_t* action_p = sh_c.action;
    const int schedule = o_arg.schedule;
    const bool clocked = b == 0 && threadIdx.x == 0 && GBP_RJ_TICKS[7] != 0;   // armed by gbp_rj_debug_stage_ticks(out, 1)
    long long t0 = clocked ? (long long)wall_clock64() : 0;
    // (the workgroup's start time waits in LDS, not in registers the iteration loop would carry across its calls)
    if (threadIdx.x == 0 && GBP_RJ_TICKS[7] != 0) sh_x.t_start = (long long)wall_clock64();
    auto tick = [&](int stage) {
        if (clocked) { const long long t1 = (long long)wall_clock64(); GBP_RJ_TICKS[stage] += t1 - t0; t0 = t1; }
    };
    for (int it = 0; it < n_iter; ++it) {
        const uint32_t iter = iter0 + (uint32_t)it;
        if (schedule == 1 && status[b] != 0) break;               // done / failed chains keep their final state (workgroup-uniform)
        if (wave == 0) { GBP_RJ_SERIAL_PRIO(GBP_RJ_PERSISTENT_PRIO); stage_propose(&sh_x, iter, lane); GBP_RJ_SERIAL_PRIO(0); }
        if (moving_height) {                                      // Point.perturb: this iteration's two heights and their windows
            __syncthreads();                                      //   (height_p is lane 0's write of the stage above; height its
            if (threadIdx.x == 0) set_windows(true);              //   write of the accept stage of the iteration before)
        }
        __syncthreads();
        tick(0);
        const int action = action_p[b];
        if (action != NONE) {                                     // Model.py:383-384: prediction + Jacobian at the remapped model
            stage_fm_dlogc<EXACT>(&sh_x, 0);
            __syncthreads();
        }
        tick(1);
        if (wave == 0) { GBP_RJ_SERIAL_PRIO(GBP_RJ_PERSISTENT_PRIO); stage_newton(&sh_x, iter, lane); GBP_RJ_SERIAL_PRIO(0); }
        __syncthreads();
        tick(2);
        if (action == INSERT || action == DELETE) stage_fm_dlogc<EXACT>(&sh_x, 1);   // Model.py:612: Jacobian + prediction at the proposal
        else stage_forward(&sh_x);                                // Inference1D.py:572-597: forward + chi^2 + logL of the proposal
        __syncthreads();
        tick(3);
        if (wave == 0) { GBP_RJ_SERIAL_PRIO(GBP_RJ_PERSISTENT_PRIO); stage_accept(&sh_x, iter, accumulate, lane); GBP_RJ_SERIAL_PRIO(0); }
        __syncthreads();
        tick(4);
        if (clocked) GBP_RJ_TICKS[5] += 1;
    }
    if (threadIdx.x == 0 && GBP_RJ_TICKS[7] != 0) {
        const unsigned long long life = (unsigned long long)((long long)wall_clock64() - sh_x.t_start);
        atomicAdd(&GBP_RJ_LIFE[0], life);
        atomicAdd(&GBP_RJ_LIFE[1], 1ull);
        atomicMax(&GBP_RJ_LIFE[2], life);
    }
    {   // everything back to the block's arrays (scratch too: the arrays end as the lock-step driver leaves them)
        const double* p = ld;
        const int32_t* q = li;
        const size_t bb = (size_t)b;
#define GBP_OUT_D(name, n, carried)                                                                                      \
    {                                                                                                                    \
        const size_t cnt = (size_t)(n);                                                                                  \
        for (size_t i = threadIdx.x; i < cnt; i += blockDim.x) c_arg.name[bb * cnt + i] = p[i];                          \
        p += cnt;                                                                                                        \
    }
        GBP_RJ_D(GBP_OUT_D)
#undef GBP_OUT_D
#define GBP_OUT_I(name, n, carried)                                                                                      \
    {                                                                                                                    \
        if (threadIdx.x == 0) c_arg.name[bb] = q[0];                                                                     \
        q += 1;                                                                                                          \
    }
        GBP_RJ_I(GBP_OUT_I)
#undef GBP_OUT_I
    }
#undef GBP_RJ_D
#undef GBP_RJ_I
}

// ---------------------------------------------------------------------------------------------------------------
// Lock-step driver, fused physics launches (frequency-domain data): ONE launch per stage instead of one per kind of evaluation
// and layer-count bucket.  Stage 0 = prediction + Jacobian at the remapped model of the chains whose structure changed; stage 1 =
// the evaluation at the proposal -- prediction + Jacobian for the chains that change dimension, fused forward + chi^2 + logL for
// all others.  A workgroup owns a chain and branches on its move (workgroup-uniform), calling the bodies of k_fdem_sens /
// k_fdem_forward: same values (they do not depend on the wave count), 7 launches per iteration instead of 10, and the two kinds
// of evaluation of stage 1 fill the GPU together instead of one after the other with a tail each.  Models of more than 8 layers
// keep their Jacobian working set in a per-chain global block (as in the persistent kernel), so the LDS block is the small one.
// ---------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ GBP_RJ_PHYSICS_BOUNDS void k_rj_physics(RjOpt o, gbp_rj_chains c, const Channel* __restrict__ chan,
                                                    const double* __restrict__ pts, int npts_total, int F, double sigma_direct,
                                                    int stage, unsigned char* deep_scratch, size_t deep_bytes,
                                                    const BinDesc* __restrict__ bins, int bin0, int n_bins,
                                                    const Channel* __restrict__ bin_chan, const double* __restrict__ bin_pts, int out_offset,
                                                    const int32_t* __restrict__ order, int n_split)
{
    // The output row lives behind the stages' working set in the dynamic block (out_offset) instead of a static 2 * GBP_MAX_FREQ doubles:
    // with two waves per chain the workgroup's LDS was 20 544 B -- 64 B more than an eighth of a CU's 160 KB -- i.e. seven resident
    // workgroups (3.5 waves per SIMD) where the 128-VGPR budget allows eight; the sampler's rows are n_channels doubles (160 B).
    __shared__ MathLds sh_math;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    double* sh_out = reinterpret_cast<double*>(sh_dyn + out_offset);
    // Workgroups B ... 2 B - 1 (launched when the Jacobian passes are split, GBP_RJ_JACOBIAN_SHARES = 2) are the second halves of the
    // chains' Jacobian evaluations: a launch lasts as long as its slowest workgroup, the Jacobian workgroups are the slow ones, and a
    // frequency's rows depend on nothing else -- two workgroups take five frequencies each and write the bits one would.
    // `order` (round 6): workgroup g takes chain order[g] -- the launch's chains by descending layer count at the start of the call;
    // n_split > 0: the first n_split of them have a second workgroup each (workgroups B ... B + n_split - 1), used by the Jacobian passes
    // of GBP_RJ_SPLIT_MIN_LAYERS or more layers (both workgroups read the same move and layer count: the same decision)
    int share = (int)blockIdx.x >= c.B ? 1 : 0;
    int n_shares = (int)gridDim.x > c.B ? 2 : 1;
    const int slot = (int)blockIdx.x - share * c.B;
    const int b = order != nullptr ? order[slot] : slot;
    if (n_split > 0 && slot >= n_split) n_shares = 1;
#ifdef GBP_RJ_PHYS_CLOCK
    const long long clk_start_ = (long long)wall_clock64();
    PhysClk clk_{threadIdx.x == 0 && (b & 15) == 0, 0, clk_start_};
    PhysClk* gbp_clk = &clk_;
#endif
    // The chain's move, layer count and height are requested together, and the math tables go to LDS while they travel (no barrier of
    // their own: forward_body / sens_body have one behind the layer-thickness fill) -- one trip to memory and one barrier less per workgroup
#ifndef GBP_RJ_PHYSICS_LATE_TABLES
    const int action = c.action[b];
    const int L_early = c.k_r[b];
    const double alt_early = (stage == 1 && o.solve_height) ? c.height_p[b] : c.height[b];
    const gbp::MathCtx M = math_setup<false>(sh_math);
#else
    const int action = c.action[b];
    const int L_early = c.k_r[b];
    const double alt_early = (stage == 1 && o.solve_height) ? c.height_p[b] : c.height[b];
#endif
    if (stage == 0 && action == NONE) return;                     // (workgroup-uniform)
    if (n_split > 0 && n_shares == 2 && L_early < GBP_RJ_SPLIT_MIN_LAYERS) n_shares = 1;   // (a shallow model: not worth a second prologue)
    if (share != 0 && (n_shares == 1 || (stage == 1 && action != INSERT && action != DELETE) || L_early > GBP_RJ_PHYSICS_LDS_LAYERS)) return;   // (a fused forward, or a deep model: one workgroup)
    if (n_shares == 2 && ((stage == 1 && action != INSERT && action != DELETE) || L_early > GBP_RJ_PHYSICS_LDS_LAYERS)) n_shares = 1;
#ifdef GBP_RJ_PHYS_CLOCK
    clk_.base = (stage == 0 ? 0 : ((action == INSERT || action == DELETE) ? 8 : 16));
#endif
    GBP_TICK(0);
    const int K = o.max_layers, N = o.n_channels, L = L_early;
    // (a sampled height: the remapped model is evaluated at the chain's current height, the proposal at the proposed one)
    const double alt = alt_early;
    if (bins != nullptr && alt >= (double)bin0) {                 // the chain's abscissa window: the bin of its sounding's altitude
        const BinDesc d = bins[min((int)(alt - (double)bin0), n_bins - 1)];
        chan = bin_chan + d.chan_off;
        pts = bin_pts + d.pts_off;
        npts_total = d.npts_total;
    }
    GBP_TICK(1);
#ifdef GBP_RJ_PHYSICS_LATE_TABLES
    const gbp::MathCtx M = math_setup(sh_math);                   // ends with __syncthreads()
#endif
    GBP_TICK(2);
    const bool jump = action == INSERT || action == DELETE;
    const int nw = (int)(blockDim.x >> 6);
    if (stage == 0 || jump) {
        const bool at_proposal = stage == 1;
        const double* sig = (at_proposal ? c.sigma_p : c.sigma_r) + (size_t)b * K;
        double* Jb = (at_proposal ? c.J_p : c.J_r) + (size_t)b * N * K;
        double* pr = (at_proposal ? c.pred_p : c.pred_r) + (size_t)b * N;
        const double* th = c.thk_r + (size_t)b * K;
        if (L <= GBP_RJ_PHYSICS_LDS_LAYERS) sens_body<EXACT, 1>(M, sh_dyn, chan, pts, npts_total, F, K, K < GBP_RJ_PHYSICS_LDS_LAYERS ? K : GBP_RJ_PHYSICS_LDS_LAYERS, L, sig, th, alt, Jb, pr, nw, min(K, 8), 1.0 GBP_TICK_PASS, share, n_shares);
        else sens_body<EXACT, GBP_RJ_DEEP_NG>(M, deep_scratch + (size_t)b * deep_bytes, chan, pts, npts_total, F, K, K, L, sig, th, alt, Jb, pr, nw,
                                 min(K, (L + 7) & ~7));
    } else {
        forward_body<true>(M, sh_out, sh_dyn, chan, pts, npts_total, F, K, L, c.sigma_p + (size_t)b * K, c.thk_r + (size_t)b * K, alt,
                           c.data + (size_t)b * N, c.rel_p[b], c.add_p[b], c.pred_p + (size_t)b * N, c.misfit_p + b, c.like_p + b,
                           sigma_direct, nw, 1.0 GBP_TICK_PASS);
    }
#ifdef GBP_RJ_PHYS_CLOCK
    if (threadIdx.x == 0) {                            // life of every workgroup (thread 0's wave) by kind and layer count
        const long long life = (long long)wall_clock64() - clk_start_;
        long long* h = GBP_PHYS_LIFE + ((stage == 0 ? 0 : (jump ? 1 : 2)) * 16 + min(L, 15)) * 3;
        atomicAdd((unsigned long long*)&h[0], (unsigned long long)life);
        atomicAdd((unsigned long long*)&h[1], 1ull);
        atomicMax((unsigned long long*)&h[2], (unsigned long long)life);
    }
#endif
}

}  // namespace rj

// ---------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------
namespace {
// Helper streams of the drivers: ONE pool of three per device and host thread, created on first use and kept for the life of the
// process.  The runtime maps streams onto a handful of hardware queues (four by default); streams that share a queue run one
// after the other, so every driver draws from the same three -- with the caller's stream that makes four -- instead of each
// creating its own (a process that had run the sub-block driver before the time-domain one lost 13 % in the latter: its side and
// deep streams landed on the queues of the four sub-block streams created earlier).
struct AuxStreams {
#ifndef GBP_AUX_STREAMS
#define GBP_AUX_STREAMS 3
#endif
    static const int N = GBP_AUX_STREAMS;
    hipStream_t q[N] = {};
    hipEvent_t fork[N] = {}, join[N] = {};
    hipEvent_t start = nullptr;
};
AuxStreams* aux_streams()
{
    static thread_local AuxStreams table[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    AuxStreams& s = table[dev];
    if (s.start == nullptr) {
        for (int i = 0; i < AuxStreams::N; ++i) {
            if (hipStreamCreateWithFlags(&s.q[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&s.fork[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        if (hipEventCreateWithFlags(&s.start, hipEventDisableTiming) != hipSuccess) { s.start = nullptr; return nullptr; }
    }
    return &s;
}

// The side stream of the evaluations at the proposals (+ fork / join events): aux stream 0.
struct SideStream {
    hipStream_t q = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
};
SideStream* side_stream()
{
    static thread_local SideStream view[64];
    AuxStreams* a = aux_streams();
    int dev = 0;
    if (a == nullptr || hipGetDevice(&dev) != hipSuccess) return nullptr;
    view[dev].q = a->q[0]; view[dev].fork = a->fork[0]; view[dev].join = a->join[0];
    return &view[dev];
}

// The streams of the Jacobian launches of the models of more than 8 layers (see rj_run_lockstep): aux streams 1 and 2.
struct DeepStreams {
    hipStream_t q[2] = {nullptr, nullptr};
    hipEvent_t fork[2] = {nullptr, nullptr}, join[2] = {nullptr, nullptr};
};
DeepStreams* deep_streams()
{
    static thread_local DeepStreams view[64];
    AuxStreams* a = aux_streams();
    int dev = 0;
    if (a == nullptr || hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < 2; ++i) { view[dev].q[i] = a->q[1 + i]; view[dev].fork[i] = a->fork[1 + i]; view[dev].join[i] = a->join[1 + i]; }
    return &view[dev];
}

// The streams of the concurrent sub-blocks of a lock-step run, each with a "done" event the caller's stream waits on: the aux
// streams again (the sub-block driver forks nothing else).
struct BlockStreams {
    static const int MAX = AuxStreams::N;
    hipStream_t q[MAX] = {};
    hipEvent_t done[MAX] = {};
    hipEvent_t start = nullptr;
};
BlockStreams* block_streams()
{
    static thread_local BlockStreams view[64];
    AuxStreams* a = aux_streams();
    int dev = 0;
    if (a == nullptr || hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < BlockStreams::MAX; ++i) { view[dev].q[i] = a->q[i]; view[dev].done[i] = a->join[i]; }
    view[dev].start = a->start;
    return &view[dev];
}

// Rows b0 .. b0 + n - 1 of a block of chains as a block of its own.  Every per-chain array is offset by its row width; the two
// [3, B] launch masks nl_a / nl_c (row stride = the block's own B) become the [3, n] region starting at 3 * b0 -- the sub-blocks
// of a run partition the array, what they hold is scratch of one iteration.
gbp_rj_chains slice_chains(const gbp_rj_options& o, const gbp_rj_chains& c, int b0, int n)
{
    gbp_rj_chains s = c;
    s.B = n;
    const size_t K = (size_t)o.max_layers, N = (size_t)o.n_channels, Gr = (size_t)o.n_rel_groups, Ga = (size_t)o.n_add_groups;
    const size_t nb = (size_t)o.n_error_bins, nd = (size_t)o.n_depth_bins, nv = (size_t)o.n_value_bins, b = (size_t)b0;
#define GBP_OFF(field, width) if (s.field != nullptr) s.field = c.field + b * (width);
    GBP_OFF(chain_id, 1) GBP_OFF(data, N) GBP_OFF(height, 1) GBP_OFF(log_mean_prior, 1) GBP_OFF(k, 1) GBP_OFF(edges, K) GBP_OFF(sigma, K)
    GBP_OFF(rel, Gr) GBP_OFF(add, Ga) GBP_OFF(pred, N) GBP_OFF(J, N * K) GBP_OFF(prior, 1) GBP_OFF(like, 1) GBP_OFF(misfit, 1)
    GBP_OFF(action, 1) GBP_OFF(k_r, 1) GBP_OFF(nl_a, 3) GBP_OFF(nl_c, 3) GBP_OFF(nl_b, 1) GBP_OFF(edges_r, K) GBP_OFF(sigma_r, K) GBP_OFF(thk_r, K)
    GBP_OFF(rel_p, Gr) GBP_OFF(add_p, Ga) GBP_OFF(pred_r, N) GBP_OFF(J_r, N * K) GBP_OFF(chol, K * K) GBP_OFF(log_prop, K) GBP_OFF(sigma_p, K)
    GBP_OFF(pred_p, N) GBP_OFF(misfit_p, 1) GBP_OFF(like_p, 1) GBP_OFF(J_p, N * K) GBP_OFF(log_ratio, 1) GBP_OFF(n_accepted, 1)
    GBP_OFF(k_hist, K + 1) GBP_OFF(edge_hist, nd) GBP_OFF(rel_hist, Gr * nb) GBP_OFF(add_hist, Ga * nb) GBP_OFF(hitmap, nv * nd) GBP_OFF(hit_dwell, 1)
    GBP_OFF(burned_in_iteration, 1) GBP_OFF(status, 1) GBP_OFF(best_posterior, 1) GBP_OFF(best_k, 1) GBP_OFF(best_edges, K) GBP_OFF(best_sigma, K)
    GBP_OFF(best_rel, Gr) GBP_OFF(best_add, Ga) GBP_OFF(iteration0, 1)
    GBP_OFF(height_p, 1) GBP_OFF(height0, 1) GBP_OFF(height_hist, nb) GBP_OFF(best_height, 1) GBP_OFF(step_flags, 1)
    GBP_OFF(trace_misfit, (size_t)o.trace_length) GBP_OFF(trace_accept, (size_t)o.trace_length) GBP_OFF(best_iteration, 1)
#undef GBP_OFF
    return s;
}

gbp_status rj_check(const gbp_rj_options* o, const gbp_rj_chains* c)
{
    if (!o || !c) return fail(GBP_ERR_INVALID_ARG, "options / chains is NULL%s");
    if (o->max_layers < 2 || o->max_layers > 64) return fail(GBP_ERR_INVALID_ARG, "max_layers must be in [2, 64]%s");
    if (o->n_channels < 1 || o->n_channels > 2 * GBP_MAX_FREQ) return fail(GBP_ERR_INVALID_ARG, "n_channels out of range%s");
    if (c->B < 0) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0%s");
    if (!(o->min_width > 0.0) || !(o->max_edge > o->min_edge) || !(o->min_edge > 0.0))
        return fail(GBP_ERR_INVALID_ARG, "need 0 < min_edge < max_edge and min_width > 0%s");
    if (c->B == 0) return GBP_OK;       // an empty block (a rank without a flight line): no array is touched, and an empty device array has no address
    if (o->schedule == 1 && (!c->burned_in_iteration || !c->status))
        return fail(GBP_ERR_INVALID_ARG, "schedule 1 needs burned_in_iteration and status%s");
    if ((c->edge_hist || c->hitmap) && (o->n_depth_bins < 1 || !(o->depth_bin_width > 0.0)))
        return fail(GBP_ERR_INVALID_ARG, "posterior depth grid is empty%s");
    if (c->hitmap && (o->n_value_bins < 1 || !(o->value_half_width > 0.0))) return fail(GBP_ERR_INVALID_ARG, "hit-map value grid is empty%s");
    if (c->hitmap && !c->hit_dwell) return fail(GBP_ERR_INVALID_ARG, "hitmap needs hit_dwell%s");
    if (o->n_rel_groups < 1 || o->n_rel_groups > 4 || o->n_add_groups < 1 || o->n_add_groups > 4)
        return fail(GBP_ERR_INVALID_ARG, "n_rel_groups / n_add_groups must be in [1, 4]%s");
    if (o->solve_height && (!c->height_p || !c->height0 || !(o->height_half_width > 0.0) || !(o->height_scale >= 0.0)))
        return fail(GBP_ERR_INVALID_ARG, "solve_height needs height_p, height0, height_half_width > 0 and height_scale >= 0%s");
    if (c->height_hist && (!o->solve_height || o->n_error_bins < 1)) return fail(GBP_ERR_INVALID_ARG, "height_hist needs solve_height and n_error_bins >= 1%s");
    if ((c->trace_misfit != nullptr) != (c->trace_accept != nullptr) || (c->trace_misfit && (o->trace_every < 1 || o->trace_length < 1)))
        return fail(GBP_ERR_INVALID_ARG, "trace_misfit and trace_accept come together, with trace_every >= 1 and trace_length >= 1%s");
    if ((c->rel_hist != nullptr) != (c->add_hist != nullptr) || (c->rel_hist && o->n_error_bins < 1))
        return fail(GBP_ERR_INVALID_ARG, "rel_hist and add_hist come together, with n_error_bins >= 1%s");
    const void* need[] = {c->data, c->height, c->log_mean_prior, c->k, c->edges, c->sigma, c->rel, c->add, c->pred, c->J, c->prior,
                          c->like, c->misfit, c->action, c->k_r, c->nl_a, c->nl_b, c->nl_c, c->edges_r, c->sigma_r, c->thk_r, c->rel_p,
                          c->add_p, c->pred_r, c->J_r, c->chol, c->log_prop, c->sigma_p, c->pred_p, c->misfit_p, c->like_p,
                          c->J_p, c->log_ratio, c->n_accepted, c->k_hist, c->best_posterior, c->best_k, c->best_edges, c->best_sigma};
    if (c->B > 0)
        for (const void* p : need)
            if (!p) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer in gbp_rj_chains%s");
    return GBP_OK;
}
}  // namespace

extern "C" {

gbp_status gbp_rj_propose(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, void* stream)
{
    return gbp_rj_debug_propose_variant(o, c, iteration, 0, stream);
}

// variant 0: the product's choice -- one thread per chain, rows staged through LDS (the faster variant at every block size
// measured, 256 ... 65536 chains; the unstaged kernel when max_layers > 42 rows do not fit); 1: unstaged thread per chain;
// 2: cooperative one wave per chain.  The three are independent implementations of the same draws (tests hold them bit-equal).
gbp_status gbp_rj_debug_propose_variant(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, int variant, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK || c->B == 0) return st;
    if (variant < 0 || variant > 2) return fail(GBP_ERR_INVALID_ARG, "variant must be 0, 1 or 2%s");
    const size_t lds = (size_t)3 * GBP_RJ_PROPOSE_ROWS * (o->max_layers | 1) * sizeof(double);
    if (variant == 2)
        hipLaunchKernelGGL(rj::k_rj_propose_wave, dim3((c->B + 3) / 4), dim3(256), 0, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration);
    else if (variant == 1 || lds > 64 * 1024)          // (rows too long to stage 64 of them: the unstaged kernel)
        hipLaunchKernelGGL(rj::k_rj_propose_thread, dim3((c->B + 127) / 128), dim3(128), 0, (hipStream_t)stream, rj::extend(*o), *c,
                           (uint32_t)iteration);
    else
        hipLaunchKernelGGL(rj::k_rj_propose_staged, dim3((c->B + GBP_RJ_PROPOSE_ROWS - 1) / GBP_RJ_PROPOSE_ROWS),
                           dim3(GBP_RJ_PROPOSE_THREADS), lds, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

// The lock-step launches take the build of the packed stages that issues its memory trips in batches (for_column); the persistent
// kernel, whose rows are in LDS, the plain loops.  Measured (scripts/bench_rj_parts.py, A/B builds interleaved on one box, M
// chain-iterations/s plain / batched, ten frequencies | Resolve): 2 048 chains 17.4 / 18.8 | 19.9 / 21.0, 4 096: 29.5 / 30.5, 8 192:
// 43.1 / 44.0 | 49.9 / 51.0 -- k_rj_newton8 24.9 -> 20.0 us, k_rj_accept8 34.0 -> 28.1 us per launch of 2 731 chains --, 65 536:
// 56.1 / 56.2; the persistent kernel with the batched build 17.9 -> 17.2 | 20.1 -> 19.7.  (At 8 192 chains and beyond the iteration
// is bound by the physics launches of the three sub-blocks, which fill the GPU: shorter per-chain stages change little there.  Atomic
// adds -- sent and forgotten -- instead of `+=` in the hit map and the counters were measured with it: no difference.)
static bool packed_trips(int B)
{
    (void)B;
#ifdef GBP_RJ_AB_TRIPS
    return GBP_RJ_AB_TRIPS != 0;                     // (A/B builds under scripts/ab only)
#endif
    return true;
}

// The launches of a per-chain stage: ONE (packed workgroups, then a wave per chain for the models above 8 layers) or TWO (packed;
// deep -- mostly workgroups that exit at once).  Measured (scripts/bench_rj_parts.py / bench_tdem_sampler.py, A/B builds interleaved
// on one box, M chain-iterations/s two / one): the time-domain driver, 14 launches per iteration with a fork and a join around the
// deep Newton launch, 3.35 / 3.70 at 1 024 chains and 11.2 / 11.8 at 8 192; frequency-domain sub-blocks (chains per launch =
// block / 3): 2 048 chains 18.7 / 19.1, 4 096: 30.8 / 30.8, 8 192: 43.7 / 42.6, 16 384: 52.2 / 50.4 -- with three sub-blocks in flight
// two short launches interleave better with the other sub-blocks' physics than one --; one block of 65 536: 56.4 / 57.1.
// (Folding the deep body into the packed WAVES was tried first: inlined it takes the accept stage from 111 to 178 VGPRs, as a call it
// adds 1.1 - 1.3 KB of scratch per lane; so was running the deep bodies as calls inside the stage-1 physics launch: 104 -> 288 B of
// scratch there.)
// Round 5: the deep chains' part of a stage is B / 64 scanning workgroups (k_rj_newton) instead of B that exit at once, so the one launch
// no longer drags 2 731 empty workgroups through the packed kernel's register budget: ONE launch per stage at every size -- five launches
// per iteration and sub-block instead of seven.  A/B on one box (scripts/ab_rj.py, ten frequencies, M chain-iterations/s, two repeats;
// round-4 library | scanning workgroups with the round-4 rule | scanning + one launch): 2 048 chains 18.99 | 18.94 | 18.92, 4 096:
// 31.23 | 31.20 | 31.19, 8 192: 44.30 | 43.35 | 44.89.  (GBP_RJ_TWO_STAGE_LAUNCHES restores the round-4 rule for A/B builds.)
static bool one_stage_launch(int n, bool time_domain)
{
#ifdef GBP_RJ_TWO_STAGE_LAUNCHES
    return time_domain || n <= 1536 || n >= 32768;
#else
    (void)n; (void)time_domain;
    return true;
#endif
}

static gbp_status rj_newton_launch(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, bool one, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK || c->B == 0) return st;
    const int n_packed = (c->B + 7) / 8, n_deep = o->max_layers > 8 ? (c->B + 63) / 64 : 0;      // (deep: scanning workgroups of 64 chains)
    const size_t lds8 = (size_t)16 * o->n_channels * sizeof(double), lds_deep = rj::Lds::bytes(o->max_layers, o->n_channels);
    auto launch = [&](int grid, size_t lds, int np) {
        if (packed_trips(c->B))
            hipLaunchKernelGGL(rj::k_rj_newton8<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration, np);
        else
            hipLaunchKernelGGL(rj::k_rj_newton8<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration, np);
    };
    if (one || n_deep == 0) launch(n_packed + n_deep, n_deep ? std::max(lds8, lds_deep) : lds8, n_packed);
    else {
        launch(n_packed, lds8, n_packed);
        hipLaunchKernelGGL(rj::k_rj_newton, dim3(n_deep), dim3(64), lds_deep, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration, 8);
    }
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

static gbp_status rj_accept_launch(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, int accumulate, bool one, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK || c->B == 0) return st;
    const int n_packed = (c->B + 7) / 8, n_deep = o->max_layers > 8 ? (c->B + 63) / 64 : 0;      // (deep: scanning workgroups of 64 chains)
    const size_t lds8 = (size_t)8 * o->n_channels * sizeof(double), lds_deep = rj::Lds::bytes(o->max_layers, o->n_channels);
    auto launch = [&](int grid, size_t lds, int np) {
        if (packed_trips(c->B))
            hipLaunchKernelGGL(rj::k_rj_accept8<true>, dim3(grid), dim3(64), lds, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration,
                               accumulate, np);
        else
            hipLaunchKernelGGL(rj::k_rj_accept8<false>, dim3(grid), dim3(64), lds, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration,
                               accumulate, np);
    };
    if (one || n_deep == 0) launch(n_packed + n_deep, n_deep ? std::max(lds8, lds_deep) : lds8, n_packed);
    else {
        launch(n_packed, lds8, n_packed);
        hipLaunchKernelGGL(rj::k_rj_accept, dim3(n_deep), dim3(64), lds_deep, (hipStream_t)stream, rj::extend(*o), *c, (uint32_t)iteration,
                           accumulate, 8);
    }
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_rj_newton(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, void* stream)
{
    return rj_newton_launch(o, c, iteration, one_stage_launch(c ? c->B : 0, false), stream);
}

gbp_status gbp_rj_accept(const gbp_rj_options* o, const gbp_rj_chains* c, int64_t iteration, int accumulate, void* stream)
{
    return rj_accept_launch(o, c, iteration, accumulate, one_stage_launch(c ? c->B : 0, false), stream);
}

static gbp_status rj_run_lockstep(const gbp_fdem_system* sys, const gbp_td_operator* td, const gbp_rj_options* o, const gbp_rj_chains* c,
                                  int64_t first_iteration, int n_iterations, int accumulate, bool fused, int parts, void* stream);

// compute units of the current device (the persistent kernel's capacity: one query per process and device)
static int device_cus()
{
    static thread_local int cached_dev = -1, cached = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cached;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) { cached = n; cached_dev = dev; }
    }
    return cached;
}

static size_t sens_lds_bytes(int nw, int Lalloc)
{
    return (size_t)nw * Lalloc * (GBP_SENS_STRIDE * sizeof(cplx) + sizeof(gbp::LayerK)) + (size_t)Lalloc * sizeof(double);
}

static size_t persistent_lds_bytes(const gbp_fdem_system* sys, const gbp_rj_options* o, int nw)
{
    const int K = o->max_layers, N = o->n_channels;
    size_t stage = dyn_lds_bytes(nw, K, (sys->t.npts + 63) / 64);
    stage = std::max(stage, sens_lds_bytes(nw, K < 8 ? K : 8));
    stage = std::max(stage, rj::Lds::bytes(K, N));
    stage = std::max(stage, (size_t)16 * N * sizeof(double));
    return rj::persistent_chain_bytes(K, N) + stage;
}

// Chains resident at once on the GPU with `nw` waves per workgroup: a CU's 160 KB of LDS over the workgroup's block (dynamic
// + 4 928 B of static: math tables, output row, the two parameter blocks, the stage context) / its 16 wave slots at 128 VGPRs.  Checked against
// measurements (scripts/bench_rj_modes.py): Resolve, 30 layers: 6 / 4 / 3 workgroups per CU with 1 / 2 / 3 waves.
static long long persistent_capacity(const gbp_fdem_system* sys, const gbp_rj_options* o, int nw)
{
    const size_t lds = persistent_lds_bytes(sys, o, nw);
    if (lds > 64 * 1024) return 0;
    return (long long)device_cus() * std::min<long long>(160 * 1024 / (long long)(lds + 4928), 4 * GBP_RJ_PERSISTENT_WAVES_PER_EU / nw);
}

// Whether (and how) a block can run in the persistent per-chain kernel: frequency-domain data, one error level of each kind.
// Returns the workgroup's waves, or 0.  The physics stages share a sounding's passes / frequencies among the waves and their
// results do not depend on how many there are (forward_passes, sens_body), so the count is a pure performance choice: the
// most waves (up to 4: the per-chain stages run on one) with which the whole block is still resident at once, else the count
// with the largest capacity.  With `pinned` (an explicit gbp_rj_run_mode(.., 2, ..)) `forward_waves` > 0 fixes it.
static int persistent_waves(const gbp_fdem_system* sys, const gbp_rj_options* o, int B, bool pinned)
{
    if (o->n_rel_groups != 1 || o->n_add_groups != 1) return 0;
    const int top = std::min(4, std::min(sys->t.nF, (sys->t.npts + 63) / 64));
    if (pinned && o->forward_waves > 0) return std::min(o->forward_waves, 4);
    int best = 0;
    long long best_cap = 0;
    for (int nw = top; nw >= 1; --nw) {
        const long long cap = persistent_capacity(sys, o, nw);
        if (cap >= B) return nw;
        if (cap > best_cap) { best_cap = cap; best = nw; }
    }
    return best;
}

static gbp_status rj_run_persistent(const gbp_fdem_system* sys, const gbp_rj_options* o, const gbp_rj_chains* c, int64_t first_iteration,
                                    int n_iterations, int accumulate, bool pinned, void* stream)
{
    const int B = c->B, K = o->max_layers, F = sys->t.nF;
    const int nw = persistent_waves(sys, o, B, pinned);
    if (nw == 0) return fail(GBP_ERR_INVALID_ARG, "the persistent sampler needs frequency-domain data with one error level of each kind%s");
    // LDS per workgroup = the chain's arrays (rj::persistent_chain_bytes) + the largest stage working set.  The Jacobian pass of a
    // model with more than 8 layers needs 1 KB per layer and wave: kept in LDS it would take the budget of one more resident
    // chain per CU away from every chain for a rare case, so it works in a per-chain global block instead (stream-ordered
    // allocation for the duration of the launch).
    const size_t lds = persistent_lds_bytes(sys, o, nw);
    if (lds > 64 * 1024) return fail(GBP_ERR_INVALID_ARG, "max_layers x channels too large for the persistent sampler's LDS block%s");
    const size_t deep_bytes = K > 8 ? ((sens_lds_bytes(nw, K) + 255) & ~(size_t)255) : 0;
    unsigned char* deep = nullptr;
    if (deep_bytes > 0) GBP_HIP(hipMallocAsync((void**)&deep, deep_bytes * (size_t)B, (hipStream_t)stream));
    auto launch = [&](auto kernel) -> gbp_status {
        if (lds > 48 * 1024) GBP_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(B), dim3(64 * nw), lds, (hipStream_t)stream, rj::extend(*o), *c, sys->d_chan, sys->d_pts, sys->t.npts, F,
                           sys->sigma_direct, (uint32_t)first_iteration, n_iterations, accumulate, nw, deep, deep_bytes, sys->d_bins,
                           sys->bin0, sys->n_bins, sys->d_bin_chan, sys->d_bin_pts);
        return GBP_OK;
    };
    gbp_status st = o->exact_jacobian ? launch(rj::k_rj_persistent<true>) : launch(rj::k_rj_persistent<false>);
    const hipError_t le = hipGetLastError();
    if (deep != nullptr) (void)hipFreeAsync(deep, (hipStream_t)stream);
    if (st != GBP_OK) return st;
    if (le != hipSuccess) return fail(GBP_ERR_HIP, "persistent sampler launch: %s", hipGetErrorString(le));
    return GBP_OK;
}

// Waves per workgroup of the stage-0 physics launch (Jacobian pass at the remapped model; results do not depend on it) given the
// sub-block's wave count `nw` of the stage-1 launch.
static int stage0_waves(int nw)
{
#ifdef GBP_RJ_PHYSICS_NW_STAGE0
    (void)nw;
    return GBP_RJ_PHYSICS_NW_STAGE0;                         // (A/B builds under scripts/ab only)
#else
    return nw;
#endif
}

// Concurrent sub-blocks of the fused lock-step driver by block size.  Each sub-block runs on one of the pool's three helper streams
// (aux_streams(): as many as map onto hardware queues of their own beside torch's stream; a fourth shares a queue and serialises --
// the first measurement of four, with the drivers' separate stream sets still in place, lost 30 %).  Measured (reference Jacobian,
// scripts/bench_rj_parts.py, M chain-iterations/s for 2 / 3 sub-blocks; Resolve | 10-frequency system): 2 048 chains 19.6 / 19.9 |
// 17.4 / 17.5, 3 072: 26.9 / 27.8 | 23.4 / 24.1, 4 096: 32.1 / 33.9 | 27.8 / 29.5, 8 192: 47.6 / 50.0 | 41.0 / 42.9, 16 384:
// 62.0 / 63.4 | 52.1 / 53.1, 32 768: 62.8 / 62.8 | 50.5 / 50.6 (one sub-block, Resolve: 18.1 / 28.1 / 38.0 / 50.7 at 2 048 ... 16 384).
// Three never lose: the host issues 7 launches per iteration and sub-block at ~8 us each, which the GPU side still hides at three.
// Workgroups per Jacobian evaluation of a chain in the lock-step physics launches (k_rj_physics: the frequencies are shared out).  A small
// sub-block leaves most of the GPU idle and its launch lasts as long as its slowest workgroup: two workgroups of five frequencies finish
// sooner than one of ten (2 048 chains = three sub-blocks of 683: 21.1 -> 21.5 M chain-it/s); from about 900 chains per launch the second
// workgroup's prologue is capacity lost (sub-blocks of 900 / 910 / 1 100 chains: - 2 / - 4 / - 6 %; with the split at every size 4 096: 32.8 -> 31.7,
// 8 192: 46.0 -> 42.4, 16 384: 51.6 -> 48.9).  Same bits either way (scripts/ab_bits.py).
static int jacobian_shares(int chains_in_launch)
{
#ifdef GBP_RJ_JACOBIAN_SHARES
    return GBP_RJ_JACOBIAN_SHARES;                           // (A/B builds under scripts/ab only)
#endif
    return chains_in_launch <= GBP_RJ_SHARES_UP_TO ? 2 : 1;
}

static int lockstep_parts(int B)
{
#ifdef GBP_RJ_LOCKSTEP_PARTS
    return B >= 2048 ? GBP_RJ_LOCKSTEP_PARTS : 1;              // (A/B builds under scripts/ab only)
#endif
    return B >= 2048 ? 3 : (B >= 1200 ? 2 : 1);      // (1 600 ... 2 047 chains, two against one: 17.5 vs 15.4, 19.2 vs 15.8, 20.8 vs 17.1 M Resolve; 15.5 vs 13.6 ... 18.3 vs 15.1 M ten frequencies)
}

gbp_status gbp_rj_run_mode(const gbp_fdem_system* sys, const gbp_rj_options* o, const gbp_rj_chains* c, int64_t first_iteration,
                           int n_iterations, int accumulate, int mode, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK) return st;
    if (!sys) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (mode < 0 || mode > 4)
        return fail(GBP_ERR_INVALID_ARG, "mode must be 0 (auto), 1 (lock-step), 2 (persistent), 3 (lock-step, one launch per evaluation kind) or 4 (lock-step, concurrent sub-blocks)%s");
    if (o->n_channels != 2 * sys->t.nF) return fail(GBP_ERR_INVALID_ARG, "n_channels must be 2 * nF of the system%s");
    if (c->B == 0 || n_iterations <= 0) return GBP_OK;
    if (mode == 0) {
        // All drivers walk the same chains; which is faster depends on how many chains share the GPU.  A persistent workgroup
        // needs ~57 us per iteration of its chain whatever the block; the lock-step driver ~95 us of dependent launches +
        // ~0.02 us per chain.  Measured (Resolve and the 10-frequency system, scripts/bench_rj_modes.py): the persistent kernel
        // wins while the whole block is resident at once (1 536 Resolve chains with one wave each: 20.2 vs 14.0 M
        // chain-iterations/s), the lock-step driver as soon as it would take a second round (2 048: 17.5 vs 16.6 M).
        // (Rounds 4 - 5 ran blocks up to a fifth beyond the resident capacity persistently; with four launches per lock-step iteration
        //  two sub-blocks are ahead from the first chain beyond it -- the measurements are beside `small` below.)  This is the size a
        //  survey block shrinks to when the chains that failed to burn in have stopped (survey.infer re-packs the running ones).
        const int nw = persistent_waves(sys, o, c->B, false);
        bool small = false;
        if (nw > 0 && n_iterations >= 4) {
            const long long cap = persistent_capacity(sys, o, nw);
            // (round 6, with four launches per lock-step iteration: a block beyond what is resident at once is faster as two sub-blocks --
            //  ten frequencies, capacity 1 280: 1 408 / 1 536 chains 13.6 / 15.0 M persistently against 16.5 / 17.9 M; Resolve, capacity
            //  1 536: 1 700 chains 19.5 against 21.8 M -- the fifth of slack of rounds 4 - 5 is gone)
            small = (long long)c->B <= cap;
        }
        if (small) return rj_run_persistent(sys, o, c, first_iteration, n_iterations, accumulate, false, stream);
        // (concurrent sub-blocks cost a host thread each and a fork / join of streams per call: they pay from about four iterations per
        //  call -- 8 192 chains, M chain-iterations/s at 1 / 4 / 16 / 200 iterations per call: 28.5 37.2 40.9 43.1 against 33.3 - 33.7 for
        //  the one-block driver; 2 048 chains: 11.2 16.0 17.9 19.0 against 15.4 - 15.8)
        mode = (lockstep_parts(c->B) > 1 && n_iterations >= 4) ? 4 : 1;
    }
    if (mode == 2) return rj_run_persistent(sys, o, c, first_iteration, n_iterations, accumulate, true, stream);
    return rj_run_lockstep(sys, nullptr, o, c, first_iteration, n_iterations, accumulate, mode != 3, mode == 4 ? std::max(2, lockstep_parts(c->B)) : 1, stream);
}

gbp_status gbp_rj_debug_stage_ticks(int64_t* out, int reset)
{
    if (!out) return fail(GBP_ERR_INVALID_ARG, "out is NULL%s");
    long long h[8];
    unsigned long long life[3];
    GBP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(rj::GBP_RJ_TICKS), sizeof(h)));
    GBP_HIP(hipMemcpyFromSymbol(life, HIP_SYMBOL(rj::GBP_RJ_LIFE), sizeof(life)));
    for (int i = 0; i < 6; ++i) out[i] = (int64_t)h[i];
    out[6] = (int64_t)life[2];                                       // the longest workgroup life of the launches since the reset
    out[7] = life[1] ? (int64_t)(life[0] / life[1]) : 0;             // ... and the mean
    if (reset) {                                 // 1: zero the counters and arm the clock; 2: zero and disarm
        std::memset(h, 0, sizeof(h));
        std::memset(life, 0, sizeof(life));
        h[7] = reset == 1 ? 1 : 0;
        GBP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rj::GBP_RJ_TICKS), h, sizeof(h)));
        GBP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rj::GBP_RJ_LIFE), life, sizeof(life)));
    }
    return GBP_OK;
}

#ifdef GBP_RJ_PHYS_CLOCK
gbp_status gbp_debug_phys_ticks(int64_t* out, int reset)   // (measurement builds only: not declared in the header)
{
    long long h[64];
    GBP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(GBP_PHYS_TICKS), sizeof(h)));
    for (int i = 0; i < 64; ++i) out[i] = (int64_t)h[i];
    if (reset) { std::memset(h, 0, sizeof(h)); GBP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(GBP_PHYS_TICKS), h, sizeof(h))); }
    return GBP_OK;
}
gbp_status gbp_debug_phys_life(int64_t* out, int reset)    // [3][16][3], see GBP_PHYS_LIFE
{
    long long h[144];
    GBP_HIP(hipMemcpyFromSymbol(h, HIP_SYMBOL(GBP_PHYS_LIFE), sizeof(h)));
    for (int i = 0; i < 144; ++i) out[i] = (int64_t)h[i];
    if (reset) { std::memset(h, 0, sizeof(h)); GBP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(GBP_PHYS_LIFE), h, sizeof(h))); }
    return GBP_OK;
}
#endif

gbp_status gbp_rj_run(const gbp_fdem_system* sys, const gbp_rj_options* o, const gbp_rj_chains* c, int64_t first_iteration,
                      int n_iterations, int accumulate, void* stream)
{
    return gbp_rj_run_mode(sys, o, c, first_iteration, n_iterations, accumulate, 0, stream);
}

gbp_status gbp_rj_run_td(const gbp_fdem_system* sys, const gbp_td_operator* td, const gbp_rj_options* o, const gbp_rj_chains* c,
                         int64_t first_iteration, int n_iterations, int accumulate, void* stream)
{
    return rj_run_lockstep(sys, td, o, c, first_iteration, n_iterations, accumulate, td == nullptr, 1, stream);
}

}  // extern "C"

static gbp_status rj_run_lockstep(const gbp_fdem_system* sys, const gbp_td_operator* td, const gbp_rj_options* o, const gbp_rj_chains* c,
                                  int64_t first_iteration, int n_iterations, int accumulate, bool fused, int parts, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK) return st;
    if (!sys) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (td == nullptr) {
        if (o->n_channels != 2 * sys->t.nF) return fail(GBP_ERR_INVALID_ARG, "n_channels must be 2 * nF of the system%s");
        if (o->n_rel_groups != 1 || o->n_add_groups != 1)
            return fail(GBP_ERR_INVALID_ARG, "frequency-domain data: one relative and one additive error level%s");
    } else {
        if ((td->mix.n_in > 0 ? td->mix.n_in : td->n_nodal) != 2 * sys->t.nF)
            return fail(GBP_ERR_INVALID_ARG, "the nodal vector the kernels write (n_nodal, or mix.n_in with geometry mixing) must be 2 * nF of the system%s");
        if (!td->W || !td->nodal || !td->J_nodal) return fail(GBP_ERR_INVALID_ARG, "NULL pointer in gbp_td_operator%s");
        if (td->mix.n_in > 0 && (!td->mix.src || !td->mix.col || !td->mix.weights || td->mix.terms < 1 || td->mix.n_weights < 1))
            return fail(GBP_ERR_INVALID_ARG, "incomplete gbp_td_mix%s");
    }
    const int B = c->B, K = o->max_layers;
    if (B == 0) return GBP_OK;
    const int caps[2] = {8 < K ? 8 : K, K};      // Jacobian launches by layer count: <= 8 (the common case, small LDS
    const int nb = K <= 8 ? 1 : 2;               //   footprint, high occupancy) and the rest
    // Jacobian launches: ~40 % of the chains need one (structure changed / dimension changed), the rest exit at once;
    // size the workgroups for the chains that work, with a wave count that divides nF (one frequency per wave at a time).
    // Measured optimum (6 frequencies): 6 waves up to 4 k chains, 3 at 8 k, 2 at 16 k, 1 from 32 k (36 frequencies: 18 at 1 k
    // chains): ~7 500 working waves for small blocks, growing to ~15 000.
    int sw = 1;
    {
        const int F = sys->t.nF;
        const double want = 7500.0 / (0.4 * B) * (1.0 + std::min(B, 16384) / 16384.0);
        for (int d = 1; d <= F; ++d)                    // the largest divisor of nF not (much) above the target
            if (F % d == 0 && d <= 16 && (double)d <= 1.15 * want) sw = d;
        // time-domain handles carry 22 ... 75 "frequencies" (spline nodes x basis integrals): at least four waves per chain, whether
        // or not four divides their number -- measured (scripts/bench_tdem_sampler.py through -DGBP_RJ_SENS_WAVES builds; 22 nodes,
        // M chain-iterations/s with 2 / 3 / 4 / 6 waves): 4 096 chains 7.0 7.4 7.7 7.4; 8 192: 10.4 10.9 11.1 10.9; 16 384: 13.6 13.9
        // 14.1 12.9; 44 nodes, 8 192 chains: 5.4 - 5.8 5.5
        if (td != nullptr) sw = std::max(sw, std::min(4, F));
#ifdef GBP_RJ_SENS_WAVES
        sw = GBP_RJ_SENS_WAVES;                              // (A/B builds under scripts/ab only)
#endif
    }
    const int fw = o->forward_waves;   // 0: the forward kernels choose from the batch size
    const int N = o->n_channels;
    const bool moving = td != nullptr && td->moves.n_moves > 0;
    if (moving) {
        const gbp_td_moves& mv = td->moves;
        if (td->mix.n_in <= 0 || !c->step_flags || !mv.geom || !mv.geom_p || !mv.geom0 || !mv.weights || !mv.weights_p || mv.weights != td->mix.weights ||
            (mv.offset != nullptr) != (mv.offset_p != nullptr) || mv.offset != td->mix.offset || mv.n_moves > 6 || mv.n_blocks < 1 ||
            mv.n_blocks * mv.n_basis != td->mix.n_weights || !mv.block_comp || !mv.block_scale || (mv.offset && (!mv.block_primary || !mv.block_windows)))
            return fail(GBP_ERR_INVALID_ARG, "incomplete gbp_td_moves (needs geometry mixing, chains->step_flags, weights = mix.weights, offset = mix.offset)%s");
        for (int q = 0; q < mv.n_moves; ++q)
            if (!((mv.entry[q] >= 1 && mv.entry[q] <= 3) || (mv.entry[q] >= 7 && mv.entry[q] <= 9) ||
                  (mv.rho_scale != nullptr && (mv.entry[q] == 0 || (mv.entry[q] >= 4 && mv.entry[q] <= 6)))) || !(mv.half_width[q] > 0.0) ||
                !(mv.scale[q] >= 0.0) || mv.n_bins[q] < 1 || mv.n_bins[q] > 199)
                return fail(GBP_ERR_INVALID_ARG, "gbp_td_moves: entries 1..3 / 7..9 (angles) or, with the position arrays, 0 / 4..6; half_width > 0, scale >= 0, 1 <= n_bins <= 199%s");
        if ((mv.rho_scale != nullptr) != (mv.rho_scale_p != nullptr) || (mv.rho_scale != nullptr && (!mv.rho_set || !mv.dz_set || !c->height_p)))
            return fail(GBP_ERR_INVALID_ARG, "gbp_td_moves: position moves need rho_scale, rho_scale_p, rho_set, dz_set and chains->height_p%s");
    }
    auto td_apply = [&](const int32_t* nl, bool with_j, double* pred, double* J, hipStream_t q, bool proposed) -> gbp_status {   // nodal -> windows
        const size_t lds = ((size_t)td->n_nodal * (with_j ? K + 1 : 1)) * sizeof(double);
        if (lds > 60000) return fail(GBP_ERR_INVALID_ARG, "n_nodal * max_layers too large for the time-domain stage%s");
        gbp_td_mix mix = td->mix;
        if (proposed && moving) {                                 // a sampled geometry: proposals are evaluated with theirs
            mix.weights = td->moves.weights_p;
            mix.offset = td->moves.offset_p;
        }
        if (with_j)
            hipLaunchKernelGGL(rj::k_td_apply<true>, dim3(B), dim3(64), lds, q, B, K, td->n_nodal, N, nl, td->W,
                               td->nodal, td->J_nodal, pred, J, mix);
        else if (((size_t)td->n_nodal * N + (size_t)4 * td->n_nodal) * sizeof(double) <= 40 * 1024 && B >= 4 * GBP_TD_PLAIN_ROWS)
            hipLaunchKernelGGL(rj::k_td_apply_plain, dim3((B + GBP_TD_PLAIN_ROWS - 1) / GBP_TD_PLAIN_ROWS), dim3(256),
                               ((size_t)td->n_nodal * N + (size_t)4 * td->n_nodal) * sizeof(double), q, B, td->n_nodal, N, nl, td->W, td->nodal, pred, mix);
        else
            hipLaunchKernelGGL(rj::k_td_apply<false>, dim3(B), dim3(64), lds, q, B, K, td->n_nodal, N, nl, td->W,
                               td->nodal, td->J_nodal, pred, J, mix);
        GBP_HIP(hipGetLastError());
        return GBP_OK;
    };
    // prediction + Jacobian of the chains selected by nl (row 0: all of them, rows 1.. by layer bucket)
    // (a sampled height -- or, time-domain data, a sampled position of the loop pair: proposals are evaluated at theirs, and with the
    //  proposal's distance scale of the chain's table set)
    const bool moving_pos = moving && td->moves.rho_scale != nullptr;
    const double* height_prop = (o->solve_height || moving_pos) ? c->height_p : c->height;
    const double* scale_cur = moving_pos ? td->moves.rho_scale : nullptr;
    const double* scale_prop = moving_pos ? td->moves.rho_scale_p : nullptr;
    // The launch of the models of more than 8 layers holds few chains, but with the working set of a K-layer model in LDS (32 KB per
    // wave at K = 30) each runs on ONE wave: all frequencies x up to 30 layers in sequence -- ~100 us for a 22-node time-domain system,
    // 21 % of an iteration at 8 192 chains when it follows the launch of the shallow models.  The two launches touch disjoint chains:
    // the deep one goes to a stream of its own (`slot`: one per evaluation that can be in flight) and overlaps the shallow one.
    // (Its working set in a global block per chain, to give it more waves, was tried and is slower: 7.3 vs 8.6 M chain-iterations/s.)
    DeepStreams* ds = (nb > 1 && td != nullptr) ? deep_streams() : nullptr;
    auto fm_dlogc = [&](const int32_t* nl, const double* sigma, const double* height, double* pred, double* J, hipStream_t q, int slot) -> gbp_status {
        const bool proposed = slot == 1;
        for (int i = nb - 1; i >= 0; --i) {
            // (compact rows: the consumers read columns < layer count only, so the columns beyond it rounded up to 8 are not touched)
            const bool aside = i == 1 && ds != nullptr;
            if (aside) {
                GBP_HIP(hipEventRecord(ds->fork[slot], q));
                GBP_HIP(hipStreamWaitEvent(ds->q[slot], ds->fork[slot], 0));
            }
            gbp_status s2 = fm_dlogc_launch(sys, B, K, nl + (size_t)(1 + i) * B, sigma, c->thk_r, height, td ? td->nodal : pred,
                                            td ? td->J_nodal : J, caps[i], o->exact_jacobian, sw, 1, td ? td->table_set : nullptr,
                                            aside ? ds->q[slot] : q, proposed ? scale_prop : scale_cur);
            if (s2 != GBP_OK) return s2;
            if (aside) GBP_HIP(hipEventRecord(ds->join[slot], ds->q[slot]));
        }
        if (nb > 1 && ds != nullptr) GBP_HIP(hipStreamWaitEvent(q, ds->join[slot], 0));
        return td ? td_apply(nl, true, pred, J, q, proposed) : GBP_OK;
    };
    // The two evaluations at the proposals work on disjoint chains (those that keep their dimension / those that change it)
    // and write disjoint rows: the second runs on a side stream of this device, forked after the proposal kernel and joined
    // before the accept kernel, so that the tail of one launch overlaps the other.  Measured: +3 % at 65 536 frequency-domain
    // chains and for time-domain blocks of any size (more kernels per branch), but the two cross-stream waits cost more than
    // the overlap gains below ~32 k chains x 6 frequencies (-3 % at 8 192, -9 % at 1 024): only large launches fork.
    const hipStream_t main_q = (hipStream_t)stream;
    // Fused physics launches pay below the size from which the two evaluations at the proposals overlap on two streams anyway
    // (scripts/bench_rj_modes.py, mode 1 vs 3: +17 % at 1 024 chains, +4 % at 8 192, -8 % at 65 536, where the forward chains
    // would run at the Jacobian pass's occupancy).
    if (fused && td == nullptr && (long long)B * sys->t.nF < 196608) {
        // One physics launch per stage (k_rj_physics): 5 - 7 launches per iteration and sub-block (one_stage_launch).  `parts` > 1: the block is cut into
        // that many contiguous sub-blocks which advance CONCURRENTLY, each on a stream of its own, launches issued round-robin --
        // the latency-bound per-chain stages of one sub-block (propose / Newton / accept: 23 + 28 + 38 us at 8 192 chains, one
        // wave per SIMD) overlap the physics of another, and a physics launch of a quarter of the chains has a shorter tail.
        // Chains are keyed by first_chain + row (or chain_id), every array is sliced by rows: the chains are bit-identical to the
        // one-block run (tests/test_rjmcmc_gpu.py).  Measured: lockstep_parts above.
        const int P = std::max(1, std::min(parts, (int)BlockStreams::MAX));
        BlockStreams* bs = P > 1 ? block_streams() : nullptr;
        if (P > 1 && bs == nullptr) return fail(GBP_ERR_HIP, "sub-block streams: %s", hipGetErrorString(hipGetLastError()));
        struct Part { gbp_rj_options o; gbp_rj_chains c; hipStream_t q; unsigned char* deep; int nw; size_t lds; int32_t* flags; int32_t* order; };
        std::vector<Part> part(P);
        const size_t deep_per_chain = K > 8 ? 1 : 0;
        for (int p = 0; p < P; ++p) {
            const int b0 = (int)((long long)B * p / P), n = (int)((long long)B * (p + 1) / P) - b0;
            Part& t = part[p];
            t.o = *o;
            t.c = P > 1 ? slice_chains(*o, *c, b0, n) : *c;
            if (P > 1 && c->chain_id == nullptr) t.o.first_chain = o->first_chain + (uint64_t)b0;
            t.q = P > 1 ? bs->q[p] : main_q;
            t.deep = nullptr;
            t.flags = nullptr;
            t.order = nullptr;
            // Waves per workgroup of the physics launches (results do not depend on it).  Measured per sub-block size n
            // (scripts/bench_rj_parts.py through -DGBP_RJ_PHYSICS_NW builds, M chain-iterations/s with 1 / 2 / 3 / 4 waves; ten
            // frequencies | Resolve): n = 1 024: 11.9 15.3 16.5 17.1 | 14.6 17.5 18.5 19.6;  2 048: 20.4 25.9 27.0 27.5 | 25.7 30.1 31.0
            // 31.7;  4 096: 31.9 40.6 40.4 38.5 | 41.0 47.0 45.8 45.2;  8 192: 48.7 51.5 48.1 44.9 | 61.2 61.3 56.3 53.1 -- a
            // chain's frequencies spread over four waves while the launch would not fill the SIMDs otherwise (<= 8 192 waves), two
            // beyond.  (More than four -- a 320- or 384-thread launch bound -- is 30 % slower at every size.)  With three sub-blocks
            // in flight (2 / 3 / 4 waves, block = 3 n): n = 1 707: - / 35.0 / 34.7 | - / 39.0 / 39.4;  2 048: 35.9 38.0 37.2 | 43.0 44.6
            // 43.6;  2 731: 43.6 44.1 41.3 | 51.3 49.7 48.1;  3 413: 47.8 44.6 43.7 | 56.4 51.8 50.0.
            // Round 5, re-measured with four launches per iteration (scripts/ab_rj.py, one box, 2 / 3 / 4 waves, ten frequencies, block = 3 n):
            // n = 910: 23.4 25.4 25.6;  1 365: 31.3 32.2 30.6;  1 820: 37.3 36.2 33.8;  2 731: 44.1 40.5 38.0 -- with a shorter chain per
            // sub-block the physics launches of the three overlap more, and the switch points move down.
            t.nw = std::min(n <= GBP_RJ_SHARES_UP_TO ? 3 : (n <= 1100 ? 4 : (n <= 1600 ? 3 : 2)), GBP_RJ_PHYSICS_MAX_WAVES);   // (two workgroups per Jacobian in the smallest launches: three waves each, 2 048 chains 22.6 -> 23.3 M)
#ifdef GBP_RJ_PHYSICS_NW
            t.nw = GBP_RJ_PHYSICS_NW;                          // (A/B builds under scripts/ab only)
#endif
            t.lds = (std::max(dyn_lds_bytes(t.nw, K, (sys->t.npts + 63) / 64), sens_lds_bytes(t.nw, K < GBP_RJ_PHYSICS_LDS_LAYERS ? K : GBP_RJ_PHYSICS_LDS_LAYERS)) + 15) & ~(size_t)15;     // + the output row, physics()
        }
        (void)deep_per_chain;
        if (P > 1) {
            GBP_HIP(hipEventRecord(bs->start, main_q));
            for (int p = 0; p < P; ++p) GBP_HIP(hipStreamWaitEvent(part[p].q, bs->start, 0));
        }
        // (the accept stage of an iteration and the proposal of the next share a launch -- k_rj_step8 -- from the second iteration of a call:
        //  two rows of ownership flags per sub-block, written by the proposals, read by the accept stages)
#ifdef GBP_RJ_NO_FUSED_STEP
        const bool fused_step = false;                               // (A/B builds only)
#else
        const bool fused_step = n_iterations > 1;
#endif
        bool alloc_failed = false;
        for (int p = 0; p < P && !alloc_failed; ++p) {
            const size_t deep_bytes = K > GBP_RJ_PHYSICS_LDS_LAYERS ? ((sens_lds_bytes(std::max(part[p].nw, stage0_waves(part[p].nw)), K) + 255) & ~(size_t)255) : 0;
            if (deep_bytes > 0 && hipMallocAsync((void**)&part[p].deep, deep_bytes * (size_t)part[p].c.B, part[p].q) != hipSuccess) {
                part[p].deep = nullptr;
                alloc_failed = true;
            }
            if (!alloc_failed && fused_step && part[p].c.B > 0 &&
                hipMallocAsync((void**)&part[p].flags, sizeof(int32_t) * 2 * (size_t)part[p].c.B, part[p].q) != hipSuccess) {
                part[p].flags = nullptr;
                alloc_failed = true;
            }
            if (!alloc_failed && GBP_RJ_ORDERED_PHYSICS && n_iterations >= 4 && part[p].c.B > 64 &&
                hipMallocAsync((void**)&part[p].order, sizeof(int32_t) * (size_t)part[p].c.B, part[p].q) != hipSuccess) {
                part[p].order = nullptr;
                alloc_failed = true;
            }
        }
        if (alloc_failed) {          // give back what was allocated and join the sub-block streams the caller's stream was forked into
            const hipError_t e = hipGetLastError();
            for (int p = 0; p < P; ++p) {
                if (part[p].deep != nullptr) (void)hipFreeAsync(part[p].deep, part[p].q);
                if (part[p].flags != nullptr) (void)hipFreeAsync(part[p].flags, part[p].q);
                if (part[p].order != nullptr) (void)hipFreeAsync(part[p].order, part[p].q);
                if (P > 1) { (void)hipEventRecord(bs->done[p], part[p].q); (void)hipStreamWaitEvent(main_q, bs->done[p], 0); }
            }
            return fail(GBP_ERR_HIP, "sampler sub-blocks: working set of the deep models: %s", hipGetErrorString(e));
        }
        auto physics = [&](const Part& t, int stage) {
            // (stage 0 -- the Jacobian pass at the remapped model: half of the workgroups leave at once -- may take a wave count of its own;
            //  the deep models' global working set is sized for the larger of the two)
            const int nw_s = stage == 0 ? stage0_waves(t.nw) : t.nw;
            const size_t lds_s = stage == 0 ? ((std::max(dyn_lds_bytes(nw_s, K, (sys->t.npts + 63) / 64), sens_lds_bytes(nw_s, K < GBP_RJ_PHYSICS_LDS_LAYERS ? K : GBP_RJ_PHYSICS_LDS_LAYERS)) + 15) & ~(size_t)15) : t.lds;
            const size_t deep_bytes = K > GBP_RJ_PHYSICS_LDS_LAYERS ? ((sens_lds_bytes(std::max(t.nw, stage0_waves(t.nw)), K) + 255) & ~(size_t)255) : 0;
            const rj::RjOpt ox = rj::extend(t.o);
            const size_t out_bytes = (size_t)o->n_channels * sizeof(double);          // the output row behind the stages' block (k_rj_physics)
            const int out_offset = (int)lds_s;
            // (the deepest 1 / GBP_RJ_SPLIT_DEEP of an ordered launch's chains get a second workgroup where the launch does not split every chain's)
            const int shares = jacobian_shares(t.c.B);
            const int n_split = (GBP_RJ_SPLIT_DEEP > 0 && shares == 1 && t.order != nullptr) ? t.c.B / GBP_RJ_SPLIT_DEEP : 0;
            const int grid = t.c.B * shares + n_split;
            if (o->exact_jacobian)
                hipLaunchKernelGGL(rj::k_rj_physics<true>, dim3(grid), dim3(64 * nw_s), lds_s + out_bytes, t.q, ox, t.c, sys->d_chan, sys->d_pts, sys->t.npts,
                                   sys->t.nF, sys->sigma_direct, stage, t.deep, deep_bytes, sys->d_bins, sys->bin0, sys->n_bins, sys->d_bin_chan,
                                   sys->d_bin_pts, out_offset, t.order, n_split);
            else
                hipLaunchKernelGGL(rj::k_rj_physics<false>, dim3(grid), dim3(64 * nw_s), lds_s + out_bytes, t.q, ox, t.c, sys->d_chan, sys->d_pts, sys->t.npts,
                                   sys->t.nF, sys->sigma_direct, stage, t.deep, deep_bytes, sys->d_bins, sys->bin0, sys->n_bins, sys->d_bin_chan,
                                   sys->d_bin_pts, out_offset, t.order, n_split);
        };
        // One host thread per sub-block issues that sub-block's launches (7 per iteration at ~9 us each: one thread issuing for
        // four sub-blocks would be slower than the GPU -- measured 31 vs 37.6 M chain-iterations/s at 8 192 chains; with a thread
        // each the sub-blocks really overlap).  The caller's thread takes sub-block 0; the others live for this call.
        int dev = 0;
        GBP_HIP(hipGetDevice(&dev));
        std::vector<gbp_status> pst(P, GBP_OK);
        std::vector<std::string> perr(P);
        // the launches of ONE iteration of sub-block p
        auto issue = [&](int p, int it) -> gbp_status {
            const Part& t = part[p];
            const bool step = fused_step && t.flags != nullptr && t.c.B > 0;
            const int nB = t.c.B;
            const int64_t iter = first_iteration + it;
            gbp_status s3;
            if (it == 0 || !step) {
                if ((s3 = gbp_rj_propose(&t.o, &t.c, iter, t.q)) != GBP_OK) return s3;
                if (step) hipLaunchKernelGGL(rj::k_rj_propose_flags, dim3((nB + 255) / 256), dim3(256), 0, t.q, t.c, t.flags);
            }
            if (t.order != nullptr && it % GBP_RJ_REORDER_EVERY == 0)     // the launches' chain order: deepest model first (one small workgroup)
                hipLaunchKernelGGL(rj::k_rj_order_by_layers, dim3(1), dim3(1024), 0, t.q, t.c, t.order);
            physics(t, 0);                                    // fm_dlogc at the remapped models whose structure changed (Model.py:383-384)
            if ((s3 = gbp_rj_newton(&t.o, &t.c, iter, t.q)) != GBP_OK) return s3;
            physics(t, 1);                                    // Inference1D.py:572-597 / Model.py:612: every proposal's evaluation
            if (step && it + 1 < n_iterations) {              // accept + the next iteration's proposal in one launch (k_rj_step8)
                const int n_packed = (nB + 7) / 8, n_deep = K > 8 ? (nB + 63) / 64 : 0;
                const size_t lds = std::max((size_t)8 * o->n_channels * sizeof(double), n_deep ? rj::Lds::bytes(K, o->n_channels) : (size_t)0);
                const int32_t* cur = t.flags + (size_t)(it & 1) * nB;
                int32_t* nxt = t.flags + (size_t)((it + 1) & 1) * nB;
                if (packed_trips(nB))
                    hipLaunchKernelGGL(rj::k_rj_step8<true>, dim3(n_packed + n_deep), dim3(64), lds, t.q, rj::extend(t.o), t.c, (uint32_t)iter, accumulate,
                                       n_packed, cur, nxt);
                else
                    hipLaunchKernelGGL(rj::k_rj_step8<false>, dim3(n_packed + n_deep), dim3(64), lds, t.q, rj::extend(t.o), t.c, (uint32_t)iter, accumulate,
                                       n_packed, cur, nxt);
                return GBP_OK;
            }
            return gbp_rj_accept(&t.o, &t.c, iter, accumulate, t.q);
        };
        auto run_part = [&](int p) {
            if (p > 0 && hipSetDevice(dev) != hipSuccess) { pst[p] = GBP_ERR_HIP; perr[p] = "hipSetDevice failed in a sub-block thread"; return; }
            gbp_status s2 = GBP_OK;
            for (int it = 0; it < n_iterations && s2 == GBP_OK; ++it) s2 = issue(p, it);
            if (s2 == GBP_OK && hipGetLastError() != hipSuccess) s2 = GBP_ERR_HIP;
            pst[p] = s2;
            if (s2 != GBP_OK) perr[p] = gbp_last_error();      // (the message is per host thread: hand it to the caller's)
        };
#ifdef GBP_RJ_SINGLE_ISSUER
        {   // (A/B builds only: the caller's thread issues for every sub-block in turn)
            gbp_status s2 = GBP_OK;
            for (int it = 0; it < n_iterations && s2 == GBP_OK; ++it)
                for (int p = 0; p < P && s2 == GBP_OK; ++p) s2 = issue(p, it);
            if (s2 == GBP_OK && hipGetLastError() != hipSuccess) s2 = GBP_ERR_HIP;
            pst[0] = s2;
            if (s2 != GBP_OK) perr[0] = gbp_last_error();
        }
#else
        {
            // (std::thread's constructor throws std::system_error when the process is out of threads: nothing may leave an
            //  extern "C" entry -- the sub-blocks that got no thread run on the caller's, after its own)
            std::vector<std::thread> workers;
            std::vector<int> inline_parts;
            for (int p = 1; p < P; ++p) {
                try { workers.emplace_back(run_part, p); }
                catch (const std::exception&) { inline_parts.push_back(p); }
            }
            run_part(0);
            for (int p : inline_parts) run_part(p);
            for (auto& w : workers) w.join();
        }
#endif
        for (int p = 0; p < P; ++p)
            if (pst[p] != GBP_OK && st == GBP_OK) st = fail(pst[p], "sampler sub-block: %s", perr[p].c_str());
        const hipError_t le = hipGetLastError();
        for (int p = 0; p < P; ++p) {
            if (part[p].deep != nullptr) (void)hipFreeAsync(part[p].deep, part[p].q);
            if (part[p].flags != nullptr) (void)hipFreeAsync(part[p].flags, part[p].q);
            if (part[p].order != nullptr) (void)hipFreeAsync(part[p].order, part[p].q);
        }
        if (P > 1)
            for (int p = 0; p < P; ++p) {
                (void)hipEventRecord(bs->done[p], part[p].q);
                (void)hipStreamWaitEvent(main_q, bs->done[p], 0);
            }
        if (st != GBP_OK) return st;
        if (le != hipSuccess) return fail(GBP_ERR_HIP, "sampler launch: %s", hipGetErrorString(le));
        return GBP_OK;
    }
    const bool fork = td != nullptr || (long long)B * sys->t.nF >= 196608;
    SideStream* ss = fork ? side_stream() : nullptr;
    if (fork && ss == nullptr) return fail(GBP_ERR_HIP, "side stream: %s", hipGetErrorString(hipGetLastError()));
    const hipStream_t jump_q = fork ? ss->q : main_q;
    for (int it = 0; it < n_iterations; ++it) {
        const int64_t iter = first_iteration + it;
        if ((st = gbp_rj_propose(o, c, iter, stream)) != GBP_OK) return st;
        if (moving) {                                             // Loop_pair.perturb: the angles of the proposal, its weights / primary field
            hipLaunchKernelGGL(rj::k_td_moves_propose, dim3((B + 63) / 64), dim3(64), 0, main_q, rj::extend(*o), *c, td->moves, td->mix.n_weights, N,
                               (uint32_t)iter);
            GBP_HIP(hipGetLastError());
        }
        // fm_dlogc at the remapped models whose structure changed (Model.py:383-384): prediction and Jacobian in one pass
        if ((st = fm_dlogc(c->nl_a, c->sigma_r, c->height, c->pred_r, c->J_r, main_q, 0)) != GBP_OK) return st;
        if ((st = rj_newton_launch(o, c, iter, one_stage_launch(B, td != nullptr), stream)) != GBP_OK) return st;
        if (fork) {
            GBP_HIP(hipEventRecord(ss->fork, main_q));
            GBP_HIP(hipStreamWaitEvent(ss->q, ss->fork, 0));
        }
        // forward + chi^2 + logL of every proposal (Inference1D.py:572-597)
        //   ... of the proposals that keep their dimension
        if (td == nullptr) {
            if ((st = gbp_fdem_forward_loglike_ex(sys, B, K, c->nl_b, c->sigma_p, c->thk_r, height_prop, c->data, c->rel_p, c->add_p,
                                                  c->pred_p, c->misfit_p, c->like_p, fw, stream)) != GBP_OK) return st;
        } else {
            if ((st = gbp_fdem_forward_rows_scaled(sys, B, K, c->nl_b, c->sigma_p, c->thk_r, height_prop, td->nodal, td->table_set, scale_prop, fw, stream)) != GBP_OK) return st;
            if ((st = td_apply(c->nl_b, false, c->pred_p, nullptr, main_q, true)) != GBP_OK) return st;
            hipLaunchKernelGGL(rj::k_td_loglike, dim3(B), dim3(64), 0, (hipStream_t)stream, rj::extend(*o), *c, c->nl_b, c->pred_p, c->rel_p, c->add_p,
                               c->misfit_p, c->like_p);
            GBP_HIP(hipGetLastError());
        }
        //   ... and prediction + Jacobian (Model.py:612) of those that change it; their chi^2 / logL are formed in accept
        if ((st = fm_dlogc(c->nl_c, c->sigma_p, height_prop, c->pred_p, c->J_p, jump_q, 1)) != GBP_OK) return st;
        if (fork) {
            GBP_HIP(hipEventRecord(ss->join, ss->q));
            GBP_HIP(hipStreamWaitEvent(main_q, ss->join, 0));
        }
        if ((st = rj_accept_launch(o, c, iter, accumulate, one_stage_launch(B, td != nullptr), stream)) != GBP_OK) return st;
        if (moving) {
            hipLaunchKernelGGL(rj::k_td_moves_accept, dim3((B + 63) / 64), dim3(64), 0, main_q, rj::extend(*o), *c, td->moves, td->mix.n_weights, N);
            GBP_HIP(hipGetLastError());
        }
    }
    return GBP_OK;
}

extern "C" {

gbp_status gbp_td_apply(int B, int K, int n_nodal, int N, const int32_t* nlayers, const double* W, const double* nodal,
                        const double* J_nodal, double* pred, double* J, void* stream)
{
    return gbp_td_apply_mix(B, K, n_nodal, N, nlayers, W, nodal, J_nodal, pred, J, nullptr, stream);
}

gbp_status gbp_td_apply_mix(int B, int K, int n_nodal, int N, const int32_t* nlayers, const double* W, const double* nodal,
                            const double* J_nodal, double* pred, double* J, const gbp_td_mix* mix, void* stream)
{
    if (B < 0 || K < 1 || n_nodal < 1 || N < 1) return fail(GBP_ERR_INVALID_ARG, "B >= 0 and K, n_nodal, N >= 1 are required%s");
    if (B == 0) return GBP_OK;
    if (!nlayers || !W || !nodal || !pred) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    const bool with_j = J_nodal != nullptr || J != nullptr;
    if (with_j && (!J_nodal || !J)) return fail(GBP_ERR_INVALID_ARG, "J_nodal and J come together%s");
    gbp_td_mix mx;
    std::memset(&mx, 0, sizeof(mx));
    if (mix != nullptr && mix->n_in > 0) {
        if (!mix->src || !mix->col || !mix->weights || mix->terms < 1 || mix->n_weights < 1) return fail(GBP_ERR_INVALID_ARG, "incomplete gbp_td_mix%s");
        mx = *mix;
    } else if (mix != nullptr) {
        mx.offset = mix->offset;
    }
    const size_t lds = ((size_t)n_nodal * (with_j ? K + 1 : 1)) * sizeof(double);
    if (lds > 60000) return fail(GBP_ERR_INVALID_ARG, "n_nodal * max_layers too large for the time-domain stage%s");
    const size_t lds_plain = ((size_t)n_nodal * N + (size_t)4 * n_nodal) * sizeof(double);
    if (with_j)
        hipLaunchKernelGGL(rj::k_td_apply<true>, dim3(B), dim3(64), lds, (hipStream_t)stream, B, K, n_nodal, N, nlayers, W, nodal, J_nodal, pred, J, mx);
    else if (lds_plain <= 40 * 1024 && B >= 4 * GBP_TD_PLAIN_ROWS)                     // forward only: the window operator once per workgroup
        hipLaunchKernelGGL(rj::k_td_apply_plain, dim3((B + GBP_TD_PLAIN_ROWS - 1) / GBP_TD_PLAIN_ROWS), dim3(256), lds_plain, (hipStream_t)stream, B, n_nodal, N,
                           nlayers, W, nodal, pred, mx);
    else
        hipLaunchKernelGGL(rj::k_td_apply<false>, dim3(B), dim3(64), lds, (hipStream_t)stream, B, K, n_nodal, N, nlayers, W, nodal, J_nodal, pred, J, mx);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_rj_flush_posteriors(const gbp_rj_options* o, const gbp_rj_chains* c, void* stream)
{
    gbp_status st = rj_check(o, c);
    if (st != GBP_OK || c->B == 0 || !c->hitmap) return st;
    hipLaunchKernelGGL(rj::k_rj_flush, dim3(c->B), dim3(64), 0, (hipStream_t)stream, rj::extend(*o), *c);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_rj_debug_random(uint64_t seed, int64_t chain, int64_t iteration, int stream_id, int n, double* uniforms,
                               double* normals, void* stream)
{
    if (n < 0 || !uniforms || !normals) return fail(GBP_ERR_INVALID_ARG, "bad n or NULL pointer%s");
    hipLaunchKernelGGL(rj::k_rj_debug_random, dim3(1), dim3(1), 0, (hipStream_t)stream, seed, (uint32_t)chain, (uint32_t)iteration,
                       (uint32_t)stream_id, n, uniforms, normals);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

}  // extern "C"
