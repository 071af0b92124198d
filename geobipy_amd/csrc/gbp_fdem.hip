// gbp_fdem.hip -- gfx950 kernels + C ABI (include/geobipy_amd.h) for the batched FDEM forward solve,
// Gaussian misfit / log-likelihood and Jacobian.  Hand-written for CDNA4 wave64; no CUDA path.
//
// Work decomposition:
//   workgroup  = one sounding  (NW waves; NW chosen on the host from the batch size)
//   wave       = one frequency at a time (f = wave, wave + NW, ...)
//   lane       = one Hankel abscissa: 64 abscissae per pass, 2 passes for the 120-point J0 filter
//   layer recursion, csqrt/cexp in registers; sigma / thickness of the sounding are wave-uniform
//   (scalar loads), the per-abscissa tables (40 B per point) are read coalesced from L2.
//   Hankel sum = wave64 butterfly reduction; chi^2 / logdet = second wave reduction over channels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/geobipy_amd.h"
#include "gbp_fdem_point.h"
#include "gbp_fdem_tables.h"

using gbp::Channel;
using gbp::cplx;

namespace {

thread_local char g_err[512] = "";

gbp_status fail(gbp_status code, const char* fmt, const char* detail = "")
{
    std::snprintf(g_err, sizeof(g_err), fmt, detail);
    return code;
}

#define GBP_HIP(call)                                                          \
    do {                                                                       \
        hipError_t e_ = (call);                                                \
        if (e_ != hipSuccess) return fail(GBP_ERR_HIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

}  // namespace

// Per-altitude-bin abscissa windows of one system (gbp_fdem_system_create_binned): bin i holds the tables windowed for soundings
// at altitude >= bin0 + i metres; the kernel picks a sounding's bin from its own altitude, so the terms that are evaluated for
// a sounding depend on that sounding only (never on the batch it is evaluated in).
struct BinDesc {
    int chan_off;        // first Channel of the bin in d_bin_chan
    int npts_total;      // points of the bin's flattened list (stride of its SoA)
    long long pts_off;   // first double of the bin's SoA in d_bin_pts
};

struct gbp_fdem_system {
    gbp::SystemTables t;      // host copy (channels, H0, point tables)
    Channel* d_chan = nullptr;
    double* d_pts = nullptr;  // SoA, GBP_PT_FIELDS arrays of [npts] (gbp_fdem_point.h)
    int bin0 = 0, n_bins = 0;
    // Further table sets of the same layout (gbp_hankel_system_add_set: e.g. the tables of other transmitter-receiver offsets);
    // row b of a launch uses set set_of_row[b], an ARGUMENT of the *_rows_ex entries (NULL: set 0) -- the handle holds no per-call
    // state, so host threads may share it.  Descriptor layout in d_bins: the n_bins altitude windows of set 0, then per further
    // set its full tables followed by its n_bins windows.
    std::vector<gbp::SystemTables> extra_sets;
    // host copies of every set's windowed tables for the current (eps, relative, first altitude, bin count): a handle that grows
    // set by set (gbp_tdem_forward meeting new receiver offsets) windows only the NEW sets in gbp_hankel_system_add_bins
    struct Pack {
        std::vector<BinDesc> desc;      // offsets relative to the pack's own chans / pts
        std::vector<Channel> chans;
        std::vector<double> pts;
    };
    std::vector<Pack> packs;
    double pack_eps = -1.0;
    int pack_relative = -1, pack_first = -1, pack_bins = -1;
    BinDesc* d_bins = nullptr;
    Channel* d_bin_chan = nullptr;
    double* d_bin_pts = nullptr;
    std::vector<int> bin_npts;
    // Soundings whose layers all have sigma >= sigma_direct take the kernels' csqrt_upper2<DIRECT> branch: for every
    // frequency f and abscissa with a = lambda^2 - omega^2 mu0 eps0 < 0,  -a <= wmu_f sigma / 4, i.e. b^2 >= 16 a^2.
    // (= 4 omega_max eps0, 2.9e-5 S/m at 130 kHz; 0 for tables without a displacement-current term)
    double sigma_direct = 0.0;
    void set_sigma_direct()
    {
        sigma_direct = 0.0;
        for (const Channel& ch : t.chan)
            for (int j = 0; j < ch.npts; ++j) {
                const double a = t.soa[(size_t)ch.off + j];   // field 0 of the SoA = a
                if (a < 0.0) sigma_direct = std::max(sigma_direct, -a / (0.25 * ch.wmu));
            }
    }
};

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
using namespace gbp;  // constants named by GBP_MATHK_INIT
__constant__ gbp::MathK GBP_K = GBP_MATHK_INIT;
__device__ const double GBP_EXP2_64[64] = GBP_EXP2_64_LIST;
__device__ const double GBP_SINCOS_64[128] = GBP_SINCOS_64_LIST;

#ifdef GBP_RJ_PHYS_CLOCK
// Measurement builds only (scripts/build_ab.sh NAME -DGBP_RJ_PHYS_CLOCK, scripts/phys_clock.py): s_memtime stamps of the sampler's physics
// workgroups, summed per slot by thread 0 of every 16th workgroup.  [slot]: ticks of the 100 MHz constant clock; [slot + 32]: samples.
__device__ long long GBP_PHYS_TICKS[64];
__device__ long long GBP_PHYS_LIFE[3 * 16 * 3];      // [kind][min(layers, 15)][sum of lives, workgroups, longest life]
struct PhysClk { bool on; int base; long long t0; };
__device__ __forceinline__ void phys_tick(PhysClk* k, int slot)
{
    if (k != nullptr && k->on) {
        const long long t1 = (long long)wall_clock64();
        atomicAdd((unsigned long long*)&GBP_PHYS_TICKS[k->base + slot], (unsigned long long)(t1 - k->t0));
        atomicAdd((unsigned long long*)&GBP_PHYS_TICKS[k->base + slot + 32], 1ull);
        k->t0 = t1;
    }
}
#define GBP_TICK(slot) phys_tick(gbp_clk, (slot))
#define GBP_TICK_ARGS , PhysClk* gbp_clk = nullptr
#define GBP_TICK_PASS , gbp_clk
#else
#define GBP_TICK(slot)
#define GBP_TICK_ARGS
#define GBP_TICK_PASS
#endif

// per-workgroup copy of the lookup tables in LDS + the scalar constants in SGPRs
struct MathLds {
    double exp2_64[64];
    gbp::SinCos sincos_64[64];
};
// SYNC = false: the caller's next workgroup barrier (the one behind the layer-thickness fill of forward_body / sens_body) is the tables'
// too -- the sampler's physics kernel fills them first thing, while its chain's move and layer count are still on their way
template <bool SYNC = true>
__device__ __forceinline__ gbp::MathCtx math_setup(MathLds& lds)
{
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        lds.exp2_64[i] = GBP_EXP2_64[i];
        lds.sincos_64[i].s = GBP_SINCOS_64[2 * i];
        lds.sincos_64[i].c = GBP_SINCOS_64[2 * i + 1];
    }
    if (SYNC) __syncthreads();
    gbp::MathCtx M;
    M.k = GBP_K;
    M.e4_v = M.k.e4;
    M.s2_v = M.k.s2;
    M.c3_v = M.k.c3;
    asm volatile("" : "+v"(M.e4_v), "+v"(M.s2_v), "+v"(M.c3_v));  // opaque: stays in VGPRs, not re-materialised
    M.exp2_64 = lds.exp2_64;
    M.sincos_64 = lds.sincos_64;
    return M;
}

// Descriptor of (table set, altitude) in gbp_fdem_system::d_bins, or -1 for "the handle's own tables" (set 0 below the first bin).
__device__ __forceinline__ int table_slot(double alt, int set, int bin0, int n_bins)
{
    const bool binned = n_bins > 0 && alt >= (double)bin0;
    const int bi = binned ? min((int)(alt - (double)bin0), n_bins - 1) : 0;
    if (set == 0) return binned ? bi : -1;
    return n_bins + (set - 1) * (n_bins + 1) + (binned ? 1 + bi : 0);
}

// Sums over the 64 lanes by ONE balanced tree over the lane index (partners lane ^ 1, ^ 2, ^ 4, ... ^ 32): fp64 addition commutes,
// so every implementation of that tree below returns the same bits.
//   dpp_get<CTRL>: the value the DPP control names (VALU moves, no LDS crossbar round trip); lanes without a source read +0.0.
template <int CTRL>
__device__ __forceinline__ double dpp_get(double v)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_mov_dpp((unsigned)u, CTRL, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_mov_dpp((unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_rows(double v)      // (masked-off rows add +0.0)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Sum of v, returned to every lane as a wave-uniform value.  Quad swaps, then row shifts by 4 and 8 (lanes 12 - 15 of a row hold the
// row's sum), then the previous row's lane 15 broadcast into rows 1 - 3 and lane 31 into rows 2 - 3: lane 63 holds the total.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_get<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_get<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_get<0x114>(v);      // row_shr:4
    v += dpp_get<0x118>(v);      // row_shr:8
    v = dpp_add_rows<0x142, 0xa>(v);     // row_bcast:15 -> rows 1 and 3
    v = dpp_add_rows<0x143, 0xc>(v);     // row_bcast:31 -> rows 2 and 3: lane 63 holds the total
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 63), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// This library is written for gfx950 alone (v_permlane16_swap / v_permlane32_swap below exist on no earlier CDNA part): say so at
// compile time rather than with an unknown-builtin error in the middle of the reduction tree.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "geobipy_amd builds for gfx950 (MI355X) only: hipcc --offload-arch=gfx950"
#endif
// v[lane] + v[lane ^ W] for W = 16, 32 in every lane, by gfx950's row swaps (v_permlane16_swap / v_permlane32_swap exchange the odd
// 16- / 32-lane rows of one register with the even rows of another: VALU moves, where a shuffle would wait for the LDS crossbar)
template <int W>
__device__ __forceinline__ void row_pair(double v, double& x, double& y)     // {x, y} = {v[lane], v[lane ^ W]} in some order
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto a = W == 16 ? __builtin_amdgcn_permlane16_swap(lo, lo, false, false) : __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = W == 16 ? __builtin_amdgcn_permlane16_swap(hi, hi, false, false) : __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    x = __builtin_bit_cast(double, ((unsigned long long)b[0] << 32) | a[0]);
    y = __builtin_bit_cast(double, ((unsigned long long)b[1] << 32) | a[1]);
}
template <int W>
__device__ __forceinline__ double row_pair_sum(double v)
{
    double x, y;
    row_pair<W>(v, x, y);
    return x + y;
}
// Sums of a AND b with one tree instead of two: after the first exchange the even lanes carry a's partial sums and the odd lanes b's
// (the partner of an even lane sends its a, the partner of an odd lane its b), every later step pairs lanes of the same parity.
// Returns, in lane 62, the sum of a and, in lane 63, the sum of b -- the bits wave_sum(a) / wave_sum(b) return.  (The cross-row steps
// are row swaps: row_bcast only broadcasts lane 15, an odd one.)  26 VALU issues against 2 x 32.
__device__ __forceinline__ double wave_sum_pair(double a, double b, int lane)
{
    const bool odd = lane & 1;
    double v = odd ? b : a;
    v += dpp_get<0xB1>(odd ? a : b);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x114>(v);
    v += dpp_get<0x118>(v);
    v = row_pair_sum<16>(v);
    v = row_pair_sum<32>(v);
    return v;
}
// ... stored as one complex number (lanes 62 and 63 write their halves)
__device__ __forceinline__ void wave_sum_store(double re, double im, int lane, cplx* dst)
{
    const double v = wave_sum_pair(re, im, lane);
    if (lane >= 62) reinterpret_cast<double*>(dst)[lane - 62] = v;
}

// Smallest conductivity of the sounding (wave-uniform, returned in SGPRs; each wave of the workgroup evaluates it).
__device__ __forceinline__ double wave_min_sigma(const double* __restrict__ sig, int L, int lane)
{
    double v = 1.7976931348623157e308;
    for (int k = lane; k < L; k += 64) v = fmin(v, sig[k]);
    v = fmin(v, dpp_get<0xB1>(v));       // (a minimum does not depend on the order: quad swaps, the two mirrors of a row, row swaps)
    v = fmin(v, dpp_get<0x4E>(v));
    v = fmin(v, dpp_get<0x141>(v));
    v = fmin(v, dpp_get<0x140>(v));
    double x, y;
    row_pair<16>(v, x, y); v = fmin(x, y);
    row_pair<32>(v, x, y); v = fmin(x, y);
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Per-frequency layer constants of this sounding -> LDS (lanes k < L), then a wave barrier.
__device__ __forceinline__ void setup_layers(gbp::LayerK* lay, double wmu, const double* __restrict__ sig, int L,
                                             int lane)
{
    for (int k = lane; k < L; k += 64) {
        const double b = wmu * sig[k];
        lay[k].b2 = b * b;
        lay[k].bc = b * 0.70710678118654752440;
    }
    __builtin_amdgcn_wave_barrier();
}

// Forward solve over a contiguous range of 64-point passes of the sounding's FLATTENED (frequency, abscissa)
// point list (P points: 120 per zz frequency, 140 per xz/zx, 260 per xx).  Flattening matters because 120 is
// not a multiple of 64: per-frequency passes would idle 8 of 128 lanes, the flat list idles < 64 of P.
// A pass may straddle two frequencies ("cur" below the boundary lane, "next" above it; never three: a frequency has at
// least 64 points): each lane picks its frequency's layer constants (two LDS slots) and altitude term.
// Summation order: every pass reduces its lanes' terms -- per frequency -- to ONE partial sum, sh_part[p][0] for the
// frequency its first point belongs to and sh_part[p][1] for the one that starts inside it; forward_body adds a frequency's
// partials in pass order.  The result therefore does not depend on which wave ran which pass: any number of waves per
// sounding gives the same bits.
template <bool DIRECT>   // csqrt_upper2<DIRECT> is valid for every layer and abscissa of this sounding (gbp_fdem_system::sigma_direct)
__device__ __forceinline__ void forward_passes(const gbp::MathCtx& M, const Channel* __restrict__ chan,
                                               const double* __restrict__ pts, int P, int F, int L,
                                               const double* __restrict__ sig, const double* sh_t2,
                                               gbp::LayerK* sh_lay /* [2][Lmax] */, int Lmax, double alt, int p0,
                                               int p1, int lane, cplx* sh_part /* [npass][2] of the sounding */, double row_scale)
{
    if (p0 >= p1) return;
    int cur = 0;
    while (cur + 1 < F && chan[cur].off + chan[cur].npts <= 64 * p0) ++cur;
    int slot = 0;
    bool cur_ready = false;
    for (int p = p0; p < p1; ++p) {
        const Channel cc = chan[cur];
        if (!cur_ready) {
            setup_layers(sh_lay + (size_t)slot * Lmax, cc.wmu, sig, L, lane);
            cur_ready = true;
        }
        const int base = 64 * p;
        const int end_cur = cc.off + cc.npts;
        const bool has_next = (base + 64 > end_cur) && (cur + 1 < F);
        double hD = cc.hd0 - 2.0 * alt;
        const gbp::LayerK* lay = sh_lay + (size_t)slot * Lmax;
        int j = base + lane;
        const bool valid = j < P;
        bool in_next = false;
        if (has_next) {
            const Channel cn = chan[cur + 1];
            setup_layers(sh_lay + (size_t)(slot ^ 1) * Lmax, cn.wmu, sig, L, lane);
            in_next = j >= end_cur;
            if (in_next) {
                hD = cn.hd0 - 2.0 * alt;
                lay = sh_lay + (size_t)(slot ^ 1) * Lmax;
            }
        }
        gbp::Point pt;
        if (base + 64 > P) {   // ragged tail of the point list (wave-uniform branch)
            pt = gbp::load_point(pts, P, valid ? j : P - 1);
            if (!valid) pt.coef = gbp::mk(0.0, 0.0);
        } else {
            pt = gbp::load_point(pts, P, j);
        }
        if (row_scale != 1.0) gbp::scale_point(pt, row_scale);     // (wave-uniform: a receiver moved off its table set's distance)
        cplx num, den;
        gbp::rte_num_den<DIRECT>(M, pt.a, L, lay, sh_t2, pt.u0, num, den);
        const bool real_ue = __ballot(pt.ue.im != 0.0) == 0ull;            // (wave-uniform: see hankel_term)
        const cplx t = gbp::hankel_term(M, num, den, pt.ue, hD, pt.coef, real_ue);
        if (has_next) {
            wave_sum_store(in_next ? 0.0 : t.re, in_next ? 0.0 : t.im, lane, sh_part + 2 * p);
            wave_sum_store(in_next ? t.re : 0.0, in_next ? t.im : 0.0, lane, sh_part + 2 * p + 1);
        } else {
            wave_sum_store(t.re, t.im, lane, sh_part + 2 * p);
        }
        if (base + 64 >= end_cur) {   // frequency `cur` has no points beyond this pass
            ++cur;
            cur_ready = has_next;
            slot ^= 1;
            if (cur >= F) break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// chi^2 / logL of one sounding from N predicted channels held in `p` (LDS or global); executed by one wave.
__device__ __forceinline__ void loglike_wave(int N, const double* p, const double* __restrict__ obs, double rel,
                                             double add, int lane, double* chi2, double* logL)
{
    double s2 = 0.0, logdet = 0.0, na = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double o = obs[i];
        const bool active = o > 0.0;  // false for NaN and for non-positive data (EmDataPoint.py:54-56)
        const double ro = rel * o;
        const double var = ro * ro + add * add;  // DataPoint.py:274
        if (active) {
            const double r = (p[i] - o) * (1.0 / sqrt(var));  // DataPoint.py:523-524
            s2 += r * r;
            logdet += log(var);
            na += 1.0;
        }
    }
    // chi^2 and log-determinant share one reduction tree (lane 62 ends with chi^2, lane 63 with the log-determinant: the bits of two
    // wave_sums); the count of active channels -- a sum of ones, exact in any order -- goes through its own
    const double mine = wave_sum_pair(s2, logdet, lane);
    const double other = dpp_get<0xB1>(mine);
    na = wave_sum(na);
    if (lane == 62) *chi2 = mine;
    // MvNormalDistribution.py:209-216 with a diagonal covariance
    if (lane == 63) *logL = -(0.5 * na) * 1.8378770664093453 - 0.5 * mine - 0.5 * other;
}

// Forward solve (+ chi^2 / logL) of ONE sounding by the calling workgroup: the body of k_fdem_forward, also called once per
// iteration by the persistent sampler kernel (gbp_rjmcmc.h).  Every thread of the workgroup must call it (it contains
// workgroup barriers); `sh_out` holds 2 * GBP_MAX_FREQ doubles, `sh_dyn` dyn_lds_bytes(nwaves, Lmax, passes) bytes:
//   LayerK lay[nwaves][2][Lmax] | cplx part[passes][2] | double t2[Lmax]
// The passes are shared by the first `nw_use` waves of the workgroup (the others only take part in the barriers); the result
// does not depend on nw_use (see forward_passes).
template <bool LIKE>
__device__ __forceinline__ void forward_body(const gbp::MathCtx& M, double* sh_out, unsigned char* sh_dyn,
                                             const Channel* __restrict__ chan, const double* __restrict__ pts, int npts_total,
                                             int F, int Lmax, int L, const double* __restrict__ sig,
                                             const double* __restrict__ th, double alt, const double* __restrict__ obs_row,
                                             double rel_b, double add_b, double* __restrict__ pred_row, double* chi2_b,
                                             double* logL_b, double sigma_direct, int nw_use, double row_scale = 1.0 GBP_TICK_ARGS)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nw_use;
    const int npass = (npts_total + 63) >> 6;
    // every layer conductive enough for the select-free complex sqrt (all but displacement-current dominated models)
    const bool direct = wave_min_sigma(sig, L, lane) >= sigma_direct;   // workgroup-uniform
    gbp::LayerK* sh_lay = reinterpret_cast<gbp::LayerK*>(sh_dyn) + (size_t)wave * 2 * Lmax;
    cplx* sh_part = reinterpret_cast<cplx*>(sh_dyn + (size_t)nwaves * 2 * Lmax * sizeof(gbp::LayerK));
    double* sh_t2 = reinterpret_cast<double*>(sh_part + (size_t)2 * npass);
    for (int k = threadIdx.x; k < L - 1; k += blockDim.x) sh_t2[k] = -2.0 * th[k];
    __syncthreads();
    GBP_TICK(3);

    if (wave < nwaves) {
        const int per = (npass + nwaves - 1) / nwaves;
        const int p0 = wave * per;
        const int p1 = min(npass, p0 + per);
        if (direct)
            forward_passes<true>(M, chan, pts, npts_total, F, L, sig, sh_t2, sh_lay, Lmax, alt, p0, p1, lane, sh_part, row_scale);
        else
            forward_passes<false>(M, chan, pts, npts_total, F, L, sig, sh_t2, sh_lay, Lmax, alt, p0, p1, lane, sh_part, row_scale);
    }
    GBP_TICK(4);
    __syncthreads();
    GBP_TICK(5);

    // out_f = 1e6 * scale * (H - H0) / H0 = g_f * sum of the frequency's per-pass partials in pass order
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const Channel ch = chan[f];
        double sr = 0.0, si = 0.0;
        for (int p = ch.off >> 6; 64 * p < ch.off + ch.npts; ++p) {
            const cplx v = sh_part[2 * p + (64 * p >= ch.off ? 0 : 1)];   // the pass's own frequency, or the one starting inside it
            sr += v.re; si += v.im;
        }
        sh_out[f] = ch.g_re * sr - ch.g_im * si;
        sh_out[F + f] = ch.g_re * si + ch.g_im * sr;
    }
    __syncthreads();

    const int N = 2 * F;
    if (pred_row != nullptr)
        for (int i = threadIdx.x; i < N; i += blockDim.x) pred_row[i] = sh_out[i];
    if (LIKE && wave == 0) loglike_wave(N, sh_out, obs_row, rel_b, add_b, lane, chi2_b, logL_b);
    GBP_TICK(6);
}

template <bool LIKE, bool SCALED = false>   // SCALED: the rows carry a distance scale (gbp_fdem_forward_rows_scaled); the plain kernels do not pay for it (4 VGPRs, 36 B of scratch)
__global__ __launch_bounds__(1024) void k_fdem_forward(const Channel* __restrict__ chan,
                                                       const double* __restrict__ pts, int npts_total, int F,
                                                       int Lmax, const int* __restrict__ nlayers,
                                                       const double* __restrict__ sigma,
                                                       const double* __restrict__ thk,
                                                       const double* __restrict__ height,
                                                       const double* __restrict__ obs,
                                                       const double* __restrict__ rel,
                                                       const double* __restrict__ add, double* __restrict__ pred,
                                                       double* __restrict__ chi2, double* __restrict__ logL,
                                                       double sigma_direct, const BinDesc* __restrict__ bins, int bin0, int n_bins,
                                                       const Channel* __restrict__ bin_chan, const double* __restrict__ bin_pts,
                                                       const int* __restrict__ row_set, const double* __restrict__ row_scale)
{
    __shared__ double sh_out[2 * GBP_MAX_FREQ];
    __shared__ MathLds sh_math;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int b = blockIdx.x;
    if (bins != nullptr) {                                  // this sounding's table set and abscissa window (the bin of its own altitude)
        const int slot = table_slot(height[b], row_set != nullptr ? row_set[b] : 0, bin0, n_bins);
        if (slot >= 0) {                                    // (set 0 below the first bin, NaN: the handle's own tables, all abscissae)
            const BinDesc d = bins[slot];
            chan = bin_chan + d.chan_off;
            pts = bin_pts + d.pts_off;
            npts_total = d.npts_total;
        }
    }
    const int L = nlayers[b];
    if (L <= 0) return;   // sounding skipped by the caller (workgroup-uniform): none of its outputs are written
    if (L > Lmax) {       // bad row (would overrun the rows of sigma / thk and the LDS layer tables): flag it with NaNs, touch nothing else
        const double qnan = __builtin_nan("");
        if (pred != nullptr)
            for (int i = threadIdx.x; i < 2 * F; i += blockDim.x) pred[(size_t)b * 2 * F + i] = qnan;
        if (LIKE && threadIdx.x == 0) { chi2[b] = qnan; logL[b] = qnan; }
        return;
    }
    const gbp::MathCtx M = math_setup(sh_math);  // ends with __syncthreads()
    forward_body<LIKE>(M, sh_out, sh_dyn, chan, pts, npts_total, F, Lmax, L, sigma + (size_t)b * Lmax, thk + (size_t)b * Lmax,
                       height[b], LIKE ? obs + (size_t)b * 2 * F : nullptr, LIKE ? rel[b] : 0.0, LIKE ? add[b] : 0.0,
                       pred != nullptr ? pred + (size_t)b * 2 * F : nullptr, LIKE ? chi2 + b : nullptr, LIKE ? logL + b : nullptr,
                       sigma_direct, (int)(blockDim.x >> 6), (SCALED && row_scale != nullptr) ? row_scale[b] : 1.0);
}

// Jacobian (+ prediction) of ONE sounding by the first `nw_use` waves of the calling workgroup: the body of k_fdem_sens, also
// called by the persistent sampler kernel.  Each lane keeps d rTE / d ln sigma_m for its abscissa in LDS (row m, column
// lane, row stride 65 slots so that the row sums below are conflict-free), lane m then sums row m over the 64 abscissae.
// J[f, m] = Re(g * sum), J[F + f, m] = Im(g * sum).  Every thread of the workgroup must call it (one workgroup barrier).
//   sh_dyn: cplx D[nw_use][Lalloc][65] | LayerK lay[nw_use][Lalloc] | double t2[Lalloc]
#define GBP_SENS_STRIDE 65
template <bool EXACT, int NG>   // NG row groups of 8 layers are summed per evaluation: 1 / 2 / 4 for launches of <= 8 / 16 / more layers
__device__ __forceinline__ void sens_body(const gbp::MathCtx& M, unsigned char* sh_dyn, const Channel* __restrict__ chan,
                                          const double* __restrict__ pts, int npts_total, int F, int Lmax, int Lalloc, int L,
                                          const double* __restrict__ sig, const double* __restrict__ th, double alt,
                                          double* __restrict__ Jb /* [2F, Lmax] of this sounding */,
                                          double* __restrict__ pred_row /* [2F] or NULL */, int nw_use, int zero_to, double row_scale = 1.0 GBP_TICK_ARGS,
                                          int share = 0, int n_shares = 1)
{   // share / n_shares: this workgroup evaluates the frequencies  wave * n_shares + share,  + nw_use * n_shares, ...  -- a frequency's
    // rows of J and pred depend on nothing but that frequency, so n_shares workgroups split a sounding's evaluation and write the same bits
    // zero_to: the unused columns L .. zero_to - 1 of every row are set to 0 (Lmax: the whole row, the public entries; the
    // sampler, whose consumers never read a column >= L, passes L rounded up to 8 and leaves the rest of the row alone)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = nw_use;
    cplx* sh_D = reinterpret_cast<cplx*>(sh_dyn) + (size_t)wave * Lalloc * GBP_SENS_STRIDE;
    gbp::LayerK* sh_lay = reinterpret_cast<gbp::LayerK*>(sh_dyn + (size_t)nwaves * Lalloc * GBP_SENS_STRIDE * sizeof(cplx)) +
                          (size_t)wave * Lalloc;
    double* sh_t2 = reinterpret_cast<double*>(sh_dyn + (size_t)nwaves * Lalloc * (GBP_SENS_STRIDE * sizeof(cplx) + sizeof(gbp::LayerK)));
    for (int k = threadIdx.x; k < L - 1; k += blockDim.x) sh_t2[k] = -2.0 * th[k];
    __syncthreads();
    GBP_TICK(3);
    if (wave >= nwaves) return;

    for (int f = wave * n_shares + share; f < F; f += nwaves * n_shares) {
        const Channel ch = chan[f];
        setup_layers(sh_lay, ch.wmu, sig, L, lane);
        const double hD = ch.hd0 - 2.0 * alt;
        // Row sums: lanes are arranged as 8 rows x 8 segments; lane (row, seg) adds the entries seg, seg + 8, ...
        // of row m0 + row (consecutive lanes -> consecutive 16-byte slots: conflict-free ds_read_b128), the 8
        // segment partials are combined with three xor-shuffles once per frequency.
        const int row = lane >> 3, seg = lane & 7;
        for (int m0 = 0; m0 < L; m0 += 8 * NG) {       // (runs once for L <= 8 NG; deeper models re-evaluate)
            double acc_re[NG], acc_im[NG];
            double fw_re = 0.0, fw_im = 0.0;       // this lane's share of the forward sum (fm_dlogc)
#pragma unroll
            for (int g = 0; g < NG; ++g) { acc_re[g] = 0.0; acc_im[g] = 0.0; }
            for (int j0 = 0; j0 < ch.npts; j0 += 64) {
                {
                    const int j = j0 + lane;
                    const bool valid = j < ch.npts;
                    gbp::Point pt = gbp::load_point(pts, npts_total, ch.off + (valid ? j : ch.npts - 1));
                    if (!valid) pt.coef = gbp::mk(0.0, 0.0);
                    if (row_scale != 1.0) gbp::scale_point(pt, row_scale);     // (wave-uniform, see forward_passes)
                    // (real exponents for every lane of the pass: exp only, same bits -- see gbp::hankel_term)
                    const cplx Q = __ballot(pt.ue.im != 0.0) == 0ull ? pt.coef * gbp::exp_neg(M, pt.ue.re * hD)
                                                                     : gbp::cexp_neg(M, pt.ue.re * hD, pt.ue.im * hD) * pt.coef;
                    const cplx t = gbp::sens_point<EXACT>(M, pt.a, L, sh_lay, sh_t2, pt.u0, Q, sh_D + lane,
                                                          GBP_SENS_STRIDE);
                    fw_re += t.re;
                    fw_im += t.im;
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (m0 + 8 * g < L) {              // wave-uniform
                        const int m = m0 + 8 * g + row;
                        if (m < L) {
                            const cplx* rowp = sh_D + (size_t)m * GBP_SENS_STRIDE + seg;
#pragma unroll
                            for (int i = 0; i < 8; ++i) { acc_re[g] += rowp[8 * i].re; acc_im[g] += rowp[8 * i].im; }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (pred_row != nullptr && m0 == 0) {  // one wave owns the frequency: fixed summation order for any launch shape
                const double mine = wave_sum_pair(fw_re, fw_im, lane);   // lane 62: the real sum, lane 63: the imaginary one
                const double other = dpp_get<0xB1>(mine);
                if (lane == 62) pred_row[f] = ch.g_re * mine - ch.g_im * other;
                if (lane == 63) pred_row[F + f] = ch.g_re * mine + ch.g_im * other;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (m0 + 8 * g < L) {
                    double sr = acc_re[g], si = acc_im[g];
                    sr += dpp_get<0xB1>(sr); si += dpp_get<0xB1>(si);       // the 8 segment partials of a row: lane ^ 1, ^ 2, then the
                    sr += dpp_get<0x4E>(sr); si += dpp_get<0x4E>(si);       // other quad of the 8 (row_half_mirror: every lane of a quad holds
                    sr += dpp_get<0x141>(sr); si += dpp_get<0x141>(si);     // the quad's sum by now) -- the sums of the xor shuffles, bit for bit
                    const int m = m0 + 8 * g + row;
                    if (m < L && seg == 0) {
                        Jb[(size_t)f * Lmax + m] = ch.g_re * sr - ch.g_im * si;
                        Jb[(size_t)(F + f) * Lmax + m] = ch.g_re * si + ch.g_im * sr;
                    }
                }
            }
        }
        for (int m = L + lane; m < zero_to; m += 64) {   // unused columns
            Jb[(size_t)f * Lmax + m] = 0.0;
            Jb[(size_t)(F + f) * Lmax + m] = 0.0;
        }
    }
    GBP_TICK(4);
}

template <bool EXACT, int NG>
__global__ __launch_bounds__(1024) void k_fdem_sens(const Channel* __restrict__ chan, const double* __restrict__ pts,
                                                    int npts_total, int F, int Lmax, int Lalloc,
                                                    const int* __restrict__ nlayers,
                                                    const double* __restrict__ sigma,
                                                    const double* __restrict__ thk,
                                                    const double* __restrict__ height, double* __restrict__ J,
                                                    double* __restrict__ pred, const BinDesc* __restrict__ bins, int bin0, int n_bins,
                                                    const Channel* __restrict__ bin_chan, const double* __restrict__ bin_pts,
                                                    int compact_rows, const int* __restrict__ row_set, const double* __restrict__ row_scale)
{
    __shared__ MathLds sh_math;
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_dyn[];
    const int b = blockIdx.x;
    const int L = nlayers[b];
    if (L <= 0) return;   // sounding skipped by the caller (workgroup-uniform)
    if (bins != nullptr) {                                  // this sounding's table set and abscissa window (see k_fdem_forward)
        const int slot = table_slot(height[b], row_set != nullptr ? row_set[b] : 0, bin0, n_bins);
        if (slot >= 0) {
            const BinDesc d = bins[slot];
            chan = bin_chan + d.chan_off;
            pts = bin_pts + d.pts_off;
            npts_total = d.npts_total;
        }
    }
    if (L > Lalloc || L > Lmax) {   // more layers than the launch was sized for: NaN row instead of an LDS / row overrun
        const double qnan = __builtin_nan("");
        const size_t n = (size_t)2 * F * Lmax;
        for (size_t i = threadIdx.x; i < n; i += blockDim.x) J[(size_t)b * n + i] = qnan;
        if (pred != nullptr)
            for (int i = threadIdx.x; i < 2 * F; i += blockDim.x) pred[(size_t)b * 2 * F + i] = qnan;
        return;
    }
    const gbp::MathCtx M = math_setup(sh_math);
    sens_body<EXACT, NG>(M, sh_dyn, chan, pts, npts_total, F, Lmax, Lalloc, L, sigma + (size_t)b * Lmax, thk + (size_t)b * Lmax,
                         height[b], J + (size_t)b * 2 * F * Lmax, pred != nullptr ? pred + (size_t)b * 2 * F : nullptr,
                         (int)(blockDim.x >> 6), compact_rows ? min(Lmax, (L + 7) & ~7) : Lmax, row_scale != nullptr ? row_scale[b] : 1.0);
}

// chi^2 / logL with an explicit per-channel standard deviation (TdemDataPoint.std is time dependent)
__global__ void k_gauss_loglike_std(int B, int N, const double* __restrict__ pred, const double* __restrict__ obs,
                                    const double* __restrict__ sd, double* __restrict__ chi2,
                                    double* __restrict__ logL)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const double* p = pred + (size_t)b * N;
    const double* o = obs + (size_t)b * N;
    const double* s = sd + (size_t)b * N;
    double s2 = 0.0, logdet = 0.0, na = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double oi = o[i];
        if (oi > 0.0) {
            const double r = (p[i] - oi) * (1.0 / s[i]);
            s2 += r * r;
            logdet += 2.0 * log(s[i]);
            na += 1.0;
        }
    }
    s2 = wave_sum(s2);
    logdet = wave_sum(logdet);
    na = wave_sum(na);
    if (lane == 0) {
        chi2[b] = s2;
        logL[b] = -(0.5 * na) * 1.8378770664093453 - 0.5 * logdet - 0.5 * s2;
    }
}

__global__ void k_gauss_loglike(int B, int N, const double* __restrict__ pred, const double* __restrict__ obs,
                                const double* __restrict__ rel, const double* __restrict__ add,
                                double* __restrict__ chi2, double* __restrict__ logL)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    loglike_wave(N, pred + (size_t)b * N, obs + (size_t)b * N, rel[b], add[b], lane, chi2 + b, logL + b);
}

// Per-sounding input / output status word (SURVEY 8b "Errors": validate, flag per sounding, never abort the batch).
__global__ void k_fdem_validate(int B, int Lmax, int N, const int* __restrict__ nlayers, const double* __restrict__ sigma,
                                const double* __restrict__ thk, const double* __restrict__ height,
                                const double* __restrict__ pred, int* __restrict__ status)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int st = 0;
    const int L = nlayers[b];
    if (L < 1 || L > Lmax) st |= GBP_ROW_BAD_NLAYERS;
    const int Lc = (L < 1 || L > Lmax) ? 0 : L;   // a row with a bad layer count has no meaningful layers to check
    for (int k = 0; k < Lc; ++k) {
        const double s = sigma[(size_t)b * Lmax + k];
        if (!(s > 0.0) || !(s < 1.7976931348623157e308)) st |= GBP_ROW_BAD_SIGMA;
        if (k < Lc - 1) {
            const double t = thk[(size_t)b * Lmax + k];
            if (!(t > 0.0) || !(t < 1.7976931348623157e308)) st |= GBP_ROW_BAD_THICKNESS;
        }
    }
    const double h = height[b];
    if (!(h >= 0.0) || !(h < 1.7976931348623157e308)) st |= GBP_ROW_BAD_HEIGHT;
    if (pred != nullptr)
        for (int i = 0; i < N; ++i) {
            const double v = pred[(size_t)b * N + i];
            if (!(v == v) || v > 1.7976931348623157e308 || v < -1.7976931348623157e308) st |= GBP_ROW_NONFINITE_OUTPUT;
        }
    status[b] = st;
}

// Evaluates the device math kernels element-wise (test hook: tests/test_gpu_math.py checks their
// accuracy on the real hardware, where v_rsq_f64 / v_rcp_f64 seeds differ from the host's).
__global__ void k_debug_math(int op, int n, const double* __restrict__ x, const double* __restrict__ y,
                             double* __restrict__ o0, double* __restrict__ o1)
{
    __shared__ MathLds sh_math;
    const gbp::MathCtx M = math_setup(sh_math);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        double a = x[i], b = y ? y[i] : 0.0, r0 = 0.0, r1 = 0.0;
        switch (op) {
        case 0: r0 = gbp::exp_neg(M, a); break;
        case 1: gbp::sincos_tab(M, a, r0, r1); break;
        case 2: { cplx z = gbp::csqrt_upper(a, b); r0 = z.re; r1 = z.im; break; }
        case 3: r0 = gbp::rcp(a); break;
        case 4: r0 = gbp::rsq_seed(a); break;
        case 5: r0 = gbp::rcp_seed(a); break;
        case 6: gbp::sqrt_rsqrt(a, r0, r1); break;
        case 7: r0 = gbp::log_pos(a); break;
        case 8: gbp::sincos_quadrant(a, r0, r1); break;
        default: break;
        }
        o0[i] = r0;
        if (o1) o1[i] = r1;
    }
}

// ------------------------------------------------------------------------------------------
// host: system tables
// ------------------------------------------------------------------------------------------
namespace {

typedef std::complex<double> zc;

size_t dyn_lds_bytes(int nw, int Lmax, int passes)
{
    return (size_t)nw * 2 * Lmax * sizeof(gbp::LayerK) + (size_t)2 * passes * sizeof(cplx) + (size_t)Lmax * sizeof(double);
}

// waves per workgroup: enough workgroups x waves to fill 256 CUs x 8 waves even for small batches.  `waves` > 0 is the
// caller's explicit choice (the *_ex entries).  Results do not depend on it (forward_passes).
int pick_waves(int B, int F, int Lmax, int max_waves, int waves)
{
    // measured (scripts/sweep_waves.py, 10 frequencies x 8 layers, default abscissa windows = 10 passes per sounding): one wave
    // per sounding is best once every SIMD has a queue of soundings -- from 8 192 soundings (66.9 / 70.0 / 71.0 / 70.7 M evals/s at
    // 8 192 / 16 384 / 32 768 / 65 536 against 65.5 / 67.6 / 67.5 / 69.6 M with four waves, whose 10 passes split 3 3 2 2); below
    // that 4 waves per sounding balance the tail better (61.4 vs 60.9 M at 4 096, 56.7 vs 47.3 M at 2 048), and small batches need
    // 8192 / B waves to fill the chip at all.  (Round 2 switched at 49 152, measured with 19 passes per sounding.)
    const int heuristic = B >= 8192 ? 1 : std::max(4, (8192 + B - 1) / B);
    int nw = waves > 0 ? waves : heuristic;
    if (nw > max_waves) nw = max_waves;
    if (nw > 16) nw = 16;
    while (nw > 1 && dyn_lds_bytes(nw, Lmax, max_waves) > 60000) --nw;
    if (nw < 1) nw = 1;
    return nw;
}

const int GBP_MAX_LAYERS = 1024;  // LDS budget: (32 nw + 8) * Lmax bytes <= 64 KiB at nw = 1

gbp_status check_batch(const gbp_fdem_system* sys, int B, int Lmax, const void* a, const void* b, const void* c,
                       const void* d)
{
    if (!sys) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    if (B < 0 || Lmax < 1) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0 and Lmax >= 1%s");
    if (Lmax > GBP_MAX_LAYERS) return fail(GBP_ERR_INVALID_ARG, "Lmax must be <= 1024%s");
    if (B > 0 && (!a || !b || !c || !d)) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    return GBP_OK;
}

// per-row table sets need the descriptors gbp_hankel_system_add_bins builds after the last gbp_hankel_system_add_set
gbp_status check_row_sets(const gbp_fdem_system* sys, const int32_t* set_of_row)
{
    if (set_of_row != nullptr && !sys->extra_sets.empty() && sys->d_bins == nullptr)
        return fail(GBP_ERR_INVALID_ARG, "table sets need their descriptors: call gbp_hankel_system_add_bins after the last add_set%s");
    return GBP_OK;
}

}  // namespace

static gbp_status fm_dlogc_launch(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                                  const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                                  int waves, int compact_rows, const int32_t* set_of_row, void* stream, const double* row_scale = nullptr);

extern "C" {

const char* gbp_version(void) { return "geobipy_amd 0.2 (gfx950)"; }
const char* gbp_last_error(void) { return g_err; }

gbp_status gbp_device_count(int* count)
{
    if (!count) return fail(GBP_ERR_INVALID_ARG, "count is NULL%s");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(GBP_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return GBP_OK;
}

gbp_status gbp_fdem_system_create(int nF, const int32_t* tid, const double* frequencies, const double* tx_z,
                                  const double* rx_z, const double* tx_moment, const double* scale,
                                  const double* rx_off, const double* separation, const double* w0,
                                  const double* lamda0, const double* w1, const double* lamda1,
                                  gbp_fdem_system** out)
{
    return gbp_fdem_system_create_windowed(nF, tid, frequencies, tx_z, rx_z, tx_moment, scale, rx_off, separation, w0,
                                           lamda0, w1, lamda1, 0.0, 0.0, out);
}

gbp_status gbp_fdem_system_create_windowed(int nF, const int32_t* tid, const double* frequencies, const double* tx_z,
                                           const double* rx_z, const double* tx_moment, const double* scale,
                                           const double* rx_off, const double* separation, const double* w0,
                                           const double* lamda0, const double* w1, const double* lamda1,
                                           double eps_ppm, double min_altitude, gbp_fdem_system** out)
{
    if (!out) return fail(GBP_ERR_INVALID_ARG, "out is NULL%s");
    *out = nullptr;
    gbp_fdem_system* s = new (std::nothrow) gbp_fdem_system();
    if (!s) return fail(GBP_ERR_INVALID_ARG, "out of host memory%s");
    const char* msg = "";
    int rc = gbp::build_system_tables(nF, tid, frequencies, tx_z, rx_z, tx_moment, scale, rx_off, separation, w0,
                                      lamda0, w1, lamda1, &s->t, &msg);
    if (rc != GBP_OK) { delete s; return fail(rc, "%s", msg); }
    if (eps_ppm > 0.0 && !(min_altitude >= 0.0)) { delete s; return fail(GBP_ERR_INVALID_ARG, "min_altitude must be >= 0%s"); }
    gbp::window_system_tables(&s->t, eps_ppm, min_altitude);
    s->set_sigma_direct();
    const std::vector<double>& soa = s->t.soa;

    hipError_t e = hipMalloc((void**)&s->d_chan, sizeof(Channel) * nF);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_pts, sizeof(double) * soa.size());
    if (e == hipSuccess) e = hipMemcpy(s->d_chan, s->t.chan.data(), sizeof(Channel) * nF, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_pts, soa.data(), sizeof(double) * soa.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gbp_fdem_system_destroy(s);
        return fail(GBP_ERR_HIP, "system table upload failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return GBP_OK;
}

gbp_status gbp_fdem_system_create_binned(int nF, const int32_t* tid, const double* frequencies, const double* tx_z,
                                         const double* rx_z, const double* tx_moment, const double* scale,
                                         const double* rx_off, const double* separation, const double* w0,
                                         const double* lamda0, const double* w1, const double* lamda1, double eps_ppm,
                                         int first_altitude_m, int n_bins, gbp_fdem_system** out)
{
    if (!out) return fail(GBP_ERR_INVALID_ARG, "out is NULL%s");
    *out = nullptr;
    if (!(eps_ppm > 0.0) || first_altitude_m < 0 || n_bins < 1 || n_bins > 1024)
        return fail(GBP_ERR_INVALID_ARG, "eps_ppm > 0, first_altitude_m >= 0 and 1 <= n_bins <= 1024 are required%s");
    gbp_fdem_system* s = nullptr;
    gbp_status st = gbp_fdem_system_create(nF, tid, frequencies, tx_z, rx_z, tx_moment, scale, rx_off, separation, w0, lamda0, w1, lamda1, &s);
    if (st != GBP_OK) return st;
    st = gbp_hankel_system_add_bins(s, eps_ppm, 0, first_altitude_m, n_bins);
    if (st != GBP_OK) {
        gbp_fdem_system_destroy(s);
        return st;
    }
    *out = s;
    return GBP_OK;
}

static void drop_device_bins(gbp_fdem_system* s)
{
    if (s->d_bins) { (void)hipFree(s->d_bins); s->d_bins = nullptr; }
    if (s->d_bin_chan) { (void)hipFree(s->d_bin_chan); s->d_bin_chan = nullptr; }
    if (s->d_bin_pts) { (void)hipFree(s->d_bin_pts); s->d_bin_pts = nullptr; }
    s->n_bins = 0;
    s->bin_npts.clear();
}

gbp_status gbp_hankel_system_clear_bins(gbp_fdem_system* s)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    drop_device_bins(s);
    s->packs.clear();
    s->pack_eps = -1.0;
    return GBP_OK;
}

gbp_status gbp_hankel_system_add_bins(gbp_fdem_system* s, double eps, int relative, int first_altitude_m, int n_bins)
{
    if (!s) return fail(GBP_ERR_INVALID_ARG, "system handle is NULL%s");
    const bool sets_only = n_bins == 0 && eps == 0.0 && !s->extra_sets.empty();     // descriptors of the further sets' full tables
    if (!sets_only && (!(eps > 0.0) || first_altitude_m < 0 || n_bins < 1 || n_bins > 1024))
        return fail(GBP_ERR_INVALID_ARG, "eps > 0, first_altitude_m >= 0 and 1 <= n_bins <= 1024 are required%s");
    drop_device_bins(s);                                                        // (replaces an earlier set)
    if (s->pack_eps != eps || s->pack_relative != (relative != 0) || s->pack_first != first_altitude_m || s->pack_bins != n_bins) {
        s->packs.clear();
        s->pack_eps = eps; s->pack_relative = relative != 0; s->pack_first = first_altitude_m; s->pack_bins = n_bins;
    }
    try {
        for (int k = (int)s->packs.size(); k <= (int)s->extra_sets.size(); ++k) {   // only the sets that are new since the last call
            const gbp::SystemTables& full = k == 0 ? s->t : s->extra_sets[k - 1];
            gbp_fdem_system::Pack pk;
            auto push = [&](const gbp::SystemTables& t) {
                BinDesc d;
                d.chan_off = (int)pk.chans.size();
                d.npts_total = t.npts;
                d.pts_off = (long long)pk.pts.size();
                pk.desc.push_back(d);
                pk.chans.insert(pk.chans.end(), t.chan.begin(), t.chan.end());
                pk.pts.insert(pk.pts.end(), t.soa.begin(), t.soa.end());
            };
            if (k > 0) push(full);                           // a further set's own tables (soundings below the first bin)
            for (int i = 0; i < n_bins; ++i) {
                gbp::SystemTables t = full;                  // the exact tables, then windowed for altitude >= first + i metres
                gbp::window_system_tables(&t, eps, (double)(first_altitude_m + i), relative != 0);
                push(t);
            }
            s->packs.push_back(std::move(pk));
        }
    } catch (const std::bad_alloc&) {
        s->packs.clear();
        s->pack_eps = -1.0;
        return fail(GBP_ERR_INVALID_ARG, "out of host memory for the abscissa windows of the table sets%s");
    }
    std::vector<BinDesc> desc;
    size_t n_chan = 0, n_pts = 0;
    for (const auto& pk : s->packs) {
        for (BinDesc d : pk.desc) { d.chan_off += (int)n_chan; d.pts_off += (long long)n_pts; desc.push_back(d); }
        n_chan += pk.chans.size();
        n_pts += pk.pts.size();
    }
    s->bin_npts.resize(n_bins);
    for (int i = 0; i < n_bins; ++i) s->bin_npts[i] = s->packs[0].desc[i].npts_total;
    s->bin0 = first_altitude_m;
    hipError_t e = hipMalloc((void**)&s->d_bins, sizeof(BinDesc) * std::max<size_t>(desc.size(), 1));
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_bin_chan, sizeof(Channel) * std::max<size_t>(n_chan, 1));
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_bin_pts, sizeof(double) * std::max<size_t>(n_pts, 1));
    if (e == hipSuccess && !desc.empty()) e = hipMemcpy(s->d_bins, desc.data(), sizeof(BinDesc) * desc.size(), hipMemcpyHostToDevice);
    n_chan = n_pts = 0;
    for (const auto& pk : s->packs) {
        if (e == hipSuccess && !pk.chans.empty())
            e = hipMemcpy(s->d_bin_chan + n_chan, pk.chans.data(), sizeof(Channel) * pk.chans.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && !pk.pts.empty())
            e = hipMemcpy(s->d_bin_pts + n_pts, pk.pts.data(), sizeof(double) * pk.pts.size(), hipMemcpyHostToDevice);
        n_chan += pk.chans.size();
        n_pts += pk.pts.size();
    }
    if (e != hipSuccess) {
        drop_device_bins(s);
        return fail(GBP_ERR_HIP, "bin table upload failed: %s", hipGetErrorString(e));
    }
    s->n_bins = n_bins;
    return GBP_OK;
}

gbp_status gbp_hankel_system_add_set(gbp_fdem_system* s, const double* hd0, const double* tables)
{
    if (!s || !hd0 || !tables) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    gbp::SystemTables t = s->t;                          // same frequencies, weights and point counts: other altitude terms and points
    for (int f = 0; f < t.nF; ++f) t.chan[f].hd0 = hd0[f];
    t.soa.assign(tables, tables + (size_t)GBP_PT_FIELDS * t.npts);
    s->extra_sets.push_back(std::move(t));
    drop_device_bins(s);                                 // descriptors are rebuilt by the next gbp_hankel_system_add_bins (the
                                                         // windows of the sets that were there are kept on the host)
    return GBP_OK;
}

gbp_status gbp_fdem_system_bin_points(const gbp_fdem_system* sys, int altitude_m, int* npts)
{
    if (!sys || !npts) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    if (sys->n_bins == 0 || altitude_m < sys->bin0) { *npts = sys->t.npts; return GBP_OK; }
    *npts = sys->bin_npts[std::min(altitude_m - sys->bin0, sys->n_bins - 1)];
    return GBP_OK;
}

gbp_status gbp_hankel_system_create_raw(int nF, const int32_t* npts, const double* wmu, const double* hd0,
                                        const double* g, const double* tables, gbp_fdem_system** out)
{
    if (!out) return fail(GBP_ERR_INVALID_ARG, "out is NULL%s");
    *out = nullptr;
    if (nF < 1 || nF > GBP_MAX_FREQ) return fail(GBP_ERR_INVALID_ARG, "nF must be in [1, 128]%s");
    if (!npts || !wmu || !hd0 || !g || !tables) return fail(GBP_ERR_INVALID_ARG, "NULL system array%s");
    gbp_fdem_system* s = new (std::nothrow) gbp_fdem_system();
    if (!s) return fail(GBP_ERR_INVALID_ARG, "out of host memory%s");
    gbp::SystemTables& t = s->t;
    t.nF = nF;
    t.chan.assign(nF, Channel());
    t.h0.assign(2 * (size_t)nF, 1.0);
    int P = 0;
    for (int f = 0; f < nF; ++f) {
        if (npts[f] < 64 || !(wmu[f] > 0.0) || !std::isfinite(wmu[f])) {
            delete s;
            return fail(GBP_ERR_BAD_SYSTEM, "each frequency needs >= 64 points and wmu > 0%s");
        }
        Channel& ch = t.chan[f];
        ch.wmu = wmu[f]; ch.w2me = 0.0; ch.hd0 = hd0[f]; ch.g_re = g[2 * f]; ch.g_im = g[2 * f + 1];
        ch.off = P; ch.npts = npts[f]; ch.real_exp = 0; ch.tid = 0;
        if (npts[f] > t.max_pts) t.max_pts = npts[f];
        P += npts[f];
    }
    t.npts = P;
    t.soa.assign(tables, tables + (size_t)GBP_PT_FIELDS * P);
    s->set_sigma_direct();
    const std::vector<double>& soa = t.soa;
    hipError_t e = hipMalloc((void**)&s->d_chan, sizeof(Channel) * nF);
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_pts, sizeof(double) * soa.size());
    if (e == hipSuccess) e = hipMemcpy(s->d_chan, s->t.chan.data(), sizeof(Channel) * nF, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(s->d_pts, soa.data(), sizeof(double) * soa.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        gbp_fdem_system_destroy(s);
        return fail(GBP_ERR_HIP, "system table upload failed: %s", hipGetErrorString(e));
    }
    *out = s;
    return GBP_OK;
}

void gbp_fdem_system_destroy(gbp_fdem_system* sys)
{
    if (!sys) return;
    if (sys->d_chan) (void)hipFree(sys->d_chan);
    if (sys->d_pts) (void)hipFree(sys->d_pts);
    if (sys->d_bins) (void)hipFree(sys->d_bins);
    if (sys->d_bin_chan) (void)hipFree(sys->d_bin_chan);
    if (sys->d_bin_pts) (void)hipFree(sys->d_bin_pts);
    delete sys;
}

gbp_status gbp_fdem_system_npoints(const gbp_fdem_system* sys, int* npts)
{
    if (!sys || !npts) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    *npts = sys->t.npts;
    return GBP_OK;
}

gbp_status gbp_fdem_system_nfreq(const gbp_fdem_system* sys, int* nF)
{
    if (!sys || !nF) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    *nF = sys->t.nF;
    return GBP_OK;
}

gbp_status gbp_fdem_system_h0(const gbp_fdem_system* sys, double* out)
{
    if (!sys || !out) return fail(GBP_ERR_INVALID_ARG, "NULL argument%s");
    std::memcpy(out, sys->t.h0.data(), sizeof(double) * sys->t.h0.size());
    return GBP_OK;
}

gbp_status gbp_fdem_forward(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                            const double* sigma, const double* thk, const double* height, double* pred,
                            void* stream)
{
    return gbp_fdem_forward_ex(sys, B, Lmax, nlayers, sigma, thk, height, pred, 0, stream);
}

gbp_status gbp_fdem_forward_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                               const double* sigma, const double* thk, const double* height, double* pred,
                               int waves, void* stream)
{
    return gbp_fdem_forward_rows_ex(sys, B, Lmax, nlayers, sigma, thk, height, pred, nullptr, waves, stream);
}

gbp_status gbp_fdem_forward_rows_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                    const double* sigma, const double* thk, const double* height, double* pred,
                                    const int32_t* set_of_row, int waves, void* stream)
{
    return gbp_fdem_forward_rows_scaled(sys, B, Lmax, nlayers, sigma, thk, height, pred, set_of_row, nullptr, waves, stream);
}

gbp_status gbp_fdem_forward_rows_scaled(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                        const double* sigma, const double* thk, const double* height, double* pred,
                                        const int32_t* set_of_row, const double* row_scale, int waves, void* stream)
{
    gbp_status st = check_batch(sys, B, Lmax, nlayers, sigma, thk, height);
    if (st != GBP_OK) return st;
    if (waves < 0 || waves > 16) return fail(GBP_ERR_INVALID_ARG, "waves must be in [0, 16]%s");
    if (B == 0) return GBP_OK;
    if (!pred) return fail(GBP_ERR_INVALID_ARG, "pred is NULL%s");
    if ((st = check_row_sets(sys, set_of_row)) != GBP_OK) return st;
    const int nw = pick_waves(B, sys->t.nF, Lmax, (sys->t.npts + 63) / 64, waves);
    auto kernel = row_scale != nullptr ? k_fdem_forward<false, true> : k_fdem_forward<false, false>;
    hipLaunchKernelGGL(kernel, dim3(B), dim3(64 * nw), dyn_lds_bytes(nw, Lmax, (sys->t.npts + 63) / 64), (hipStream_t)stream, sys->d_chan,
                       sys->d_pts, sys->t.npts, sys->t.nF, Lmax, nlayers, sigma, thk, height, nullptr, nullptr,
                       nullptr, pred, nullptr, nullptr, sys->sigma_direct, sys->d_bins, sys->bin0, sys->n_bins, sys->d_bin_chan, sys->d_bin_pts,
                       set_of_row, row_scale);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_gauss_loglike(int B, int N, const double* pred, const double* obs, const double* rel,
                             const double* add, double* chi2, double* logL, void* stream)
{
    if (B < 0 || N < 1) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0 and N >= 1%s");
    if (B == 0) return GBP_OK;
    if (!pred || !obs || !rel || !add || !chi2 || !logL) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    const int wpb = 4;
    hipLaunchKernelGGL(k_gauss_loglike, dim3((B + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, B, N,
                       pred, obs, rel, add, chi2, logL);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_gauss_loglike_std(int B, int N, const double* pred, const double* obs, const double* sd, double* chi2,
                                 double* logL, void* stream)
{
    if (B < 0 || N < 1) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0 and N >= 1%s");
    if (B == 0) return GBP_OK;
    if (!pred || !obs || !sd || !chi2 || !logL) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    const int wpb = 4;
    hipLaunchKernelGGL(k_gauss_loglike_std, dim3((B + wpb - 1) / wpb), dim3(64 * wpb), 0, (hipStream_t)stream, B, N,
                       pred, obs, sd, chi2, logL);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_fdem_forward_loglike(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                    const double* sigma, const double* thk, const double* height,
                                    const double* obs, const double* rel, const double* add, double* pred,
                                    double* chi2, double* logL, void* stream)
{
    return gbp_fdem_forward_loglike_ex(sys, B, Lmax, nlayers, sigma, thk, height, obs, rel, add, pred, chi2, logL, 0, stream);
}

gbp_status gbp_fdem_forward_loglike_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                       const double* sigma, const double* thk, const double* height,
                                       const double* obs, const double* rel, const double* add, double* pred,
                                       double* chi2, double* logL, int waves, void* stream)
{
    gbp_status st = check_batch(sys, B, Lmax, nlayers, sigma, thk, height);
    if (st != GBP_OK) return st;
    if (waves < 0 || waves > 16) return fail(GBP_ERR_INVALID_ARG, "waves must be in [0, 16]%s");
    if (B == 0) return GBP_OK;
    if (!obs || !rel || !add || !chi2 || !logL) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    const int nw = pick_waves(B, sys->t.nF, Lmax, (sys->t.npts + 63) / 64, waves);
    hipLaunchKernelGGL(k_fdem_forward<true>, dim3(B), dim3(64 * nw), dyn_lds_bytes(nw, Lmax, (sys->t.npts + 63) / 64), (hipStream_t)stream, sys->d_chan,
                       sys->d_pts, sys->t.npts, sys->t.nF, Lmax, nlayers, sigma, thk, height, obs, rel, add, pred,
                       chi2, logL, sys->sigma_direct, sys->d_bins, sys->bin0, sys->n_bins, sys->d_bin_chan, sys->d_bin_pts, nullptr, nullptr);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_fdem_validate(int B, int Lmax, int N, const int32_t* nlayers, const double* sigma, const double* thk,
                             const double* height, const double* pred, int32_t* status, void* stream)
{
    if (B < 0 || Lmax < 1 || N < 0) return fail(GBP_ERR_INVALID_ARG, "B must be >= 0, Lmax >= 1, N >= 0%s");
    if (B == 0) return GBP_OK;
    if (!nlayers || !sigma || !thk || !height || !status) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    hipLaunchKernelGGL(k_fdem_validate, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, Lmax, N, nlayers, sigma,
                       thk, height, pred, status);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_bench_time_forward_loglike(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                         const double* sigma, const double* thk, const double* height,
                                         const double* obs, const double* rel, const double* add, double* pred,
                                         double* chi2, double* logL, void* stream, int reps, float* avg_ms)
{
    if (!avg_ms || reps < 1) return fail(GBP_ERR_INVALID_ARG, "avg_ms NULL or reps < 1%s");
    hipEvent_t e0, e1;
    GBP_HIP(hipEventCreate(&e0));
    GBP_HIP(hipEventCreate(&e1));
    GBP_HIP(hipEventRecord(e0, (hipStream_t)stream));
    gbp_status st = GBP_OK;
    for (int i = 0; i < reps && st == GBP_OK; ++i)
        st = gbp_fdem_forward_loglike(sys, B, Lmax, nlayers, sigma, thk, height, obs, rel, add, pred, chi2, logL,
                                      stream);
    GBP_HIP(hipEventRecord(e1, (hipStream_t)stream));
    GBP_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    GBP_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / reps;
    return st;
}

gbp_status gbp_debug_math(int op, int n, const double* x, const double* y, double* out0, double* out1, void* stream)
{
    if (n < 0 || op < 0 || op > 8) return fail(GBP_ERR_INVALID_ARG, "bad op or n%s");
    if (n == 0) return GBP_OK;
    if (!x || !out0) return fail(GBP_ERR_INVALID_ARG, "NULL device pointer%s");
    hipLaunchKernelGGL(k_debug_math, dim3(1024), dim3(256), 0, (hipStream_t)stream, op, n, x, y, out0, out1);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

gbp_status gbp_fdem_sensitivity(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                const double* sigma, const double* thk, const double* height, double* J,
                                void* stream)
{
    return gbp_fdem_sensitivity_ex(sys, B, Lmax, nlayers, sigma, thk, height, J, Lmax, 0, stream);
}

gbp_status gbp_fdem_sensitivity_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers,
                                   const double* sigma, const double* thk, const double* height, double* J,
                                   int max_layers, int exact, void* stream)
{
    return gbp_fdem_fm_dlogc(sys, B, Lmax, nlayers, sigma, thk, height, nullptr, J, max_layers, exact, stream);
}

gbp_status gbp_fdem_fm_dlogc(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                             const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                             void* stream)
{
    return gbp_fdem_fm_dlogc_ex(sys, B, Lmax, nlayers, sigma, thk, height, pred, J, max_layers, exact, 0, stream);
}

gbp_status gbp_fdem_fm_dlogc_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                                const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                                int waves, void* stream)
{
    return fm_dlogc_launch(sys, B, Lmax, nlayers, sigma, thk, height, pred, J, max_layers, exact, waves, 0, nullptr, stream);
}

gbp_status gbp_fdem_fm_dlogc_rows_ex(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                                     const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                                     const int32_t* set_of_row, int waves, void* stream)
{
    return fm_dlogc_launch(sys, B, Lmax, nlayers, sigma, thk, height, pred, J, max_layers, exact, waves, 0, set_of_row, stream);
}

gbp_status gbp_fdem_fm_dlogc_rows_scaled(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                                         const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                                         const int32_t* set_of_row, const double* row_scale, int waves, void* stream)
{
    return fm_dlogc_launch(sys, B, Lmax, nlayers, sigma, thk, height, pred, J, max_layers, exact, waves, 0, set_of_row, stream, row_scale);
}

}  // extern "C"

// compact_rows != 0 (the sampler's launches): only the columns up to the layer count rounded up to 8 are written
static gbp_status fm_dlogc_launch(const gbp_fdem_system* sys, int B, int Lmax, const int32_t* nlayers, const double* sigma,
                                  const double* thk, const double* height, double* pred, double* J, int max_layers, int exact,
                                  int waves, int compact_rows, const int32_t* set_of_row, void* stream, const double* row_scale)
{
    gbp_status st = check_batch(sys, B, Lmax, nlayers, sigma, thk, height);
    if (st != GBP_OK) return st;
    if (B == 0) return GBP_OK;
    if (!J) return fail(GBP_ERR_INVALID_ARG, "J is NULL%s");
    if ((st = check_row_sets(sys, set_of_row)) != GBP_OK) return st;
    if (max_layers < 1 || max_layers > Lmax) max_layers = Lmax;
    const size_t per_wave = (size_t)max_layers * (GBP_SENS_STRIDE * sizeof(cplx) + sizeof(gbp::LayerK));
    if (per_wave + (size_t)max_layers * 8 > 150000)
        return fail(GBP_ERR_INVALID_ARG, "too many layers for the Jacobian kernel's LDS working set (max ~140)%s");
    // one frequency per wave at a time, so the result does not depend on nw and nw should divide nF; `waves` is a
    // performance hint only (the sampler's launches, where only a fraction of the workgroups has work)
    if (waves < 0 || waves > 16) return fail(GBP_ERR_INVALID_ARG, "waves must be in [0, 16]%s");
    int nw = pick_waves(B, sys->t.nF, Lmax, sys->t.nF, waves);
    if (nw > sys->t.nF) nw = sys->t.nF;
    // LDS per workgroup: 60 KB keeps two workgroups per CU resident; small launches -- where a workgroup per CU is all there is, and a
    // deep model on one wave is a long tail (22 frequencies x 30 layers in sequence) -- may take most of a CU's 160 KB
    const size_t lds_cap = B <= 4096 ? 150000 : 60000;
    while (nw > 1 && nw * per_wave + (size_t)max_layers * 8 > lds_cap) --nw;
    const size_t lds = nw * per_wave + (size_t)max_layers * 8;
    auto launch = [&](auto kernel) -> gbp_status {
        if (lds > 48 * 1024) GBP_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(B), dim3(64 * nw), lds, (hipStream_t)stream, sys->d_chan, sys->d_pts, sys->t.npts, sys->t.nF,
                           Lmax, max_layers, nlayers, sigma, thk, height, J, pred, sys->d_bins, sys->bin0, sys->n_bins, sys->d_bin_chan,
                           sys->d_bin_pts, compact_rows, set_of_row, row_scale);
        return GBP_OK;
    };
    // launches capped at 8 (the sampler's common case) / 16 layers use variants with one / two row groups: fewer VGPRs.  Deeper launches
    // sum FOUR row groups (32 layers) per evaluation -- 121 / 119 VGPRs, no scratch; the eight-group variant of rounds 1 - 5 spilled 11 - 13
    // registers (48 - 56 B per lane) under the 128-VGPR budget -- and a model of 33 or more layers takes a second evaluation (sens_body's m0
    // loop): the rows are the same sums in the same order, the same bits (tests/test_gpu_parity.py: the 30-layer fixtures)
    const int ng = max_layers <= 8 ? 1 : (max_layers <= 16 ? 2 : 4);
    if (exact) st = ng == 1 ? launch(k_fdem_sens<true, 1>) : (ng == 2 ? launch(k_fdem_sens<true, 2>) : launch(k_fdem_sens<true, 4>));
    else st = ng == 1 ? launch(k_fdem_sens<false, 1>) : (ng == 2 ? launch(k_fdem_sens<false, 2>) : launch(k_fdem_sens<false, 4>));
    if (st != GBP_OK) return st;
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

#include "gbp_rjmcmc.h"
#include "gbp_tdem.h"
#include "gbp_hostpack.h"
#include "gbp_hitmap.h"

// Per-depth mean and 5 / 50 / 95 % points of log10 conductivity of B hit maps [B, nv, nz] (depth fastest) -> four [B, nz] arrays
extern "C" gbp_status gbp_hitmap_statistics(int B, int nv, int nz, const int32_t* hitmap, const double* log_mean_prior, double half_width,
                                            double* mean, double* p05, double* p50, double* p95, void* stream)
{
    if (B == 0) return GBP_OK;                     // (an empty block: empty device arrays have no address)
    if (B < 0 || nv < 1 || nz < 1 || !hitmap || !log_mean_prior || !mean || !p05 || !p50 || !p95)
        return fail(GBP_ERR_INVALID_ARG, "gbp_hitmap_statistics: NULL pointer or non-positive size%s");
    hipLaunchKernelGGL(hitmap::k_hitmap_stats, dim3(B, (nz + 255) / 256), dim3(256), 0, (hipStream_t)stream, nv, nz, hitmap, log_mean_prior, half_width,
                       mean, p05, p50, p95);
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

// The hit maps' rows (M = nv * nz cells each) as runs.  Call with start == NULL to COUNT (counts[B] <- runs per row), build the
// exclusive prefix ptr[B + 1] of the counts, allocate ptr[B] entries, then call again with ptr / start / value to WRITE.
extern "C" gbp_status gbp_hitmap_runs(int B, int64_t M, const int32_t* hitmap, int64_t* counts, const int64_t* ptr, int32_t* start,
                                      int32_t* value, void* stream)
{
    if (B == 0) return GBP_OK;
    if (B < 0 || M < 1 || M > 0x7fffffff || !hitmap) return fail(GBP_ERR_INVALID_ARG, "gbp_hitmap_runs: NULL pointer or size out of range%s");
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t is long long here");
    if (start == nullptr) {
        if (!counts) return fail(GBP_ERR_INVALID_ARG, "gbp_hitmap_runs: counts is NULL%s");
        hipLaunchKernelGGL(hitmap::k_hitmap_runs<false>, dim3(B), dim3(256), 0, (hipStream_t)stream, (long long)M, hitmap, (long long*)counts,
                           (const long long*)nullptr, (int*)nullptr, (int*)nullptr);
    } else {
        if (!ptr || !value) return fail(GBP_ERR_INVALID_ARG, "gbp_hitmap_runs: ptr / value is NULL%s");
        hipLaunchKernelGGL(hitmap::k_hitmap_runs<true>, dim3(B), dim3(256), 0, (hipStream_t)stream, (long long)M, hitmap, (long long*)nullptr,
                           (const long long*)ptr, start, value);
    }
    GBP_HIP(hipGetLastError());
    return GBP_OK;
}

// [host] rows of a hit map held as runs -> one zlib stream per row (see gbp_hostpack.h); no device work
extern "C" gbp_status gbp_runs_to_zlib(int n_rows, int64_t cells_per_row, const int64_t* ptr, const int32_t* start, const int32_t* value,
                                       uint8_t* out, int64_t out_capacity, int64_t* out_ptr)
{
    if (n_rows < 0 || cells_per_row < 1 || !ptr || !out || !out_ptr || out_capacity < 0 || (n_rows > 0 && (!start || !value)))
        return fail(GBP_ERR_INVALID_ARG, "gbp_runs_to_zlib: NULL pointer or non-positive size%s");
    int64_t pos = 0;
    out_ptr[0] = 0;
    for (int r = 0; r < n_rows; ++r) {
        const int64_t a = ptr[r], b = ptr[r + 1];
        if (b <= a || start[a] != 0) return fail(GBP_ERR_INVALID_ARG, "gbp_runs_to_zlib: every row needs a first run starting at cell 0%s");
        for (int64_t q = a + 1; q < b; ++q)
            if (start[q] <= start[q - 1] || start[q] >= cells_per_row) return fail(GBP_ERR_INVALID_ARG, "gbp_runs_to_zlib: run starts must increase within the row%s");
        const size_t n = hostpack::row_to_zlib(cells_per_row, b - a, start + a, value + a, out + pos, (size_t)(out_capacity - pos));
        if (n == 0) return fail(GBP_ERR_INVALID_ARG, "gbp_runs_to_zlib: output buffer too small%s");
        pos += (int64_t)n;
        out_ptr[r + 1] = pos;
    }
    return GBP_OK;
}
