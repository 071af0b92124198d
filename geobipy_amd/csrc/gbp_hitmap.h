// gbp_hitmap.h -- what leaves the device of a block's conductivity-depth hit maps (int32 [B, n_value, n_depth], depth fastest,
// 440 KB per sounding): per-depth statistics for the survey summary, and the maps themselves in run-length form for the results
// containers.  Both are one streaming pass (two for the runs: count, then write) over the maps -- HBM-bound integer work; the torch
// formulation they replace (transpose to float64, cumsum, nonzero over 9e8 cells) took 0.1 s per 8 192 soundings, these take a few ms.
// No reference counterpart: the reference derives the same statistics from its Histogram2D posterior on the host, one sounding at a
// time (classes/statistics/Histogram.py: mean / percentile), and stores the maps dense.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace hitmap {

// mean[b, z] = sum_v h (c_v) / max(1, sum_v h) + shift_b,  c_v = ((v + 0.5) / nv) 2 hw - hw, shift_b = log_mean_prior[b] / ln 10;
// pq[b, z] = c_{idx} + shift_b with idx = #{v : cumsum_v / tot < q} clamped to nv - 1, q = 0.05, 0.5, 0.95.
// One workgroup per (sounding, 256 depth cells); thread z walks the value bins: the loads of a wave are 64 consecutive ints.
__global__ __launch_bounds__(256) void k_hitmap_stats(int nv, int nz, const int* __restrict__ hm, const double* __restrict__ log_mean_prior,
                                                       double half_width, double* __restrict__ mean, double* __restrict__ p05,
                                                       double* __restrict__ p50, double* __restrict__ p95)
{
    const int b = blockIdx.x, z = blockIdx.y * 256 + threadIdx.x;
    if (z >= nz) return;
    const int* col = hm + (size_t)b * nv * nz + z;
    const double shift = log_mean_prior[b] / 2.302585092994046;
    const double w = 2.0 * half_width;
    long long tot = 0;
    double wsum = 0.0;
#pragma unroll 10                                  // (ten loads in flight per wave: bound by memory-level parallelism before bytes)
    for (int v = 0; v < nv; ++v) {
        const int h = col[(size_t)v * nz];
        tot += h;
        wsum += (double)h * ((((double)v + 0.5) / (double)nv) * w - half_width);
    }
    const double t = (double)(tot > 1 ? tot : 1);
    long long cum = 0;
    int i05 = 0, i50 = 0, i95 = 0;
#pragma unroll 10
    for (int v = 0; v < nv; ++v) {
        cum += col[(size_t)v * nz];
        const double cdf = (double)cum / t;
        i05 += cdf < 0.05;
        i50 += cdf < 0.5;
        i95 += cdf < 0.95;
    }
    const size_t o = (size_t)b * nz + z;
    auto centre = [&](int i) { return (((double)(i < nv - 1 ? i : nv - 1) + 0.5) / (double)nv) * w - half_width; };
    mean[o] = wsum / t + shift;
    p05[o] = centre(i05) + shift;
    p50[o] = centre(i50) + shift;
    p95[o] = centre(i95) + shift;
}

// Runs of a row's flattened cells: a run starts at cell 0 and wherever the count differs from the cell before.
// Pass 1 (WRITE = false): counts[b] = number of runs.  Pass 2 (WRITE = true): start / value at ptr[b] + (rank of the run in the row).
// One workgroup per row walks it in tiles of 1024 cells (4 per thread, coalesced); the ranks inside a tile come from a wave ballot
// and a four-wave prefix in LDS.
template <bool WRITE>
__global__ __launch_bounds__(256) void k_hitmap_runs(long long M, const int* __restrict__ hm, long long* __restrict__ counts,
                                                      const long long* __restrict__ ptr, int* __restrict__ start, int* __restrict__ value)
{
    __shared__ int totals[2][4][4];               // [tile parity][sub-tile][wave]: one barrier per tile
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int* row = hm + (size_t)b * M;
    long long base = WRITE ? ptr[b] : 0;          // runs written (or counted) before this tile
    int parity = 0;
    for (long long t0 = 0; t0 < M; t0 += 1024, parity ^= 1) {
        int flags[4], vals[4], before[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {             // sub-tile q: cells t0 + 256 q + thread -- consecutive threads, consecutive cells
            const long long j = t0 + 256 * q + threadIdx.x;
            int f = 0, v = 0;
            if (j < M) {
                v = row[j];
                f = (j == 0) || (v != row[j - 1]);
            }
            flags[q] = f; vals[q] = v;
            const unsigned long long m = __ballot(f);
            before[q] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) totals[parity][q][wave] = __popcll(m);
        }
        __syncthreads();                          // (the other parity's slots are free again: every thread passed the barrier of the tile before)
        // order of the runs inside the tile: sub-tile major, then wave, then lane
        int rank0 = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int lower = 0, all = 0;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const int c = totals[parity][q][w2];
                lower += (w2 < wave) ? c : 0;
                all += c;
            }
            if (WRITE && flags[q]) {
                const long long r = base + rank0 + lower + before[q];
                start[r] = (int)(t0 + 256 * q + threadIdx.x);
                value[r] = vals[q];
            }
            rank0 += all;
        }
        base += rank0;
    }
    if (!WRITE && threadIdx.x == 0) counts[b] = base;
}

}  // namespace hitmap
