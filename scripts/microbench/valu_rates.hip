// Issue cost (cycles per wave64 instruction on one SIMD) of the VALU instructions the forward kernel is made of.
// hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip && ./valu_rates
// Method: each kernel runs a loop of 64 independent copies of ONE instruction (8 independent chains x 8 unrolled) for
// N iterations; with W waves per SIMD resident and every SIMD busy, cycles / instruction = elapsed_cycles * ... measured
// with s_memtime (100 MHz constant clock is too coarse) -> wall time via hipEvents and the measured shader clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 4096;

#define KERNEL(name, decl, body)                                                         \
    __global__ __launch_bounds__(256) void name(double* out, double seed, int n)         \
    {                                                                                    \
        decl;                                                                            \
        for (int it = 0; it < n; ++it) {                                                 \
            _Pragma("unroll") for (int r = 0; r < 4; ++r) { body; }                      \
        }                                                                                \
        double acc = 0.0;                                                                \
        for (int j = 0; j < 8; ++j) acc += (double)v[j];                                 \
        if (acc == 12345.678) out[threadIdx.x] = acc;                                    \
    }

#define D8 double v[8]; for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x * 1e-3; double c1 = seed * 0.999, c2 = seed * 1e-3
#define F8 float v[8]; for (int j = 0; j < 8; ++j) v[j] = (float)seed + j + threadIdx.x * 1e-3f; float c1 = (float)seed * 0.999f, c2 = (float)seed * 1e-3f
#define EACH(expr) _Pragma("unroll") for (int j = 0; j < 8; ++j) { expr; }

KERNEL(k_fma64, D8, EACH(v[j] = __builtin_fma(v[j], c1, c2)))
KERNEL(k_mul64, D8, EACH(v[j] = v[j] * c1))
KERNEL(k_add64, D8, EACH(v[j] = v[j] + c2))
KERNEL(k_rsq64, D8, EACH(v[j] = __builtin_amdgcn_rsq(v[j])))
KERNEL(k_rcp64, D8, EACH(v[j] = __builtin_amdgcn_rcp(v[j])))
KERNEL(k_sqrt64, D8, EACH(v[j] = __builtin_amdgcn_sqrt(v[j])))
KERNEL(k_ldexp64, D8, EACH(v[j] = __builtin_amdgcn_ldexp(v[j], 1) + c2))
KERNEL(k_max64, D8, EACH(v[j] = __builtin_fmax(v[j], c1)))
KERNEL(k_cvt_64_32_64, D8, EACH(v[j] = (double)(float)v[j] + c2))
KERNEL(k_rsq32_via_cvt, D8, EACH(v[j] = (double)__builtin_amdgcn_rsqf((float)v[j])))
KERNEL(k_fma32, F8, EACH(v[j] = __builtin_fmaf(v[j], c1, c2)))
KERNEL(k_rsq32, F8, EACH(v[j] = __builtin_amdgcn_rsqf(v[j])))
KERNEL(k_bperm, D8, EACH(v[j] = __shfl_xor(v[j], 1 + j, 64)))
__device__ __forceinline__ double dpp_shr1(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    unsigned lo = (unsigned)u;
    unsigned hi = (unsigned)(u >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
KERNEL(k_dpp_row_shr, D8, EACH(v[j] = dpp_shr1(v[j]) + c2))
KERNEL(k_cndmask64, D8, EACH(v[j] = v[j] > c1 ? v[j] : c2 + v[j] * 0.0))

// mixes: does a transcendental overlap with the plain VALU instructions issued behind it?
__global__ __launch_bounds__(256) void k_mix_rsq64_fma(double* out, double seed, int n)
{
    double v[8], w[32];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x * 1e-3;
    for (int j = 0; j < 32; ++j) w[j] = seed - j * 1e-2;
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = __builtin_amdgcn_rsq(v[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[4 * j + q] = __builtin_fma(w[4 * j + q], c1, c2);
        }
    }
    double acc = 0.0;
    for (int j = 0; j < 8; ++j) acc += v[j];
    for (int j = 0; j < 32; ++j) acc += w[j];
    if (acc == 12345.678) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_mix_rsq32cvt_fma(double* out, double seed, int n)
{
    double v[8], w[32];
    for (int j = 0; j < 8; ++j) v[j] = seed + j + threadIdx.x * 1e-3;
    for (int j = 0; j < 32; ++j) w[j] = seed - j * 1e-2;
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = (double)__builtin_amdgcn_rsqf((float)v[j]);
#pragma unroll
            for (int q = 0; q < 4; ++q) w[4 * j + q] = __builtin_fma(w[4 * j + q], c1, c2);
        }
    }
    double acc = 0.0;
    for (int j = 0; j < 8; ++j) acc += v[j];
    for (int j = 0; j < 32; ++j) acc += w[j];
    if (acc == 12345.678) out[threadIdx.x] = acc;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out;
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    struct K { const char* name; void (*fn)(double*, double, int); int per_iter; };
    std::vector<K> ks = {{"v_fma_f64", k_fma64, 32}, {"v_mul_f64", k_mul64, 32}, {"v_add_f64", k_add64, 32}, {"v_rsq_f64", k_rsq64, 32},
                         {"v_rcp_f64", k_rcp64, 32}, {"v_sqrt_f64", k_sqrt64, 32}, {"v_ldexp_f64 + v_add_f64", k_ldexp64, 32}, {"v_max_f64", k_max64, 32},
                         {"cvt f64->f32->f64 + v_add_f64 (3 instr)", k_cvt_64_32_64, 32}, {"cvt + v_rsq_f32 + cvt (3 instr)", k_rsq32_via_cvt, 32},
                         {"v_fma_f32", k_fma32, 32}, {"v_rsq_f32", k_rsq32, 32}, {"ds_bpermute_b32 x2 (64-bit xor shuffle)", k_bperm, 32},
                         {"v_mov_dpp x2 + v_add_f64", k_dpp_row_shr, 32}, {"cmp + cndmask x2 (+mul,add)", k_cndmask64, 32},
                         {"MIX per slot: (1 v_rsq_f64 + 4 v_fma_f64)/5", k_mix_rsq64_fma, 40},
                         {"MIX per slot: (cvt,v_rsq_f32,cvt + 4 v_fma_f64)/5", k_mix_rsq32cvt_fma, 40}};
    // clock: time the fma kernel at full occupancy and assume 4 cycles per v_fma_f64 (the documented fp64 rate) -> effective GHz
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = cus * waves_per_simd;         // 256 threads = 4 waves = one per SIMD
        double fma_ns = 0.0;
        printf("waves per SIMD: %d\n", waves_per_simd);
        for (auto& k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5, 64);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5, ITER);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double ns_per_instr = (double)best * 1e6 / ((double)ITER * k.per_iter * waves_per_simd);
            if (std::string(k.name) == "v_fma_f64") fma_ns = ns_per_instr;
            printf("  %-42s %7.3f ns per wave-instruction-slot  = %5.2f x v_fma_f64\n", k.name, ns_per_instr, ns_per_instr / fma_ns);
        }
    }
    return 0;
}
