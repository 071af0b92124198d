// Micro-benchmark: issue rate of the fp64 VALU instructions the kernels use (gfx950).
// hipcc --offload-arch=gfx950 -O3 fp64_rates.hip -o fp64_rates && ./fp64_rates
// Measured on MI355X, in issue slots of one v_mul_f64 (4.6 cycles per wave-instruction): fma 1.1, add 1.0, rndne / ldexp /
// frexp_exp / max / cvt_i32 1.0 - 1.6, cvt f64<->f32 1.0 each, v_rsq_f64 / v_rcp_f64 / v_sqrt_f64 3.5 - 3.7 (quarter rate),
// and an fp32 seed (cvt + v_rsq_f32 + cvt) costs the same 3.7 -- so the two seeds of a complex square root are ~8 % of a layer
// of the forward kernel and cannot be had cheaper.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 2000
#define U 8
template <int OP>
__global__ void k(double* out, double seed)
{
    double v[U];
    for (int i = 0; i < U; ++i) v[i] = seed + threadIdx.x * 1e-3 + i;
    int acc = 0;
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (OP == 0) v[i] = __builtin_fma(v[i], 1.0000001, 1e-9);
            if (OP == 1) v[i] = __builtin_rint(v[i] * 1.0000001);            // mul + rndne
            if (OP == 2) { acc += (int)v[i]; v[i] = v[i] * 1.0000001; }      // mul + cvt_i32
            if (OP == 3) v[i] = __builtin_amdgcn_ldexp(v[i] * 1.0000001, it & 1 ? 1 : -1);   // mul + ldexp
            if (OP == 4) v[i] = __builtin_amdgcn_rsq(v[i] * 1.0000001);      // mul + rsq
            if (OP == 5) v[i] = __builtin_amdgcn_rcp(v[i] * 1.0000001);      // mul + rcp
            if (OP == 6) v[i] = v[i] * 1.0000001;                            // mul only
            if (OP == 7) { acc += __builtin_amdgcn_frexp_exp(v[i]); v[i] = v[i] * 1.0000001; }  // mul + frexp_exp
            if (OP == 8) v[i] = v[i] + 1.0000001;                            // add
            if (OP == 9) v[i] = __builtin_fmax(v[i] * 1.0000001, 0.5);       // mul + max
            if (OP == 10) v[i] = (double)__builtin_amdgcn_rsqf((float)(v[i] * 1.0000001));   // mul + cvt + rsq_f32 + cvt
            if (OP == 11) v[i] = (double)(float)(v[i] * 1.0000001);          // mul + cvt + cvt
            if (OP == 12) v[i] = __builtin_amdgcn_sqrt(v[i] * 1.0000001);    // mul + v_sqrt_f64
        }
    }
    double s = acc;
    for (int i = 0; i < U; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP>
float run(double* d)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, d, 1.5);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8 * 256 * 8);
    const char* names[] = {"fma", "mul+rndne", "mul+cvt_i32", "mul+ldexp", "mul+rsq", "mul+rcp", "mul", "mul+frexp_exp", "add", "mul+max", "mul+cvt+rsq32+cvt", "mul+cvt+cvt", "mul+sqrt64"};
    float t[13];
    t[0] = run<0>(d); t[1] = run<1>(d); t[2] = run<2>(d); t[3] = run<3>(d); t[4] = run<4>(d);
    t[10] = run<10>(d); t[11] = run<11>(d); t[12] = run<12>(d);
    t[5] = run<5>(d); t[6] = run<6>(d); t[7] = run<7>(d); t[8] = run<8>(d); t[9] = run<9>(d);
    for (int i = 0; i < 13; ++i) printf("%-14s %8.3f ms  (x%.2f of mul)\n", names[i], t[i], t[i] / t[6]);
    // waves: 2048 blocks x 4 waves; per wave N_IT*U ops
    double ops = 2048.0 * 4 * N_IT * U;
    printf("mul: %.2f cycles per wave-instruction per SIMD at 2.4 GHz (1024 SIMDs)\n", t[6] * 1e-3 * 2.4e9 * 1024 / ops);
    return 0;
}
