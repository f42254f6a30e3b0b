// Microbenchmark (dev tool): do f16 MFMAs (v_mfma_f32_16x16x32_f16) and VALU/transcendental work overlap on a SIMD?
// modes as in mfma_valu_overlap.hip.  8 waves per CU, no barriers, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    floatx4 acc[25];
    for (int t = 0; t < 25; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
    float v[28];
    for (int i = 0; i < 28; ++i) v[i] = 0.001f * (threadIdx.x + i);
    halfx8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + i * 1e-3f); b[i] = (_Float16)0.5f; }
    const int bit = MODE == 3 ? (wave >> 2) & 1 : MODE == 4 ? (wave >> 1) & 1 : wave & 1;
    const bool do_mfma = MODE == 0 || MODE == 1 || (MODE >= 3 && bit == 0);
    const bool do_valu = MODE == 0 || MODE == 2 || (MODE >= 3 && bit == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int kk = 0; kk < 10; ++kk)
#pragma unroll
                for (int t = 0; t < 25; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[t], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int i = 0; i < 28; ++i) {
                float x = v[i];
                const float e1 = __builtin_amdgcn_exp2f(-1.44f * x), e2 = __builtin_amdgcn_exp2f(-1.3f * x);
                const float e3 = __builtin_amdgcn_exp2f(fminf(2.8f * x, 60.f)), e4 = __builtin_amdgcn_exp2f(-1.2f * x);
                const float s1 = __builtin_amdgcn_rcpf(1.f + e1);
                const float ij = (e3 - 1.f) * __builtin_amdgcn_rcpf((1.f + e2) * (1.f + e3));
                const float cn = fmaf(x, s1, ij);
                const float e5 = __builtin_amdgcn_exp2f(fminf(2.8f * cn, 60.f));
                v[i] = (e5 - 1.f) * __builtin_amdgcn_rcpf((1.f + e5) * (1.f + e4)) + 0.01f;
            }
        }
    }
    float r = 0.f;
    for (int t = 0; t < 25; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int i = 0; i < 28; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE>
float run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 2000;
    const float t0 = run<0>(d, iters), t1 = run<1>(d, iters), t2 = run<2>(d, iters), t3 = run<3>(d, iters), t4 = run<4>(d, iters), t5 = run<5>(d, iters);
    printf("f16 16x16x32: mode0 alternate MFMA+VALU in every wave : %.3f ms\n", t0);
    printf("mode1 MFMA only (250 per wave per iter)  : %.3f ms  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", t1, t1 * 1e-3 * 2.4e9 / (iters * 500.0));
    printf("mode2 VALU only                          : %.3f ms\n", t2);
    printf("mode3/4/5 MFMA waves next to VALU waves  : %.3f / %.3f / %.3f ms\n", t3, t4, t5);
    printf("mode0: additive %.3f ms, perfect overlap %.3f ms;  mode3: additive %.3f ms, perfect overlap %.3f ms\n", t1 + t2, t1 > t2 ? t1 : t2, 0.5f * (t1 + t2), 0.5f * (t1 > t2 ? t1 : t2));
    return 0;
}
