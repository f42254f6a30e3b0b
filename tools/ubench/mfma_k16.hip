// Cycles per v_mfma_f32_16x16x16_f16 (K = 16, the CDNA3-era shape) vs v_mfma_f32_16x16x32_f16 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));
#define ITERS 2000
template <int MODE>
__global__ void k(float* out, long long* cyc) {
    halfx8 a8, b8; halfx4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f); b8[i] = (_Float16)0.25f; }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    floatx4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = floatx4{0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(a4), "v"(b4));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    floatx4 s = c[0];
    for (int i = 1; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 8);
    k<MODE><<<256, 256>>>(out, cyc);
    k<MODE><<<256, 256>>>(out, cyc);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %6.2f ticks per MFMA (one wave per SIMD)\n", name, (double)h / ITERS / 8);
}
int main() { run<0>("v_mfma_f32_16x16x32_f16"); run<1>("v_mfma_f32_16x16x16_f16"); }
