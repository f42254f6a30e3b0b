// Does a chain of dependent v_mfma_f32_16x16x32_f16 (SrcC = previous vDst) issue back to back on gfx950?
// Patterns: (0) 3 in-place accumulations per tile, tiles rotate; (1) same chain but the compiler-style renamed
// destinations; (2) two tiles interleaved (every MFMA independent of its predecessor); (3) in-place, 1 tile only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
#define ITERS 1000
template <int MODE>
__global__ void k(float* out, long long* cyc) {
    halfx8 a0, a1, b0, b1;
    for (int i = 0; i < 8; ++i) { a0[i] = (_Float16)(threadIdx.x * 0.001f); a1[i] = (_Float16)0.5f; b0[i] = (_Float16)0.25f; b1[i] = (_Float16)0.125f; }
    floatx4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = floatx4{0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %3, %2, %0\n v_mfma_f32_16x16x32_f16 %0, %1, %4, %0"
                             : "+v"(c[t]) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                floatx4 u, w;
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %3, %4, %2\n v_mfma_f32_16x16x32_f16 %1, %5, %4, %0\n v_mfma_f32_16x16x32_f16 %2, %3, %6, %1"
                             : "=&v"(u), "=&v"(w), "+v"(c[t]) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %3, %1\n"
                             "v_mfma_f32_16x16x32_f16 %0, %4, %3, %0\n v_mfma_f32_16x16x32_f16 %1, %4, %3, %1\n"
                             "v_mfma_f32_16x16x32_f16 %0, %2, %5, %0\n v_mfma_f32_16x16x32_f16 %1, %2, %5, %1"
                             : "+v"(c[t]), "+v"(c[t + 1]) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
            }
        } else {
#pragma unroll
            for (int t = 0; t < 24; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[0]) : "v"(a0), "v"(b0));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    floatx4 s = c[0];
    for (int i = 1; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, cyc);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-46s waves/SIMD %d: %6.2f ticks per MFMA per wave; per SIMD %5.2f ticks/MFMA (wall: %5.2f clk @2.4GHz)\n", name, threads / 256,
           (double)h / ITERS / 24, (double)h / ITERS / 24 / (threads / 256), ms * 1e6 / ITERS / 24 * 2.4 / (threads / 256));
}
int main() {
    for (int threads : {256, 512}) {
        run<0>("3 in-place dependent MFMAs per tile", threads);
        run<1>("3 dependent MFMAs, renamed destinations", threads);
        run<2>("two tiles interleaved (independent neighbours)", threads);
        run<3>("24 dependent MFMAs on one accumulator", threads);
    }
}
