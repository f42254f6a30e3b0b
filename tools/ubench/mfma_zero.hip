// Is v_mfma_f32_16x16x32_f16 faster on zero operands (data-dependent timing) or does the chip just clock higher?
// Same instruction stream with B = 0, B = constant, B = lane/iteration-dependent values: cycle counter per MFMA and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
#define ITERS 20000
template <int MODE>
__global__ void k(float* out, long long* cyc) {
    halfx8 a8, b8;
    for (int i = 0; i < 8; ++i) {
        a8[i] = MODE == 0 ? (_Float16)0.0f : (_Float16)(0.01f * ((threadIdx.x * 7 + i * 13) % 61) - 0.3f);
        b8[i] = MODE == 0 ? (_Float16)0.0f : (MODE == 1 ? (_Float16)0.25f : (_Float16)(0.013f * ((threadIdx.x * 11 + i * 5 + blockIdx.x) % 53) - 0.33f));
    }
    floatx4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = floatx4{0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(a8), "v"(b8));
        if (MODE == 2) b8 = b8 + a8 * (_Float16)0.001f;   // keep the operand changing
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
    hipMalloc(&out, 256 * 512 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) k<MODE><<<256, 512>>>(out, cyc);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) k<MODE><<<256, 512>>>(out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double mfmas = 10.0 * ITERS * 8 * 2;   // per SIMD: 2 waves
    printf("%-34s %6.2f ticks per MFMA per SIMD, %7.3f ms per launch -> %5.0f TFLOP/s, implied clock %.2f GHz\n", name,
           (double)h / ITERS / 8 / 2, ms / 10, 10.0 * 256 * 8 * ITERS * 8.0 * 16 * 16 * 32 * 2 / (ms * 1e-3) / 1e12,
           ((double)h) / (ms / 10 * 1e-3) / 1e9);
}
int main() { run<0>("all operands zero"); run<1>("constant non-zero operands"); run<2>("varying non-zero operands"); run<0>("all operands zero (again)"); }
