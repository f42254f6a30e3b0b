// Microbenchmark (dev tool, round 3): what does one k16-step of the split product cost the matrix cores at the package power
// limit when the two cross terms are ONE v_mfma_i32_32x32x32_i8 instead of two v_mfma_f32_32x32x16_f16?
//   stream x3 : { f16, f16, f16 } per k16-step on one fp32 accumulator            (the product kernel)
//   stream i8 : { f16 on the fp32 accumulator, i8 on an int32 accumulator }        (tests/experiments/precision_schedule_experiment.py)
//   stream x2 : { f16, f16 }
//   stream i8only / f16only: bare streams of one kind
// Operands with real, varying bits (the matrix cores' power depends on them: profiles/r02/README.md).  One wave per SIMD,
// 256 workgroups.  Output: wall time per k16-step per wave, shader clock, cycles per k16-step.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int intx16 __attribute__((ext_vector_type(16)));
typedef int intx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    floatx16 acc;
    intx16 acci;
    for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acci[i] = 0; }
    halfx8 a[4], b[4];
    intx4 ai[4], bi[4];
    for (int j = 0; j < 4; ++j) {
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (_Float16)(0.37f * float(((lane * 7 + i * 13 + j * 29) % 61) - 30) / 30.f);
            b[j][i] = (_Float16)(0.41f * float(((lane * 11 + i * 5 + j * 3) % 53) - 26) / 26.f);
        }
        for (int i = 0; i < 4; ++i) {
            unsigned va = 0, vb = 0;
            for (int q = 0; q < 4; ++q) {
                va |= unsigned((((lane * 37 + i * 11 + j * 5 + q * 101) * 2654435761u) >> 13) & 0xFF) << (8 * q);
                vb |= unsigned((((lane * 53 + i * 7 + j * 3 + q * 59) * 2246822519u) >> 11) & 0xFF) << (8 * q);
            }
            ai[j][i] = int(va);
            bi[j][i] = int(vb);
        }
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            if constexpr (KIND == 0 || KIND == 2 || KIND == 3) {       // x3, x2, f16 only
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 3], b[(g * 3) & 3], acc, 0, 0, 0); FENCE();
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(g + 1) & 3], b[(g * 3) & 3], acc, 0, 0, 0); FENCE();
                if constexpr (KIND == 0) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 3], b[(g * 3 + 1) & 3], acc, 0, 0, 0); FENCE(); }
            } else if constexpr (KIND == 1) {                          // f16 hi.hi + one i8 cross product
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[g & 3], b[(g * 3) & 3], acc, 0, 0, 0); FENCE();
                acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[(g + 1) & 3], bi[(g * 3 + 1) & 3], acci, 0, 0, 0); FENCE();
            } else {                                                   // i8 only, two per k16-step
                acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[g & 3], bi[(g * 3) & 3], acci, 0, 0, 0); FENCE();
                acci = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai[(g + 1) & 3], bi[(g * 3 + 1) & 3], acci, 0, 0, 0); FENCE();
            }
        }
        if ((it & 63) == 63) {      // keep the accumulators bounded without changing the stream materially
            for (int i = 0; i < 16; ++i) { acc[i] *= 1e-3f; acci[i] >>= 8; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i] + float(acci[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
void run(const char* label, float* d, unsigned long long* dc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, dc, 2000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 0, 0, d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double steps = double(iters) * 12;
    printf("%-34s %7.2f ns per k16-step per wave, %6.1f cycles, %.2f GHz, wall %.2f ms\n", label, ms * 1e6 / steps, double(c) / steps,
           double(c) / (ms * 1e6), ms);
    fflush(stdout);
}

int main() {
    float* d;
    unsigned long long* dc;
    hipMalloc(&d, 256 * 256 * 4);
    hipMalloc(&dc, 64);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("x3: f16 f16 f16", d, dc);
        run<1>("i8: f16 + i8(32x32x32)", d, dc);
        run<2>("x2: f16 f16", d, dc);
        run<4>("i8 only: i8 i8", d, dc);
    }
    return 0;
}
