// Microbenchmark (dev tool): how many VALU / transcendental / LDS-read "filler" instructions hide under the 16-bit MFMAs
// of ONE in-order wave on a gfx950 SIMD, for the two f16 shapes (32x32x16: 32 cycles, 16x16x32: 16 cycles), with the
// MFMAs chained on one accumulator (the hi.hi -> lo.hi -> hi.lo triple of the split-f16 kernel) or rotating over several.
// Round 1 concluded "MFMA and VALU time are additive on a SIMD" from block-alternating streams (mfma16_valu_overlap.hip);
// this one interleaves at instruction granularity, one or two waves per SIMD, operands with real (non-zero, varying) bits.
//
//   stream per group g:  CHAIN x { MFMA on acc[g % NACC] ; NV plain v_fma_f32 ; NT v_exp_f32 }  then NDS x ds_read_b128
// Output: shader cycles per MFMA (s_memtime of wave 0 of workgroup 0), wall ms, and the implied clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int SHAPE> struct Acc;
template <> struct Acc<32> { typedef floatx16 T; };
template <> struct Acc<16> { typedef floatx4 T; };
template <> struct Acc<3200> { typedef floatx16 T; };   // 32x32x16 bf16 (same operand bits reinterpreted): does the narrower multiplier draw less power?
template <> struct Acc<1600> { typedef floatx4 T; };    // 16x16x32 bf16
template <> struct Acc<1605> { typedef floatx4 T; };    // 16x16x32 f16, low 5 mantissa bits of every operand zero (does a shorter lo half draw less power?)
template <> struct Acc<1608> { typedef floatx4 T; };    // 16x16x32 f16, low 8 mantissa bits zero
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int SHAPE>
__device__ __forceinline__ typename Acc<SHAPE>::T mfma(halfx8 a, halfx8 b, typename Acc<SHAPE>::T c) {
    if constexpr (SHAPE == 32) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else if constexpr (SHAPE == 3200) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else if constexpr (SHAPE == 1600) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);       // 16, 1605, 1608
}

template <int SHAPE, int NACC, int CHAIN, int NV, int NT, int NDS, int GROUPS>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    typedef typename Acc<SHAPE>::T acc_t;
    const int lane = threadIdx.x & 63;
    acc_t acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < (SHAPE == 32 || SHAPE == 3200 ? 16 : 4); ++i) acc[t][i] = 0.f;
    halfx8 a[4], b[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            a[j][i] = (_Float16)(0.37f * float(((lane * 7 + i * 13 + j * 29) % 61) - 30) / 30.f);
            b[j][i] = (_Float16)(0.41f * float(((lane * 11 + i * 5 + j * 3) % 53) - 26) / 26.f);
        }
    if (SHAPE == 1605 || SHAPE == 1608) {
        const unsigned short msk = SHAPE == 1605 ? 0xFFE0 : 0xFF00;
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 8; ++i) {
                a[j][i] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, a[j][i]) & msk));
                b[j][i] = __builtin_bit_cast(_Float16, (unsigned short)(__builtin_bit_cast(unsigned short, b[j][i]) & msk));
            }
    }
    float v[8], e[4];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (threadIdx.x + i);
    for (int i = 0; i < 4; ++i) e[i] = 0.01f * (lane + i);
    // LDS filled with non-trivial bits for the ds_read fillers
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * i;
    __syncthreads();
    halfx8 r[2] = {a[0], a[1]};
    float k1 = 0.999f, k2 = 0.0007f;
    asm volatile("" : "+v"(k1), "+v"(k2));
    const unsigned laddr = (unsigned)(size_t)lds + lane * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (NDS) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "i"(NDS));   // the previous group's reads may still fly
#pragma unroll
            for (int c = 0; c < CHAIN; ++c) {
                acc[g % NACC] = mfma<SHAPE>(a[(g + c) & 3], b[(g * 3 + c) & 3], acc[g % NACC]);
                FENCE();
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(c * NV + i) & 7]) : "v"(k1), "v"(k2));   // asm: hipcc would SLP-pack builtins into v_pk_fma
                    FENCE();
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    asm volatile("v_exp_f32 %0, -%0" : "+v"(e[(c * NT + i) & 3]));   // x -> 2^-x: bounded
                    FENCE();
                }
            }
#pragma unroll
            for (int i = 0; i < NDS; ++i) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(r[i & 1]) : "v"(laddr), "i"(((g * NDS + i) & 15) * 1024));
                FENCE();
            }
        }
        if (NDS) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int i = 0; i < (SHAPE == 32 || SHAPE == 3200 ? 16 : 4); ++i) s += acc[t][i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += e[i];
    s += (float)r[0][0] + (float)r[1][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int SHAPE, int NACC, int CHAIN, int NV, int NT, int NDS>
void run(const char* label, int waves, float* d, unsigned long long* dc) {
    constexpr int GROUPS = 12;
    const int iters = 4000;
    auto kern = k<SHAPE, NACC, CHAIN, NV, NT, NDS, GROUPS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 100 * 1024, 0, d, dc, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 100 * 1024, 0, d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double nmfma = double(iters) * GROUPS * CHAIN;                    // per wave
    const double per_simd = nmfma * (waves / 4);                            // MFMAs issued on one SIMD
    const double flop = 16384.0 * (SHAPE == 32 || SHAPE == 3200 ? 2 : 1) * nmfma * waves * 256;
    printf("%-64s waves/SIMD %d: %6.1f cyc per MFMA per SIMD (%5.1f%% pipe), wall %7.3f ms, %.2f GHz, %6.0f TF/s\n", label, waves / 4,
           double(c) / per_simd, 100.0 * (SHAPE == 32 || SHAPE == 3200 ? 32 : 16) * per_simd / double(c), ms, double(c) / (ms * 1e6), flop / (ms * 1e9));
    fflush(stdout);
}

int main() {
    float* d;
    unsigned long long* dc;
    hipMalloc(&d, 256 * 512 * 4);
    hipMalloc(&dc, 64);
#define R(SHAPE, NACC, CHAIN, NV, NT, NDS) \
    run<SHAPE, NACC, CHAIN, NV, NT, NDS>(#SHAPE " acc=" #NACC " chain=" #CHAIN " valu=" #NV " trans=" #NT " ds/group=" #NDS, 4, d, dc); \
    run<SHAPE, NACC, CHAIN, NV, NT, NDS>(#SHAPE " acc=" #NACC " chain=" #CHAIN " valu=" #NV " trans=" #NT " ds/group=" #NDS, 8, d, dc);
    // bare MFMA streams
    R(3200, 1, 3, 0, 0, 0) R(1600, 1, 3, 0, 0, 0) R(16, 1, 3, 0, 0, 0) R(1605, 1, 3, 0, 0, 0) R(1608, 1, 3, 0, 0, 0) R(16, 1, 3, 0, 0, 0) R(1605, 1, 3, 0, 0, 0) R(1608, 1, 3, 0, 0, 0)
    R(32, 1, 3, 0, 0, 0) R(32, 2, 3, 0, 0, 0) R(32, 3, 1, 0, 0, 0) R(16, 1, 3, 0, 0, 0) R(16, 3, 1, 0, 0, 0)
    // plain VALU fillers between chained MFMAs (same accumulator) and between rotating accumulators
    R(32, 1, 3, 1, 0, 0) R(32, 1, 3, 2, 0, 0) R(32, 1, 3, 3, 0, 0) R(32, 1, 3, 4, 0, 0) R(32, 1, 3, 5, 0, 0) R(32, 1, 3, 6, 0, 0)
    R(32, 3, 1, 2, 0, 0) R(32, 3, 1, 4, 0, 0) R(32, 3, 1, 6, 0, 0) R(32, 2, 3, 4, 0, 0)
    // transcendental fillers
    R(32, 1, 3, 0, 1, 0) R(32, 1, 3, 0, 2, 0) R(32, 1, 3, 0, 3, 0) R(32, 3, 1, 0, 2, 0)
    // the cell-update mix per triple: ~13 plain + ~8 transcendental per cell, 4 cells + split/pack per 39 MFMAs -> ~2 + 1 per MFMA
    R(32, 1, 3, 2, 1, 2) R(32, 1, 3, 3, 1, 2) R(32, 2, 3, 2, 1, 2) R(32, 2, 3, 3, 1, 2) R(32, 1, 3, 4, 1, 2)
    // the same work on the 16-cycle shape (half the flops per MFMA -> half the fillers per MFMA)
    R(16, 1, 3, 1, 0, 0) R(16, 1, 3, 2, 0, 0) R(16, 1, 3, 3, 0, 0) R(16, 3, 1, 2, 0, 0) R(16, 1, 3, 0, 1, 0) R(16, 1, 3, 1, 1, 1) R(16, 3, 1, 1, 1, 1)
    return 0;
}
