// Issue-rate microbenchmark for the LSTM cell epilogue on gfx950: cycles per wave-instruction of
// fp32 VALU (v_fma), packed fp32 (v_pk_fma_f32), transcendental (v_exp_f32 / v_rcp_f32), and mixes,
// with 1 and 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 2000
#define NCH 16   // independent chains

template <int MODE>
__global__ void k(float* out, long long* cyc, float seed) {
    float v[NCH];
    float2v pv[NCH / 2];
#pragma unroll
    for (int i = 0; i < NCH; ++i) v[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
#pragma unroll
    for (int i = 0; i < NCH / 2; ++i) pv[i] = float2v{v[2 * i], v[2 * i + 1]};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {          // 16 fma
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
        } else if (MODE == 1) {   // 8 pk_fma (= 16 fp32 fma)
#pragma unroll
            for (int i = 0; i < NCH / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[i]) : "v"(pv[(i + 1) & 7]));
        } else if (MODE == 2) {   // 16 exp
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        } else if (MODE == 3) {   // 16 rcp
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        } else if (MODE == 4) {   // interleaved 8 exp + 24 fma (independent)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[8 + i]) : "v"(seed));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[8 + ((i + 3) & 7)]) : "v"(seed));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[8 + ((i + 5) & 7)]) : "v"(seed));
            }
        } else if (MODE == 5) {   // 8 exp then 24 fma (blocked)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[8 + i]) : "v"(seed));
        } else if (MODE == 6) {   // dependent chain exp -> fma -> exp ... single chain
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
                asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[0]) : "v"(seed));
            }
        } else if (MODE == 7) {   // 16 cvt f32->f16 (v_cvt_f16_f32)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[i]));
        } else if (MODE == 8) {   // 16 v_accvgpr_write/read pairs
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_read_b32 %0, a0" : "+v"(v[i]) :: "a0");
        } else if (MODE == 9) {   // 8 exp + 12 pk_fma interleaved
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[4 + (i & 3)]) : "v"(pv[(i + 1) & 3]));
                if (i & 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[4 + ((i + 2) & 3)]) : "v"(pv[(i + 1) & 3]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < NCH / 2; ++i) s += pv[i].x + pv[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int n_instr, int threads) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, threads>>>(out, cyc, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, 0.5f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d: %6.2f counter ticks per wave-instr, wall %6.2f ns per wave-instr (x2.4 GHz = %5.2f clk); per SIMD %5.2f clk/instr\n", name, threads / 256,
           (double)h / ITERS / n_instr, ms * 1e6 / ITERS / n_instr, ms * 1e6 / ITERS / n_instr * 2.4, ms * 1e6 / ITERS / n_instr * 2.4 / (threads / 256));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32 x16 indep", 16, threads);
        run<1>("v_pk_fma_f32 x8 indep", 8, threads);
        run<2>("v_exp_f32 x16 indep", 16, threads);
        run<3>("v_rcp_f32 x16 indep", 16, threads);
        run<4>("8 exp + 24 fma interleaved", 32, threads);
        run<5>("8 exp then 24 fma blocked", 32, threads);
        run<6>("dependent exp->fma chain x8", 16, threads);
        run<7>("v_cvt_f16_f32 x16", 16, threads);
        run<8>("accvgpr write+read x16", 32, threads);
        run<9>("8 exp + 12 pk_fma interleaved", 20, threads);
    }
    return 0;
}
