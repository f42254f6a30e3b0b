// Energy microbenchmark for the LSTM cell arithmetic on gfx950: the same number of VALU instructions per second as plain fp32
// (v_fma_f32 / v_mul_f32 / v_add_f32) and as packed fp32 (v_pk_fma_f32: two lanes' worth of work per instruction), one wave per SIMD
// on every CU, for a few seconds - socket power is sampled from outside (tools/power_trace.sh).  Does a packed instruction cost
// what a plain one costs (then packing the cell update halves its energy), or twice that (then it buys nothing)?
//   hipcc --offload-arch=gfx950 -O3 valu_energy.hip -o valu_energy;  valu_energy <mode 0..4> [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float float2v __attribute__((ext_vector_type(2)));
#define ITERS 20000
#define NCH 16

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float v[NCH];
    float2v pv[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        v[i] = seed + i * 0.01f + threadIdx.x * 1e-4f;
        pv[i] = float2v{v[i], v[i] * 0.5f};
    }
    const float2v ps = float2v{seed, seed * 0.999f};
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {          // 16 plain fma
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
        } else if (MODE == 1) {   // 16 packed fma (twice the arithmetic of mode 0 in as many instructions)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[i]) : "v"(ps));
        } else if (MODE == 2) {   // 8 packed fma + 8 s_nop-free slots: the arithmetic of mode 0 in half the instructions
#pragma unroll
            for (int i = 0; i < NCH / 2; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(pv[i]) : "v"(ps));
        } else if (MODE == 3) {   // 8 mul + 8 add (the cell's other plain ops)
#pragma unroll
            for (int i = 0; i < NCH; i += 2) {
                asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i + 1]) : "v"(seed));
            }
        } else {                  // 16 transcendentals (8 exp, 8 rcp)
#pragma unroll
            for (int i = 0; i < NCH; i += 2) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
                asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i + 1]));
            }
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc += v[i] + pv[i][0] + pv[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double seconds = argc > 2 ? atof(argv[2]) : 4.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount;       // one workgroup of 4 waves per CU: one wave per SIMD, like the classifier
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f); break;
            case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f); break;
            default: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f); break;
        }
    };
    launch();
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 20; ++i) launch();
        hipDeviceSynchronize();
        launches += 20;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double instr = double(launches) * ITERS * (mode == 2 ? NCH / 2 : NCH) * blocks * 4;     // wave-instructions
    printf("mode %d: %lld launches in %.3f s, %.3e VALU wave-instructions/s (%d CUs x 4 waves), %.2f cycles per instruction per wave at 2.4 GHz\n", mode, launches, el,
           instr / el, blocks, 2.4e9 * el / (instr / (blocks * 4.0)));
    return 0;
}
