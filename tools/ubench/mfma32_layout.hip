// Dev check: operand / result lane layout of v_mfma_f32_32x32x16_f16 on gfx950 as the v2 classifier kernel assumes it.
//   A[i][k]: lane l holds i = l % 32, k = 8 * (l / 32) .. + 7      B[k][j]: lane l holds j = l % 32, k = 8 * (l / 32) .. + 7
//   D[i][j]: lane l, register r holds j = l % 32, i = 8 * (r / 4) + 4 * (l / 32) + (r % 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    halfx8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)A[(l % 32) * 16 + 8 * (l / 32) + e];
        b[e] = (_Float16)B[(8 * (l / 32) + e) * 32 + (l % 32)];
    }
    floatx16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + (r % 4)) * 32 + (l % 32)] = c[r];
}
int main() {
    float hA[32 * 16], hB[16 * 32], hD[32 * 32], ref[32 * 32];
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return float(int((s >> 16) % 17) - 8); };
    for (float& v : hA) v = rnd();
    for (float& v : hB) v = rnd();
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float acc = 0;
            for (int kk = 0; kk < 16; ++kk) acc += hA[i * 16 + kk] * hB[kk * 32 + j];
            ref[i * 32 + j] = acc;
        }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_f32_32x32x16_f16 layout check: %d mismatches of 1024 (%s)\n", bad, bad ? "LAYOUT ASSUMPTION WRONG" : "ok");
    return bad != 0;
}
