// Dev tool (GPU box): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for KNOWN byte counts on gfx950?  (bench.py applied the
// guide's x2 correction for 16-byte streaming loads to a kernel whose HBM reads are 4-byte dword loads of 28-byte feature rows.)
//   stream16: every lane reads 16 B (float4), perfectly coalesced, N bytes in all               -> the guide's case
//   rows4   : the classifier's access pattern: a wave's 64 lanes read ONE dword each from 16 windows x 4 features of a [n][21][7] fp32 array
//             (lane = (window n, feature g)), row after row of the 21, i.e. 4-byte loads, 588-byte window stride, every byte of the array touched once
//             across the features g and 4 + g
//   write4  : 8 bytes per window written (the logits)
// Run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/ubench/fetch_calib   and compare Counter_Value (KB) with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void stream16(const float4* __restrict__ x, long long n4, float* __restrict__ out) {
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void rows4(const float* __restrict__ x, long long nwin, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, n16 = lane & 15, grp = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    float acc = 0.f;
    for (long long w0 = wave * 16; w0 < nwin; w0 += nwaves * 16) {
        const float* xw = x + (w0 + n16) * 147;
        const int f1 = 4 + grp < 7 ? 4 + grp : 6;
        for (int row = 0; row < 21; ++row) acc += xw[row * 7 + grp] + xw[row * 7 + f1];
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ void write8(float2* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = float2{float(i), 1.0f};
}
int main() {
    const long long nwin = 1 << 20;                       // 1,048,576 windows x 588 B = 616,562,688 B
    const size_t bytes = size_t(nwin) * 147 * 4;
    float *x, *out;
    float2* o2;
    hipMalloc(&x, bytes); hipMalloc(&out, 64); hipMalloc(&o2, size_t(nwin) * 8);
    hipMemset(x, 1, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(2048), dim3(256), 0, 0, (const float4*)x, (long long)(bytes / 16), out);
        hipLaunchKernelGGL(rows4, dim3(2048), dim3(256), 0, 0, x, nwin, out);
        hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, 0, o2, nwin);
    }
    hipDeviceSynchronize();
    printf("stream16 reads %zu bytes (16 B per lane, coalesced)\nrows4 reads %zu bytes (4 B per lane, the classifier's feature-row pattern; feature 6 read by two lane groups: %zu B requested)\nwrite8 writes %lld bytes\n",
           bytes, bytes, size_t(nwin) * 21 * 8 * 4, nwin * 8);
    return 0;
}
