// Microbenchmark (dev tool, round 3): can global_load_lds_dwordx4 (LDS-DMA, destination base in M0) write LDS addresses at and
// above 64 KB on gfx950 (160 KB of LDS per workgroup)?  A four-slot 26 KB weight ring needs destinations up to 104 KB.
// One workgroup, one wave: DMA 1 KB of a known pattern to LDS byte address A for several A, read LDS back at A and at A mod 64 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out, unsigned a, int off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const unsigned lane = threadIdx.x;
    for (unsigned i = lane; i < 160 * 1024 / 4 - 256; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xDEAD0000u + (i & 0xFFFF);
    __syncthreads();
    const unsigned lds_base = (unsigned)(size_t)lds;
    const unsigned dst = lds_base + a;
    const unsigned lane16 = lane * 16;
    unsigned keep;
    if (off == 0)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "s"(dst), "v"(lane16), "s"(src) : "memory");
    else {      // the product kernel's form: M0 and the global base biased by +4096, the instruction offset -4096 moves both back
        const unsigned* g = src + 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3 offset:-4096\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                     : "=&s"(keep) : "s"(dst + 4096u), "v"(lane16), "s"(g) : "memory");
    }
    __syncthreads();
    out[lane] = reinterpret_cast<unsigned*>(lds + a)[lane * 4];                 // where it should be
    out[64 + lane] = reinterpret_cast<unsigned*>(lds + (a & 0xFFFF))[lane * 4];  // where a 16-bit base would put it
}
int main() {
    unsigned h[2048], *d, *o, r[128];
    for (int i = 0; i < 2048; ++i) h[i] = 0xABC00000u + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    const unsigned addrs[] = {4096, 53248, 65536 - 1024, 65536, 79872, 106496 - 1024, 131072, 150000 & ~15u};
    for (int off = 0; off < 2; ++off)
    for (unsigned a : addrs) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024 - 1024, 0, d, o, a, off);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int ok = 0, wrapped = 0;
        for (int l = 0; l < 64; ++l) { ok += r[l] == 0xABC00000u + 4 * l; wrapped += r[64 + l] == 0xABC00000u + 4 * l; }
        printf("%s LDS-DMA to byte address %6u: %2d/64 lanes landed there, %2d/64 at address mod 64 KB (%s)\n", off ? "offset:-4096" : "offset:0    ", a, ok, wrapped,
               ok == 64 ? "OK" : (wrapped == 64 ? "WRAPPED to 16 bits" : "elsewhere"));
    }
    return 0;
}
