// kern_f32.hip - translation unit of lstm32::bilstm_f32_kernel (DM_PREC_F32): the kernel, its weight packer, its launch wrapper.
#include "kernels.h"
#include "lstm_f32.hip.inc"

static_assert(lstm32::TILE_M == dmk::TILE_M, "work item size");

namespace dmk {

// column of the TF kernel ([.., 400] = i|j|f|o blocks of 100) held by N-tile t, tile column c
inline int gate_col(int t, int c) {
    if (t < 24) return (t & 3) * 100 + 16 * (t >> 2) + c;
    return (c >> 2) * 100 + 96 + (c & 3);
}

Packed32 pack_weights_f32(const float* flat) {
    using namespace lstm32;
    Packed32 P;
    P.w.assign(size_t(2) * KS_DIR * KSTEP_F, 0.0f);
    P.b.assign(size_t(6) * 400, 0.0f);
    P.h.assign(size_t(2) * 25 * 64, 0.0f);
    const float* p = flat;
    for (int d = 0; d < 2; ++d) {
        int ks_base = 0;
        for (int l = 0; l < 3; ++l) {
            const int kin = l == 0 ? NFEAT : HID;
            const int ksin = l == 0 ? 2 : 25;
            const float* kern = p;
            p += size_t(kin + HID) * 400;
            const float* bias = p;
            p += 400;
            for (int ks = 0; ks < ksin + 25; ++ks) {
                float* dst = P.w.data() + size_t(d * KS_DIR + ks_base + ks) * KSTEP_F;
                for (int lane = 0; lane < 64; ++lane) {
                    const int sub = lane >> 4, c = lane & 15;
                    int krow;  // row of the TF kernel feeding this (k-step, sub-k); -1 = zero padding
                    if (ks < ksin) {
                        const int k = 4 * ks + sub;
                        krow = k < kin ? k : -1;
                    } else {
                        krow = kin + 4 * (ks - ksin) + sub;
                    }
                    for (int t = 0; t < NT; ++t) {
                        const float v = krow < 0 ? 0.0f : kern[size_t(krow) * 400 + gate_col(t, c)] * gate_scale(gate_col(t, c));
                        if (t < 24) dst[((t >> 2) * 64 + lane) * 4 + (t & 3)] = v;   // [tile quad][lane][4]
                        else dst[6 * 256 + lane] = v;                                // tile 24: [lane]
                    }
                }
            }
            float* bd = P.b.data() + size_t(d * 3 + l) * 400;
            for (int t = 0; t < NT; ++t)
                for (int c = 0; c < 16; ++c) {
                    const int gc = gate_col(t, c);
                    bd[t * 16 + c] = (bias[gc] + ((gc >= 200 && gc < 300) ? 1.0f : 0.0f)) * gate_scale(gc);  // forget_bias=1.0
                }
            ks_base += ksin + 25;
        }
    }
    const float* wout = p;  // [200][2]
    const float* bo = p + 400;
    for (int d = 0; d < 2; ++d)
        for (int kh = 0; kh < 25; ++kh)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = lane & 15, sub = lane >> 4;
                P.h[(size_t(d) * 25 + kh) * 64 + lane] = c < 2 ? wout[(d * HID + 4 * kh + sub) * 2 + c] : 0.0f;
            }
    P.bout[0] = bo[0];
    P.bout[1] = bo[1];
    return P;
}


hipError_t f32_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lstm32::bilstm_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lstm32::LDS_BYTES));
}
void f32_launch(const F32Args& a, int grid, hipStream_t stream) {
    lstm32::Params p;
    p.wpack = a.wpack; p.bpack = a.bpack; p.hpack = a.hpack;
    p.bout0 = a.bout0; p.bout1 = a.bout1;
    p.x = a.x; p.xstride = a.xstride; p.widx = a.widx; p.n = a.n;
    p.prob = a.prob; p.cls = a.cls; p.scratch = a.scratch; p.ntiles = a.ntiles; p.dbg = a.dbg; p.dir_split = a.dir_split; p.plogit = a.plogit;
    hipLaunchKernelGGL(lstm32::bilstm_f32_kernel, dim3(grid), dim3(lstm32::THREADS), lstm32::LDS_BYTES, stream, p);
}
size_t f32_scratch_floats_per_wg() { return lstm32::SCRATCH_FLOATS_PER_WG; }
int f32_waves() { return lstm32::WAVES; }

}  // namespace dmk
