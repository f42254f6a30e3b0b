// kern_f16q0.hip - translation unit of lstm16q::bilstm_f16q_kernel<0> (DM_PREC_F16X3, the default kernel) and of the weight packer of both
// instantiations (kern_f16q1.hip holds <1>: the two compile side by side).
#include "kernels.h"
#include <utility>
#include "lstm_common.hip.inc"
#include "lstm_f16q.hip.inc"

static_assert(lstm16q::TILE_M == dmk::TILE_M, "work item size");

namespace {
inline void fill(lstmc::Params& p, const dmk::F16Args& a) {
    p.wpack = a.wpack;
    p.wpack_i8 = a.wpack;
    p.hpack = a.hpack;
    p.bout0 = p.bout1 = 0.0f;          // (the head's bias is added by lstmhead::head_finish_kernel)
    p.x = a.x;
    p.xstride = a.xstride;
    p.widx = a.widx;
    p.n = a.n;
    p.ntiles = a.ntiles;
    p.plogit = a.plogit;
    p.len_scale = std::ldexp(1.0f, -a.len_shift);
    p.len_mul = std::ldexp(1.0f, a.len_shift);
    p.range_flag = a.range_flag;
    for (int k = 0; k < 24; ++k) p.i8s[k] = a.i8s ? a.i8s[k] : 0.0f;
}
}  // namespace

namespace dmk {

// 16x16x32 packing (lstm_f16q.hip.inc): [dir][layer][super-tile S][records in the kernel's processing order].
// A-operand lane l of a 16-row tile: row m = l % 16 -> unit 8S + 4 rh + m / 4, gate m % 4; k = (g = l / 16, j = 0..7) -> K slot of the B operand.
//   own t = 0..2 / input t = 0..2 (layers 1, 2), 4 KB each, [row half][hi|lo][lane][8 x f16]: slot j = own / input unit 8 (4t + j/2) + 4 (j%2) + g
//   mixed, 2 KB, [row half][lane][8 x f16] - the three products of the left-over slots side by side (round 5):
//       j = 0, 1, 2: (hi, lo, hi) of the weight of own unit 96 + g      against the B slots (h_hi, h_hi, h_lo)
//       j = 3, 4, 5: (hi, lo, hi) of the weight of input unit 96 + g (layer 0: of feature g) against (x_hi, x_hi, x_lo)
//       j = 6, 7   : g = 0: (hi, lo) of the bias row against (1, 1); else zero
//   layer 0 only, a second mixed record: j = 0, 1, 2, 3: (hi, lo, hi, lo) of the weight of feature 4 + g (g = 3: the event length x 2^len_shift)
//       against (x_hi, x_hi, x_lo, x_lo) - all FOUR products of the signal features (event lengths reach 10^4); j >= 4: zero
// int8 = true (DM_PREC_F16I8 on this shape, round 5): the second KB of a row half of an ORDINARY record holds, instead of the lo f16 halves, the int8
// cross-term weights of the same 32 K slots: bytes (2j, 2j + 1) of lane l = (w_hi8, w_lo8) of the unit of slot j - they meet the B bytes (lo8, hi8) of that
// unit in one v_mfma_i32_16x16x64_i8.  Scales per (direction, layer, gate kind) exactly as in pack_weights_tile below: sw = max(|w_hi|, 2^12 |w_lo|) over the
// rows that ride the int8 product (own / input units 0..95), i8s = sw 2^-12 / 127^2.  The mixed records keep their f16 form.
Packed16 pack_weights_f16q(const float* flat, const bool int8, float* i8s) {
    using namespace lstm16q;
    Packed16 P;
    P.w.assign(WEIGHT_BYTES, 0);
    P.len_shift = choose_len_shift(flat);
    const float len_mul = std::ldexp(1.0f, P.len_shift);
    const float* p = flat;
    auto split = [&](float v, _Float16& hi, _Float16& lo) {
        if (!std::isfinite(v)) P.finite = false;
        else P.max_abs = std::max(P.max_abs, std::fabs(v));
        hi = (_Float16)v;
        lo = (_Float16)(v - (float)hi);
    };
    for (int d = 0; d < 2; ++d) {
        size_t off = size_t(d) * WEIGHT_BYTES_DIR;
        for (int l = 0; l < 3; ++l) {
            const int kin = l == 0 ? NFEAT : HID;
            const float* kern = p;
            const float* bias = p + size_t(kin + HID) * 400;
            p += size_t(kin + HID) * 400 + 400;
            // weight of TF kernel row krow (-2: the bias row, -1: nothing) for gate column gc, exponent scale folded
            auto wval = [&](int krow, int gc, float mul) {
                if (krow >= 0) return kern[size_t(krow) * 400 + gc] * gate_scale(gc) * mul;
                if (krow == -2) return (bias[gc] + (gc >= 200 && gc < 300 ? 1.0f : 0.0f)) * gate_scale(gc);
                return 0.0f;
            };
            float sw[4] = {1.f, 1.f, 1.f, 1.f};
            if (int8) {
                for (int gk = 0; gk < 4; ++gk) {
                    float m = 0.0f;
                    for (int u = 0; u < HID; ++u)
                        for (int krow = (l == 0 ? kin : 0); krow < kin + 96; ++krow) {
                            if (l > 0 && krow >= 96 && krow < kin) continue;          // input units 96..99 ride the mixed record
                            const float v = wval(krow, gk * 100 + u, 1.0f);
                            const _Float16 hi = (_Float16)v;
                            const _Float16 lo = (_Float16)(v - (float)hi);
                            m = std::max(m, std::max(std::fabs((float)hi), 4096.0f * std::fabs((float)lo)));
                        }
                    sw[gk] = m > 0.0f ? m : 1.0f;
                    if (i8s) i8s[(d * 3 + l) * 4 + gk] = sw[gk] / (4096.0f * 127.0f * 127.0f);
                }
            }
            for (int S = 0; S < NTILE; ++S) {
                // ordinary records: own 0..2, then (layers 1, 2) input 0..2
                for (int rec = 0; rec < (l == 0 ? 3 : 6); ++rec) {
                    _Float16* dst = reinterpret_cast<_Float16*>(P.w.data() + off);
                    signed char* dst8 = reinterpret_cast<signed char*>(P.w.data() + off);
                    off += REC_BYTES;
                    const bool is_own = rec < 3;
                    const int t = is_own ? rec : rec - 3;
                    for (int rh = 0; rh < 2; ++rh)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int m = lane & 15, g = lane >> 4;
                            const int unit = 8 * S + 4 * rh + m / 4, gate = m % 4;
                            for (int j = 0; j < 8; ++j) {
                                _Float16 hi = (_Float16)0.0f, lo = (_Float16)0.0f;
                                if (unit < HID) split(wval((is_own ? kin : 0) + 8 * (4 * t + j / 2) + 4 * (j % 2) + g, gate * 100 + unit, 1.0f), hi, lo);
                                dst[((size_t(rh) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                                if (!int8) dst[((size_t(rh) * 2 + 1) * 64 + lane) * 8 + j] = lo;
                                else {
                                    const float s8 = 127.0f / sw[gate];
                                    const float qh = std::nearbyint((float)hi * s8), ql = std::nearbyint((float)lo * s8 * 4096.0f);
                                    signed char* d8 = dst8 + (size_t(rh) * 2 + 1) * 1024 + size_t(lane) * 16 + 2 * j;
                                    d8[0] = (signed char)std::max(-127.0f, std::min(127.0f, qh));
                                    d8[1] = (signed char)std::max(-127.0f, std::min(127.0f, ql));
                                }
                            }
                        }
                }
                // mixed record(s)
                for (int mrec = 0; mrec < (l == 0 ? 2 : 1); ++mrec) {
                    _Float16* dst = reinterpret_cast<_Float16*>(P.w.data() + off);
                    off += MIX_BYTES;
                    for (int rh = 0; rh < 2; ++rh)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int m = lane & 15, g = lane >> 4;
                            const int unit = 8 * S + 4 * rh + m / 4, gate = m % 4;
                            _Float16 slot[8];
                            for (int j = 0; j < 8; ++j) slot[j] = (_Float16)0.0f;
                            if (unit < HID) {
                                const int gc = gate * 100 + unit;
                                _Float16 hi, lo;
                                if (mrec == 0) {
                                    split(wval(kin + 96 + g, gc, 1.0f), hi, lo);
                                    slot[0] = hi; slot[1] = lo; slot[2] = hi;
                                    split(wval(l == 0 ? g : 96 + g, gc, 1.0f), hi, lo);
                                    slot[3] = hi; slot[4] = lo; slot[5] = hi;
                                    if (g == 0) {
                                        split(wval(-2, gc, 1.0f), hi, lo);
                                        slot[6] = hi; slot[7] = lo;
                                    }
                                } else {
                                    split(g < 3 ? wval(4 + g, gc, 1.0f) : wval(NFEAT - 1, gc, len_mul), hi, lo);
                                    slot[0] = hi; slot[1] = lo; slot[2] = hi; slot[3] = lo;      // j = 3: x_lo w_lo, the fourth product (free slot)
                                }
                            }
                            for (int j = 0; j < 8; ++j) dst[(size_t(rh) * 64 + lane) * 8 + j] = slot[j];
                        }
                }
            }
        }
    }
    return P;
}


hipError_t f16q_prepare(int mm) {
    if (mm == 1) return f16q1_prepare();
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16q::bilstm_f16q_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lstm16q::LDS_BYTES));
}
void f16q_launch(int mm, const F16Args& a, int grid, hipStream_t stream) {
    if (mm == 1) return f16q1_launch(a, grid, stream);
    lstmc::Params p;
    fill(p, a);
    hipLaunchKernelGGL(lstm16q::bilstm_f16q_kernel<0>, dim3(grid), dim3(lstm16q::THREADS), lstm16q::LDS_BYTES, stream, p);
}

}  // namespace dmk
