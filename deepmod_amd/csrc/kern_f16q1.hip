// kern_f16q1.hip - translation unit of lstm16q::bilstm_f16q_kernel<1> (DM_PREC_F16I8 on the 16x16 MFMA shape, round 5).
#include "kernels.h"
#include <utility>
#include "lstm_common.hip.inc"
#include "lstm_f16q.hip.inc"

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

hipError_t f16q1_prepare() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(lstm16q::bilstm_f16q_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lstm16q::LDS_BYTES_I8));
}
void f16q1_launch(const F16Args& a, int grid, hipStream_t stream) {
    lstmc::Params p;
    fill(p, a);
    hipLaunchKernelGGL(lstm16q::bilstm_f16q_kernel<1>, dim3(grid), dim3(lstm16q::THREADS), lstm16q::LDS_BYTES_I8, stream, p);
}

}  // namespace dmk
