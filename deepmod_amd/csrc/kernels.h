// kernels.h - what the translation units of libdeepmod_hip.so share (the library is built from four of them in parallel - the classifier kernels
// compile for minutes each, the host code around them in seconds; round 6: the 32x32x16 kernels of rounds 2-3 are a fifth unit of EXPERIMENT builds only).  Each kernel family lives in ONE
// translation unit (kern_*.hip) with its weight packer and a launch wrapper; deepmod_hip.hip (C ABI, host runtime, the small kernels) sees
// only this header.  Host-side interface only: no kernel, no device type in here.
#pragma once
#include <hip/hip_runtime.h>

// ---- experiment switches (round 5: one umbrella) ------------------------------------------------------------------------------------------
// Every macro below turns a product kernel into a TIMING-ONLY or otherwise experimental build (tools/ablate.py).  None of them may reach
// the product by accident: without -DDM_EXPERIMENT any of them is a compile error, and dm_build_flags() reports what a library was built with
// (tests/test_build_guard.py checks the shipped library says "experiment=0").
#if defined(DM16Q_ABL_NOCELL) || defined(DM16Q_ABL_NODMA) || defined(DM16Q_ABL_MIX1) || defined(DM16Q_ABL_I8T) || defined(DM16Q_ABL_NOBAR) || defined(DM16Q_ABL_NOVMWAIT) || defined(DM16Q_AINIT) || defined(DM16Q_DEBUG_NOP) ||           \
    defined(DM16Q_NOCHUNK) || defined(DM16Q_NOPN) || defined(DM16Q_LOPACK) || defined(DM16Q_PRE) || defined(DM16Q_SNAKE) || defined(DM16Q_TRANS_COST) ||                     \
    defined(DM16S_ABL_2PROD) || defined(DM16S_ABL_B64) || defined(DM16S_ABL_LO_ONLY) || defined(DM16S_ABL_MFMA16) ||                              \
    defined(DM16S_ABL_MFMA16_PAD) || defined(DM16S_ABL_NOBAR) || defined(DM16S_ABL_NOCELL) || defined(DM16S_ABL_NODMA) ||                          \
    defined(DM16S_ABL_NOLDSA) || defined(DM16S_ADIST) || defined(DM16S_ALO_TRUNC) || defined(DM16S_PRE) || defined(DM_ABL_NOBAR) ||                \
    defined(DM_ABL_NODMA) || defined(DM_ABL_NOEPI) || defined(DM_ABL_NOSEQ) || defined(DM_TIMING) || defined(DM_TRACE) || defined(DM_TRACE2) ||   \
    defined(DM_WLO_TRUNC_ENV) || defined(DM_WLO_TRUNC_DEFAULT) || defined(DM_WITH_F16X3_ROLES) || defined(DM16R_DMA_M) ||                         \
    defined(DM_F16X3_SHAPE_DEFAULT) || defined(DM_WAVES) || defined(DM_MT) || defined(DM_WITH_F16S)
#define DM_ANY_EXPERIMENT_SWITCH 1
#ifndef DM_EXPERIMENT
#error "an ablation / experiment macro is defined without -DDM_EXPERIMENT: timing-only kernels must not be built into the product by a stray -D"
#endif
#else
#define DM_ANY_EXPERIMENT_SWITCH 0
#endif


#if defined(DM_WITH_F16X3_ROLES) && !defined(DM_WITH_F16S)
#error "the roles experiment lives in the translation unit of the 32x32x16 kernels: build it with -DDM_WITH_F16S as well"
#endif

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/deepmod_hip.h"

#ifndef DM_F16X3_SHAPE_DEFAULT
#define DM_F16X3_SHAPE_DEFAULT 16
#endif

namespace dmk {

constexpr int TILE_M = 128;       // windows per work item of every classifier kernel (each kernel file asserts its own value against it)

// exponent scale folded into kernel and bias columns (both kernels): the cell update takes 2^x of the accumulators as
// they are - i, f, o columns x -log2(e) (sigmoid = 1 / (1 + 2^a)), j column x 2 log2(e) (tanh = (2^a - 1) / (2^a + 1))
inline float gate_scale(int gc) { return (gc >= 100 && gc < 200) ? 2.8853900817779268f : -1.4426950408889634f; }


struct Packed32 {
    std::vector<float> w, b, h;
    float bout[2];
};

// split-f16 packing: [dir][stream position][tile][hi|lo][lane][8 x f16], k-steps in the kernel's stream order
struct Packed16 {
    std::vector<unsigned char> w;
    float max_abs = 0.0f;   // largest |packed value| (after the exponent-scale fold): must stay <= 65504 to be an f16
    bool finite = true;
    int len_shift = 0;      // the weight row of feature 6 (event length) is stored a second time x 2^len_shift (slot 7)
};

// len_shift: the largest k <= 10 for which (length row x exponent scale x 2^k) is still an f16: an event length beyond
// 65504 samples is then fed as v * 2^-k through slot 7 (exact power-of-two rescale; covers |v| <= 65504 * 2^k)
inline int choose_len_shift(const float* flat) {
    float m = 0.0f;
    const float* p = flat;
    for (int d = 0; d < 2; ++d) {
        const float* row = p + size_t(DM_NFEAT - 1) * 400;          // layer-0 kernel row of feature 6
        for (int gc = 0; gc < 400; ++gc) m = std::max(m, std::fabs(row[gc] * gate_scale(gc)));
        p += size_t(DM_NFEAT + DM_HIDDEN) * 400 + 400 + 2 * (size_t(2 * DM_HIDDEN) * 400 + 400);
    }
    int k = 10;
    while (k > 0 && !(m * std::ldexp(1.0f, k) <= 32768.0f)) --k;
    return k;
}



// arguments of one classifier launch on device-resident buffers, independent of the kernel family
struct F16Args {
    const unsigned char* wpack;   // the weight pack of the kernel that runs
    const float* hpack;           // head weights W[200][2] fp32
    const float* x;
    long long xstride;
    const int* widx;
    long long n;
    int ntiles;
    float* plogit;
    int len_shift;
    int* range_flag;
    const float* i8s;             // [24] fold scales of the int8 mode (null otherwise)
};
struct F32Args {
    const float *wpack, *bpack, *hpack;
    float bout0, bout1;
    const float* x;
    long long xstride;
    const int* widx;
    long long n;
    float* prob;
    unsigned char* cls;
    float* scratch;
    int ntiles;
    unsigned long long* dbg;
    int dir_split;
    float* plogit;
};

// kern_f32.hip: lstm32::bilstm_f32_kernel
Packed32 pack_weights_f32(const float* flat);
hipError_t f32_prepare();                                          // once per process: dynamic LDS size of the kernel
void f32_launch(const F32Args& a, int grid, hipStream_t stream);
size_t f32_scratch_floats_per_wg();
int f32_waves();
#ifdef DM_WITH_F16S
// tools/experiments/f16s/kern_f16s.hip (experiment builds only, round 6): lstm16s::bilstm_f16s_kernel<mm> - the 32x32x16 kernels of rounds 2-3
// (mm 0: three f16 products, 1: int8 cross terms, 2: the roles experiment when built)
Packed16 pack_weights_f16s(const float* flat, bool int8 = false, float* i8s = nullptr);
hipError_t f16s_prepare(int mm);
void f16s_launch(int mm, const F16Args& a, int grid, hipStream_t stream);
bool f16s_has_roles();
#endif
// kern_f16q0.hip / kern_f16q1.hip: lstm16q::bilstm_f16q_kernel<mm>
Packed16 pack_weights_f16q(const float* flat, bool int8 = false, float* i8s = nullptr);
hipError_t f16q_prepare(int mm);
void f16q_launch(int mm, const F16Args& a, int grid, hipStream_t stream);
hipError_t f16q1_prepare();
void f16q1_launch(const F16Args& a, int grid, hipStream_t stream);

}  // namespace dmk
