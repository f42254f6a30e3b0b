// deepmod_hip.hip — libdeepmod_hip.so: C ABI (include/deepmod_hip.h) + host runtime for gfx950 + the small kernels (head, summary,
// cluster, calibration, signal).  The three classifier kernel families are separate translation units (kern_*.hip, interface: kernels.h);
// __graft_entry__.build() compiles the five in parallel (hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c) and links them.
#include "kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/deepmod_hip.h"
#include "head.hip.inc"

using dmk::Packed16;
using dmk::Packed32;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(DM_EDEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear sticky "invalid value" for plain host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// ---------------------------------------------------------------------------------------------
// feature rows of raw reads, assembled on the device (round 5; get_Feature, myDetect.py:839-903, fnum = 7): row q of the batch = one-hot of
// its reference base | mean, stdv, length of the event it shows - from the device form of rowsbatch.inc (ev3, code, rdesc).  HBM-bound:
// 13 B in, 28 B out per row; one thread per row, the read of a row found by binary search over the (few hundred) read descriptors.
// ---------------------------------------------------------------------------------------------
__global__ void rows_assemble_kernel(float* __restrict__ rows, const unsigned char* __restrict__ code, const float* __restrict__ ev3,
                                     const long long* __restrict__ rdesc, const int n_reads, const long long n_rows) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_rows) return;
    int lo = 0, hi = n_reads - 1;                    // the last read whose first row is <= q
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rdesc[4 * (long long)mid] <= q) lo = mid;
        else hi = mid - 1;
    }
    const long long* d = rdesc + 4 * (long long)lo;
    const long long e = q + d[1];
    const bool has = e >= d[2] && e < d[3];
    const unsigned c = code[q];
    float* o = rows + q * DM_NFEAT;
    o[0] = c == 0u ? 1.0f : 0.0f;
    o[1] = c == 1u ? 1.0f : 0.0f;
    o[2] = c == 2u ? 1.0f : 0.0f;
    o[3] = c == 3u ? 1.0f : 0.0f;
    o[4] = has ? ev3[3 * e] : 0.0f;
    o[5] = has ? ev3[3 * e + 1] : 0.0f;
    o[6] = has ? ev3[3 * e + 2] : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// summary kernel: dense int32 counters with a wavefront-level pre-reduction.
// Bases arrive read by read, so a wave mostly sees 64 consecutive distinct positions (one atomic per
// counter per base), but deep pile-ups (amplicons, the cov > 1000 case of the BED format) put long
// runs of the SAME position next to each other: adjacent lanes holding the same position form a run,
// the run's head lane counts the run's flags with ballot + popcount and issues ONE atomic per
// counter for the whole run.  Integer adds commute, so the result is bit-identical either way.
// HBM-bound in principle (10 B per base in); at today's batch sizes it is launch-latency bound.
// ---------------------------------------------------------------------------------------------
__global__ void summary_add_kernel(int* __restrict__ touch, int* __restrict__ cov, int* __restrict__ mod,
                                   const long long length, const long long* __restrict__ pos,
                                   const unsigned char* __restrict__ flags,
                                   const unsigned char* __restrict__ cls, const long long n,
                                   int* __restrict__ oob) {
    const int lane = threadIdx.x & 63;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // whole waves iterate together so that ballots see all 64 lanes
    for (long long base = blockIdx.x * (long long)blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
        const long long i = base + lane;
        unsigned f = 0;
        long long q = -1 - lane;                      // inactive lanes get distinct keys: never merged
        if (i < n) {
            f = flags[i];
            if (cls) f = (f & 3u) | (cls[i] == 1 ? 4u : 0u);
            if (f & 1u) {
                q = pos[i];
                if (q < 0 || q >= length) {
                    atomicAdd(oob, 1);
                    f = 0;
                    q = -1 - lane;
                }
            } else {
                f = 0;
            }
        }
        const long long qprev = __shfl_up(q, 1, 64);
        const bool head = lane == 0 || q != qprev;
        const unsigned long long heads = __ballot(head);
        const unsigned long long m_touch = __ballot((f & 1u) != 0);
        const unsigned long long m_cov = __ballot((f & 3u) == 3u);
        const unsigned long long m_mod = __ballot((f & 7u) == 7u);
        if (head && (f & 1u)) {
            // run = [lane, next head)
            const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1));
            const int len = above ? __ffsll((long long)above) : 64 - lane;
            const unsigned long long run = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
            atomicAdd(touch + q, __popcll(m_touch & run));
            const int nc = __popcll(m_cov & run);
            if (nc) atomicAdd(cov + q, nc);
            const int nm = __popcll(m_mod & run);
            if (nm) atomicAdd(mod + q, nm);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CpG-cluster MLP (hm_cluster_predict.py:94-103): one row per thread, weights broadcast from LDS,
// k-ascending fp32 accumulation.  60 B in/out per row, 3,420 FMA per row: VALU-bound, tiny.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cluster_mlp_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                          const long long n, float* __restrict__ out) {
    __shared__ float lw[DM_CLUSTER_WEIGHT_FLOATS];
    for (int i = threadIdx.x; i < DM_CLUSTER_WEIGHT_FLOATS; i += blockDim.x) lw[i] = w[i];
    __syncthreads();
    const float* W1 = lw;             // [14][100]
    const float* b1 = lw + 1400;      // [100]
    const float* W2 = lw + 1500;      // [100][20]
    const float* b2 = lw + 3500;      // [20]
    const float* WO = lw + 3520;      // [20]
    const float bO = lw[3540];
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        float xi[14];
#pragma unroll
        for (int k = 0; k < 14; ++k) xi[k] = x[r * 14 + k];
        float h2[20];
#pragma unroll
        for (int j = 0; j < 20; ++j) h2[j] = 0.0f;
        for (int u = 0; u < 100; ++u) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 14; ++k) a = fmaf(xi[k], W1[k * 100 + u], a);
            a = fmaxf(a + b1[u], 0.0f);                 // layer_1 = relu(X W_1 + b_1); dropout(keep_prob 1) = identity
#pragma unroll
            for (int j = 0; j < 20; ++j) h2[j] = fmaf(a, W2[u * 20 + j], h2[j]);
        }
        float o = 0.0f;
#pragma unroll
        for (int j = 0; j < 20; ++j) o = fmaf(fmaxf(h2[j] + b2[j], 0.0f), WO[j], o);   // layer_2 = relu(.), then W_O
        o += bO;
        out[r] = 1.0f / (1.0f + expf(-o));              // output = sigmoid(.)
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------
struct dm_model {
    int device = 0;
    int num_cu = 0;
    hipStream_t stream = nullptr;
    float* d_wpack = nullptr;
    float* d_bpack = nullptr;
    float* d_hpack = nullptr;
    unsigned char* d_wpack16s = nullptr;  // experiment builds (DM_WITH_F16S): split-f16 weights in the tile-major layout of the 32x32x16 kernels
    unsigned char* d_wpack16q = nullptr;  // the same weights in the 16x16x32 layout (lstm_f16q.hip.inc)
    bool f16_q = true;                    // the split-f16 modes run lstm16q::bilstm_f16q_kernel (16x16x32 MFMAs): always, except in an experiment build with DM_OPT_F16X3_SHAPE = 32
    unsigned char* d_wpack16i = nullptr;  // the same layout with int8 cross-term records (DM_PREC_F16I8)
    float i8s[24] = {};                   // its fold scales [dir][layer][gate kind]
    unsigned char* d_wpack16qi = nullptr; // DM_PREC_F16I8 in the 16x16x32 layout (lstm16q::bilstm_f16q_kernel<1>: int8 16x16x64 cross terms)
    float i8s_q[24] = {};                 // its fold scales
    float* d_wout = nullptr;              // head W[200][2] fp32 (DM_PREC_F16X3)
    float* d_scratch = nullptr;
    unsigned long long* d_dbg = nullptr;  // DM_TIMING builds only
    float* d_plogit = nullptr;            // partial logits [2 directions][plogit_tiles * 128][2], grown on demand
    int64_t plogit_tiles = 0;
    float bout[2] = {0, 0};
    int grid_cap = 0;
    std::vector<float> host_weights;      // canonical blob, kept for lazy packing of other precisions
    // staging for host-pointer callers
    static constexpr int64_t STAGE_WINDOWS = 65536;
    float* d_x = nullptr;
    float* d_x2 = nullptr;       // second staging buffer: H2D of batch i+1 overlaps the kernel of batch i
    hipStream_t copy_stream = nullptr;
    static constexpr int AHEAD_EVENTS = 8;      // dm_model_h2d_ahead: copy-done events, reused round robin (a wait captures the record it was queued behind)
    hipEvent_t ahead_event[AHEAD_EVENTS] = {};
    int ahead_next = 0;
    float* d_prob = nullptr;
    uint8_t* d_cls = nullptr;
    int64_t stage_rows = 0;  // capacity of d_x in floats
    // profiling
    bool profile = false;
    bool async = false;                   // DM_OPT_ASYNC: device-resident calls return after enqueue
    int precision = DM_PREC_F16X3;        // default: fastest mode that meets the 1e-4 probability tolerance
    float f16_max_abs = 0.0f;             // largest |packed weight| (x exponent scale)
    bool f16_ok = true;                   // every packed weight is a finite f16 (checked at create; else the default is DM_PREC_F32)
    bool i8_calibrated = false;           // DM_PREC_F16I8 was selected by dm_model_calibrate_i8 (for the MFMA shape of that moment)
    int len_shift = 0;                    // DM_INFO_F16_LENGTH_SHIFT
    // DM_ERANGE bookkeeping: host-mapped words the split-f16 kernels set on an input they cannot represent.  A launch writes
    // the CURRENT slot; dm_model_mark(i) ties the current slot to marker i and moves on to a free one, so that
    // dm_model_wait_mark(i) reports exactly the launches queued between the marker before it and marker i, and a later
    // batch that is still in flight cannot leak into (or be cleared by) the check of an earlier one.
    static constexpr int RANGE_SLOTS = 2 * DM_MARKS + 2;
    int* range_flag = nullptr;            // [RANGE_SLOTS]
    int* d_range_flag = nullptr;          // its device address
    int range_cur = 0;                    // slot of the launches being queued now
    int mark_slot[DM_MARKS];              // slot tied to marker i, -1: none
    std::vector<int> range_orphans;       // slots of markers that were re-recorded before anybody waited for them: checked by dm_model_sync
    bool slot_free[RANGE_SLOTS];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    hipEvent_t marks[DM_MARKS] = {};     // dm_model_mark / dm_model_wait_mark
    size_t events_used = 0;
    double prof_ms = 0.0;
    int64_t prof_launches = 0, prof_windows = 0;
};

struct dm_summary {
    int device = 0;
    int64_t length = 0;
    int* d_counts = nullptr;  // touch | cov | mod  (+ SLACK ints: a reduce-scatter reads up to nranks - 1 positions past the last array)
    static constexpr int64_t SLACK = 64;
    int* d_slice = nullptr;   // dm_summary_reduce_scatter: this rank's slice of the three arrays, [3][slice_chunk]
    int64_t slice_chunk = 0, slice_first = 0, slice_count = -1;   // slice_count < 0: no scatter result held
    int* d_oob = nullptr;
    long long* d_pos = nullptr;
    unsigned char* d_flags = nullptr;
    int64_t stage_cap = 0;
    hipStream_t stream = nullptr;
    dm_model* follow = nullptr;   // dm_summary_follow: enqueue on this model's stream (in-order with its launches)
};

namespace {

int ensure_stage(dm_model* m, int64_t x_floats) {
    if (!m->d_prob) {
        HIP_TRY(hipMalloc(&m->d_prob, sizeof(float) * 2 * dm_model::STAGE_WINDOWS));
        HIP_TRY(hipMalloc(&m->d_cls, dm_model::STAGE_WINDOWS));
    }
    if (x_floats > m->stage_rows) {
        if (m->d_x) HIP_TRY(hipFree(m->d_x));
        m->d_x = nullptr;
        HIP_TRY(hipMalloc(&m->d_x, sizeof(float) * x_floats));
        m->stage_rows = x_floats;
    }
    return DM_OK;
}

int flush_profile(dm_model* m) {
    for (size_t i = 0; i < m->events_used; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(m->events[i].second));
        HIP_TRY(hipEventElapsedTime(&ms, m->events[i].first, m->events[i].second));
        m->prof_ms += ms;
    }
    m->events_used = 0;
    return DM_OK;
}

// partial-logit buffer of direction-split launches: 16 B per window, grown on demand (a running launch may still use the
// old buffer: wait for the stream before freeing it)
int ensure_plogit(dm_model* m, int64_t ntiles) {
    if (ntiles <= m->plogit_tiles) return DM_OK;
    HIP_TRY(hipStreamSynchronize(m->stream));
    (void)hipFree(m->d_plogit);
    m->d_plogit = nullptr;
    m->plogit_tiles = 0;
    const int64_t cap = std::max<int64_t>(ntiles + ntiles / 4, 1024);
    HIP_TRY(hipMalloc(&m->d_plogit, size_t(2) * size_t(cap) * lstmhead::TILE_M * 2 * sizeof(float)));
    m->plogit_tiles = cap;
    return DM_OK;
}

// what both split-f16 kernels need: weights that fit an f16 after the exponent-scale fold (checked once in model_init),
// the fp32 head weights
int ensure_f16_common(dm_model* m) {
    if (!m->f16_ok)
        return fail(DM_EINVAL, "DM_PREC_F16X3: a packed weight (|w| x exponent scale = %g) is outside the f16 range; use DM_PREC_F32",
                    double(m->f16_max_abs));
    if (m->d_wout) return DM_OK;
    const float* wout = m->host_weights.data() + (DM_WEIGHT_FLOATS - 402);
    HIP_TRY(hipMalloc(&m->d_wout, 400 * sizeof(float)));
    HIP_TRY(hipMemcpy(m->d_wout, wout, 400 * sizeof(float), hipMemcpyHostToDevice));
    return DM_OK;
}

// one weight pack per (shape, mode), built on first use
int ensure_pack(unsigned char*& d_pack, const Packed16& P, hipError_t prepared) {
    if (prepared != hipSuccess) return fail(DM_EDEVICE, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", hipGetErrorString(prepared));
    // the model's pointer is set only once the pack is on the device: a failed copy must not leave a non-null pointer to uninitialised memory
    // behind (the next launch would take it for a finished pack)
    unsigned char* d = nullptr;
    HIP_TRY(hipMalloc(&d, P.w.size()));
    const hipError_t e = hipMemcpy(d, P.w.data(), P.w.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(DM_EDEVICE, "uploading a weight pack failed: %s", hipGetErrorString(e));
    }
    d_pack = d;
    return DM_OK;
}
#ifdef DM_WITH_F16S
int ensure_f16s(dm_model* m) {
    if (m->d_wpack16s) return DM_OK;
    int rc = ensure_f16_common(m);
    if (rc) return rc;
    rc = ensure_pack(m->d_wpack16s, dmk::pack_weights_f16s(m->host_weights.data()), dmk::f16s_prepare(0));
    if (rc) return rc;
    if (dmk::f16s_has_roles() && dmk::f16s_prepare(2) != hipSuccess) return fail(DM_EDEVICE, "the roles kernel cannot be prepared");
    return DM_OK;
}
int ensure_f16i8(dm_model* m) {
    if (m->d_wpack16i) return DM_OK;
    int rc = ensure_f16_common(m);
    if (rc) return rc;
    return ensure_pack(m->d_wpack16i, dmk::pack_weights_f16s(m->host_weights.data(), true, m->i8s), dmk::f16s_prepare(1));
}
#endif
int ensure_f16q(dm_model* m) {
    if (m->d_wpack16q) return DM_OK;
    int rc = ensure_f16_common(m);
    if (rc) return rc;
    return ensure_pack(m->d_wpack16q, dmk::pack_weights_f16q(m->host_weights.data()), dmk::f16q_prepare(0));
}
int ensure_f16qi(dm_model* m) {
    if (m->d_wpack16qi) return DM_OK;
    int rc = ensure_f16_common(m);
    if (rc) return rc;
    return ensure_pack(m->d_wpack16qi, dmk::pack_weights_f16q(m->host_weights.data(), true, m->i8s_q), dmk::f16q_prepare(1));
}


// launch on device-resident buffers
int launch_bilstm(dm_model* m, const float* d_x, long long xstride, int64_t n, float* d_prob, uint8_t* d_cls, const int* d_widx = nullptr) {
    if (n <= 0) return DM_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (m->profile) {
        if (m->events_used == m->events.size()) {
            if (m->events.size() >= 4096) {
                int rc = flush_profile(m);
                if (rc) return rc;
            } else {
                hipEvent_t a, b;
                HIP_TRY(hipEventCreate(&a));
                HIP_TRY(hipEventCreate(&b));
                m->events.emplace_back(a, b);
            }
        }
        e0 = m->events[m->events_used].first;
        e1 = m->events[m->events_used].second;
        ++m->events_used;
        HIP_TRY(hipEventRecord(e0, m->stream));
    }
    const int ntiles = int((n + dmk::TILE_M - 1) / dmk::TILE_M);
    {
        int rcp = ensure_plogit(m, ntiles);
        if (rcp) return rcp;
    }
    const long long npad = (long long)ntiles * dmk::TILE_M;
    // work item = (tile, direction); a persistent workgroup per CU takes items round robin.  When the items do not fill the last round, the launch takes
    // ceil(items / cap) rounds whatever the grid: it is sized to that many rounds exactly (4,198 items on 256 CUs = 17 rounds: 247 workgroups of 17 items
    // finish when 256 workgroups of 16 or 17 would) and the CUs left over run whatever else is queued - the signal stage's kernels of the next batches -
    // instead of idling through the last round (round 6)
    int grid = std::min(2 * ntiles, m->grid_cap);
    if (2 * ntiles > m->grid_cap) {
        const int rounds = (2 * ntiles + m->grid_cap - 1) / m->grid_cap;
        grid = (2 * ntiles + rounds - 1) / rounds;
    }
    if (m->precision == DM_PREC_F16X3 || m->precision == DM_PREC_F16I8 || m->precision == DM_PREC_F16X3_ROLES) {
        const bool i8 = m->precision == DM_PREC_F16I8;
        const bool q16 = m->precision == DM_PREC_F16X3 && m->f16_q;
        const bool qi8 = i8 && m->f16_q;                 // DM_OPT_F16X3_SHAPE picks the MFMA shape of both modes
#ifdef DM_WITH_F16S
        int rc = qi8 ? ensure_f16qi(m) : i8 ? ensure_f16i8(m) : (q16 ? ensure_f16q(m) : ensure_f16s(m));
#else
        if (!m->f16_q || m->precision == DM_PREC_F16X3_ROLES) return fail(DM_EINVAL, "the 32x32x16 kernels are not part of this build (tools/experiments/f16s: DM_WITH_F16S=1)");
        int rc = qi8 ? ensure_f16qi(m) : ensure_f16q(m);
#endif
        if (rc) return rc;
        dmk::F16Args a;
        a.wpack = qi8 ? m->d_wpack16qi : i8 ? m->d_wpack16i : q16 ? m->d_wpack16q : m->d_wpack16s;
        a.hpack = m->d_wout;
        a.x = d_x;
        a.xstride = xstride;
        a.widx = d_widx;
        a.n = n;
        a.ntiles = ntiles;
        a.plogit = m->d_plogit;
        a.len_shift = m->len_shift;
        a.range_flag = m->d_range_flag + m->range_cur;
        a.i8s = qi8 ? m->i8s_q : i8 ? m->i8s : nullptr;
        if (q16 || qi8) dmk::f16q_launch(qi8 ? 1 : 0, a, grid, m->stream);
#ifdef DM_WITH_F16S
        else dmk::f16s_launch(i8 ? 1 : (m->precision == DM_PREC_F16X3_ROLES ? 2 : 0), a, grid, m->stream);
#endif
    } else {
        dmk::F32Args a;
        a.wpack = m->d_wpack;
        a.bpack = m->d_bpack;
        a.hpack = m->d_hpack;
        a.bout0 = m->bout[0];
        a.bout1 = m->bout[1];
        a.x = d_x;
        a.xstride = xstride;
        a.widx = d_widx;
        a.n = n;
        a.prob = d_prob;
        a.cls = d_cls;
        a.scratch = m->d_scratch;
        a.ntiles = ntiles;
        a.dbg = m->d_dbg;
        a.dir_split = 1;      // work item = (tile, direction), see above
        a.plogit = m->d_plogit;
        dmk::f32_launch(a, grid, m->stream);
    }
    hipLaunchKernelGGL(lstmhead::head_finish_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, m->stream, m->d_plogit, (long long)n,
                       npad, m->bout[0], m->bout[1], d_prob, d_cls);
    HIP_TRY(hipGetLastError());
    if (m->profile) {
        HIP_TRY(hipEventRecord(e1, m->stream));
        ++m->prof_launches;
        m->prof_windows += n;
    }
    return DM_OK;
}

int range_error(dm_model* m) {
    return fail(DM_ERANGE, "DM_PREC_F16X3: an input feature is outside the representable range (features 0-5: |x| <= 65504, "
                "feature 6: |x| <= 65504 * 2^%d, no NaN); the results of these launches are invalid - repeat them with DM_PREC_F32",
                m->len_shift);
}
// the launches that wrote `slot` have finished: did one of them meet an input it cannot represent?  The slot is cleared.
bool take_range_slot(dm_model* m, int slot) {
    volatile int* f = reinterpret_cast<volatile int*>(m->range_flag) + slot;
    const bool bad = *f != 0;
    *f = 0;
    return bad;
}
// after the model's stream has been synchronised: every slot is final - the current one, the markers', the orphans'
int check_range_all(dm_model* m) {
    if (!m->range_flag) return DM_OK;
    bool bad = take_range_slot(m, m->range_cur);
    for (int i = 0; i < DM_MARKS; ++i)
        if (m->mark_slot[i] >= 0) {
            bad |= take_range_slot(m, m->mark_slot[i]);
            m->slot_free[m->mark_slot[i]] = true;
            m->mark_slot[i] = -1;
        }
    for (int sl : m->range_orphans) {
        bad |= take_range_slot(m, sl);
        m->slot_free[sl] = true;
    }
    m->range_orphans.clear();
    return bad ? range_error(m) : DM_OK;
}
int sync_and_check(dm_model* m) {
    HIP_TRY(hipStreamSynchronize(m->stream));
    return check_range_all(m);
}

int predict_common(dm_model* m, const float* x, long long xstride, int64_t x_floats_total, int64_t n,
                   float* prob, uint8_t* cls, bool windows_materialised) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (n < 0) return fail(DM_EINVAL, "negative window count");
    if (n == 0) return DM_OK;
    if (!x) return fail(DM_EINVAL, "null input");
    HIP_TRY(hipSetDevice(m->device));
    const bool xdev = is_device_ptr(x);
    const bool pdev = prob ? is_device_ptr(prob) : true;
    const bool cdev = cls ? is_device_ptr(cls) : true;
    if (xdev && pdev && cdev) {
        int rc = launch_bilstm(m, x, xstride, n, prob, cls);
        if (rc) return rc;
        if (!m->async) return sync_and_check(m);
        return DM_OK;
    }
    // host buffers: stage through device memory in batches
    if (!windows_materialised) {
        // per-read rows: copy the whole row matrix once, classify, copy results back
        int rc = ensure_stage(m, x_floats_total);
        if (rc) return rc;
        const float* dx = x;
        if (!xdev) {
            HIP_TRY(hipMemcpyAsync(m->d_x, x, sizeof(float) * x_floats_total, hipMemcpyHostToDevice, m->stream));
            dx = m->d_x;
        }
        for (int64_t off = 0; off < n; off += dm_model::STAGE_WINDOWS) {
            const int64_t cnt = std::min<int64_t>(dm_model::STAGE_WINDOWS, n - off);
            float* dp = prob ? (pdev ? prob + 2 * off : m->d_prob) : nullptr;
            uint8_t* dc = cls ? (cdev ? cls + off : m->d_cls) : nullptr;
            rc = launch_bilstm(m, dx + off * xstride, xstride, cnt, dp, dc);
            if (rc) return rc;
            if (prob && !pdev)
                HIP_TRY(hipMemcpyAsync(prob + 2 * off, m->d_prob, sizeof(float) * 2 * cnt, hipMemcpyDeviceToHost, m->stream));
            if (cls && !cdev) HIP_TRY(hipMemcpyAsync(cls + off, m->d_cls, cnt, hipMemcpyDeviceToHost, m->stream));
            rc = sync_and_check(m);
            if (rc) return rc;
        }
        return DM_OK;
    }
    const int64_t stage_floats = int64_t(dm_model::STAGE_WINDOWS) * DM_WINDOW * DM_NFEAT;
    int rc = ensure_stage(m, stage_floats);
    if (rc) return rc;
    if (!xdev && !m->d_x2) {
        HIP_TRY(hipMalloc(&m->d_x2, sizeof(float) * stage_floats));
        HIP_TRY(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    }
    float* stage[2] = {m->d_x, m->d_x2};
    const int64_t win_floats = DM_WINDOW * DM_NFEAT;
    if (!xdev) {   // prime the pipeline: batch 0
        const int64_t cnt0 = std::min<int64_t>(dm_model::STAGE_WINDOWS, n);
        HIP_TRY(hipMemcpyAsync(stage[0], x, sizeof(float) * cnt0 * win_floats, hipMemcpyHostToDevice, m->copy_stream));
        HIP_TRY(hipStreamSynchronize(m->copy_stream));
    }
    int b = 0;
    for (int64_t off = 0; off < n; off += dm_model::STAGE_WINDOWS, b ^= 1) {
        const int64_t cnt = std::min<int64_t>(dm_model::STAGE_WINDOWS, n - off);
        const float* dx = xdev ? x + off * xstride : stage[b];
        float* dp = prob ? (pdev ? prob + 2 * off : m->d_prob) : nullptr;
        uint8_t* dc = cls ? (cdev ? cls + off : m->d_cls) : nullptr;
        rc = launch_bilstm(m, dx, xstride, cnt, dp, dc);          // asynchronous on the model's stream
        if (rc) return rc;
        const int64_t noff = off + dm_model::STAGE_WINDOWS;
        if (!xdev && noff < n) {                                  // upload the next batch meanwhile
            const int64_t ncnt = std::min<int64_t>(dm_model::STAGE_WINDOWS, n - noff);
            HIP_TRY(hipMemcpyAsync(stage[b ^ 1], x + noff * xstride, sizeof(float) * ncnt * win_floats,
                                   hipMemcpyHostToDevice, m->copy_stream));
            HIP_TRY(hipStreamSynchronize(m->copy_stream));
        }
        if (prob && !pdev)
            HIP_TRY(hipMemcpyAsync(prob + 2 * off, m->d_prob, sizeof(float) * 2 * cnt, hipMemcpyDeviceToHost, m->stream));
        if (cls && !cdev) HIP_TRY(hipMemcpyAsync(cls + off, m->d_cls, cnt, hipMemcpyDeviceToHost, m->stream));
        rc = sync_and_check(m);
        if (rc) return rc;
    }
    return DM_OK;
}

int model_init(dm_model* m, const float* weights) {
    HIP_TRY(hipSetDevice(m->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, m->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DM_EDEVICE, "device %d is %s; this library is built for gfx950 only", m->device, prop.gcnArchName);
    m->num_cu = prop.multiProcessorCount;
    m->grid_cap = m->num_cu;  // 120 KB of LDS per workgroup -> one resident (persistent) workgroup per CU
    HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    m->host_weights.assign(weights, weights + DM_WEIGHT_FLOATS);
    Packed32 P = dmk::pack_weights_f32(weights);
    m->bout[0] = P.bout[0];
    m->bout[1] = P.bout[1];
    HIP_TRY(hipMalloc(&m->d_wpack, P.w.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&m->d_bpack, P.b.size() * sizeof(float)));
    HIP_TRY(hipMalloc(&m->d_hpack, P.h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(m->d_wpack, P.w.data(), P.w.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_bpack, P.b.data(), P.b.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(m->d_hpack, P.h.data(), P.h.size() * sizeof(float), hipMemcpyHostToDevice));
    size_t scratch_bytes = size_t(m->grid_cap) * dmk::f32_scratch_floats_per_wg() * sizeof(float);      // h sequences of the fp32 kernel
    HIP_TRY(hipMalloc(&m->d_scratch, scratch_bytes));
    HIP_TRY(hipMemsetAsync(m->d_scratch, 0, scratch_bytes, m->stream));      // ordered with the launches of m->stream (non-blocking)
#if defined(DM_TIMING) || defined(DM_TRACE) || defined(DM_TRACE2)
    HIP_TRY(hipMalloc(&m->d_dbg, size_t(m->grid_cap) * dmk::f32_waves() * 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(m->d_dbg, 0, size_t(m->grid_cap) * dmk::f32_waves() * 8 * sizeof(unsigned long long), m->stream));
#endif
    HIP_TRY(dmk::f32_prepare());
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->range_flag), sizeof(int) * dm_model::RANGE_SLOTS, hipHostMallocMapped));
    for (int i = 0; i < dm_model::RANGE_SLOTS; ++i) {
        m->range_flag[i] = 0;
        m->slot_free[i] = i != 0;
    }
    for (int i = 0; i < DM_MARKS; ++i) m->mark_slot[i] = -1;
    m->range_cur = 0;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&m->d_range_flag), m->range_flag, 0));
    {   // trained kernels far outside the usual range cannot be split into f16 halves: such a model runs the fp32 kernel
        Packed16 P16 = dmk::pack_weights_f16q(weights);
        m->f16_ok = P16.finite && P16.max_abs <= 65504.0f;
        m->f16_max_abs = P16.finite ? P16.max_abs : INFINITY;
        m->len_shift = P16.len_shift;
        m->precision = m->f16_ok ? DM_PREC_F16X3 : DM_PREC_F32;
    }
    m->f16_q = true;
#ifdef DM_WITH_F16S
    {   // experiment builds: DM_F16X3_SHAPE=32 in the environment at model creation runs the split-f16 modes on the 32x32x16 kernels of rounds 2-3
        const char* e = std::getenv("DM_F16X3_SHAPE");
        m->f16_q = DM_F16X3_SHAPE_DEFAULT == 16;
        if (e && *e) {
            if (!std::strcmp(e, "16")) m->f16_q = true;
            else if (!std::strcmp(e, "32")) m->f16_q = false;
            else std::fprintf(stderr, "deepmod_hip: DM_F16X3_SHAPE=%s ignored (16 or 32); the default shape %d stays\n", e, DM_F16X3_SHAPE_DEFAULT);
        }
    }
#endif
    return DM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* dm_last_error(void) { return g_err.c_str(); }
#define DM_STR2(X) #X
#define DM_STR(X) DM_STR2(X)
// the compiler is part of the identity of the hand-scheduled kernels: the evidence of a round is valid for the hipcc it was taken with
const char* dm_version(void) { return "deepmod_hip 0.5 (gfx950; hip " DM_STR(HIP_VERSION_MAJOR) "." DM_STR(HIP_VERSION_MINOR) "." DM_STR(HIP_VERSION_PATCH) "; " __VERSION__ ")"; }
const char* dm_build_flags(void) {
#ifdef DM_EXPERIMENT
    return "experiment=1 switches=" DM_STR(DM_ANY_EXPERIMENT_SWITCH);
#else
    return "experiment=0 switches=0";
#endif
}

int dm_device_pci_bus_id(int device, char* buf, int len) {
    if (!buf || len < 13) return fail(DM_EINVAL, "dm_device_pci_bus_id: buffer of >= 13 bytes");
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return DM_OK;
}

int dm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, d) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

dm_model* dm_model_create(int device, const float* weights, size_t n_floats, int n_feat, int hidden, int window,
                          int layers) {
    if (!weights || n_floats != DM_WEIGHT_FLOATS) {
        fail(DM_EINVAL, "weights: expected %d floats, got %zu", DM_WEIGHT_FLOATS, n_floats);
        return nullptr;
    }
    if (n_feat != DM_NFEAT || hidden != DM_HIDDEN || window != DM_WINDOW || layers != DM_LAYERS) {
        fail(DM_EINVAL, "unsupported geometry fnum=%d hidden=%d window=%d layers=%d (built for 7/100/21/3)", n_feat,
             hidden, window, layers);
        return nullptr;
    }
    dm_model* m = new (std::nothrow) dm_model();
    if (!m) {
        fail(DM_ENOMEM, "out of host memory");
        return nullptr;
    }
    m->device = device;
    if (model_init(m, weights) != DM_OK) {
        std::string keep = g_err;
        dm_model_destroy(m);
        g_err = keep;
        return nullptr;
    }
    return m;
}

void dm_model_destroy(dm_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    for (auto& e : m->events) {
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    for (hipEvent_t e : m->marks)
        if (e) (void)hipEventDestroy(e);
    (void)hipFree(m->d_wpack);
    (void)hipFree(m->d_bpack);
    (void)hipFree(m->d_hpack);
    (void)hipFree(m->d_scratch);
    (void)hipFree(m->d_wpack16s);
    (void)hipFree(m->d_wpack16q);
    (void)hipFree(m->d_wpack16qi);
    (void)hipFree(m->d_wpack16i);
    (void)hipFree(m->d_wout);
    (void)hipFree(m->d_dbg);
    (void)hipFree(m->d_plogit);
    (void)hipFree(m->d_x);
    (void)hipFree(m->d_x2);
    for (hipEvent_t ev : m->ahead_event)
        if (ev) (void)hipEventDestroy(ev);
    if (m->copy_stream) (void)hipStreamDestroy(m->copy_stream);
    if (m->range_flag) (void)hipHostFree(m->range_flag);
    (void)hipFree(m->d_prob);
    (void)hipFree(m->d_cls);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

int dm_model_set_option(dm_model* m, int key, int64_t value) {
    if (!m) return fail(DM_EINVAL, "null model");
    switch (key) {
        case DM_OPT_PROFILE:
            m->profile = value != 0;
            return DM_OK;
        case DM_OPT_ASYNC:
            m->async = value != 0;
            return DM_OK;
        case DM_OPT_F16X3_SHAPE:
            if (value != 16 && value != 32) return fail(DM_EINVAL, "DM_OPT_F16X3_SHAPE: %lld (16 or 32)", (long long)value);
#ifndef DM_WITH_F16S
            if (value == 32)
                return fail(DM_EINVAL, "DM_OPT_F16X3_SHAPE = 32 (the 32x32x16 kernels of rounds 2-3) is not part of this build: tools/experiments/f16s, DM_WITH_F16S=1");
#endif
            if ((value == 16) != m->f16_q && m->precision == DM_PREC_F16I8 && m->i8_calibrated) {
                // the calibration gate looked at the int8 kernel of the OTHER shape (its own pack, its own quantisation): the selection does not carry over
                m->precision = DM_PREC_F16X3;
                m->i8_calibrated = false;
            }
            m->f16_q = value == 16;
            return DM_OK;
        case DM_OPT_RESERVED_CUS:
            if (value < 0 || value > m->num_cu / 2) return fail(DM_EINVAL, "reserved CUs %lld outside [0, %d]", (long long)value, m->num_cu / 2);
            m->grid_cap = m->num_cu - int(value);
            return DM_OK;
        case DM_OPT_PRECISION:
            if (value != DM_PREC_F32 && value != DM_PREC_F16X3 && value != DM_PREC_F16X3_ROLES && value != DM_PREC_F16I8)
                return fail(DM_EINVAL, "unknown precision %lld", (long long)value);
            m->i8_calibrated = false;      // an explicit choice of the caller
#ifndef DM_WITH_F16X3_ROLES
            if (value == DM_PREC_F16X3_ROLES)
                return fail(DM_EINVAL, "DM_PREC_F16X3_ROLES (the wave-pair experiment kernel of round 4) is not part of this build; rebuild with -DDM_WITH_F16X3_ROLES");
#endif
            if (value != DM_PREC_F32 && !m->f16_ok)
                return fail(DM_EINVAL, "DM_PREC_F16X3 refused: a packed weight of this model is outside the f16 range (|w| x 2.886 > 65504)");
            m->precision = int(value);
            return DM_OK;
        default:
            return fail(DM_EINVAL, "unknown option %d", key);
    }
}

int dm_predict_windows(dm_model* m, const float* x, int64_t n, float* prob, uint8_t* cls) {
    return predict_common(m, x, DM_WINDOW * DM_NFEAT, n * DM_WINDOW * DM_NFEAT, n, prob, cls, true);
}

int dm_predict_read(dm_model* m, const float* rows, int64_t m_rows, int64_t first, int64_t count, float* prob,
                    uint8_t* cls) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (count < 0 || first < DM_WINDOW / 2 || first + count + DM_WINDOW / 2 > m_rows)
        return fail(DM_EINVAL, "window range [%lld, %lld) +-10 outside the %lld feature rows", (long long)first,
                    (long long)(first + count), (long long)m_rows);
    if (count == 0) return DM_OK;
    if (!rows) return fail(DM_EINVAL, "null input");
    // window i = rows[first + i - 10 .. first + i + 10]  ->  base pointer at row (first - 10), window stride 7
    if (is_device_ptr(rows))
        return predict_common(m, rows + (first - DM_WINDOW / 2) * DM_NFEAT, DM_NFEAT, 0, count, prob, cls, false);
    // host rows: ship only the rows this call needs
    const float* base = rows + (first - DM_WINDOW / 2) * DM_NFEAT;
    return predict_common(m, base, DM_NFEAT, (count + DM_WINDOW - 1) * DM_NFEAT, count, prob, cls, false);
}

// Classify `count` windows of a feature-row matrix picked by their CENTRE rows: window i = rows[centre[i] - 10 .. centre[i] + 10].
// What the streaming worker calls: only windows centred on a base of interest can reach the BED (sum_handler tests refbase == Base
// before it looks at mod_pred, myDetect.py:1091-1100), so the other ~3/4 of a read's windows are never computed.  Everything
// device-resident: queued on the model's stream (DM_OPT_ASYNC), else synchronous.  Host arrays are staged through temporaries.
int dm_predict_read_at(dm_model* m, const float* rows, int64_t m_rows, const int32_t* centre, int64_t count, float* prob, uint8_t* cls) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (count < 0 || m_rows < 0) return fail(DM_EINVAL, "negative count");
    if (count == 0) return DM_OK;
    if (!rows || !centre) return fail(DM_EINVAL, "null input");
    if (m_rows < DM_WINDOW) return fail(DM_EINVAL, "%lld feature rows hold no window", (long long)m_rows);
    HIP_TRY(hipSetDevice(m->device));
    const bool rd = is_device_ptr(rows), cd = is_device_ptr(centre), pd = prob ? is_device_ptr(prob) : true, kd = cls ? is_device_ptr(cls) : true;
    if (rd && cd && pd && kd) {
        int rc = launch_bilstm(m, rows - (DM_WINDOW / 2) * DM_NFEAT, DM_NFEAT, count, prob, cls, centre);
        if (rc) return rc;
        return m->async ? DM_OK : sync_and_check(m);
    }
    // host arrays (tests, small callers): the centres are checked, everything goes through temporaries, synchronous
    if (!cd)
        for (int64_t i = 0; i < count; ++i)
            if (centre[i] < DM_WINDOW / 2 || centre[i] >= m_rows - DM_WINDOW / 2)
                return fail(DM_EINVAL, "window %lld: centre row %d +-10 outside the %lld feature rows", (long long)i, centre[i], (long long)m_rows);
    float* d_rows = nullptr;
    int* d_centre = nullptr;
    float* d_prob = nullptr;
    uint8_t* d_cls = nullptr;
    int rc = DM_OK;
    auto cleanup = [&]() {
        if (!rd) (void)hipFree(d_rows);
        if (!cd) (void)hipFree(d_centre);
        if (prob && !pd) (void)hipFree(d_prob);
        if (cls && !kd) (void)hipFree(d_cls);
    };
#define DM_TRY_CLEAN(expr)                                                                                   \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            cleanup();                                                                                       \
            return fail(DM_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(_e));                          \
        }                                                                                                    \
    } while (0)
    if (!rd) {
        DM_TRY_CLEAN(hipMalloc(&d_rows, sizeof(float) * size_t(m_rows) * DM_NFEAT));
        DM_TRY_CLEAN(hipMemcpyAsync(d_rows, rows, sizeof(float) * size_t(m_rows) * DM_NFEAT, hipMemcpyHostToDevice, m->stream));
    } else d_rows = const_cast<float*>(rows);
    if (!cd) {
        DM_TRY_CLEAN(hipMalloc(&d_centre, sizeof(int) * size_t(count)));
        DM_TRY_CLEAN(hipMemcpyAsync(d_centre, centre, sizeof(int) * size_t(count), hipMemcpyHostToDevice, m->stream));
    } else d_centre = const_cast<int*>(centre);
    if (prob && !pd) DM_TRY_CLEAN(hipMalloc(&d_prob, sizeof(float) * 2 * size_t(count)));
    else d_prob = prob;
    if (cls && !kd) DM_TRY_CLEAN(hipMalloc(&d_cls, size_t(count)));
    else d_cls = cls;
    rc = launch_bilstm(m, d_rows - (DM_WINDOW / 2) * DM_NFEAT, DM_NFEAT, count, d_prob, d_cls, d_centre);
    if (rc == DM_OK && prob && !pd) DM_TRY_CLEAN(hipMemcpyAsync(prob, d_prob, sizeof(float) * 2 * size_t(count), hipMemcpyDeviceToHost, m->stream));
    if (rc == DM_OK && cls && !kd) DM_TRY_CLEAN(hipMemcpyAsync(cls, d_cls, size_t(count), hipMemcpyDeviceToHost, m->stream));
    if (rc == DM_OK) rc = sync_and_check(m);
    else (void)hipStreamSynchronize(m->stream);
    cleanup();
#undef DM_TRY_CLEAN
    return rc;
}

// debug builds (-DDM_TIMING): copy the per-wave section cycle counters [grid][waves][8]; returns element count
extern "C" long long dm_debug_timing(dm_model* m, unsigned long long* out, long long cap) {
    if (!m || !m->d_dbg) return 0;
    const long long n = (long long)m->grid_cap * dmk::f32_waves() * 8;
    if (out && cap >= n) (void)hipMemcpy(out, m->d_dbg, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    return n;
}

// ---- load-time calibration gate of the opt-in int8 mode (round 4) ----
// windows with the distribution of BASELINE configs[1] (deepmod_amd/synth.py synthetic_windows: base one-hot 4 x 24 % + 4 % none, event mean
// N(0, 1.2) clipped to +-5, stdv |N(0.25, 0.15)|, both rounded to 3 decimals, length ~ Geometric(0.12)), generated on the device: one
// thread per (window, row), a counter-based hash as the random source (reproducible: the same windows on every box)
namespace calib {
__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float unit(uint64_t h) { return ((float)(h >> 41) + 0.5f) * (1.0f / 8388608.0f); }      // (0, 1): 23 bits + 1/2, exact in fp32
// wide = 0: the configs[1] distribution.  wide = 1 / 2 (all windows / every second window of a batch - the gate's draw since round 6; round 5 alternated
// whole batches, so a call of one batch never saw it): READ-SHAPED tails the nominal draw hardly ever produces - event
// means uniform over the whole clip range with 6 % exactly on the clip (+-5: the MAD-normalised signal is clipped there), standard deviations
// up to 9x, event lengths log-uniform from 1 to 30,000 samples (stalled events) - so that the gate sees the inputs a real run can feed
__global__ void windows_kernel(float* __restrict__ x, long long n_rows, uint64_t seed, int wide) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const uint64_t h0 = mix(seed * 0x100000001B3ull + (uint64_t)i * 4u), h1 = mix(h0), h2 = mix(h1), h3 = mix(h2);
    float* r = x + i * DM_NFEAT;
    const float uc = unit(h0);
    const int cat = uc < 0.96f ? (int)(uc * (1.0f / 0.24f)) : 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) r[b] = cat == b ? 1.0f : 0.0f;
    const float rad = sqrtf(-2.0f * logf(unit(h1))), ang = 6.2831853f * unit(h2);
    const float z0 = rad * cosf(ang), z1 = rad * sinf(ang);
    r[4] = rintf(fminf(fmaxf(1.2f * z0, -5.0f), 5.0f) * 1000.0f) * 0.001f;
    r[5] = rintf(fabsf(0.25f + 0.15f * z1) * 1000.0f) * 0.001f;
    r[6] = 1.0f + floorf(logf(unit(h3)) * (1.0f / -0.12783337f));      // ln(1 - 0.12)
    if (wide == 2 ? int((i / DM_WINDOW) & 1) : wide) {        // 2: every second WINDOW of the batch (a call of a single batch sees both draws)
        const uint64_t h4 = mix(h3), h5 = mix(h4);
        const float u = unit(h4), v = unit(h5);
        r[4] = u < 0.03f ? -5.0f : (u > 0.97f ? 5.0f : rintf((10.0f * unit(h1) - 5.0f) * 1000.0f) * 0.001f);
        r[5] = rintf(fabsf(0.25f + 0.15f * z1) * (1.0f + 8.0f * v * v) * 1000.0f) * 0.001f;
        r[6] = floorf(exp2f(unit(h3) * 14.8727f));                     // 1 .. 30,000
    }
}
// largest |a - b| of two probability arrays (non-negative floats order like their bit patterns)
__global__ void maxdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, unsigned* __restrict__ out) {
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d > m || d != d) ? (d != d ? INFINITY : d) : m;
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}
}  // namespace calib

int dm_model_calibrate_i8(dm_model* m, int64_t n_windows, double bound, double* max_abs_dp, int* selected) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (selected) *selected = 0;
    if (max_abs_dp) *max_abs_dp = INFINITY;
    if (n_windows <= 0 || !(bound > 0.0)) return fail(DM_EINVAL, "dm_model_calibrate_i8: n_windows %lld, bound %g", (long long)n_windows, bound);
    if (!m->f16_ok) return DM_OK;                       // the split-f16 kernels are not available to this model: nothing to select
    HIP_TRY(hipSetDevice(m->device));
    int rc = sync_and_check(m);
    if (rc) return rc;
    const int64_t B = 65536;
    float *d_x = nullptr, *d_pa = nullptr, *d_pb = nullptr;
    unsigned* d_max = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_x); (void)hipFree(d_pa); (void)hipFree(d_pb); (void)hipFree(d_max);
    };
#define DM_TRY_CAL(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            cleanup();                                                                                       \
            return fail(DM_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(_e));                          \
        }                                                                                                    \
    } while (0)
    DM_TRY_CAL(hipMalloc(&d_x, sizeof(float) * size_t(B) * DM_WINDOW * DM_NFEAT));
    DM_TRY_CAL(hipMalloc(&d_pa, sizeof(float) * 2 * size_t(B)));
    DM_TRY_CAL(hipMalloc(&d_pb, sizeof(float) * 2 * size_t(B)));
    DM_TRY_CAL(hipMalloc(&d_max, sizeof(unsigned)));
    DM_TRY_CAL(hipMemsetAsync(d_max, 0, sizeof(unsigned), m->stream));
    const int keep_precision = m->precision;
    const bool keep_profile = m->profile;
    m->profile = false;
    for (int64_t done = 0, batch = 0; done < n_windows && rc == DM_OK; done += B, ++batch) {
        const int64_t n = std::min<int64_t>(B, n_windows - done);
        const long long rows = (long long)n * DM_WINDOW;
        hipLaunchKernelGGL(calib::windows_kernel, dim3(unsigned((rows + 255) / 256)), dim3(256), 0, m->stream, d_x, rows, uint64_t(0x5EEDC0DEull + batch), 2);
        m->precision = DM_PREC_F32;
        rc = launch_bilstm(m, d_x, (long long)DM_WINDOW * DM_NFEAT, n, d_pa, nullptr);
        m->precision = DM_PREC_F16I8;
        if (rc == DM_OK) rc = launch_bilstm(m, d_x, (long long)DM_WINDOW * DM_NFEAT, n, d_pb, nullptr);
        if (rc == DM_OK) hipLaunchKernelGGL(calib::maxdiff_kernel, dim3(256), dim3(256), 0, m->stream, d_pa, d_pb, (long long)n * 2, d_max);
    }
    m->precision = keep_precision;
    m->profile = keep_profile;
    unsigned bits = 0x7F800000u;
    if (rc == DM_OK) {
        rc = sync_and_check(m);
        if (rc == DM_OK) DM_TRY_CAL(hipMemcpy(&bits, d_max, sizeof(unsigned), hipMemcpyDeviceToHost));
    }
#undef DM_TRY_CAL
    cleanup();
    if (rc) return rc;
    float err;
    std::memcpy(&err, &bits, 4);
    if (max_abs_dp) *max_abs_dp = double(err);
    if (double(err) <= bound && m->precision == DM_PREC_F16X3) {
        m->precision = DM_PREC_F16I8;
        m->i8_calibrated = true;
    }
    if (selected) *selected = m->precision == DM_PREC_F16I8 ? 1 : 0;      // the mode the model runs after the call
    if (m->precision != DM_PREC_F16I8) {                    // a refused model does not keep the int8 weight packs (1.7 MB)
        (void)hipFree(m->d_wpack16i);
        (void)hipFree(m->d_wpack16qi);
        m->d_wpack16i = m->d_wpack16qi = nullptr;
    }
    return DM_OK;
}

// device form of a raw batch (dm_rows_emit_device) -> feature rows [n_rows][7] on the device, queued on the model's stream
int dm_rows_assemble(dm_model* m, float* d_rows, const uint8_t* d_code, const float* d_ev3, const int64_t* d_rdesc, int64_t n_reads, int64_t n_rows) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (n_rows <= 0) return DM_OK;
    if (!d_rows || !d_code || !d_ev3 || !d_rdesc || n_reads <= 0 || n_reads > 0x7fffffffLL) return fail(DM_EINVAL, "dm_rows_assemble: null array or no read");
    HIP_TRY(hipSetDevice(m->device));
    hipLaunchKernelGGL(rows_assemble_kernel, dim3(unsigned((n_rows + 255) / 256)), dim3(256), 0, m->stream, d_rows, d_code, d_ev3,
                       reinterpret_cast<const long long*>(d_rdesc), int(n_reads), (long long)n_rows);
    HIP_TRY(hipGetLastError());
    return DM_OK;
}

int dm_model_sync(dm_model* m) {
    if (!m) return fail(DM_EINVAL, "null model");
    HIP_TRY(hipSetDevice(m->device));
    return sync_and_check(m);
}

int dm_model_get_info(dm_model* m, int key, int64_t* value) {
    if (!m || !value) return fail(DM_EINVAL, "null argument");
    switch (key) {
        case DM_INFO_PRECISION: *value = m->precision; return DM_OK;
        case DM_INFO_F16_REPRESENTABLE: *value = m->f16_ok ? 1 : 0; return DM_OK;
        case DM_INFO_F16_LENGTH_SHIFT: *value = m->len_shift; return DM_OK;
        case DM_INFO_DEVICE: *value = m->device; return DM_OK;
        case DM_INFO_HAS_F16S:
#ifdef DM_WITH_F16S
            *value = 1;
#else
            *value = 0;
#endif
            return DM_OK;
        case DM_INFO_HAS_F16X3_ROLES:
#ifdef DM_WITH_F16X3_ROLES
            *value = 1;
#else
            *value = 0;
#endif
            return DM_OK;
        default: return fail(DM_EINVAL, "unknown info key %d", key);
    }
}

int dm_profile_reset(dm_model* m) {
    if (!m) return fail(DM_EINVAL, "null model");
    HIP_TRY(hipStreamSynchronize(m->stream));
    m->events_used = 0;
    m->prof_ms = 0.0;
    m->prof_launches = 0;
    m->prof_windows = 0;
    return DM_OK;
}

int dm_profile_get(dm_model* m, double* kernel_ms, int64_t* launches, int64_t* windows) {
    if (!m) return fail(DM_EINVAL, "null model");
    HIP_TRY(hipStreamSynchronize(m->stream));
    int rc = flush_profile(m);
    if (rc) return rc;
    if (kernel_ms) *kernel_ms = m->prof_ms;
    if (launches) *launches = m->prof_launches;
    if (windows) *windows = m->prof_windows;
    return DM_OK;
}

void* dm_device_alloc(int device, size_t bytes) {
    void* p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        fail(DM_ENOMEM, "hipMalloc(%zu) on device %d failed", bytes, device);
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

int dm_device_free(int device, void* p) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(p));
    return DM_OK;
}

int dm_memcpy_h2d(int device, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return DM_OK;
}

// host -> device copy queued on the model's stream, i.e. ordered with its launches: a staging buffer can be refilled for
// the next batch without a host-side wait.  `src` must stay valid until the next dm_model_sync.
int dm_model_h2d_async(dm_model* m, void* dst, const void* src, size_t bytes) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (bytes == 0) return DM_OK;
    if (!dst || !src) return fail(DM_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, m->stream));
    return DM_OK;
}

// host -> device copy on the model's COPY stream: it does not wait for the launches already queued (it overlaps them), and
// every launch queued after this call waits for it.  The caller guarantees that nothing queued touches dst (a marker passed).
int dm_model_h2d_ahead(dm_model* m, void* dst, const void* src, size_t bytes) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (bytes == 0) return DM_OK;
    if (!dst || !src) return fail(DM_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    if (!m->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    hipEvent_t& ev = m->ahead_event[m->ahead_next];
    if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    m->ahead_next = (m->ahead_next + 1) % dm_model::AHEAD_EVENTS;
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, m->copy_stream));
    HIP_TRY(hipEventRecord(ev, m->copy_stream));
    HIP_TRY(hipStreamWaitEvent(m->stream, ev, 0));
    return DM_OK;
}

void* dm_host_alloc(int device, size_t bytes) {
    if (hipSetDevice(device) != hipSuccess) {
        fail(DM_EDEVICE, "hipSetDevice(%d) failed", device);
        return nullptr;
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        fail(DM_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

int dm_host_free(int device, void* p) {
    if (!p) return DM_OK;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipHostFree(p));
    return DM_OK;
}

int dm_model_mark(dm_model* m, int i) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (i < 0 || i >= DM_MARKS) return fail(DM_EINVAL, "marker %d outside [0, %d)", i, DM_MARKS);
    HIP_TRY(hipSetDevice(m->device));
    if (!m->marks[i]) HIP_TRY(hipEventCreateWithFlags(&m->marks[i], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->marks[i], m->stream));
    // the launches queued since the marker before this one wrote range slot range_cur: it now belongs to marker i
    if (m->mark_slot[i] >= 0) m->range_orphans.push_back(m->mark_slot[i]);      // re-recorded unwaited: dm_model_sync reports it
    int next = -1;
    for (int k = 0; k < dm_model::RANGE_SLOTS; ++k)
        if (m->slot_free[k]) {
            next = k;
            break;
        }
    if (next < 0) {
        // every slot is tied to a marker nobody waited for: keep writing the same slot (its launches are then reported with
        // marker i as well as with the later one - conservative, never silent)
        m->mark_slot[i] = -1;
        return DM_OK;
    }
    m->mark_slot[i] = m->range_cur;
    m->slot_free[next] = false;
    m->range_cur = next;
    return DM_OK;
}

int dm_model_wait_mark(dm_model* m, int i) {
    if (!m) return fail(DM_EINVAL, "null model");
    if (i < 0 || i >= DM_MARKS) return fail(DM_EINVAL, "marker %d outside [0, %d)", i, DM_MARKS);
    if (!m->marks[i]) return DM_OK;
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipEventSynchronize(m->marks[i]));
    if (m->mark_slot[i] < 0) return DM_OK;          // reported already (or shared with a later marker, see dm_model_mark)
    const int sl = m->mark_slot[i];
    m->mark_slot[i] = -1;
    const bool bad = take_range_slot(m, sl);
    m->slot_free[sl] = true;
    return bad ? range_error(m) : DM_OK;
}

int dm_memcpy_d2h(int device, void* dst, const void* src, size_t bytes) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return DM_OK;
}

// ------------------------------------------------------------------------------- summary ----
dm_summary* dm_summary_create(int device, int64_t length) {
    if (length <= 0 || length > (int64_t(1) << 33)) {
        fail(DM_EINVAL, "bad contig length %lld", (long long)length);
        return nullptr;
    }
    dm_summary* s = new (std::nothrow) dm_summary();
    if (!s) {
        fail(DM_ENOMEM, "out of host memory");
        return nullptr;
    }
    s->device = device;
    s->length = length;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&s->d_counts, sizeof(int) * (3 * length + dm_summary::SLACK)) == hipSuccess &&
              hipMalloc(&s->d_oob, sizeof(int)) == hipSuccess &&
              // on the summary's own (non-blocking) stream and waited for: a memset on the null stream is not ordered with the
              // kernels that the summary's or a followed model's stream run next (3 GB of counters take a millisecond to clear)
              hipMemsetAsync(s->d_counts, 0, sizeof(int) * (3 * length + dm_summary::SLACK), s->stream) == hipSuccess &&
              hipMemsetAsync(s->d_oob, 0, sizeof(int), s->stream) == hipSuccess &&
              hipStreamSynchronize(s->stream) == hipSuccess;
    if (!ok) {
        fail(DM_EDEVICE, "summary allocation of %lld positions on device %d failed: %s", (long long)length, device,
             hipGetErrorString(hipGetLastError()));
        dm_summary_destroy(s);
        return nullptr;
    }
    return s;
}

void dm_summary_destroy(dm_summary* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipFree(s->d_counts);
    (void)hipFree(s->d_slice);
    (void)hipFree(s->d_oob);
    (void)hipFree(s->d_pos);
    (void)hipFree(s->d_flags);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int64_t dm_summary_length(const dm_summary* s) { return s ? s->length : 0; }

static int summary_add_impl(dm_summary* s, const int64_t* pos, const uint8_t* flags, const uint8_t* cls, int64_t n,
                            bool* deferred = nullptr) {
    if (!s) return fail(DM_EINVAL, "null summary");
    if (n < 0) return fail(DM_EINVAL, "negative count");
    if (n == 0) return DM_OK;
    if (!pos || !flags) return fail(DM_EINVAL, "null input");
    HIP_TRY(hipSetDevice(s->device));
    const long long* dpos = reinterpret_cast<const long long*>(pos);
    const unsigned char* dfl = flags;
    const unsigned char* dcl = cls;
    const bool pd = is_device_ptr(pos), fd = is_device_ptr(flags), cd = cls ? is_device_ptr(cls) : true;
    if (!(pd && fd && cd)) {
        if (n > s->stage_cap) {
            (void)hipFree(s->d_pos);
            (void)hipFree(s->d_flags);
            s->d_pos = nullptr;
            s->d_flags = nullptr;
            const int64_t cap = std::max<int64_t>(n, 1 << 20);
            HIP_TRY(hipMalloc(&s->d_pos, sizeof(long long) * cap));
            HIP_TRY(hipMalloc(&s->d_flags, 2 * cap));
            s->stage_cap = cap;
        }
        if (!pd) {
            HIP_TRY(hipMemcpyAsync(s->d_pos, pos, sizeof(long long) * n, hipMemcpyHostToDevice, s->stream));
            dpos = s->d_pos;
        }
        if (!fd) {
            HIP_TRY(hipMemcpyAsync(s->d_flags, flags, n, hipMemcpyHostToDevice, s->stream));
            dfl = s->d_flags;
        }
        if (cls && !cd) {
            HIP_TRY(hipMemcpyAsync(s->d_flags + s->stage_cap, cls, n, hipMemcpyHostToDevice, s->stream));
            dcl = s->d_flags + s->stage_cap;
        }
    }
    const int threads = 256;
    const int blocks = int(std::min<int64_t>((n + threads - 1) / threads, 2048));
    hipStream_t st = s->stream;
    if (s->follow) {
        if (pd && fd && cd) {
            st = s->follow->stream;                                 // device-resident inputs: one in-order queue with the classifier
            if (deferred && s->follow->async) *deferred = true;     // out-of-range check waits for dm_summary_sync / fetch
        }
        else HIP_TRY(hipStreamSynchronize(s->follow->stream));      // staged inputs travel on the summary's own stream
    }
    hipLaunchKernelGGL(summary_add_kernel, dim3(blocks), dim3(threads), 0, st, s->d_counts,
                       s->d_counts + s->length, s->d_counts + 2 * s->length, (long long)s->length, dpos, dfl, dcl,
                       (long long)n, s->d_oob);
    HIP_TRY(hipGetLastError());
    return DM_OK;
}

static int summary_check_oob(dm_summary* s) {
    int oob = 0;
    if (s->follow) HIP_TRY(hipStreamSynchronize(s->follow->stream));
    HIP_TRY(hipMemcpyAsync(&oob, s->d_oob, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (oob) {
        HIP_TRY(hipMemsetAsync(s->d_oob, 0, sizeof(int), s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
        return fail(DM_EINVAL, "%d positions outside [0, %lld) were dropped", oob, (long long)s->length);
    }
    return DM_OK;
}

int dm_summary_add(dm_summary* s, const int64_t* pos, const uint8_t* flags, int64_t n) {
    bool deferred = false;
    int rc = summary_add_impl(s, pos, flags, nullptr, n, &deferred);
    if (rc || n <= 0 || deferred) return rc;
    return summary_check_oob(s);  // synchronous on return (the caller may reuse its buffers)
}

int dm_summary_add_classified(dm_summary* s, const int64_t* pos, const uint8_t* flags, const uint8_t* cls,
                              int64_t n) {
    if (n > 0 && !cls) return fail(DM_EINVAL, "null cls");
    bool deferred = false;
    int rc = summary_add_impl(s, pos, flags, cls, n, &deferred);
    if (rc || n <= 0 || deferred) return rc;
    return summary_check_oob(s);
}

int dm_summary_sync(dm_summary* s) {
    if (!s) return fail(DM_EINVAL, "null summary");
    HIP_TRY(hipSetDevice(s->device));
    return summary_check_oob(s);
}

int dm_summary_fetch(dm_summary* s, int32_t* touch, int32_t* cov, int32_t* mod) {
    if (!s) return fail(DM_EINVAL, "null summary");
    HIP_TRY(hipSetDevice(s->device));
    int rc0 = summary_check_oob(s);
    if (rc0) return rc0;
    const size_t bytes = sizeof(int) * s->length;
    if (touch) HIP_TRY(hipMemcpy(touch, s->d_counts, bytes, hipMemcpyDeviceToHost));
    if (cov) HIP_TRY(hipMemcpy(cov, s->d_counts + s->length, bytes, hipMemcpyDeviceToHost));
    if (mod) HIP_TRY(hipMemcpy(mod, s->d_counts + 2 * s->length, bytes, hipMemcpyDeviceToHost));
    return DM_OK;
}

void* dm_summary_device_ptr(dm_summary* s) { return s ? s->d_counts : nullptr; }

int dm_summary_follow(dm_summary* s, dm_model* m) {
    if (!s) return fail(DM_EINVAL, "null summary");
    if (m && m->device != s->device) return fail(DM_EINVAL, "summary (device %d) cannot follow a model on device %d", s->device, m->device);
    if (s->follow) HIP_TRY(hipStreamSynchronize(s->follow->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->follow = m;
    return DM_OK;
}

// ------------------------------------------------------------------------------- cluster ----
struct dm_cluster {
    int device = 0;
    hipStream_t stream = nullptr;
    float* d_w = nullptr;
    float* d_x = nullptr;
    float* d_out = nullptr;
    int64_t cap = 0;
};

dm_cluster* dm_cluster_create(int device, const float* weights, size_t n_floats) {
    if (!weights || n_floats != DM_CLUSTER_WEIGHT_FLOATS) {
        fail(DM_EINVAL, "cluster weights: expected %d floats, got %zu", DM_CLUSTER_WEIGHT_FLOATS, n_floats);
        return nullptr;
    }
    dm_cluster* c = new (std::nothrow) dm_cluster();
    if (!c) {
        fail(DM_ENOMEM, "out of host memory");
        return nullptr;
    }
    c->device = device;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&c->d_w, sizeof(float) * n_floats) == hipSuccess &&
              hipMemcpy(c->d_w, weights, sizeof(float) * n_floats, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) {
        fail(DM_EDEVICE, "cluster model setup on device %d failed: %s", device, hipGetErrorString(hipGetLastError()));
        dm_cluster_destroy(c);
        return nullptr;
    }
    return c;
}

void dm_cluster_destroy(dm_cluster* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipFree(c->d_w);
    (void)hipFree(c->d_x);
    (void)hipFree(c->d_out);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dm_cluster_predict(dm_cluster* c, const float* x, int64_t n, float* out) {
    if (!c) return fail(DM_EINVAL, "null cluster model");
    if (n < 0) return fail(DM_EINVAL, "negative row count");
    if (n == 0) return DM_OK;
    if (!x || !out) return fail(DM_EINVAL, "null buffer");
    HIP_TRY(hipSetDevice(c->device));
    const bool xd = is_device_ptr(x), od = is_device_ptr(out);
    if ((!xd || !od) && n > c->cap) {
        (void)hipFree(c->d_x);
        (void)hipFree(c->d_out);
        c->d_x = c->d_out = nullptr;
        HIP_TRY(hipMalloc(&c->d_x, sizeof(float) * 14 * n));
        HIP_TRY(hipMalloc(&c->d_out, sizeof(float) * n));
        c->cap = n;
    }
    const float* dx = x;
    float* dout = out;
    if (!xd) {
        HIP_TRY(hipMemcpyAsync(c->d_x, x, sizeof(float) * 14 * n, hipMemcpyHostToDevice, c->stream));
        dx = c->d_x;
    }
    if (!od) dout = c->d_out;
    const int blocks = int(std::min<int64_t>((n + 255) / 256, 4096));
    hipLaunchKernelGGL(cluster_mlp_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_w, dx, (long long)n, dout);
    HIP_TRY(hipGetLastError());
    if (!od) HIP_TRY(hipMemcpyAsync(out, c->d_out, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return DM_OK;
}

// ---- RCCL, loaded lazily so single-GPU users never need it ---------------------------------
namespace {
// The library is found with dlopen (a one-GPU user never needs it, and DEEPMOD_RCCL_LIBRARY may name a site build), but every type, enumerator and
// prototype is RCCL's own, from <rccl/rccl.h>: a drift between what is called here and what the library exports is a compile error in this file -
// and in tests/shim/shmccl.cpp, which defines the same functions against the same header - not a surprise on the first eight-rank run.
static_assert(sizeof(ncclUniqueId) == 128 && NCCL_UNIQUE_ID_BYTES == 128, "the C ABI hands the RCCL id around as 128 bytes (dm_rccl_unique_id)");
struct Rccl {
    void* h = nullptr;
    std::string path;                                  // what dlopen was given
    int version = 0;                                   // ncclGetVersion, 0 if the library has none
    decltype(&ncclGetUniqueId) getid = nullptr;
    decltype(&ncclCommInitRank) init = nullptr;
    decltype(&ncclAllReduce) allreduce = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclReduceScatter) reducescatter = nullptr;
    decltype(&ncclCommDestroy) destroy = nullptr;
    decltype(&ncclGetErrorString) errstr = nullptr;
    decltype(&ncclGetVersion) getversion = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;   // optional: one launch for the three counter arrays of a reduce-scatter
    decltype(&ncclGroupEnd) group_end = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return DM_OK;
    // DEEPMOD_RCCL_LIBRARY names the collective library to bind instead of the system's librccl (a site build of RCCL; the one-GPU test
    // transport tests/shim/shmccl.cpp).  When it is set nothing else is tried: a wrong path is an error, not a silent fall back.
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    const char* named = std::getenv("DEEPMOD_RCCL_LIBRARY");
    if (named && *named) {
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(DM_ERCCL, "cannot dlopen DEEPMOD_RCCL_LIBRARY=%s: %s", named, dlerror());
        g_rccl.path = named;
    }
    for (const char* nm : names) {
        if (h) break;
        h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (h) g_rccl.path = nm;
    }
    if (!h) return fail(DM_ERCCL, "cannot dlopen librccl: %s", dlerror());
    auto bind = [h](const char* name, auto& fn) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name)); };
    bind("ncclGetUniqueId", g_rccl.getid);
    bind("ncclCommInitRank", g_rccl.init);
    bind("ncclAllReduce", g_rccl.allreduce);
    bind("ncclReduce", g_rccl.reduce);
    bind("ncclReduceScatter", g_rccl.reducescatter);
    bind("ncclCommDestroy", g_rccl.destroy);
    bind("ncclGetErrorString", g_rccl.errstr);
    bind("ncclGetVersion", g_rccl.getversion);
    bind("ncclGroupStart", g_rccl.group_start);
    bind("ncclGroupEnd", g_rccl.group_end);
    if (!g_rccl.getid || !g_rccl.init || !g_rccl.allreduce || !g_rccl.reduce || !g_rccl.destroy)
        return fail(DM_ERCCL, "librccl is missing required symbols");
    if (g_rccl.getversion && g_rccl.getversion(&g_rccl.version) != ncclSuccess) g_rccl.version = 0;
    {   // the file the loader really mapped (librccl.so.1 -> /opt/rocm-7.2.0/lib/librccl.so.1.0...): what a run's log should name
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(g_rccl.getid), &info) && info.dli_fname && *info.dli_fname) g_rccl.path = info.dli_fname;
    }
    g_rccl.h = h;
    return DM_OK;
}
const char* rccl_err(ncclResult_t e) { return g_rccl.errstr ? g_rccl.errstr(e) : "error"; }
}  // namespace

struct dm_comm {
    int device = 0, rank = 0, nranks = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double* d_scalar = nullptr;    // one device double for barrier / max
    int64_t reduces = 0;           // collectives issued (dm_comm_stats)
    int64_t reduced_bytes = 0;
};

// Which collective library this process bound (loading it if that has not happened yet): the file the loader mapped and ncclGetVersion's code
// (e.g. 22207 = 2.22.7; 0 if the library has no such entry).  What `detect --gpus N` prints at the start of a multi-GPU run.
int dm_rccl_info(char* path, int path_len, int* version) {
    int rc = load_rccl();
    if (rc) return rc;
    if (path && path_len > 0) {
        std::strncpy(path, g_rccl.path.c_str(), size_t(path_len) - 1);
        path[path_len - 1] = 0;
    }
    if (version) *version = g_rccl.version;
    return DM_OK;
}

int dm_rccl_unique_id(void* out128) {
    if (!out128) return fail(DM_EINVAL, "null id buffer");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t e = g_rccl.getid(&id);
    if (e != ncclSuccess) return fail(DM_ERCCL, "ncclGetUniqueId: %s", rccl_err(e));
    std::memcpy(out128, &id, 128);
    return DM_OK;
}

dm_comm* dm_comm_create(int device, const void* unique_id128, int rank, int nranks) {
    if (!unique_id128 || nranks < 1 || rank < 0 || rank >= nranks) {
        fail(DM_EINVAL, "bad communicator arguments (rank %d of %d)", rank, nranks);
        return nullptr;
    }
    if (load_rccl() != DM_OK) return nullptr;
    dm_comm* c = new (std::nothrow) dm_comm();
    if (!c) {
        fail(DM_ENOMEM, "out of host memory");
        return nullptr;
    }
    c->device = device;
    c->rank = rank;
    c->nranks = nranks;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, 128);
    bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc(&c->d_scalar, sizeof(double)) == hipSuccess;
    if (!ok) {
        fail(DM_EDEVICE, "communicator setup on device %d failed: %s", device, hipGetErrorString(hipGetLastError()));
        dm_comm_destroy(c);
        return nullptr;
    }
    ncclResult_t e = g_rccl.init(&c->comm, nranks, id, rank);       // collective over all ranks: every rank calls it once
    if (e != ncclSuccess) {
        fail(DM_ERCCL, "ncclCommInitRank(rank %d of %d): %s", rank, nranks, rccl_err(e));
        c->comm = nullptr;
        dm_comm_destroy(c);
        return nullptr;
    }
    return c;
}

void dm_comm_destroy(dm_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) g_rccl.destroy(c->comm);
    (void)hipFree(c->d_scalar);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int dm_comm_rank(const dm_comm* c) { return c ? c->rank : -1; }
int dm_comm_size(const dm_comm* c) { return c ? c->nranks : 0; }

int dm_comm_stats(const dm_comm* c, int64_t* collectives, int64_t* bytes) {
    if (!c) return fail(DM_EINVAL, "null communicator");
    if (collectives) *collectives = c->reduces;
    if (bytes) *bytes = c->reduced_bytes;
    return DM_OK;
}

// max over ranks of *value (in place); with value == NULL a plain barrier.  Host-synchronous.
int dm_comm_max_f64(dm_comm* c, double* value) {
    if (!c) return fail(DM_EINVAL, "null communicator");
    HIP_TRY(hipSetDevice(c->device));
    double v = value ? *value : 0.0;
    HIP_TRY(hipMemcpyAsync(c->d_scalar, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    ncclResult_t e = g_rccl.allreduce(c->d_scalar, c->d_scalar, 1, ncclFloat64, ncclMax, c->comm, c->stream);
    if (e != ncclSuccess) return fail(DM_ERCCL, "ncclAllReduce(max): %s", rccl_err(e));
    HIP_TRY(hipMemcpyAsync(&v, c->d_scalar, sizeof v, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (value) *value = v;
    return DM_OK;
}
int dm_comm_barrier(dm_comm* c) { return dm_comm_max_f64(c, nullptr); }

// In-place sum of touch|cov|mod over the ranks of `c` (int32, order independent -> the BED is the same for any sharding):
// root >= 0: ncclReduce, the result is valid on `root` only (what the final per-position summary needs);
// root < 0 : ncclAllReduce, valid everywhere.  All ranks must call it for summaries of the same length, in the same order.
int dm_summary_reduce(dm_summary* s, dm_comm* c, int root) {
    if (!s || !c) return fail(DM_EINVAL, "null summary / communicator");
    if (root >= c->nranks) return fail(DM_EINVAL, "root %d outside the %d ranks", root, c->nranks);
    if (s->device != c->device) return fail(DM_EINVAL, "summary on device %d, communicator on device %d", s->device, c->device);
    HIP_TRY(hipSetDevice(s->device));
    if (s->follow) HIP_TRY(hipStreamSynchronize(s->follow->stream));   // adds queued on the classifier's stream
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t count = size_t(3) * s->length;
    ncclResult_t e = root >= 0 ? g_rccl.reduce(s->d_counts, s->d_counts, count, ncclInt32, ncclSum, root, c->comm, c->stream)
                      : g_rccl.allreduce(s->d_counts, s->d_counts, count, ncclInt32, ncclSum, c->comm, c->stream);
    if (e != ncclSuccess) return fail(DM_ERCCL, "%s: %s", root >= 0 ? "ncclReduce" : "ncclAllReduce", rccl_err(e));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++c->reduces;
    c->reduced_bytes += int64_t(count) * 4;
    return DM_OK;
}

// Grow the counters to `new_length` positions (new positions zero).  A worker that learns contig lengths from its reads
// sizes the counters as it goes; before a reduce all ranks grow to the common length.
int dm_summary_grow(dm_summary* s, int64_t new_length) {
    if (!s) return fail(DM_EINVAL, "null summary");
    if (new_length <= s->length) return DM_OK;
    if (new_length > (int64_t(1) << 33)) return fail(DM_EINVAL, "bad contig length %lld", (long long)new_length);
    HIP_TRY(hipSetDevice(s->device));
    if (s->follow) HIP_TRY(hipStreamSynchronize(s->follow->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    int* nd = nullptr;
    if (hipMalloc(&nd, sizeof(int) * (3 * new_length + dm_summary::SLACK)) != hipSuccess) {
        (void)hipGetLastError();
        return fail(DM_ENOMEM, "cannot grow the summary to %lld positions", (long long)new_length);
    }
    hipError_t e = hipMemsetAsync(nd, 0, sizeof(int) * (3 * new_length + dm_summary::SLACK), s->stream);
    for (int k = 0; k < 3 && e == hipSuccess; ++k)
        e = hipMemcpyAsync(nd + k * new_length, s->d_counts + k * s->length, sizeof(int) * s->length, hipMemcpyDeviceToDevice, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) {
        (void)hipFree(nd);                         // the old counters stay valid
        return fail(DM_EDEVICE, "growing the summary to %lld positions failed: %s", (long long)new_length, hipGetErrorString(e));
    }
    HIP_TRY(hipFree(s->d_counts));
    s->d_counts = nd;
    s->length = new_length;
    s->slice_count = -1;
    return DM_OK;
}

// Scatter form of the merge: positions are cut into nranks slices of chunk = ceil(length / nranks); after the call
// rank r holds the sums over all ranks of touch | cov | mod for its slice [r * chunk, min(length, (r + 1) * chunk)) and formats
// that part of the BED itself - no rank fetches or formats a whole contig.  One ncclReduceScatter per counter array (int32 sum).
int dm_summary_reduce_scatter(dm_summary* s, dm_comm* c, int64_t* first, int64_t* count) {
    if (!s || !c) return fail(DM_EINVAL, "null summary / communicator");
    if (s->device != c->device) return fail(DM_EINVAL, "summary on device %d, communicator on device %d", s->device, c->device);
    if (c->nranks > dm_summary::SLACK) return fail(DM_EINVAL, "reduce-scatter over %d ranks (at most %lld)", c->nranks, (long long)dm_summary::SLACK);
    if (!g_rccl.reducescatter) return fail(DM_ERCCL, "librccl has no ncclReduceScatter");
    HIP_TRY(hipSetDevice(s->device));
    if (s->follow) HIP_TRY(hipStreamSynchronize(s->follow->stream));   // adds queued on the classifier's stream
    HIP_TRY(hipStreamSynchronize(s->stream));
    const int64_t chunk = (s->length + c->nranks - 1) / c->nranks;
    if (chunk > s->slice_chunk) {
        (void)hipFree(s->d_slice);
        s->d_slice = nullptr;
        s->slice_chunk = 0;
        HIP_TRY(hipMalloc(&s->d_slice, sizeof(int) * 3 * chunk));
        s->slice_chunk = chunk;
    }
    // the three counter arrays as ONE group where librccl has the group calls (one fused launch per rank instead of three)
    const bool grouped = g_rccl.group_start && g_rccl.group_end;
    if (grouped) {
        ncclResult_t e = g_rccl.group_start();
        if (e != ncclSuccess) return fail(DM_ERCCL, "ncclGroupStart: %s", rccl_err(e));
    }
    ncclResult_t err = ncclSuccess;
    for (int k = 0; k < 3 && err == ncclSuccess; ++k) {
        // the send buffer of array k is read up to nranks * chunk <= length + nranks - 1 positions: past `length` that is the head of
        // the next array (or the allocation's slack) - sums of positions that do not exist, never looked at
        err = g_rccl.reducescatter(s->d_counts + k * s->length, s->d_slice + k * s->slice_chunk, size_t(chunk), ncclInt32, ncclSum, c->comm, c->stream);
    }
    if (grouped) {
        const ncclResult_t e = g_rccl.group_end();     // always closed, also after a failed call inside the group
        if (err == ncclSuccess) err = e;
    }
    if (err != ncclSuccess) return fail(DM_ERCCL, "ncclReduceScatter: %s", rccl_err(err));
    HIP_TRY(hipStreamSynchronize(c->stream));
    ++c->reduces;
    c->reduced_bytes += int64_t(3) * s->length * 4;
    s->slice_first = std::min<int64_t>(s->length, int64_t(c->rank) * chunk);
    s->slice_count = std::min<int64_t>(s->length, int64_t(c->rank + 1) * chunk) - s->slice_first;
    if (first) *first = s->slice_first;
    if (count) *count = s->slice_count;
    return DM_OK;
}

// this rank's slice after dm_summary_reduce_scatter: `count` int32 each (any may be NULL)
int dm_summary_fetch_slice(dm_summary* s, int32_t* touch, int32_t* cov, int32_t* mod) {
    if (!s) return fail(DM_EINVAL, "null summary");
    if (s->slice_count < 0) return fail(DM_ESTATE, "no reduce-scatter result: call dm_summary_reduce_scatter first");
    HIP_TRY(hipSetDevice(s->device));
    int rc0 = summary_check_oob(s);
    if (rc0) return rc0;
    const size_t bytes = sizeof(int) * size_t(s->slice_count);
    if (bytes == 0) return DM_OK;
    if (touch) HIP_TRY(hipMemcpy(touch, s->d_slice, bytes, hipMemcpyDeviceToHost));
    if (cov) HIP_TRY(hipMemcpy(cov, s->d_slice + s->slice_chunk, bytes, hipMemcpyDeviceToHost));
    if (mod) HIP_TRY(hipMemcpy(mod, s->d_slice + 2 * s->slice_chunk, bytes, hipMemcpyDeviceToHost));
    return DM_OK;
}

}  // extern "C"

#include "signal.hip.inc"
#include "readmap.inc"
#include "rowsbatch.inc"
#include "bedtext.inc"
