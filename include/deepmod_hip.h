/*
 * deepmod_hip.h — C ABI of libdeepmod_hip.so: the MI355X (gfx950) implementation of DeepMod's
 * per-read BiLSTM modification-calling hot path and its per-position summary.
 *
 * Plain pointers and sizes only.  Every function returns 0 on success or a negative DM_E* code;
 * the message of the last failure on the calling thread is available from dm_last_error().
 * A handle is single-owner (one process per GPU, as the reference runs one TF session per
 * process: bin/DeepMod_scripts/myDetect.py:948-956, :1177-1180); calls are synchronous on return
 * unless the name says _async.  The caller owns every buffer it passes; the library keeps no host
 * pointer after a call returns.
 *
 * Reference interfaces replaced (all under /root/reference/):
 *   dm_model_create / dm_model_destroy
 *        myMultiBiRNN.mCreateSession            bin/DeepMod_scripts/myMultiBiRNN.py:21-91
 *        tf.Session + Saver.restore             bin/DeepMod_scripts/myDetect.py:950-956
 *   dm_predict_windows
 *        sess.run([mfpred], feed_dict={X, Y})   bin/DeepMod_scripts/myDetect.py:814-820
 *        (graph: myMultiBiRNN.py:38-61; prediction=softmax :59, mfpred=argmax :61)
 *   dm_predict_read
 *        window assembly tx[mind-10:mind+11]    bin/DeepMod_scripts/myDetect.py:794-803
 *        + the sess.run above, fused on device
 *   dm_summary_create / _add / _add_read / _fetch / _destroy
 *        sum_handler's per-base accumulation    bin/DeepMod_scripts/myDetect.py:1089-1100
 *   dm_comm_create / dm_summary_reduce / dm_comm_destroy
 *        cross-process additive merge           DeepMod_tools/sum_chr_mod.py:47-52
 *        (the reference merges BED files of separate runs; here: one RCCL reduce per contig x strand)
 *   dm_cluster_create / _predict / _destroy
 *        cluster MLP sess.run([output])         DeepMod_tools/hm_cluster_predict.py:94-103, :158-164
 */
#ifndef DEEPMOD_HIP_H
#define DEEPMOD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_OK 0
#define DM_EINVAL (-1)   /* bad argument */
#define DM_EDEVICE (-2)  /* HIP runtime error / no gfx950 device */
#define DM_ENOMEM (-3)   /* allocation failed */
#define DM_ESTATE (-4)   /* handle in wrong state */
#define DM_ERCCL (-5)    /* RCCL could not be loaded or failed */
#define DM_ERANGE (-6)   /* DM_PREC_F16X3 met an input it cannot represent; the call's results are invalid, repeat with DM_PREC_F32 */

/* model geometry fixed by the shipped checkpoints */
#define DM_NFEAT 7
#define DM_HIDDEN 100
#define DM_WINDOW 21
#define DM_LAYERS 3
#define DM_WEIGHT_FLOATS 408402

/* dm_model_set_option keys */
#define DM_OPT_PROFILE 1   /* 1: bracket every kernel launch with HIP events on the model's stream */
#define DM_OPT_PRECISION 2 /* DM_PREC_*; default DM_PREC_F16X3 */
#define DM_OPT_ASYNC 3     /* 1: dm_predict_* with device-resident buffers return after enqueue on the model's stream;
                              wait with dm_model_sync.  Default 0: every call is synchronous on return. */
#define DM_OPT_RESERVED_CUS 4 /* n in [0, CUs / 2]: the classifier's persistent grid uses CUs - n workgroups (one per CU), so that
                              small kernels of OTHER streams (the signal stage of the streaming worker) run beside a classifier
                              launch instead of waiting for it to drain.  Default 0. */
#define DM_OPT_F16X3_SHAPE 5  /* MFMA shape of the DM_PREC_F16X3 / DM_PREC_F16I8 kernels.  16 = the product (v_mfma_f32_16x16x32_f16 and
                              v_mfma_i32_16x16x64_i8, csrc/lstm_f16q.hip.inc); 32 (v_mfma_f32_32x32x16_f16 / v_mfma_i32_32x32x32_i8: the kernels of
                              rounds 2-3, ~8 % / ~5 % slower, tools/experiments/f16s since round 6) exists only in an experiment build
                              (DM_WITH_F16S=1, DM_INFO_HAS_F16S) - a product library answers 32 with DM_EINVAL.  Same arithmetic, another
                              summation order: results differ in the last bits.  A model whose DM_PREC_F16I8 was selected by
                              dm_model_calibrate_i8 goes back to DM_PREC_F16X3 when the shape changes (the gate saw the other kernel). */
#define DM_PREC_F32 0      /* fp32 MFMA (v_mfma_f32_16x16x4_f32): fp32 products, the TF graph's own arithmetic */
#define DM_PREC_F16X3 1    /* split-f16 MFMA: every fp32 operand = hi + lo f16, 3 products per fp32 product, fp32
                              accumulation; max |dp| vs the oracle 1e-6 .. 2e-6 (tolerance of the path: 1e-4), ~3x faster.
                              Step-major kernel: weights and feature rows in, logits out, the state of the three layers never
                              leaves the chip: csrc/lstm_f16q.hip.inc on v_mfma_f32_16x16x32_f16.
                              Range contract (nothing is clamped silently):
                                * weights: every kernel / bias value times its exponent scale (<= 2.886) must be a finite f16
                                  (|w| <~ 22,700).  A model that violates this is created with DM_PREC_F32 as its default and
                                  dm_model_set_option(DM_OPT_PRECISION, DM_PREC_F16X3) returns DM_EINVAL.
                                * inputs: features 0..5 |x| <= 65504; feature 6 (event length, a raw sample count: reference
                                  myDetect.py:894-900) |x| <= 65504 * 2^k, k = DM_INFO_F16_LENGTH_SHIFT (10 for ordinary weights):
                                  beyond 65504 it is fed as x * 2^-k against a weight row stored x 2^k (exact rescale).
                                  An input outside these bounds (or NaN) makes the call fail with DM_ERANGE - synchronous
                                  calls on return, DM_OPT_ASYNC calls at the next dm_model_sync. */
#define DM_PREC_F16X3_ROLES 2 /* EXPERIMENT (round 4, tools/experiments/f16r): the arithmetic of DM_PREC_F16X3, bit for bit, with the work of a
                              SIMD split between a matrix wave and a cell wave (two waves per SIMD, accumulators and h handed over
                              through LDS): 6 % fewer cycles, the same time per launch - the part runs this path at its power limit and
                              gives a saved cycle back as a lower clock (profiles/r04/README.md).  Not part of the product build:
                              selectable only in a library built with -DDM_WITH_F16X3_ROLES (DM_INFO_HAS_F16X3_ROLES), otherwise
                              refused with DM_EINVAL. */
#define DM_PREC_F16I8 3    /* OPT-IN: the step-major kernel with hi*hi in f16 and BOTH cross terms of every product as one int8 MFMA
                              (v_mfma_i32_16x16x64_i8 since round 5, v_mfma_i32_32x32x32_i8 with DM_OPT_F16X3_SHAPE = 32; int32
                              accumulation, folded into the fp32 pre-activations per tile): 2 issued matrix units per product instead
                              of 3, 8-9 % less time per window than the default.  Operands carry ~19 bits instead
                              of 22.  REDUCED PRECISION: on 10^6 windows at weight scale 4 the worst window is 1.1e-4 from the fp32
                              graph, 2 windows exceed the path's 1e-4 tolerance and 99.99 % are below 6.5e-5 (DM_PREC_F16X3: worst
                              9e-6); at weight scale 1: 6e-6.  Documented bound 2e-4; classes equal wherever p1 is further than that
                              from 0.5.  Same range contract as DM_PREC_F16X3 (the raw features never ride the int8 product).
                              Never selected by default - neither by the library nor by the command line; opt-in per model through
                              dm_model_calibrate_i8 below (the command line: DEEPMOD_PRECISION=auto). */
/* dm_model_get_info keys */
#define DM_INFO_PRECISION 1          /* DM_PREC_* in effect */
#define DM_INFO_F16_REPRESENTABLE 2  /* 1 if the weights fit DM_PREC_F16X3 */
#define DM_INFO_F16_LENGTH_SHIFT 3   /* k above */
#define DM_INFO_DEVICE 4
#define DM_INFO_HAS_F16X3_ROLES 5    /* 1 if the library was built with the wave-pair experiment kernel */
#define DM_INFO_HAS_F16S 6           /* 1 if the library was built with the 32x32x16 kernels of rounds 2-3 (experiment builds: DM_WITH_F16S=1) */

typedef struct dm_model dm_model;
typedef struct dm_summary dm_summary;

const char* dm_last_error(void);
const char* dm_version(void);

/* "experiment=0 switches=0" for the product; "experiment=1 ..." for a library built with -DDM_EXPERIMENT (tools/ablate.py: timing-only
 * kernels).  Any ablation macro without -DDM_EXPERIMENT is a compile error.  dm_version() also names the HIP / clang version the
 * library was compiled with (the hand-scheduled kernels are validated per compiler).  No reference counterpart. */
const char* dm_build_flags(void);

/* number of visible gfx950 devices (0 if none / HIP unusable) */
int dm_device_count(void);
/* PCI bus id ("0000:05:00.0") of HIP device `device` into buf (>= 13 bytes): the key under /sys/bus/pci/devices/ where the driver
 * publishes the device's socket power and clocks (deepmod_amd/powerlog.py, bench.py roofline.power).  No reference counterpart. */
int dm_device_pci_bus_id(int device, char* buf, int len);

/*
 * Build a model on `device` from the canonical flat weight blob (DM_WEIGHT_FLOATS floats, host):
 *   for d in (fw, bw): for l in 0..2: kernel[K_l][400] row-major (K_0 = 107, else 200; columns
 *   i|j|f|o as TF BasicLSTMCell), bias[400];  then head W[200][2], head b[2].
 * The forget bias (+1.0) is NOT pre-added by the caller.  Returns NULL on failure.
 */
dm_model* dm_model_create(int device, const float* weights, size_t n_floats, int n_feat, int hidden,
                          int window, int layers);
void dm_model_destroy(dm_model* m);
int dm_model_set_option(dm_model* m, int key, int64_t value);
int dm_model_get_info(dm_model* m, int key, int64_t* value);
/*
 * Load-time calibration gate of the opt-in int8 mode (round 4; no reference counterpart - the reference runs fp32 TensorFlow kernels,
 * myMultiBiRNN.py:38-61).  Classifies n_windows synthetic windows generated on the device (the same on every box; alternating batches
 * of 65,536: the BASELINE configs[1] distribution, and - round 5 - a read-shaped tail draw: event means over the whole +-5 clip range
 * with 6 % exactly on the clip, standard deviations up to 9x, event lengths log-uniform 1 .. 30,000 samples) with DM_PREC_F32 and with
 * DM_PREC_F16I8 and reports the largest |p(f32) - p(f16i8)| in *max_abs_dp.  If that is <= bound and the model currently runs
 * DM_PREC_F16X3, DM_PREC_F16I8 becomes its precision.  *selected = 1 if the model runs DM_PREC_F16I8 after the call (also when it
 * already did), else 0; a model that does not run it afterwards does not keep the int8 weight pack.  What the mode's error is depends
 * on the weights: 1.2e-5 in the tail of 10^6 windows for weights with trained statistics, 7e-5 - 1.1e-4 for U(-a, a) kernels at scale 4
 * (profiles/r04/i8_tail.txt) - hence a gate per model, never a global default.  10^6 windows take ~0.12 s.  Synchronises the model's stream.
 */
int dm_model_calibrate_i8(dm_model* m, int64_t n_windows, double bound, double* max_abs_dp, int* selected);

/*
 * Classify n windows.  x: [n][21][7] fp32 C-contiguous, host OR device memory (detected).
 * prob: [n][2] fp32 or NULL; cls: [n] u8 (argmax, ties -> 0) or NULL; each may be host or device.
 * n == 0 is a no-op.
 */
int dm_predict_windows(dm_model* m, const float* x, int64_t n, float* prob, uint8_t* cls);

/*
 * Classify `count` consecutive windows of one read: window i is rows[first+i-10 .. first+i+10]
 * of the per-read feature matrix rows [m][7] (fp32, host or device); the caller guarantees
 * first-10 >= 0 and first+count+10 <= m (DeepMod pads 100 rows each side).  Windows are
 * assembled on the device.
 */
int dm_predict_read(dm_model* m, const float* rows, int64_t m_rows, int64_t first, int64_t count,
                    float* prob, uint8_t* cls);

/*
 * Classify `count` windows of the feature-row matrix picked by their CENTRE rows: window i = rows[centre[i]-10 .. centre[i]+10]
 * (10 <= centre[i] < m_rows - 10; checked for host arrays only).  The streaming worker classifies only the windows centred on
 * a base of interest: sum_handler tests refbase == Base before it looks at mod_pred (myDetect.py:1091-1100), so the class of
 * every other window never reaches the BED - on E. coli 5mC that is 3 of 4 windows.  The reference
 * computes them because it stores per-read tables; `--storePred 1` (dm_predict_read) still does.  prob / cls: [count].
 */
int dm_predict_read_at(dm_model* m, const float* rows, int64_t m_rows, const int32_t* centre, int64_t count, float* prob,
                       uint8_t* cls);

/* block until all work queued on the model's stream has finished; reports a pending DM_ERANGE of asynchronous launches */
int dm_model_sync(dm_model* m);

/* profiling (DM_OPT_PROFILE=1): summed HIP-event time and launch count of the BiLSTM kernel since
 * the last reset; dm_profile_get synchronises the stream first. */
int dm_profile_reset(dm_model* m);
int dm_profile_get(dm_model* m, double* kernel_ms, int64_t* launches, int64_t* windows);

/* device-memory helpers so a host language without a HIP binding can keep inputs resident */
void* dm_device_alloc(int device, size_t bytes);
int dm_device_free(int device, void* p);
int dm_memcpy_h2d(int device, void* dst, const void* src, size_t bytes);
int dm_memcpy_d2h(int device, void* dst, const void* src, size_t bytes);
/* host -> device copy queued on the model's stream (ordered with its launches; with DM_OPT_ASYNC a worker refills its
 * staging buffers without waiting for the device).  src must stay valid until the next dm_model_sync. */
int dm_model_h2d_async(dm_model* m, void* dst, const void* src, size_t bytes);
/* The same copy on the model's copy stream: it does NOT wait for the launches already queued (it runs while they compute) and
 * every launch queued after this call waits for it.  The caller guarantees that no queued launch touches dst - in the staging
 * scheme below: dm_model_wait_mark of the set has returned.  src must stay valid until the set's next marker has passed. */
int dm_model_h2d_ahead(dm_model* m, void* dst, const void* src, size_t bytes);
/* Page-locked host staging memory (hipHostMalloc) and stream markers for a pipelined worker: batch k is copied into staging
 * set k % N while the device still works on batch k - 1; dm_model_mark(m, i) records a marker on the model's stream after the
 * launches that read set i, dm_model_wait_mark(m, i) blocks the host until that marker has passed (at once if it was never
 * recorded).  DM_ERANGE is tracked per marker: wait_mark(i) reports exactly the launches queued between the marker recorded before i and
 * marker i - a later batch still in flight is neither reported early nor cleared; dm_model_sync reports everything outstanding.  i in [0, DM_MARKS).  No reference counterpart: the reference feeds numpy arrays to session.run
 * (myDetect.py:796-822). */
#define DM_MARKS 8
void* dm_host_alloc(int device, size_t bytes);
int dm_host_free(int device, void* p);
int dm_model_mark(dm_model* m, int i);
int dm_model_wait_mark(dm_model* m, int i);

/* ------------------------------------------------------------------ per-position summary -- */
/*
 * Dense per-position counters for one contig x strand of `length` reference positions:
 * touch (rows with refbase == Base: the reference creates the BED key before testing
 * readbase, myDetect.py:1093-1094), cov (readbase != '-'), mod (mod_pred == 1), all int32.
 */
dm_summary* dm_summary_create(int device, int64_t length);
void dm_summary_destroy(dm_summary* s);
int64_t dm_summary_length(const dm_summary* s);

/* flags bit0: refbase == Base (and not '-','N','n'); bit1: readbase != '-'; bit2: mod_pred == 1.
 * pos / flags: [n], host or device. */
int dm_summary_add(dm_summary* s, const int64_t* pos, const uint8_t* flags, int64_t n);
/* same, with bit2 taken from the classifier output cls[i] == 1 (the per-base scatter of
 * myDetect.py:829-831 fused with the accumulation); flags' own bit2 is ignored. */
int dm_summary_add_classified(dm_summary* s, const int64_t* pos, const uint8_t* flags, const uint8_t* cls,
                              int64_t n);
int dm_summary_sync(dm_summary* s);

/* grow the counters to new_length positions (existing counts kept, new positions zero); no-op if not larger */
int dm_summary_grow(dm_summary* s, int64_t new_length);

/* ---- multi-GPU: one process per GPU, a persistent RCCL communicator, one integer reduce per contig x strand -----
 * Reads shard across processes with no data-path collective; the only exchange is the additive merge of the
 * per-position counters at the end (what the reference does across runs with DeepMod_tools/sum_chr_mod.py:47-52).
 *   dm_rccl_unique_id   rank 0 obtains 128 bytes and hands them to the other ranks out of band (deepmod_amd/comm.py:
 *                       a file in the run's output folder; any byte channel works)
 *   dm_comm_create      collective: every rank calls it once with the same id; the communicator lives until
 *                       dm_comm_destroy and serves any number of reduces
 *   dm_summary_reduce   in-place int32 sum of touch|cov|mod over all ranks.  root >= 0: ncclReduce, result valid on
 *                       `root` only; root < 0: ncclAllReduce.  All ranks call it with summaries of equal length, in
 *                       the same order.  Integer sums are order independent: the BED is the same for any GPU count.
 *   dm_comm_max_f64 / dm_comm_barrier   small host-synchronous helpers (timing max over ranks, rendezvous)
 *   dm_comm_stats       collectives issued and bytes reduced by this rank so far
 * The collective library is bound at the first call (dlopen): librccl by soname, or the file DEEPMOD_RCCL_LIBRARY names - a site
 * build of RCCL, or the shared-memory test transport tests/shim/shmccl.cpp that lets several ranks share one device (RCCL refuses
 * that), which is how the N > 1 calls below are tested on a one-GPU box.  A path that cannot be loaded is an error (DM_ERCCL). */
typedef struct dm_comm dm_comm;
int dm_rccl_unique_id(void* out128);
/* the collective library this process bound (path the loader mapped, ncclGetVersion code; loads it on first use) */
int dm_rccl_info(char* path, int path_len, int* version);
dm_comm* dm_comm_create(int device, const void* unique_id128, int rank, int nranks);
void dm_comm_destroy(dm_comm* c);
int dm_comm_rank(const dm_comm* c);
int dm_comm_size(const dm_comm* c);
int dm_comm_barrier(dm_comm* c);
int dm_comm_max_f64(dm_comm* c, double* value);
int dm_comm_stats(const dm_comm* c, int64_t* collectives, int64_t* bytes);
int dm_summary_reduce(dm_summary* s, dm_comm* c, int root);
/*   dm_summary_reduce_scatter   the merge that scales (the reference's per-process BED files + sum_chr_mod.py:47-63 turned
 *                       inside out): positions are cut into nranks slices of ceil(length / nranks); afterwards THIS rank holds the
 *                       all-rank sums of its slice [*first, *first + *count) (count 0 for a rank past the end) - one ncclReduceScatter
 *                       (int32 sum) per counter array.  Each rank then fetches and formats only its slice (dm_summary_fetch_slice,
 *                       dm_bed_format_at) and the parts are concatenated in rank order.  Collective, nranks <= 64.
 *   dm_summary_fetch_slice      that slice: *count int32 each (any may be NULL) */
int dm_summary_reduce_scatter(dm_summary* s, dm_comm* c, int64_t* first, int64_t* count);
int dm_summary_fetch_slice(dm_summary* s, int32_t* touch, int32_t* cov, int32_t* mod);

/* copy counters to host arrays of `length` int32 each (any may be NULL) */
int dm_summary_fetch(dm_summary* s, int32_t* touch, int32_t* cov, int32_t* mod);
/* raw device pointers (3 * length int32: touch | cov | mod) for callers that run their own collective */
void* dm_summary_device_ptr(dm_summary* s);
/* Enqueue this summary's device-resident adds on model m's stream, i.e. in order with its classifier launches (with
 * DM_OPT_ASYNC the host then never waits between classify and accumulate, and the out-of-range check of those adds is
 * reported by the next dm_summary_sync / dm_summary_fetch instead); m = NULL detaches.  Detach or destroy the summary
 * before destroying the model. */
int dm_summary_follow(dm_summary* s, dm_model* m);

/* BED text of one contig x strand from the (host) counters, byte for byte what sum_handler writes (myDetect.py:1107-1120):
 * one line per position with touch > 0, "<chr> <pos> <pos+1> <Base> <min(cov,1000)> <strand> <pos> <pos+1> 0,0,0 <cov> <pct> <mod> \n".
 * Returns the length of the text.  With out == NULL or cap below the safe bound (lines * (strlen(chrom) + 128)) nothing is written and
 * that bound is returned: allocate it and call again. */
int64_t dm_bed_format(const char* chrom, char strand, char base, const int32_t* touch, const int32_t* cov, const int32_t* mod,
                      int64_t length, char* out, int64_t cap);
/* the same for counters of positions [first_pos, first_pos + length): a rank's slice of the text (myDetect.py:1112-1120 is sorted by
 * position, so the ranks' parts concatenated in rank order are the file) */
int64_t dm_bed_format_at(const char* chrom, char strand, char base, int64_t first_pos, const int32_t* touch, const int32_t* cov,
                         const int32_t* mod, int64_t length, char* out, int64_t cap);

/* ------------------------------------------------------------- CpG cluster second stage -- */
/*
 * MLP 14 -> 100 (relu) -> 20 (relu) -> 1 (sigmoid) of DeepMod_tools/hm_cluster_predict.py:94-103,:161
 * (graph nodes X, W_1/b_1, W_2/b_2, W_O/b_O, output; dropout with keep_prob = 1 is the identity).
 * weights: 3541 floats = W_1[14][100], b_1[100], W_2[100][20], b_2[20], W_O[20][1], b_O[1] (host).
 * x: [n][14] fp32, out: [n] fp32; host or device pointers.
 */
#define DM_CLUSTER_WEIGHT_FLOATS 3541
typedef struct dm_cluster dm_cluster;
dm_cluster* dm_cluster_create(int device, const float* weights, size_t n_floats);
void dm_cluster_destroy(dm_cluster* c);
int dm_cluster_predict(dm_cluster* c, const float* x, int64_t n, float* out);

/* ---- raw-signal normalisation + per-event statistics ------------------------------------
 * Replaces myDetect.py:266-282 (mnormalized: median / MAD shift-scale over the event-covered slice, second
 * median / MAD, clip to +-5 MAD, round to 3 decimals) and :332-343 (per-event round(np.mean, 3), round(np.std, 3)
 * into the '<f4' fields of m_event).  Results are bit-identical to numpy's (same float64 operations in the same
 * order, including np.add.reduce's 8,192-element buffers and 8-lane pairwise sums).
 *   raw        int16 DAC samples [n_raw] (FAST5 Raw/Reads/.../Signal), host or device (device: 16-byte aligned)
 *   ev_start / ev_length  uint64 [n_events] host arrays (m_event['start'], m_event['length'])
 *   ev_mean / ev_stdv     float [n_events] host outputs; NaN for an event whose slice is empty
 *   norm6      optional double[6]: mshift, mscale, read_med, read_mad, lower_lim, upper_lim
 *   first_empty optional: index of the first event with an empty slice (the reference stops its loop there,
 *              :334-340), n_events if none
 *   normalized optional double [n_raw] (host or device): the normalised signal itself */
typedef struct dm_signal dm_signal;
dm_signal* dm_signal_create(int device);
void dm_signal_destroy(dm_signal* s);
int dm_signal_event_stats(dm_signal* s, const int16_t* raw, int64_t n_raw, const uint64_t* ev_start,
                          const uint64_t* ev_length, int64_t n_events, float* ev_mean, float* ev_stdv, double* norm6,
                          int64_t* first_empty, double* normalized);

/* Many reads per call (a typical 120 k-sample read is launch / latency bound when it travels alone): the reads' samples back to
 * back in raw (read r = raw[raw_off[r] .. raw_off[r+1])), event tables back to back (events of read r = [ev_off[r], ev_off[r+1]),
 * starts relative to the read's first sample); norm6 [n_reads][6] and first_empty [n_reads] optional.  Host arrays only.
 * Bit-identical to n_reads calls of dm_signal_event_stats. */
int dm_signal_event_stats_batch(dm_signal* s, int64_t n_reads, const int16_t* raw, const int64_t* raw_off, const uint64_t* ev_start,
                                const uint64_t* ev_length, const int64_t* ev_off, float* ev_mean, float* ev_stdv, double* norm6,
                                int64_t* first_empty);

/* The RESIDENT form (round 6; SURVEY 8f1 + 8f3: "fuse get_Feature's output rows straight into the kernel"): the per-event statistics never leave the
 * device.  dm_signal_plan_batch (host only, no device call) makes the checks of the batched call and returns first_empty [n_reads] - all a feeder needs
 * from the signal stage before it walks its alignments (myDetect.py:334-340 cuts the event table there).  dm_signal_event_stats_device runs the same
 * four kernels and writes d_ev3 [n_events][3] = (mean, stdv, length) of every merged event of the batch into a device block of the caller - the three
 * values get_Feature (:892-900) copies into a feature row - with the basecaller's values (fb_mean / fb_stdv, host, may be NULL when no read has an empty
 * event) merged in for events at or behind first_empty.  dm_rows_emit_resident's descriptors index that block and dm_rows_assemble reads it in place: no
 * D2H -> feeder -> H2D of the statistics.  The block is complete when the call returns; host arrays should be page-locked (dm_host_alloc).
 * flags (optional): bit 0 = a value outside the split-f16 kernels' range (the batch then takes DM_PREC_F32). */
int dm_signal_plan_batch(int64_t n_reads, const int64_t* raw_off, const int64_t* ev_off, const uint64_t* ev_start, const uint64_t* ev_length,
                         int64_t* first_empty);
int dm_signal_event_stats_device(dm_signal* s, int64_t n_reads, const int16_t* raw, const int64_t* raw_off, const uint64_t* ev_start,
                                 const uint64_t* ev_length, const int64_t* ev_off, const int64_t* first_empty, const float* fb_mean, const float* fb_stdv,
                                 float* d_ev3, double* norm6, int32_t* flags);

/* ---- SAM record -> per-base alignment table (host code, no GPU needed) ---------------------
 * Replaces the alignment walk of handle_record, myDetect.py:515-714: clip stripping, one row per M/I/D/N/=/X
 * position, first/last-match trimming of table and event slice, '-' strand flip + complement, the CpG gap swap.
 *   flag, pos1, cigar, readseq     fields 2, 4, 6, 10 of the SAM line (pos1 is 1-based)
 *   refseq                         the whole (upper-cased) reference sequence of RNAME
 *   n_events                       len(f5data[readk][1]) (events of the read, one per basecalled base)
 *   refbase/readbase/refbasei/readbasei   caller-allocated columns of base_map_info (dtype :660), cap_rows rows
 *   info[DM_MAP_INFO_LEN]          see DM_MAP_* below
 * Returns 0 and info[DM_MAP_STATUS] = DM_MAP_OK | DM_MAP_NO_MATCH (read skipped, :617-622) | DM_MAP_NEED_ROWS
 * (cap_rows < info[DM_MAP_N_ROWS]; nothing written), or a negative code for a CIGAR that is malformed or runs
 * past the read / reference (the reference raises IndexError there). */
#define DM_MAP_INFO_LEN 16
#define DM_MAP_STATUS 0
#define DM_MAP_N_ROWS 1
#define DM_MAP_LEFTCLIP 2            /* after the strand swap of :667 = start_clip of get_Feature / mPredict1 */
#define DM_MAP_RIGHTCLIP 3           /* = end_clip */
#define DM_MAP_EV_LO 4               /* m_event = events[EV_LO:EV_HI] after both trimming steps */
#define DM_MAP_EV_HI 5
#define DM_MAP_FIRST_MATCH_POS 6
#define DM_MAP_LAST_MATCH_POS 7
#define DM_MAP_NUM_INSERT 8
#define DM_MAP_NUM_DELETE 9
#define DM_MAP_NUM_MISMATCH 10
#define DM_MAP_STRAND 11             /* 0 '+', 1 '-' */
#define DM_MAP_POS_AFTER_CLIP 12     /* 0-based position after the left clip (region filter of :544-553) */
#define DM_MAP_EVENTS_AFTER_CLIP 13  /* len(m_event) at that point */
#define DM_MAP_OK 0
#define DM_MAP_NO_MATCH 1
#define DM_MAP_NEED_ROWS 2
int dm_map_read(int flag, int64_t pos1, const char* cigar, const char* readseq, int64_t readseq_len,
                const char* refseq, int64_t refseq_len, int64_t n_events, char* refbase, char* readbase,
                uint64_t* refbasei, uint64_t* readbasei, int64_t cap_rows, int64_t* info);

/* ---- one worker batch of reads -> the device-ready arrays of the streaming detect (host code) ------------------------------
 * What the streaming worker uploads per batch: feature rows [R][7] fp32 (the reads' matrices back to back, 100 zero rows of
 * padding on both sides of every read - myDetect.py:850-851), per row a reference position and a flag byte (dm_summary_add_classified),
 * and "extra" (position, flag) pairs of table rows that own no window.  Replaces, per batch instead of per read:
 *   dm_events_merge     getEvent, Albacore-2 'simple' branch                           myDetect.py:237-251
 *   dm_rows_add_raw     handle_record's filters + alignment walk + get_Feature         myDetect.py:497-713, :839-903
 *   dm_rows_add_packed  the reads of a feature container (arguments of mPredict1)      myDetect.py:715
 *   dm_rows_info / dm_rows_emit   window <-> base association and per-base flags        myDetect.py:794-803, :824-833, :1089-1100
 * Per-read status codes (read_info[i][0]): */
#define DM_ROWS_OK 0
#define DM_ROWS_LESS_EVENT 1      /* fewer than 50 aligned events (:702-705) */
#define DM_ROWS_INDEX_ERROR 2     /* fewer aligned table rows than aligned events (the reference raises IndexError) */
#define DM_ROWS_NO_MATCH 3        /* no matching base (:617-622) */
#define DM_ROWS_CIGAR_ERROR 4
#define DM_ROWS_FILTERED 5        /* region filter / skipped by the caller */
#define DM_ROWS_NO_REFERENCE 6
#define DM_ROWS_NOT_MATCHING 7    /* raw reads: event bases differ from the table's read bases (:868-874) */
#define DM_ROWS_INFO 8            /* int64 per read: status, contig, strand (0 '+', 1 '-'), windows, rows, table rows, mismatches, extras */
typedef struct dm_rowsbatch dm_rowsbatch;
/* Container tables come from disk: every offset table is checked against the size of the arrays it indexes (n_events, n_tx_rows,
 * n_table_rows below), contigs against n_contigs, strands against {0, 1} - a damaged file is DM_EINVAL, never an out-of-bounds read
 * (tests/test_asan_host.py drives these functions, built with -fsanitize=address, with damaged tables). */
int64_t dm_events_merge(int64_t n_reads, int64_t n_events, const int64_t* ev_off, const double* mean, const double* stdv, const uint64_t* start,
                        const uint64_t* length, const uint32_t* model_state, int32_t ms_width, const int64_t* move, int64_t* mev_off,
                        float* m_mean, float* m_stdv, uint64_t* m_start, uint64_t* m_length, char* m_base);
dm_rowsbatch* dm_rows_create(char base);
void dm_rows_destroy(dm_rowsbatch* h);
int dm_rows_add_packed(dm_rowsbatch* h, int64_t n_reads, int64_t n_tx_rows, int64_t n_table_rows, int64_t n_events, int32_t n_contigs,
                       const int64_t* row_off, const int64_t* bmi_off, const int64_t* ev_off,
                       const float* tx, const char* refbase, const char* readbase, const int64_t* refbasei, const char* evbase,
                       const int64_t* start_clip, const int64_t* end_clip, const int32_t* contig, const int32_t* strand);
int dm_rows_add_raw(dm_rowsbatch* h, int64_t n_reads, const int32_t* flag, const int64_t* pos1, const char* const* cigar,
                    const char* const* readseq, const int64_t* readseq_len, const int32_t* contig, const int32_t* ev_read,
                    const uint8_t* skip, int32_t n_contigs, const char* const* refseq, const int64_t* refseq_len,
                    int64_t n_event_reads, int64_t n_events, const int64_t* mev_off, const float* m_mean, const float* m_stdv, const uint64_t* m_length, const char* m_base,
                    const float* s_mean, const float* s_stdv, const int64_t* first_empty, int32_t n_region,
                    const int32_t* region_contig, const int64_t* region_lo, const int64_t* region_hi);
/* reads whose alignment table the caller already has (get_Feature's rows are built from the event tables at emit time) */
int dm_rows_add_mapped(dm_rowsbatch* h, int64_t n_reads, int64_t n_table_rows, int64_t n_events, int32_t n_contigs, const int64_t* bmi_off, const char* refbase, const char* readbase,
                       const int64_t* refbasei, const int64_t* start_clip, const int64_t* end_clip, const int32_t* contig,
                       const int32_t* strand, const int64_t* mev_off, const float* m_mean, const float* m_stdv, const uint64_t* m_length,
                       const char* m_base, const float* s_mean, const float* s_stdv, const int64_t* first_empty);
int64_t dm_rows_info(dm_rowsbatch* h, int64_t* n_rows, int64_t* n_pos, int64_t* n_sel, int64_t* read_info, int64_t* mism, int64_t cap_mism,
                     int64_t* n_mism);
#define DM_ROWS_GROUP 8           /* int64 per group: contig, strand, row_lo, row_hi, xlo, xhi, sel_lo, sel_hi */
int64_t dm_rows_emit(dm_rowsbatch* h, const int32_t* contig_rank, float* rows, int32_t* sel_row, int64_t* pos, uint8_t* flags,
                     int64_t* groups, int64_t cap_groups, int64_t* contig_len, int64_t n_contig_len, int32_t* in_range);
/* The DEVICE form of a batch of raw reads (round 5; SURVEY 8f1 "fuse get_Feature's rows into the kernel"): instead of rows [R][7] the host hands over
 * ev3 [E][3] = (mean, stdv, length) of the events the rows cover, code [R] (one-hot class of a row: 0..3 = A, C, G, T, 255 = none) and rdesc [reads][4]
 * = (first row, row -> event shift, first event, end event) - 13 instead of 28 bytes per row, no feature row written or range-scanned on the host -
 * and dm_rows_assemble builds the same [R][7] matrix on the device (get_Feature, myDetect.py:839-903, bit for bit: the values are copied, not computed).
 * dm_rows_device_info (after dm_rows_info): 1 if every emitted read is a raw read, with the sizes E and reads; dm_rows_emit_device: as dm_rows_emit. */
int dm_rows_device_info(dm_rowsbatch* h, int64_t* n_events, int64_t* n_reads);
int64_t dm_rows_emit_device(dm_rowsbatch* h, const int32_t* contig_rank, float* ev3, uint8_t* code, int64_t* rdesc, int32_t* sel_row, int64_t* pos,
                            uint8_t* flags, int64_t* groups, int64_t cap_groups, int64_t* contig_len, int64_t n_contig_len, int32_t* in_range);
/* The RESIDENT form (round 6): reads added by dm_rows_add_raw with s_mean == s_stdv == NULL and first_empty given (dm_signal_plan_batch) - their
 * statistics are the device block of dm_signal_event_stats_device.  No ev3: rdesc's event indices are positions in the batch's merged event tables
 * (= rows of that block); in_range covers the event lengths and the fall-back values, the signal stage reports the range of its own values. */
int64_t dm_rows_emit_resident(dm_rowsbatch* h, const int32_t* contig_rank, uint8_t* code, int64_t* rdesc, int32_t* sel_row, int64_t* pos, uint8_t* flags,
                              int64_t* groups, int64_t cap_groups, int64_t* contig_len, int64_t n_contig_len, int32_t* in_range);
int dm_rows_assemble(dm_model* m, float* d_rows, const uint8_t* d_code, const float* d_ev3, const int64_t* d_rdesc, int64_t n_reads, int64_t n_rows);

#ifdef __cplusplus
}
#endif
#endif /* DEEPMOD_HIP_H */
