/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not shipped, not linked into libdeepmod_hip.so.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * Plain-C fp32 restatement of the reference's per-window BiLSTM classifier:
 *   graph built by  /root/reference/bin/DeepMod_scripts/myMultiBiRNN.py:21-61
 *     :30      X = placeholder [None, 21, 7]
 *     :39      unstack along time
 *     :42-43   2 x MultiRNNCell([BasicLSTMCell(100, forget_bias=1.0)] * 3)   (fw, bw)
 *     :47      static_bidirectional_rnn  (bw consumes reversed input, outputs re-reversed)
 *     :55      logits = outputs[int(21/2)] @ W[200,2] + b[2]
 *     :59      softmax ; :61 argmax
 *   executed by /root/reference/bin/DeepMod_scripts/myDetect.py:814-820 (sess.run([mfpred])).
 *
 * The arithmetic itself lives in third-party tensorflow==1.x (tf.contrib.rnn.BasicLSTMCell;
 * docs/Install.md:26 pins 1.7.0, the shipped .meta were written by 1.8.0) which is absent.
 * Its published algorithm (BasicLSTMCell.call) is, per step and layer:
 *     g = concat([inp, h], 1) @ kernel + bias              kernel [in+100, 400]
 *     i, j, f, o = split(g, 4, axis=1)                      (order verified in the .meta wiring)
 *     c' = c * sigmoid(f + forget_bias) + sigmoid(i) * tanh(j)
 *     h' = tanh(c') * sigmoid(o)
 * Only outputs[10] is fetched, so TF prunes the static graph to 11 live steps per direction:
 * fw consumes rows 0..10, bw consumes rows 20..10 (confirmed by
 * tools/graphdef_interp.py executing 67 MatMuls).
 *
 * PARITY PIN: TensorFlow cannot run here and the reference ships no tests, so this oracle is
 * pinned against the numpy-interpreted reference GraphDef (tools/graphdef_interp.py ->
 * tests/golden/bilstm_*.npz).  Arithmetic parity vs real TF kernels is therefore "unpinned"
 * beyond that.
 *
 * Canonical flat weight blob (408,402 floats), shared with the product's packer:
 *   for d in (fw, bw): for l in 0..2: kernel[K_l][400] (K_0 = 107, else 200), bias[400]
 *   then head W[200][2], head b[2].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NFEAT 7
#define HID 100
#define WIN 21
#define LIVE 11 /* int(21/2)+1 live steps per direction */
#define NG 400  /* 4 gates x 100 */

static const int64_t KROWS[3] = {NFEAT + HID, 2 * HID, 2 * HID};

int64_t dmo_weight_count(void) {
    int64_t per_dir = 0;
    for (int l = 0; l < 3; ++l) per_dir += KROWS[l] * NG + NG;
    return 2 * per_dir + 2 * HID * 2 + 2;
}

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one LSTM cell step for one window: inp[kin], h[100], c[100] updated in place */
static void cell_step(const float* kernel, const float* bias, const float* inp, int kin,
                      float* h, float* c, float* g) {
    for (int n = 0; n < NG; ++n) g[n] = 0.0f;
    for (int k = 0; k < kin; ++k) {
        const float a = inp[k];
        const float* w = kernel + (int64_t)k * NG;
        for (int n = 0; n < NG; ++n) g[n] += a * w[n];
    }
    for (int k = 0; k < HID; ++k) {
        const float a = h[k];
        const float* w = kernel + (int64_t)(kin + k) * NG;
        for (int n = 0; n < NG; ++n) g[n] += a * w[n];
    }
    for (int n = 0; n < NG; ++n) g[n] += bias[n]; /* BiasAdd after MatMul, as in the graph */
    for (int u = 0; u < HID; ++u) {
        const float gi = g[u], gj = g[HID + u], gf = g[2 * HID + u], go = g[3 * HID + u];
        const float cn = c[u] * sigmoidf_(gf + 1.0f) + sigmoidf_(gi) * tanhf(gj);
        c[u] = cn;
        h[u] = tanhf(cn) * sigmoidf_(go);
    }
}

/* x: [n][21][7] fp32; prob: [n][2] or NULL; cls: [n] or NULL; hcat: [n][200] or NULL */
int dmo_predict_windows(const float* weights, const float* x, int64_t n, float* prob,
                        uint8_t* cls, float* hcat, int nthreads) {
    if (!weights || (!x && n > 0) || n < 0) return -1;
    const float* kern[2][3];
    const float* bias[2][3];
    const float* p = weights;
    for (int d = 0; d < 2; ++d)
        for (int l = 0; l < 3; ++l) {
            kern[d][l] = p;
            p += KROWS[l] * NG;
            bias[d][l] = p;
            p += NG;
        }
    const float* wout = p;
    const float* bout = p + 2 * HID * 2;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(static)
    for (int64_t w = 0; w < n; ++w) {
        float h[2][3][HID], c[2][3][HID], g[NG];
        memset(h, 0, sizeof h); /* MultiRNNCellZeroState */
        memset(c, 0, sizeof c);
        const float* xw = x + w * WIN * NFEAT;
        for (int d = 0; d < 2; ++d) {
            for (int s = 0; s < LIVE; ++s) {
                const int row = d == 0 ? s : (WIN - 1 - s);
                const float* inp = xw + row * NFEAT;
                int kin = NFEAT;
                for (int l = 0; l < 3; ++l) {
                    cell_step(kern[d][l], bias[d][l], inp, kin, h[d][l], c[d][l], g);
                    inp = h[d][l];
                    kin = HID;
                }
            }
        }
        /* concat_10 = [fw h2, bw h2]; logits = . @ W + b */
        float lg[2] = {0.0f, 0.0f};
        for (int d = 0; d < 2; ++d)
            for (int u = 0; u < HID; ++u) {
                lg[0] += h[d][2][u] * wout[(d * HID + u) * 2 + 0];
                lg[1] += h[d][2][u] * wout[(d * HID + u) * 2 + 1];
            }
        lg[0] += bout[0];
        lg[1] += bout[1];
        const float m = lg[0] > lg[1] ? lg[0] : lg[1];
        const float e0 = expf(lg[0] - m), e1 = expf(lg[1] - m);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        if (prob) {
            prob[2 * w] = p0;
            prob[2 * w + 1] = p1;
        }
        if (cls) cls[w] = p1 > p0 ? 1 : 0; /* tf.argmax: ties -> index 0 */
        if (hcat) {
            memcpy(hcat + w * 2 * HID, h[0][2], HID * sizeof(float));
            memcpy(hcat + w * 2 * HID + HID, h[1][2], HID * sizeof(float));
        }
    }
    return 0;
}

/*
 * Per-position summary restatement of /root/reference/bin/DeepMod_scripts/myDetect.py:1089-1100:
 * for every base row of a read, if refbase == Base (and not '-','N','n'): touch[pos]++ (key
 * creation), and if readbase != '-': cov[pos]++, and if mod_pred == 1: mod[pos]++.
 * flags bit0 = refbase == Base, bit1 = readbase != '-', bit2 = mod_pred == 1.
 */
int dmo_summary_add(int32_t* touch, int32_t* cov, int32_t* mod, int64_t length, const int64_t* pos,
                    const uint8_t* flags, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        if (!(flags[i] & 1)) continue;
        if (pos[i] < 0 || pos[i] >= length) return -2;
        touch[pos[i]] += 1;
        if (flags[i] & 2) {
            cov[pos[i]] += 1;
            if (flags[i] & 4) mod[pos[i]] += 1;
        }
    }
    return 0;
}
