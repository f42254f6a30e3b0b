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

/* one LSTM cell step (BasicLSTMCell.call) for a block of WB windows: inp[w][kin], h[w][100], c[w][100] updated
 * in place; each kernel row is reused across the block from L1; k-ascending summation per output element */
#define WB 8
static void cell_step_block(const float* kernel, const float* bias, const float* const* inp, int kin,
                            float (*h)[HID], float (*c)[HID], float (*g)[NG], int nb) {
    for (int w = 0; w < nb; ++w)
        for (int n = 0; n < NG; ++n) g[w][n] = 0.0f;
    for (int k = 0; k < kin; ++k) {
        const float* wr = kernel + (int64_t)k * NG;
        for (int w = 0; w < nb; ++w) {
            const float a = inp[w][k];
            float* gw = g[w];
            for (int n = 0; n < NG; ++n) gw[n] += a * wr[n];
        }
    }
    for (int k = 0; k < HID; ++k) {
        const float* wr = kernel + (int64_t)(kin + k) * NG;
        for (int w = 0; w < nb; ++w) {
            const float a = h[w][k];
            float* gw = g[w];
            for (int n = 0; n < NG; ++n) gw[n] += a * wr[n];
        }
    }
    for (int w = 0; w < nb; ++w) {
        float* gw = g[w];
        for (int n = 0; n < NG; ++n) gw[n] += bias[n];
        for (int u = 0; u < HID; ++u) {
            const float gi = gw[u], gj = gw[HID + u], gf = gw[2 * HID + u], go = gw[3 * HID + u];
            const float cn = c[w][u] * sigmoidf_(gf + 1.0f) + sigmoidf_(gi) * tanhf(gj);
            c[w][u] = cn;
            h[w][u] = tanhf(cn) * sigmoidf_(go);
        }
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
    for (int64_t w0 = 0; w0 < n; w0 += WB) {
        const int nb = (int)((n - w0) < WB ? (n - w0) : WB);
        float h[2][3][WB][HID], c[2][3][WB][HID], g[WB][NG];
        memset(h, 0, sizeof h); /* MultiRNNCellZeroState */
        memset(c, 0, sizeof c);
        for (int d = 0; d < 2; ++d) {
            for (int s = 0; s < LIVE; ++s) {
                const int row = d == 0 ? s : (WIN - 1 - s);
                const float* inp[WB];
                for (int w = 0; w < nb; ++w) inp[w] = x + (w0 + w) * WIN * NFEAT + row * NFEAT;
                int kin = NFEAT;
                for (int l = 0; l < 3; ++l) {
                    cell_step_block(kern[d][l], bias[d][l], inp, kin, h[d][l], c[d][l], g, nb);
                    for (int w = 0; w < nb; ++w) inp[w] = h[d][l][w];
                    kin = HID;
                }
            }
        }
        for (int wi = 0; wi < nb; ++wi) {
            const int64_t w = w0 + wi;
            /* concat_10 = [fw h2, bw h2]; logits = . @ W + b */
            float lg[2] = {0.0f, 0.0f};
            for (int d = 0; d < 2; ++d)
                for (int u = 0; u < HID; ++u) {
                    lg[0] += h[d][2][wi][u] * wout[(d * HID + u) * 2 + 0];
                    lg[1] += h[d][2][wi][u] * wout[(d * HID + u) * 2 + 1];
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
                memcpy(hcat + w * 2 * HID, h[0][2][wi], HID * sizeof(float));
                memcpy(hcat + w * 2 * HID + HID, h[1][2][wi], HID * sizeof(float));
            }
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
