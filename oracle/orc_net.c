/*
 * oracle/orc_net.c -- CPU ORACLE (test infrastructure, not product code).
 * float32 restatement of the inference graph built by NetworkFactory.__call__
 * (/root/reference/src/NetworkFactory.py:22-183) as it is run by Network.getEvaluation /
 * getPolicy (Network.py:48-64).
 *
 * PARITY UNPINNED by the reference: TensorFlow cannot be imported here and the reference
 * holds no golden outputs for the network.  This file follows the call sites and TF's
 * documented layer defaults (conv2d use_bias=True, SAME padding, batch_normalization in
 * inference mode with epsilon=1e-3, dense on the last axis) and is cross-checked against
 * independent PyTorch-CPU ops in tests/test_oracle_net.py.
 *
 * Summation order (a free choice -- TF's is unspecified): every dot product is a single
 * k-ordered fmaf chain starting from the bias.  The order inside a 3x3 tower conv is
 * (16-channel block, tap, r, j) with c = 16*block + 4*j + r, which is the order in which a
 * v_mfma_f32_16x16x4_f32 chain on gfx950 consumes K when each lane group j holds channels
 * 4j..4j+3 and the 9 taps of one channel block are consumed together (cache locality of the
 * wide networks; for 16 filters there is one block and the order is simply (tap, r, j));
 * the first conv (few input planes) uses the natural (tap, c) order.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define BN_EPS 1e-3f

static void bn_fold(const float *bn, int F, float *scale, float *shift) {
    /* y = gamma * (x - mean) / sqrt(var + eps) + beta  ==  x*scale + shift */
    for (int f = 0; f < F; f++) {
        float g = bn[0 * F + f], b = bn[1 * F + f], m = bn[2 * F + f], v = bn[3 * F + f];
        float s = g / sqrtf(v + BN_EPS);
        float t = m * s;
        scale[f] = s;
        shift[f] = b - t;
    }
}

/* 3x3 SAME conv + bias + BN (+skip) + ReLU over one position.
 * in [H][W][Cin], out [H][W][F], k [3][3][Cin][F] HWIO. natural_order: (tap, c) ascending. */
static void conv3x3(const float *in, float *out, const float *skip, int H, int W, int Cin, int F,
                    const float *k, const float *bias, const float *scale, const float *shift,
                    int natural_order) {
    float *acc = (float *)malloc(sizeof(float) * (size_t)F);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            for (int f = 0; f < F; f++) acc[f] = bias[f];
            if (natural_order) {
                for (int t = 0; t < 9; t++) {
                    int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue; /* zero padding: fma(0,w,acc)==acc */
                    const float *ip = in + (yy * W + xx) * Cin;
                    const float *kp = k + (size_t)t * Cin * F;
                    for (int c = 0; c < Cin; c++) {
                        float a = ip[c];
                        const float *w = kp + (size_t)c * F;
                        for (int f = 0; f < F; f++) acc[f] = fmaf(a, w[f], acc[f]);
                    }
                }
            } else {
                for (int cb = 0; cb < Cin; cb += 16)
                    for (int t = 0; t < 9; t++) {
                        int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                        const float *ip = in + (yy * W + xx) * Cin;
                        const float *kp = k + (size_t)t * Cin * F;
                        for (int r = 0; r < 4; r++)
                            for (int j = 0; j < 4; j++) {
                                int c = cb + 4 * j + r;
                                if (c >= Cin) continue;
                                float a = ip[c];
                                const float *w = kp + (size_t)c * F;
                                for (int f = 0; f < F; f++) acc[f] = fmaf(a, w[f], acc[f]);
                            }
                    }
            }
            float *o = out + (y * W + x) * F;
            for (int f = 0; f < F; f++) {
                float v = fmaf(acc[f], scale[f], shift[f]);
                if (skip) v = v + skip[(y * W + x) * F + f]; /* tf.add(bn2, block input), NetworkFactory.py:95-98 */
                o[f] = v > 0.f ? v : 0.f;
            }
        }
    free(acc);
}

/* reduce_sum over H, W as a fixed pairwise tree: the values sit in slots 0..n-1 of a 64-slot vector (the rest 0.0) and are
 * folded with strides 1, 2, ..., 32 (adjacent pairs first) -- the xor butterfly of a 64-lane wave in the order the HIP
 * kernels form it (blackbird_amd/csrc/net.hip.h: wave_sum_f32).  TensorFlow leaves the order of a reduce_sum unspecified. */
static float tree_sum64(const float *x, int n, int stride) {
    float t[64];
    for (int i = 0; i < 64; i++) t[i] = i < n ? x[(size_t)i * stride] : 0.f;
    for (int o = 1; o < 64; o <<= 1)
        for (int i = 0; i < 64; i += 2 * o) t[i] = t[i] + t[i + o];
    return t[0];
}

/* The forward pass in two statements of the heads, equal in exact arithmetic:
 *   pooled = 0  the reference's op order: dense on the last axis of [H, W, c], THEN reduce_sum over H, W
 *               (NetworkFactory.py:125-131, 165-169), sums in row-major pixel order;
 *   pooled = 1  reduce_sum over H, W first (tree_sum64), then the dense layer on the pooled activations:
 *               relu(k[d] * R + HW * b[d]) and k1[a] * R1 + (k0[a] * R0 + HW * b[a]).
 * orc_net_forward is the pooled statement -- the one the HIP kernels implement, bit for bit up to expf/tanhf;
 * orc_net_forward_perpixel keeps the literal op order; tests/test_oracle_net.py holds the two against each other
 * (<= 1e-6 relative) and both against independent PyTorch ops. */
static void net_forward_impl(const orc_net *w, const int8_t *boards, int n, float *value, float *logits, float *policy,
                             int pooled) {
    int H = w->H, W = w->W, C = w->C, F = w->F, R = w->R, D = w->D, A = w->A;
    int HW = H * W;
    float *x0 = (float *)malloc(sizeof(float) * (size_t)HW * C);
    float *xa = (float *)malloc(sizeof(float) * (size_t)HW * F);
    float *xb = (float *)malloc(sizeof(float) * (size_t)HW * F);
    float *xc = (float *)malloc(sizeof(float) * (size_t)HW * F);
    float *sc = (float *)malloc(sizeof(float) * (size_t)F);
    float *sh = (float *)malloc(sizeof(float) * (size_t)F);
    float *rv = (float *)malloc(sizeof(float) * (size_t)HW);
    float *rp = (float *)malloc(sizeof(float) * (size_t)HW * 2);
    float *lg = (float *)malloc(sizeof(float) * (size_t)A);
    for (int b = 0; b < n; b++) {
        const int8_t *bd = boards + (size_t)b * HW * C;
        for (int i = 0; i < HW * C; i++) x0[i] = (float)bd[i]; /* int8 feed cast to float32 placeholder */
        /* resTower/conv_block, NetworkFactory.py:46-57 */
        bn_fold(w->conv0_bn, F, sc, sh);
        conv3x3(x0, xa, NULL, H, W, C, F, w->conv0_k, w->conv0_b, sc, sh, 1);
        /* residual blocks, NetworkFactory.py:59-103 */
        for (int r = 0; r < R; r++) {
            const float *k1 = w->blk_k + ((size_t)r * 2 + 0) * 9 * F * F;
            const float *k2 = w->blk_k + ((size_t)r * 2 + 1) * 9 * F * F;
            bn_fold(w->blk_bn + ((size_t)r * 2 + 0) * 4 * F, F, sc, sh);
            conv3x3(xa, xb, NULL, H, W, F, F, k1, w->blk_b + ((size_t)r * 2 + 0) * F, sc, sh, 0);
            bn_fold(w->blk_bn + ((size_t)r * 2 + 1) * 4 * F, F, sc, sh);
            conv3x3(xb, xc, xa, H, W, F, F, k2, w->blk_b + ((size_t)r * 2 + 1) * F, sc, sh, 0);
            float *t = xa;
            xa = xc;
            xc = t;
        }
        /* value head, NetworkFactory.py:105-146 */
        {
            float s1, t1;
            bn_fold(w->v_bn, 1, &s1, &t1);
            for (int p = 0; p < HW; p++) {
                float a = w->v_conv_b[0];
                for (int c = 0; c < F; c++) a = fmaf(xa[p * F + c], w->v_conv_k[c], a);
                float v = fmaf(a, s1, t1);
                rv[p] = v > 0.f ? v : 0.f;
            }
            float e = w->v_d2_b[0];
            if (pooled) { /* dense_1 after the pool: relu(k[d] * sum_p rv[p] + HW * b[d]); dense_2: the products folded by
                           * the same fixed tree (TensorFlow leaves the order of a matmul's sum open), then the bias */
                float Rv = tree_sum64(rv, HW, 1);
                float prod[64];
                for (int d = 0; d < D && d < 64; d++) {
                    float s = fmaf(Rv, w->v_d1_k[d], (float)HW * w->v_d1_b[d]);
                    s = s > 0.f ? s : 0.f;
                    prod[d] = s * w->v_d2_k[d];
                }
                e = tree_sum64(prod, D < 64 ? D : 64, 1) + w->v_d2_b[0];
            } else {
                for (int d = 0; d < D; d++) {
                    float s = 0.f; /* dense_1 per pixel (:125-127) then reduce_sum over H,W (:129-131) */
                    for (int p = 0; p < HW; p++) s += fmaf(rv[p], w->v_d1_k[d], w->v_d1_b[d]);
                    s = s > 0.f ? s : 0.f;
                    e = fmaf(s, w->v_d2_k[d], e);
                }
            }
            value[b] = tanhf(e);
        }
        /* policy head, NetworkFactory.py:148-172 */
        {
            float s2[2], t2[2];
            bn_fold(w->p_bn, 2, s2, t2);
            for (int p = 0; p < HW; p++)
                for (int q = 0; q < 2; q++) {
                    float a = w->p_conv_b[q];
                    for (int c = 0; c < F; c++) a = fmaf(xa[p * F + c], w->p_conv_k[c * 2 + q], a);
                    float v = fmaf(a, s2[q], t2[q]);
                    rp[p * 2 + q] = v > 0.f ? v : 0.f;
                }
            float m = -INFINITY;
            float R0 = 0.f, R1 = 0.f;
            if (pooled) {
                R0 = tree_sum64(rp, HW, 2);
                R1 = tree_sum64(rp + 1, HW, 2);
            }
            for (int a = 0; a < A; a++) {
                float s = 0.f;
                if (pooled) { /* dense after the pool: k1[a] * R1 + (k0[a] * R0 + HW * b[a]) */
                    s = fmaf(R1, w->p_d_k[A + a], fmaf(R0, w->p_d_k[a], (float)HW * w->p_d_b[a]));
                } else { /* dense on the last axis (:165-166) then reduce_sum over H,W (:168-169) */
                    for (int p = 0; p < HW; p++)
                        s += fmaf(rp[p * 2 + 1], w->p_d_k[A + a], fmaf(rp[p * 2 + 0], w->p_d_k[a], w->p_d_b[a]));
                }
                lg[a] = s;
                if (s > m) m = s;
            }
            float tot = 0.f;
            for (int a = 0; a < A; a++) {
                if (logits) logits[(size_t)b * A + a] = lg[a];
                lg[a] = expf(lg[a] - m);
                tot += lg[a];
            }
            if (pooled && A <= 64) tot = tree_sum64(lg, A, 1); /* the kernels' softmax sum: the same fixed tree (narrow heads) */
            if (policy)
                for (int a = 0; a < A; a++) policy[(size_t)b * A + a] = lg[a] / tot;
        }
    }
    free(x0); free(xa); free(xb); free(xc); free(sc); free(sh); free(rv); free(rp); free(lg);
}

void orc_net_forward(const orc_net *w, const int8_t *boards, int n, float *value, float *logits, float *policy) {
    net_forward_impl(w, boards, n, value, logits, policy, 1);
}

void orc_net_forward_perpixel(const orc_net *w, const int8_t *boards, int n, float *value, float *logits, float *policy) {
    net_forward_impl(w, boards, n, value, logits, policy, 0);
}
