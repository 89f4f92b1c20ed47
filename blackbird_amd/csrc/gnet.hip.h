// gnet.hip.h -- the network for ANY filter count F = 16*NCB (BASELINE configs[4]: 20 blocks x 256 filters).
//
// The fused kernels in net.hip.h keep a whole F=16 tower in one wave's LDS.  At F=256 one position's activations
// are 43 KB per layer and one layer's weights 2.4 MB, so the tower becomes what it is on any machine: one
// implicit-GEMM launch per conv layer (M = batch x H x W pixels, N = F, K = 9 F), activations ping-ponging
// between two HBM buffers that stay L2/MALL resident, on v_mfma_f32_16x16x4_f32.
//
//  * activation layout  act[pos][cb][j][slot][4]   (cb = 16-channel block, j = 4-channel group, slot = pixel
//    incl. a zero halo ring that is never written): the 16 pixels of an MFMA tile x 16 B are contiguous, the 9 taps
//    are constant offsets, no bounds checks anywhere;
//  * a wave = 2 filter blocks x NT pixel tiles; a workgroup = 4 waves = 8 consecutive 16-filter blocks of the SAME
//    PPB positions, so the B operand (activations) of a (tap, cb) step is shared through L1 by the 4 waves; each
//    wave streams its own pre-swizzled A operands (weights) once per NT pixel tiles; operands are prefetched two
//    steps ahead in three register sets, with scheduling barriers so that the compiler keeps the distance;
//  * K order per output pixel is (16-channel block, tap, r, j), c = 16 cb + 4 j + r: the 9 shifted views of one
//    channel block (17 KB for 4 positions) are consumed together, so they are L1/L2 hits instead of 9 passes over
//    the whole activation tensor -- and it is exactly the CPU restatement's fmaf
//    chain (its conv3x3), so the tower stays bit-comparable with the CPU restatement the tests check against;
//  * heads: one wave per position runs the two 1x1 convolutions over F channels, then net_head_tail (shared with
//    the fused kernels).
// Every kernel takes the batch size from device memory when n_ptr != nullptr (compacted leaf lists of the
// asynchronous search) and indexes mailboxes through slot_list when given.
#pragma once
#include "net.hip.h"

struct GNetDev {
    int F, NCB, R;            // filters, F/16, residual blocks
    const float *w0;          // [NCB][steps0][64]            first conv, A-operand lane order per filter block
    const f32x4 *wt;          // [2R][NCB(fb)][NCB(cb)][9][64] tower convs: lane (f=l&15, j=l>>4), .r = W[tap][16cb+4j+r][16fb+f]
    const float *epi;         // [1+2R][NCB][3][16]           bias, bn scale, bn shift per filter block
    float *inp;               // [cap][SLOTS][CP]             input planes as floats, zero halo
    float *act[2];            // [cap][NCB][4][SLOTS][4]      ping-pong activations, zero halo
    int cap;                  // positions the buffers hold (a multiple of GN_PPB)
};

template <class G>
struct GNetGeom {
    static constexpr int H = G::H, W = G::W, CIN = G::C, HW = H * W;
    static constexpr int SLOTS = (H + 2) * (W + 1) + 1;
    static constexpr int CP = (CIN + 3) / 4 * 4;
    static constexpr int STEPS0 = (9 * CIN + 3) / 4;
    // positions per workgroup: chosen so that PPB*HW fills whole 16-pixel tiles (or nearly)
    static constexpr int PPB = HW == 42 ? 4 : (HW == 9 ? 16 : 2);
    static constexpr int NT = (PPB * HW + 15) / 16; // C4: 11 tiles (10.5 used), TTT: 9 (exact), DragonChess: 8 (exact)
    static constexpr int PLANE = SLOTS * 4;         // floats of one (cb, j) plane of one position
};

// ---- input planes: one thread per (position, cell) ------------------------------------------------------------
template <class G>
__global__ void __launch_bounds__(256) k_gnet_input(GNetDev gd, int n_max, const int *n_ptr, const int *slot_list,
                                                    const typename G::State *states, const int8_t *planes) {
    using GG = GNetGeom<G>;
    const int n = n_ptr ? *n_ptr : n_max;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * GG::HW) return;
    int i = (int)(t / GG::HW), cell = (int)(t % GG::HW), y = cell / GG::W, x = cell % GG::W;
    float *dst = gd.inp + ((size_t)i * GG::SLOTS + (y + 1) * (GG::W + 1) + (x + 1)) * GG::CP;
    if (planes) {
        const int8_t *src = planes + ((size_t)i * GG::HW + cell) * GG::CIN;
        for (int c = 0; c < GG::CIN; c++) dst[c] = (float)src[c];
    } else {
        int8_t v[GG::CIN];
        G::encode_cell(states[slot_list ? slot_list[i] : i], y, x, v);
        for (int c = 0; c < GG::CIN; c++) dst[c] = (float)v[c];
    }
}

// ---- one conv layer (first conv when FIRST, else tower layer l) ---------------------------------------------------
// A wave owns GN_FBW = 2 filter blocks x NT pixel tiles of PPB positions: the B operand (activations) it loads for a
// (tap, cb) step feeds 2 x NT x 4 MFMAs and each A operand NT x 4, i.e. 13 KB of operand loads per 88 MFMAs --
// half the L1 traffic of one filter block per wave.  A workgroup = 4 waves = 8 consecutive filter blocks of the
// same positions (their B loads hit L1).
#define GN_FBW 2
// PPB / FBW are template parameters so that a SMALL batch (a single FindMove position, an arena of a few games) can be cut
// finely -- one position and one filter block per wave: 16x more waves, each with 1/7 of the MFMAs -- instead of
// leaving 250 CUs idle behind two fat workgroups.
// out3 != nullptr (first conv only): the outputs are written as three bf16 planes, [pos][cb][slot][plane][16 ch], for the
// bf16-pipe tower layers (gnet_x3.hip.h) instead of the float32 layout.
template <class G, bool FIRST, int PPB = GNetGeom<G>::PPB, int FBW = GN_FBW>
__global__ void __launch_bounds__(256) k_gnet_conv(GNetDev gd, int layer, int n_max, const int *n_ptr, const float *in,
                                                   float *out, int skip, int pairs_per_wg, unsigned char *out3 = nullptr) {
    using GG = GNetGeom<G>;
    constexpr int HW = GG::HW, W = GG::W, SLOTS = GG::SLOTS, CP = GG::CP, NT = (PPB * HW + 15) / 16, PLANE = GG::PLANE,
                  CIN = GG::CIN, STEPS0 = GG::STEPS0;
    const int n = n_ptr ? *n_ptr : n_max;
    // the 4 waves of a workgroup = pairs_per_wg (1, 2 or 4) filter-block pairs x 4/pairs_per_wg groups of PPB positions,
    // so that narrow networks (F < 128) still use every wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane >> 4, nn = lane & 15;
    const int pos0 = (blockIdx.x * (4 / pairs_per_wg) + wave / pairs_per_wg) * PPB;
    if (pos0 >= n) return;
    const int NCB = gd.NCB;
    const int fb0 = (blockIdx.y * pairs_per_wg + wave % pairs_per_wg) * FBW;
    if (fb0 >= NCB) return;
    int fbs[FBW]; // an odd filter-block count leaves the last wave's second block unused: computed on block NCB-1, not stored
    bool fb_ok[FBW];
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        fb_ok[f] = fb0 + f < NCB;
        fbs[f] = fb_ok[f] ? fb0 + f : NCB - 1;
    }
    const size_t pos_floats = (size_t)NCB * 4 * PLANE; // activation floats per position

    // per-tile addressing: lane (j, nn) <-> pixel nn of the tile
    int boff[NT];  // float offset of (pos, plane j of block 0, slot) in the input (tower) / of (pos, slot) in inp (first conv)
    int ooff[NT];  // float offset of (pos, plane j of block 0, slot) in the output
    bool valid[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int q = t * 16 + nn;
        int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
        valid[t] = pp < PPB && pos0 + pp < n;
        if (pp >= PPB) pp = PPB - 1; // a padded tile column: any readable address, result discarded
        int slot = (y + 1) * (W + 1) + (x + 1);
        ooff[t] = (int)(pp * pos_floats) + j * PLANE + slot * 4;
        boff[t] = FIRST ? (pp * SLOTS + slot) * CP : ooff[t];
    }
    f32x4 acc[FBW][NT];
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        const f32x4 bias = *(const f32x4 *)(gd.epi + ((size_t)layer * NCB + fbs[f]) * 48 + 4 * j);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[f][t] = bias;
    }

    if constexpr (FIRST) {
        const float *ip = gd.inp + (size_t)pos0 * SLOTS * CP;
#pragma unroll
        for (int s = 0; s < STEPS0; s++) {
            int kk0 = 4 * s + j;
            int kk = kk0 < 9 * CIN ? kk0 : 9 * CIN - 1; // padded k: weight is 0
            int tap = kk / CIN, c = kk % CIN;
            int toff = ((tap / 3 - 1) * (W + 1) + (tap % 3 - 1)) * CP + c;
            float a[FBW];
#pragma unroll
            for (int f = 0; f < FBW; f++) a[f] = gd.w0[((size_t)fbs[f] * STEPS0 + s) * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                float b = ip[boff[t] + toff];
#pragma unroll
                for (int f = 0; f < FBW; f++) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f], b, acc[f][t], 0, 0, 0);
            }
        }
    } else {
        const float *ip = in + (size_t)pos0 * pos_floats;
        const f32x4 *wp[FBW];
#pragma unroll
        for (int f = 0; f < FBW; f++) wp[f] = gd.wt + (((size_t)(layer - 1) * NCB + fbs[f]) * 9) * NCB * 64 + lane;
        // software pipeline over the NCB*9 (cb, tap) steps, two register sets used alternately (no copies): the
        // operands of step s+1 are in flight while the FBW*4*NT MFMAs of step s issue
        f32x4 a0[FBW], a1[FBW], b0[NT], b1[NT];
        const int steps = 9 * NCB;
        auto load_step = [&](f32x4 (&a)[FBW], f32x4 (&b)[NT], int s) {
            const int cb = s / 9, tap = s - 9 * cb;
            const int toff = ((tap / 3 - 1) * (W + 1) + (tap % 3 - 1)) * 4 + cb * 4 * PLANE;
#pragma unroll
            for (int f = 0; f < FBW; f++) a[f] = wp[f][(size_t)s * 64];
#pragma unroll
            for (int t = 0; t < NT; t++) b[t] = *(const f32x4 *)(ip + boff[t] + toff);
        };
        auto mma_step = [&](const f32x4 (&a)[FBW], const f32x4 (&b)[NT]) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int f = 0; f < FBW; f++)
                        acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][r], b[t][r], acc[f][t], 0, 0, 0);
        };
        // Prefetch distance two steps (three register sets in rotation; 9*NCB is a multiple of 3).  The scheduling
        // barriers keep the compiler from sinking a step's loads down to their first use, which would put an L2
        // round trip in front of every step.
        f32x4 a2[FBW], b2[NT];
        const int last = steps - 1;
        load_step(a0, b0, 0);
        load_step(a1, b1, 1 < last ? 1 : last);
        for (int s = 0; s < steps; s += 3) {
            load_step(a2, b2, s + 2 < last ? s + 2 : last);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            load_step(a0, b0, s + 3 < last ? s + 3 : last);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            load_step(a1, b1, s + 4 < last ? s + 4 : last);
            __builtin_amdgcn_sched_barrier(0);
            mma_step(a2, b2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float *op = out + (size_t)pos0 * pos_floats;
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        if (!fb_ok[f]) continue;
        const float *ep = gd.epi + ((size_t)layer * NCB + fbs[f]) * 48;
        const f32x4 scale = *(const f32x4 *)(ep + 16 + 4 * j), shift = *(const f32x4 *)(ep + 32 + 4 * j);
        const int fo = fbs[f] * 4 * PLANE;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 y;
            f32x4 sk = {0.f, 0.f, 0.f, 0.f};
            if (skip && valid[t]) sk = *(const f32x4 *)(op + ooff[t] + fo); // tf.add(batch_norm_2, block input) before the ReLU
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = __builtin_fmaf(acc[f][t][r], scale[r], shift[r]);
                if (skip) v = v + sk[r];
                y[r] = fmaxf(v, 0.f);
            }
            if constexpr (FIRST) {
                if (out3) {
                    if (valid[t]) {
                        int q = t * 16 + nn, pp = q / HW, cell = q % HW, yy = cell / W, xx = cell % W;
                        unsigned char *o = out3 + ((size_t)(pos0 + pp) * NCB + fbs[f]) * (SLOTS * 96) + ((yy + 1) * (W + 1) + (xx + 1)) * 96 + j * 8;
                        unsigned a[4], b[4], c[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            __bf16 h1 = (__bf16)y[r];
                            a[r] = (unsigned)__builtin_bit_cast(unsigned short, h1);
                            float r1 = y[r] - __uint_as_float(a[r] << 16);
                            __bf16 h2 = (__bf16)r1;
                            b[r] = (unsigned)__builtin_bit_cast(unsigned short, h2);
                            float r2 = r1 - __uint_as_float(b[r] << 16);
                            __bf16 h3 = (__bf16)r2;
                            c[r] = (unsigned)__builtin_bit_cast(unsigned short, h3);
                        }
                        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                        *(u2 *)(o) = u2{a[0] | (a[1] << 16), a[2] | (a[3] << 16)};
                        *(u2 *)(o + 32) = u2{b[0] | (b[1] << 16), b[2] | (b[3] << 16)};
                        *(u2 *)(o + 64) = u2{c[0] | (c[1] << 16), c[2] | (c[3] << 16)};
                    }
                    continue;
                }
            }
            if (valid[t]) *(f32x4 *)(op + ooff[t] + fo) = y;
        }
    }
}

// ---- heads: one wave per position ----------------------------------------------------------------------------------
template <class G>
__global__ void __launch_bounds__(256) k_gnet_heads(GNetDev gd, NetDev nd, int n_max, const int *n_ptr, const int *slot_list,
                                                    const float *act, const uint32_t *game_id, const int32_t *serial,
                                                    int noise, float *value_out, float *logits_out, float *policy_out,
                                                    int pstride) {
    using GG = GNetGeom<G>;
    constexpr int HW = GG::HW, W = GG::W, PLANE = GG::PLANE, A = G::A;
    __shared__ float scratch[4][3 * HW + 64 + 2 * (A <= 64 ? A : 0) + 8];
    const int n = n_ptr ? *n_ptr : n_max;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pos = blockIdx.x * 4 + wave;
    if (pos >= n) return;
    float *rv = scratch[wave], *rp = rv + HW;
    const float *hp = nd.head;
    const float *vk = hp + nd.off_vk, *v3 = hp + nd.off_v3, *pk = hp + nd.off_pk, *p6 = hp + nd.off_p6;
    const float *ap = act + (size_t)pos * gd.NCB * 4 * PLANE;
    for (int q = lane; q < HW; q += 64) { // channels in natural order c = 16 cb + 4 j + r (oracle: plain ascending c)
        int y = q / W, x = q % W, slot = (y + 1) * (W + 1) + (x + 1);
        float av = v3[0], a0 = p6[0], a1 = p6[1];
        for (int cj = 0; cj < gd.NCB * 4; cj++) {
            f32x4 xv = *(const f32x4 *)(ap + (size_t)cj * PLANE + slot * 4);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int c = 4 * cj + r;
                av = __builtin_fmaf(xv[r], vk[c], av);
                a0 = __builtin_fmaf(xv[r], pk[2 * c], a0);
                a1 = __builtin_fmaf(xv[r], pk[2 * c + 1], a1);
            }
        }
        rv[q] = fmaxf(__builtin_fmaf(av, v3[1], v3[2]), 0.f);
        rp[2 * q] = fmaxf(__builtin_fmaf(a0, p6[2], p6[4]), 0.f);
        rp[2 * q + 1] = fmaxf(__builtin_fmaf(a1, p6[3], p6[5]), 0.f);
    }
    wave_lds_handover();
    net_head_tail<G, 1>(nd, n, pos, slot_list, rv, rp, game_id, serial, noise, value_out, logits_out, policy_out, pstride);
}
