// gnet_x3.hip.h -- the conv layers of the general-filter network (gnet.hip.h) on the bf16 matrix pipe: float32 operands as
// three bf16 planes, six v_mfma_f32_16x16x32_bf16 products per K slice (net_x3.hip.h explains why and how accurate).
//
//  * activations between tower layers live in HBM as  act3[pos][cb][slot][plane 0..2][16 ch] bf16  (96 B per pixel slot of a
//    16-channel block, zero halo never written): one 16-byte load = 8 channels of one plane of one pixel -- the B operand of
//    a K = 32 slice (two taps x 16 channels of one channel block, lane group g = tap g >> 1, channels 8 (g & 1) .. + 7).  The
//    first conv (float32 MFMA on the exact int8 planes, gnet.hip.h) writes that form, the last tower layer writes the
//    float32 layout the heads kernel reads;
//  * K order per output: channel blocks ascending, inside a block the slices taps (0,1), (3,4), (6,7), (2,5); then tap 8 of
//    every channel block as K = 16 slices (zero-extended to the K = 32 instruction: net_x3.hip.h x3_k16).  The
//    order does not depend on the tiling, so the throughput launch (PPB positions x 4 filter blocks per wave) and the
//    latency launch (1 x 1) give the same bits;
//  * a wave = FBW filter blocks x NT pixel tiles; the B operand (activations) is shared by its filter blocks and, through
//    L1, by the workgroup's other waves (same positions); with 4 filter blocks per wave a slice is 6 x 4 x NT MFMAs = 4 224
//    matrix cycles at NT = 11 for 33 KB of activation loads -- half the L1 rate two filter blocks per wave would need;
//  * the three pixel planes roll through two register buffers: while the 3 FBW NT MFMAs on plane 1 run, plane 2 arrives;
//    during its 2 FBW NT MFMAs plane 3 arrives in plane 1's registers, during its FBW NT MFMAs the next slice's plane 1.
//  * weights: [layer][fb][cb] blocks of 13 824 B = [slice 4][plane 3][lane][16 B] + [plane 3][lane][8 B] (tap 8), split and
//    swizzled on the host, streamed from L2.
#pragma once
#include "gnet.hip.h"
#include "net_x3.hip.h"

#define GX3_PAIR_B (4 * 3 * 64 * 16 + 3 * 64 * 8) // bytes of one (filter block, channel block) weight block

struct GNetX3 {
    const unsigned char *wt; // [2R][NCB][NCB] x GX3_PAIR_B; nullptr: float32-MFMA layers
    unsigned char *act3[2];  // [cap][NCB][SLOTS][96]
};

// (Measured and dropped: the activation operand of a slice fetched once per workgroup into a double-buffered LDS image and
// read back by its four waves -- 47.1 ms against 40.1 ms per 20 x 256 batch; the L1 rate was not the limit.)
// (Round 3, measured and dropped: tap 8 as three plane-concatenated K = 32 products -- 27 MFMAs per block and tile instead of
// 30, as in net_x3.hip.h.  In its own loop after the slices: 40.9 ms against 39.8 (two 16-byte operand loads per tile instead of
// three 8-byte ones; the section waits for memory, not for the matrix pipe).  As a fifth step inside the slice loop with
// compile-time step tags: the optimiser hoists per-step 64-bit operand pointers out of the layer loop and spills 50-150
// registers of a kernel that uses all 512 -- 48.7 ms; with buffer loads (descriptor + scalar offset + 32-bit lane offset) 73.7 ms.
// What paces these layers is the first slice of every channel block missing to HBM, not the 10 % of padded MFMA time.)
template <class G, int PPB, int FBW, bool LAST>
__global__ void __launch_bounds__(256) k_gnet_conv_x3(GNetDev gd, GNetX3 gx, int layer, int n_max, const int *n_ptr,
                                                      const unsigned char *in, unsigned char *out3, float *outf, int skip,
                                                      int groups_per_wg) {
    using GG = GNetGeom<G>;
    constexpr int HW = GG::HW, W = GG::W, SLOTS = GG::SLOTS, NT = (PPB * HW + 15) / 16, PLANE = GG::PLANE, SB = 96;
    const int n = n_ptr ? *n_ptr : n_max;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, nn = lane & 15, gh = g >> 1, gl = g & 1;
    const int pos0 = (blockIdx.x * (4 / groups_per_wg) + wave / groups_per_wg) * PPB;
    if (pos0 >= n) return;
    const int NCB = gd.NCB;
    const int fb0 = (blockIdx.y * groups_per_wg + wave % groups_per_wg) * FBW;
    if (fb0 >= NCB) return;
    int fbs[FBW];
    bool fb_ok[FBW];
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        fb_ok[f] = fb0 + f < NCB;
        fbs[f] = fb_ok[f] ? fb0 + f : NCB - 1; // a filter-block count that is no multiple of FBW: computed on the last block, not stored
    }
    const int cb_bytes = SLOTS * SB;
    const size_t pos_bytes = (size_t)NCB * cb_bytes, pos_floats = (size_t)NCB * 4 * PLANE;
    constexpr int TAP0 = (W + 1) + 1;
    int aA[NT], aB[NT], aC[NT];
    bool valid[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int q = t * 16 + nn;
        int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
        valid[t] = pp < PPB && pos0 + pp < n;
        if (pp >= PPB) pp = PPB - 1; // a padded tile column: any readable address, result discarded
        int base = (int)(pp * pos_bytes) + ((y + 1) * (W + 1) + (x + 1) - TAP0) * SB;
        aA[t] = base + gh * SB + gl * 16;           // slices (3r, 3r + 1): + r (W + 1) SB
        aB[t] = base + gh * (W + 1) * SB + gl * 16; // slice (2, 5): + 2 SB
        aC[t] = base + g * 8;                        // tap 8: + (2 (W + 1) + 2) SB; this lane's outputs: + TAP0 SB
    }
    f32x4 acc[FBW][NT];
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        const f32x4 bias = *(const f32x4 *)(gd.epi + ((size_t)layer * NCB + fbs[f]) * 48 + 4 * g);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[f][t] = bias;
    }
    const unsigned char *ip = in + (size_t)pos0 * pos_bytes;
    const unsigned char *wp[FBW];
#pragma unroll
    for (int f = 0; f < FBW; f++) wp[f] = gx.wt + (((size_t)(layer - 1) * NCB + fbs[f]) * NCB) * GX3_PAIR_B;
    auto xoff = [&](int s, int t) __attribute__((always_inline)) { return s < 3 ? aA[t] + s * (W + 1) * SB : aB[t] + 2 * SB; };

    // ---- K = 32 slices: u = 4 cb + s --------------------------------------------------------------------------------------
    // (Measured and dropped: a channel-block order rotated per filter-block group, so that the workgroup's four waves take
    // each other's HBM misses -- 44.0 ms against 38.8: in step, one L1 fill serves all four.)
    const int steps = 4 * NCB;
    bf16x8 wa[FBW][3], wb[FBW][3], p0[NT], p1[NT];
    auto load_w = [&](bf16x8 (&w)[FBW][3], int u) __attribute__((always_inline)) {
        const int cb = u >> 2, s = u & 3;
#pragma unroll
        for (int f = 0; f < FBW; f++)
#pragma unroll
            for (int q = 0; q < 3; q++) w[f][q] = *(const bf16x8 *)(wp[f] + (size_t)cb * GX3_PAIR_B + (s * 3 + q) * 1024 + (unsigned)(lane * 16));
    };
    // The slice-dependent part of an operand address is wave-uniform: it goes into the (scalar) base, the per-lane part is a
    // 32-bit offset register that never changes -- a load is then ONE instruction (saddr + voffset).  With the slice offset
    // added per lane every load carried 6 vector instructions, ~250 per slice in bursts between the MFMA groups.
    auto load_x = [&](bf16x8 (&p)[NT], int u, int plane) __attribute__((always_inline)) {
        const int cb = u >> 2, s = u & 3;
        const unsigned char *b = ip + (size_t)cb * cb_bytes + plane * 32;
        if (s < 3) {
            const unsigned char *bs = b + s * (W + 1) * SB;
#pragma unroll
            for (int t = 0; t < NT; t++) p[t] = *(const bf16x8 *)(bs + (unsigned)aA[t]);
        } else {
            const unsigned char *bs = b + 2 * SB;
#pragma unroll
            for (int t = 0; t < NT; t++) p[t] = *(const bf16x8 *)(bs + (unsigned)aB[t]);
        }
    };
    auto mma = [&](const bf16x8 (&w)[FBW][3], int q, const bf16x8 (&p)[NT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int f = 0; f < FBW; f++) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[f][q], p[t], acc[f][t], 0, 0, 0);
    };
    // one slice with its plane 1 in `pa`: plane 2 -> pb, plane 3 -> pa, the next slice's plane 1 -> pb, its weights -> wn
    // (Measured and dropped: touching the next channel block's cache lines a slice and a half ahead -- one dword per 128-byte
    // line, values unused -- 45-50 ms against 38.8 ms per 20 x 256 batch.)
    auto slice = [&](const bf16x8 (&w)[FBW][3], bf16x8 (&wn)[FBW][3], bf16x8 (&pa)[NT], bf16x8 (&pb)[NT], int u) __attribute__((always_inline)) {
        const int un = u + 1 < steps ? u + 1 : u;
        load_x(pb, u, 1);
        load_w(wn, un);
        __builtin_amdgcn_sched_barrier(0);
        mma(w, 2, pa);
        mma(w, 1, pa);
        mma(w, 0, pa);
        __builtin_amdgcn_sched_barrier(0);
        load_x(pa, u, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(w, 1, pb);
        mma(w, 0, pb);
        __builtin_amdgcn_sched_barrier(0);
        load_x(pb, un, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(w, 0, pa);
        __builtin_amdgcn_sched_barrier(0);
    };
    (void)xoff;
    load_w(wa, 0);
    load_x(p0, 0, 0);
    for (int u = 0; u < steps; u += 2) { // steps = 4 NCB is even
        slice(wa, wb, p0, p1, u);
        slice(wb, wa, p1, p0, u + 1);
    }
    // ---- tap 8 of every channel block: K = 16 ------------------------------------------------------------------------------
    constexpr int T8 = (2 * (W + 1) + 2) * SB;
    for (int cb = 0; cb < NCB; cb++) {
        s16x4 w8[FBW][3], y[NT][3];
#pragma unroll
        for (int f = 0; f < FBW; f++)
#pragma unroll
            for (int q = 0; q < 3; q++) w8[f][q] = *(const s16x4 *)(wp[f] + (size_t)cb * GX3_PAIR_B + 4 * 3 * 64 * 16 + (q * 64 + lane) * 8);
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int q = 0; q < 3; q++) y[t][q] = *(const s16x4 *)(ip + (size_t)cb * cb_bytes + aC[t] + T8 + q * 32);
        auto m8 = [&](int qw, int qx) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int f = 0; f < FBW; f++) acc[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x3_k16(w8[f][qw]), x3_k16(y[t][qx]), acc[f][t], 0, 0, 0);
        };
        m8(2, 0);
        m8(1, 1);
        m8(0, 2);
        m8(1, 0);
        m8(0, 1);
        m8(0, 0);
    }
    // ---- epilogue ----------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < FBW; f++) {
        if (!fb_ok[f]) continue;
        const float *ep = gd.epi + ((size_t)layer * NCB + fbs[f]) * 48;
        const f32x4 scale = *(const f32x4 *)(ep + 16 + 4 * g), shift = *(const f32x4 *)(ep + 32 + 4 * g);
        unsigned char *o3 = out3 + (size_t)pos0 * pos_bytes + (size_t)fbs[f] * cb_bytes + TAP0 * SB;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (!valid[t]) continue;
            f32x4 sk = {0.f, 0.f, 0.f, 0.f};
            if (skip) { // tf.add(batch_norm_2, block input) before the ReLU: the block input is the three planes at the output's place
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    u32x2 pq = *(const u32x2 *)(o3 + aC[t] + q * 32);
                    sk[0] += __uint_as_float(pq[0] << 16);
                    sk[1] += __uint_as_float(pq[0] & 0xffff0000u);
                    sk[2] += __uint_as_float(pq[1] << 16);
                    sk[3] += __uint_as_float(pq[1] & 0xffff0000u);
                }
            }
            f32x4 yv;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = __builtin_fmaf(acc[f][t][r], scale[r], shift[r]);
                if (skip) v = v + sk[r];
                yv[r] = fmaxf(v, 0.f);
            }
            if constexpr (LAST) { // float32 [pos][cb][j][slot][4] for the heads kernel
                int q = t * 16 + nn, pp = q / HW, cell = q % HW, yy = cell / W, xx = cell % W;
                float *op = outf + (size_t)pos0 * pos_floats + (size_t)pp * pos_floats + (size_t)fbs[f] * 4 * PLANE + g * PLANE +
                            ((yy + 1) * (W + 1) + (xx + 1)) * 4;
                *(f32x4 *)op = yv;
            } else {
                u32x2 q1, q2, q3;
                x3_split4(yv, q1, q2, q3);
                *(u32x2 *)(o3 + aC[t]) = q1;
                *(u32x2 *)(o3 + aC[t] + 32) = q2;
                *(u32x2 *)(o3 + aC[t] + 64) = q3;
            }
        }
    }
}
