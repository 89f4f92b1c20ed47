// net.hip.h -- the residual tower + policy/value heads as ONE fused gfx950 kernel (F = 16 filters).
//
// Computes what Network.getEvaluation/getPolicy return for the graph NetworkFactory.__call__
// builds (/root/reference/src/NetworkFactory.py:22-183): conv3x3+bias -> BN -> ReLU,
// R residual blocks, 1x1 value/policy convs, per-pixel dense + spatial sum, tanh / softmax
// (+ Beta-noise mix, :176-182).  float32 throughout.
//
// Mapping to CDNA4:
//  * each WAVE owns PW whole positions; their activations ([pixel][16 ch], zero halo) never leave
//    LDS between layers, so HBM sees only the 16-byte packed boards and the outputs.
//    One 256-thread workgroup = 4 waves = 4*PW positions (Connect4: 16) -> 4096 positions are
//    exactly one workgroup per CU.  No __syncthreads(): waves share nothing.
//  * a 3x3 conv is an implicit GEMM on v_mfma_f32_16x16x4_f32:  D[f][pixel] += W[f][k] * X[k][pixel]
//    with M = 16 output channels, N = 16 pixels, K = 9*Cin.  Lane (j = lane>>4, n = lane&15) supplies
//    X for pixel n, channels 4j..4j+3 of one tap with ONE ds_read_b128, and the accumulator lane
//    layout (4 consecutive channels of one pixel) is exactly what the next layer reads, so the
//    epilogue (bias-initialised accumulator, BN scale/shift, skip add, ReLU) ends in one ds_write_b128.
//  * the f32 MFMA is a k-ordered fmaf chain, so results are bit-identical to the CPU oracle's
//    fmaf chains (same k order) up to expf/tanhf/powf in the heads.
//  * layer weights (9.2 KB each) stream from L2 into 36 VGPRs per layer, pre-swizzled on the host
//    into the lane order the MFMA A-operand wants.
#pragma once
#include <type_traits>
#include "games.hip.h"
#include "rng.hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef BB_FUSED_WMODE
#define BB_FUSED_WMODE 1 // k_net_fused16: weights stream from L2 (net_body WMODE)
#endif

// Lanes of a wave hand data to each other through LDS (a layer's output pixels are the next layer's neighbours).  The
// hardware runs a wave's LDS operations in order, but the COMPILER orders memory operations per lane: a store to
// out[x + c1] and a later load of out[x + c2] differ by a constant for "this lane", so it may hoist the load above the
// store -- although another lane's store feeds it.  This fence (wavefront scope: no instruction, only an ordering
// constraint) marks every such hand-over.
__device__ __forceinline__ void wave_lds_handover() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

struct NetDev {
    int R, D, A;             // blocks, value dense width, actions
    const float *w0;         // [steps0][64]            first conv, A-operand lane order
    const f32x4 *wt;         // [2R][9][64]             tower convs: lane (f=l&15, j=l>>4), .r = W[tap][4j+r][f]
    const float *epi;        // [1+2R][3][16]           bias, bn scale, bn shift
    const float *head;       // packed head parameters (offsets below)
    int head_floats;         // length of `head`
    int off_vk, off_v3, off_d1k, off_d1b, off_d2k, off_d2b, off_pk, off_p6, off_pdk, off_pdb;
    uint64_t seed;
    float alpha, eps;
    float inv_alpha, inv_beta; // 1 / alpha and 1 / (1 - alpha) (float32 quotients, formed once on the host)
    int dbg; // ablation switches for bb_timing_net, looked at in diagnostic builds only (ND_DBG): 1 no heads, 2 no tower, 8 no noise draws
};
#ifdef BB_DIAG
#define ND_DBG(bit) (nd.dbg & (bit))
#else
#define ND_DBG(bit) false
#endif

#ifdef BB_STAMPS
__device__ unsigned long long g_net_stamps[8];
#endif
#ifdef BB_STAMPS_NET
#define NSTAMP_ON 1 // diagnostic build: cycles per section of net_body, summed over calls (per workgroup in LDS, flushed by the persistent kernel)
__shared__ unsigned long long s_net_stamps[8];
#define NSTAMP(i) do { long long _t = clock64(); if (lane == 0) atomicAdd(&s_net_stamps[i], (unsigned long long)(_t - _ns)); _ns = clock64(); } while (0)
#else
#define NSTAMP(i) do {} while (0)
#endif

template <class G, int PW_>
struct NetGeom {
    static constexpr int H = G::H, W = G::W, CIN = G::C, A = G::A;
    static constexpr int PW = PW_;                     // positions per wave
    static constexpr int HW = H * W;
    static constexpr int SLOTS = (H + 2) * (W + 1) + 1; // pixel slots incl. zero halo
    static constexpr int CP = (CIN + 3) / 4 * 4;       // input planes padded to a float4
    static constexpr int NT = (PW * HW + 15) / 16;     // 16-pixel tiles per wave
    static constexpr int STEPS0 = (9 * CIN + 3) / 4;   // MFMA k-steps of the first conv
    static constexpr int ACT = PW * SLOTS * 16;        // floats per activation buffer
    static constexpr int PLANE = PW * SLOTS * 4;       // an activation buffer is [4 channel groups][PW][SLOTS][4 channels]:
                                                       // the 16 pixels of a tile x 16 B are contiguous -> conflict-free ds_read_b128
    static constexpr int STATE_FLOATS = (PW * (int)sizeof(typename G::State) + 15) / 16 * 4; // packed boards staged in LDS
    static constexpr int WAVE_FLOATS = 2 * ACT + PW * SLOTS * CP + STATE_FLOATS;
    static constexpr int LDS_BYTES = 4 * WAVE_FLOATS * 4;
    static_assert(LDS_BYTES <= 163840, "activations must fit the 160 KiB LDS");
};

// ---- wide policy head (DragonChess, A = 4032) ---------------------------------------------------------------------
// logits[a] = sum_p (r1[p]*k1[a] + (r0[p]*k0[a] + b[a])) = k1[a]*sum(r1) + k0[a]*sum(r0) + HW*b[a]: the spatial sum is
// taken once per position ("a global-sum-pool then a 2xA GEMV"), so the whole 4032-wide policy of a
// position is a function of TWO numbers (R0, R1); it differs from the oracle's per-pixel order only in rounding (1e-5,
// tests/test_gpu_net.py).  The softmax runs on the hardware exp2 (v_exp_f32 on (x - m) * log2 e) with one reciprocal per
// wave.  These pieces are shared by the dense form below (bb_net_eval, lock-step search: the 4032 probabilities go to
// memory) and the compact form of the one-wave-per-game kernel (mega_dc.hip.h: only max and 1/sum are kept and the
// expansion evaluates the probabilities of its ~20 legal moves itself) so that both give the same bits.
struct WideHead { // what is left of an evaluation once the 4032-wide row is not materialised
    float value, R0, R1, m, inv;
};
template <int HW>
__device__ __forceinline__ float wide_logit(float R0, float R1, float k0, float k1, float b) {
    return __builtin_fmaf(R1, k1, __builtin_fmaf(R0, k0, (float)HW * b));
}
__device__ __forceinline__ float wide_expterm(float l, float m) { return __builtin_amdgcn_exp2f((l - m) * 1.44269504088896340736f); }
template <int HW>
__device__ __forceinline__ float wide_prob(const WideHead &h, float k0, float k1, float b) {
    return wide_expterm(wide_logit<HW>(h.R0, h.R1, k0, k1, b), h.m) * h.inv;
}
// Wave-wide reductions without LDS traffic (all 64 lanes active).  The sum is the xor butterfly with strides 1, 2, 4, 8, 16,
// 32 -- the adjacent-pairs tree ((x0 + x1) + (x2 + x3)) + ... that the oracle's tree_sum64 folds in the same order: strides 1
// and 2 are quad permutes; for 4 and 8 the mirrored lane (7 - i, 15 - i) lies in the other quad / half, whose lanes all hold
// the same partial sum by then; after that the four rows are uniform and (r0 + r1) + (r2 + r3) is what strides 16 and 32 form
// on every lane.  (Six ds_bpermute round trips each before: ~2 k cycles per evaluation in the heads.)
__device__ __forceinline__ float dpp_f32(float v, int ctrl_sel) {
    const int i = __float_as_int(v);
    int r;
    switch (ctrl_sel) {
    case 0: r = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xf, 0xf, false); break;  // quad_perm [1,0,3,2]
    case 1: r = __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xf, 0xf, false); break;  // quad_perm [2,3,0,1]
    case 2: r = __builtin_amdgcn_update_dpp(i, i, 0x141, 0xf, 0xf, false); break; // row_half_mirror
    default: r = __builtin_amdgcn_update_dpp(i, i, 0x140, 0xf, 0xf, false); break; // row_mirror
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float lane_f32(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ float wave_max_f32(float v) {
    v = fmaxf(v, dpp_f32(v, 0));
    v = fmaxf(v, dpp_f32(v, 1));
    v = fmaxf(v, dpp_f32(v, 2));
    v = fmaxf(v, dpp_f32(v, 3));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}
// The same folds when only lanes 0 .. 15 hold anything (the rest 0.0 / -inf): the cross-row steps of wave_sum_f32 /
// wave_max_f32 would add exact zeros (take the maximum with -inf) -- row 0 alone gives the same bits.
__device__ __forceinline__ float row0_sum_f32(float v) {
    v += dpp_f32(v, 0);
    v += dpp_f32(v, 1);
    v += dpp_f32(v, 2);
    v += dpp_f32(v, 3);
    return lane_f32(v, 0);
}
__device__ __forceinline__ float row0_max_f32(float v) {
    v = fmaxf(v, dpp_f32(v, 0));
    v = fmaxf(v, dpp_f32(v, 1));
    v = fmaxf(v, dpp_f32(v, 2));
    v = fmaxf(v, dpp_f32(v, 3));
    return lane_f32(v, 0);
}
__device__ __forceinline__ float wave_sum_f32(float v) {
    v += dpp_f32(v, 0);
    v += dpp_f32(v, 1);
    v += dpp_f32(v, 2);
    v += dpp_f32(v, 3);
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
// max and 1 / sum(exp) of the A logits, lanes striding the actions (lane + 64 k, k ascending; partial sums per lane in
// that order, then the xor butterflies) -- the order the dense form uses.  PK: pointer type of the head weights
// (generic, or address_space(3) when the caller keeps them in LDS).
template <int A, int HW, class PK>
__device__ __forceinline__ void wide_head_stats(WideHead &h, PK pdk, PK pdb, int lane) {
    static_assert(A % 64 == 0, "every lane owns exactly A / 64 actions: the loops below carry no bounds test");
    constexpr int NPL = A / 64, UB = 9; // UB actions per batch: their 3 x UB weights are requested together, then consumed
    static_assert(NPL % UB == 0, "whole batches");
    // the lane's A / 64 logits stay in registers from the GEMV to the sum (as in the dense form of head_one): one pass over the
    // head weights -- round 2 formed every logit twice, 3 x 63 more LDS reads and 2 x 63 more fmas per lane and evaluation
    float sv[NPL];
    float m = -INFINITY;
#pragma unroll
    for (int k0 = 0; k0 < NPL; k0 += UB) {
        float w0[UB], w1[UB], wb[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int a = lane + 64 * (k0 + u);
            w0[u] = pdk[a];
            w1[u] = pdk[A + a];
            wb[u] = pdb[a];
        }
#pragma unroll
        for (int u = 0; u < UB; u++) {
            sv[k0 + u] = wide_logit<HW>(h.R0, h.R1, w0[u], w1[u], wb[u]);
            m = fmaxf(m, sv[k0 + u]);
        }
    }
    m = wave_max_f32(m);
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NPL; k++) tot += wide_expterm(sv[k], m); // (k ascending: the dense form's order)
    tot = wave_sum_f32(tot);
    h.m = m;
    h.inv = 1.0f / tot;
}

// ---- heads after the two 1x1 convolutions (NetworkFactory.py:105-183) -------------------------------------------------
// The reference applies `dense` to the last axis of the [H, W, c] head activations and then reduce_sums over H and W:
//     value:  relu(sum_p (r[p] * k[d] + b[d]))        policy logits:  sum_p (r0[p] * k0[a] + r1[p] * k1[a] + b[a])
// which is a global sum pool followed by a tiny dense layer:
//     relu(k[d] * R + HW * b[d]),                       k0[a] * R0 + k1[a] * R1 + HW * b[a],     R* = sum_p r*[p].
// This file computes the pooled form: three wave reductions and one fma per unit instead of H*W dependent fma/add
// steps per unit (460 of the ~1400 vector instructions of a Connect4 evaluation).  TensorFlow does not define the
// order of a reduce_sum, so neither form is "the" rounding; they agree to ~1e-7 relative (tests/test_oracle_net.py
// holds both against each other and against PyTorch) and the oracle's orc_net_forward restates THIS form -- sums as the
// fixed pairwise tree below -- so that GPU and oracle logits stay bit-identical.
//
// reduce_sum over H, W of one position: pixel p on lane p (lanes >= H*W hold 0.0), xor butterfly with strides 1 ... 32.
// Every lane ends with the same bits; the tree does not depend on how positions are packed into waves.
__device__ __forceinline__ float pooled_sum(float v) { return wave_sum_f32(v); }

// One position's value / policy from its pooled activations (R, R0, R1 are wave-uniform); all 64 lanes take part.
// `live` = the position exists (pos < n); outputs go to index `pos`.
template <class G>
__device__ __forceinline__ void head_one(const NetDev &nd, float R, float R0, float R1, int pos, bool live,
                                         const uint32_t *game_id, const int32_t *serial, int noise, float *value_out,
                                         float *logits_out, float *policy_out, int pstride, WideHead *compact,
                                         const float *noise_ready = nullptr) {
    constexpr int A = G::A, HW = G::H * G::W;
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_)); // (keeps the addresses below inside a persistent caller's loop: see net_body_x3)
    const int lane = lane_;
    const float *hp = nd.head;
    const int D = nd.D; // <= 64 (bb_load_weights)
    const float *d1k = hp + nd.off_d1k, *d1b = hp + nd.off_d1b, *d2k = hp + nd.off_d2k, *d2b = hp + nd.off_d2b;
    const float *pdk = hp + nd.off_pdk, *pdb = hp + nd.off_pdb;
    auto lane_f = [](float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
    // value head: dense_1 on the pooled activation (unit d on lane d), ReLU, dense_2 = the products folded by the fixed pairwise
    // tree of pooled_sum (TensorFlow leaves the order of a matmul's sum open; the oracle restates this one), + bias, tanh.
    // (Until round 3 an fma chain in unit order: D dependent steps of two v_readlane each on the tail every game waits for.)
    const float sdv = lane < D ? fmaxf(__builtin_fmaf(R, d1k[lane], (float)HW * d1b[lane]), 0.f) : 0.f;
    const float d2kv = lane < D ? d2k[lane] : 0.f;
    const float e = (D <= 16 ? row0_sum_f32(sdv * d2kv) : wave_sum_f32(sdv * d2kv)) + d2b[0];
    // tanh on the hardware exponential: 1 - 2 / (exp(2e) + 1) (v_exp_f32, v_rcp_f32: absolute error ~1e-7, the library tanhf is
    // ~60 instructions on the network wave's critical tail); saturates to +-1 through exp's overflow / underflow
    const float value = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(e * 2.88539008177792681472f) + 1.0f);
    if (live && lane == 0 && value_out) value_out[pos] = value;
    if constexpr (A > 64) {
        // wide policy (DragonChess, A = 4032): see the helpers above
        WideHead h;
        h.value = value;
        h.R0 = R0;
        h.R1 = R1;
        if (compact) { // the caller evaluates the few probabilities it needs from (R0, R1, m, 1/sum)
            // contract: a caller that asks for the compact form keeps nd.head in LDS (mega_dc.hip.h) -- the 3 x 4032
            // head weights are then read with ds_read instead of flat loads
            using LP = const __attribute__((address_space(3))) float *;
            wide_head_stats<A, HW>(h, (LP)pdk, (LP)pdb, lane);
            if (lane == 0) *compact = h;
            return;
        }
        if (!live) return;
        float *outp = policy_out ? policy_out + (size_t)pos * pstride : nullptr;
        // the A/64 logits of a lane stay in registers from the GEMV to the normalised store: one pass over the head
        // weights (L2-resident, shared by all positions) and ONE 16 KB write per position
        constexpr int NPL = (A + 63) / 64;
        float sv[NPL];
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < NPL; k++) {
            const int a = lane + 64 * k;
            if (a < A) {
                sv[k] = wide_logit<HW>(R0, R1, pdk[a], pdk[A + a], pdb[a]);
                if (logits_out) logits_out[(size_t)pos * A + a] = sv[k];
                m = fmaxf(m, sv[k]);
            } else {
                sv[k] = -INFINITY;
            }
        }
        m = wave_max_f32(m);
        if (outp) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < NPL; k++)
                if (lane + 64 * k < A) {
                    sv[k] = wide_expterm(sv[k], m);
                    tot += sv[k];
                }
            tot = wave_sum_f32(tot);
            float inv = 1.0f / tot;
            if (noise) { // getPolicy through bb_net_eval: policy = (1-eps)*softmax + eps*Beta(alpha,1-alpha); policy /= sum(policy)
                // (the tree paths pass noise = 0 for wide games: they mix the draws in at expansion, tree_dc.hip.h)
                const uint32_t gid = game_id ? game_id[pos] : (uint32_t)noise, ser = serial ? (uint32_t)serial[pos] : (uint32_t)pos;
                float t2 = 0.f;
#pragma unroll 1
                for (int k = 0; k < NPL; k++)
                    if (lane + 64 * k < A) {
                        sv[k] = (1.0f - nd.eps) * (sv[k] * inv) + nd.eps * bb_beta_noise(nd.seed, gid, ser, (uint32_t)(lane + 64 * k), nd.alpha);
                        t2 += sv[k];
                    }
                t2 = wave_sum_f32(t2);
                inv = 1.0f / t2;
            }
#pragma unroll
            for (int k = 0; k < NPL; k++)
                if (lane + 64 * k < A) outp[lane + 64 * k] = sv[k] * inv;
        }
    } else {
        static_assert(A > 64 || 2 * A <= 64, "the prior-noise draws use two lanes per action");
        // policy: action a on lane a (lanes >= A hold -inf / 0).  The softmax maximum and sums are the wave reductions of
        // pooled_sum -- the fixed pairwise tree the oracle restates (tree_sum64) -- not chains of A readlane steps.
        const bool act = lane < A;
        const float l = act ? wide_logit<HW>(R0, R1, pdk[act ? lane : 0], pdk[A + (act ? lane : 0)], pdb[act ? lane : 0]) : -INFINITY;
        constexpr bool ROW0 = A <= 16; // (every action on a lane of row 0)
        const float m = ROW0 ? row0_max_f32(l) : wave_max_f32(l);
        float pr = act ? wide_expterm(l, m) : 0.f; // exp(l - m) on v_exp_f32, as the wide head does
        const float tot = ROW0 ? row0_sum_f32(pr) : wave_sum_f32(pr);
        pr = pr * __builtin_amdgcn_rcpf(tot); // (v_rcp_f32: one ulp; an IEEE division is twelve instructions on this tail, twice)
        if (noise) { // policy = (1-eps)*softmax + eps*Beta(alpha,1-alpha); policy /= sum(policy)   (NetworkFactory.py:176-182)
            // two lanes per action: lane pair (2a, 2a+1) tries Philox pairs k and k+1 side by side
            const float ia = nd.inv_alpha, ib = nd.inv_beta;
            const int q = lane >> 1, sub = lane & 1;
            const bool drawing = q < A && live;
            const uint32_t gid = game_id ? game_id[live ? pos : 0] : (uint32_t)noise;
            const uint32_t ser = serial ? (uint32_t)serial[live ? pos : 0] : (uint32_t)pos;
            float r = ND_DBG(8) ? nd.alpha : -1.0f;
            if (noise_ready) r = nd.alpha; // (the draws were made by the caller: persistent kernel, tree waves; see below)
            for (uint32_t k = 0; k < 32 && __any(drawing && r < 0.0f); k += 2) {
                float mine = (drawing && r < 0.0f) ? bb_beta_pair(nd.seed, gid, ser, (uint32_t)q, ia, ib, k + sub) : -1.0f;
                float other = dpp_f32(mine, 0); // the pair's other lane (quad permute, no LDS round trip)
                float first = sub ? other : mine, second = sub ? mine : other; // pair k before pair k+1
                if (r < 0.0f) r = first >= 0.0f ? first : second;
            }
            r = r >= 0.0f ? r : nd.alpha;
            float nz = __shfl(r, 2 * (act ? lane : 0), 64); // action a's draw sits on lane 2a
            if (noise_ready) { // the same draws (bb_beta_noise: same trials in the same order), made by the tree wave that posted the leaf
                // -- after it posted it (mega2.hip.h): wait for its flag in the spare slot of the game's noise row, take the draws,
                // clear the flag for the game's next leaf.  (Bounded: a launch that is being aborted must still drain.)
                float *nr = const_cast<float *>(noise_ready);
                for (int spin = 0; spin < (1 << 16) && __hip_atomic_load(nr + G::S - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0.f; spin++)
                    __builtin_amdgcn_s_sleep(1);
                nz = act ? nr[lane] : 0.f;
                wave_lds_handover();
                if (lane == 0) __hip_atomic_store(nr + G::S - 1, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            pr = (1.0f - nd.eps) * pr + nd.eps * (act ? nz : 0.f);
            const float t2 = ROW0 ? row0_sum_f32(pr) : wave_sum_f32(pr);
            pr = pr * __builtin_amdgcn_rcpf(t2);
        }
        if (live && act && logits_out) logits_out[(size_t)pos * A + lane] = l;
        if (live && act && policy_out) policy_out[(size_t)pos * pstride + lane] = pr;
    }
}

// head_one's value / policy tails for a dense game, computed by the S lanes of a tree wave that own the game (persistent
// kernel: the network wave stops at the pooled activations R, R0, R1 and the tree wave that picks the result up finishes it --
// the network waves are the busy side).  Same operations in the same order as head_one, so the same bits: every lane forms
// the value chain redundantly (no cross-lane traffic), action a sits on lane a of the game's lane group, the ordered softmax
// sums gather their terms with ds_bpermute (`base` = first lane of the group).  `nz` = this lane's prior-noise draw (made
// ahead by the same tree wave).  Returns the value; *prior_out = this lane's prior (lanes >= A: 0).
template <class G>
__device__ __forceinline__ float head_tree(const NetDev &nd, const float *hp, float R, float R0, float R1, int lane, int base, bool noise,
                                           float nz, float *prior_out) {
    constexpr int A = G::A, HW = G::H * G::W;
    const int D = nd.D;
    const float *d1k = hp + nd.off_d1k, *d1b = hp + nd.off_d1b, *d2k = hp + nd.off_d2k, *d2b = hp + nd.off_d2b;
    const float *pdk = hp + nd.off_pdk, *pdb = hp + nd.off_pdb;
    // (the orders of head_one: products / terms in slots 0 .. 63 of a vector -- the rest zero -- folded by strides 1, 2, ..., 32,
    // i.e. the tree of wave_sum_f32; zeros add exactly, so only the occupied part of the tree is walked)
    auto tree16 = [](float *t) __attribute__((always_inline)) { // 16 slots -> t[0]
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
            for (int i = 0; i < 16; i += 2 * o) t[i] = t[i] + t[i + o];
        return t[0];
    };
    float e = 0.f;
    { // (four units per round trip: the three parameter rows are 16-byte aligned in the packed head -- engine.hip `push`)
        float blk[4] = {0.f, 0.f, 0.f, 0.f}; // sums of slots 0-15, 16-31, 32-47, 48-63
        for (int b16 = 0; b16 < 4 && b16 * 16 < D; b16++) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = 0.f;
            for (int dd = b16 * 16; dd < D && dd < b16 * 16 + 16; dd++) {
                const float sdv = fmaxf(__builtin_fmaf(R, d1k[dd], (float)HW * d1b[dd]), 0.f);
                t[dd & 15] = sdv * d2k[dd];
            }
            blk[b16] = tree16(t);
        }
        e = ((blk[0] + blk[1]) + (blk[2] + blk[3])) + d2b[0];
    }
    const float value = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(e * 2.88539008177792681472f) + 1.0f);
    const bool act = lane < A;
    const int la = act ? lane : 0;
    const float l = act ? wide_logit<HW>(R0, R1, pdk[la], pdk[A + la], pdb[la]) : -INFINITY;
    static_assert(A <= 16, "head_tree: the actions of a dense game fit one 16-slot block of the tree");
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < A; a++) m = fmaxf(m, __shfl(l, base + a, 64));
    float pr = act ? wide_expterm(l, m) : 0.f;
    auto gsum = [&](float v) __attribute__((always_inline)) {
        float t[16];
#pragma unroll
        for (int a = 0; a < 16; a++) t[a] = a < A ? __shfl(v, base + a, 64) : 0.f;
        return tree16(t);
    };
    pr = pr * __builtin_amdgcn_rcpf(gsum(pr));
    if (noise) {
        pr = (1.0f - nd.eps) * pr + nd.eps * (act ? nz : 0.f);
        pr = pr * __builtin_amdgcn_rcpf(gsum(pr));
    }
    *prior_out = act ? pr : 0.f;
    return value;
}

// The same for PW positions whose per-pixel head activations sit in memory (LDS): rv[PW*HW], rp[PW*HW][2].
template <class G, int PW>
__device__ __forceinline__ void net_head_tail(const NetDev &nd, int n, int pos0, const int *slot_list, const float *rv,
                                              const float *rp, const uint32_t *game_id, const int32_t *serial, int noise,
                                              float *value_out, float *logits_out, float *policy_out, int pstride,
                                              WideHead *compact = nullptr) {
    constexpr int HW = G::H * G::W;
    static_assert(HW <= 64, "one lane per pixel of a position");
    const int lane = threadIdx.x & 63;
    for (int pp = 0; pp < PW; pp++) {
        const bool live = pos0 + pp < n;
        if (!live && !compact) break;
        const int pos = live ? (slot_list ? slot_list[pos0 + pp] : pos0 + pp) : 0;
        const float x = lane < HW ? rv[pp * HW + lane] : 0.f;
        const float x0 = lane < HW ? rp[2 * (pp * HW + lane)] : 0.f, x1 = lane < HW ? rp[2 * (pp * HW + lane) + 1] : 0.f;
        head_one<G>(nd, pooled_sum(x), pooled_sum(x0), pooled_sum(x1), pos, live, game_id, serial, noise, value_out, logits_out,
                    policy_out, pstride, compact ? compact + pp : nullptr);
    }
}

// The whole network for the PW positions [pos0, pos0+PW) of a batch of n, computed by ONE wave in its
// own LDS region `wlds` (NetGeom<G,PW>::WAVE_FLOATS floats).  slot_list != nullptr: batch entry i is
// engine slot slot_list[i] (inputs are read from, and outputs written to, that slot's mailbox).
// WMODE: how the tower's operands reach the MFMAs (see conv_layer below): 0 plain, 1 weights from L2 with next-layer
// prefetch, 2 weights in LDS with next-tap prefetch.
// Team: the evaluation may be shared by PARTS waves working in the SAME LDS region (mega_dc.hip.h: two waves of a
// DragonChess game on two SIMDs): every wave takes NT / PARTS of the 16-pixel tiles of each conv layer, part 0 alone does the
// prologue (zero fill, input planes) and the heads, and `team.sync()` is called wherever one part's LDS writes are another
// part's reads: after the prologue and after every conv layer.  The default team is the wave on its own.
struct SoloTeam {
    static constexpr int PARTS = 1, PART = 0;
    __device__ __forceinline__ void sync() const {}
};
template <class G, int PW, int WMODE = 0, class Team = SoloTeam>
__device__ __forceinline__ void net_body(const NetDev &nd, int n, int pos0, const int *slot_list, float *wlds,
                                         const typename G::State *states, const int8_t *planes,
                                         const uint32_t *game_id, const int32_t *serial, int noise, float *value_out,
                                         float *logits_out, float *policy_out, int pstride, bool zero_lds = true,
                                         WideHead *compact = nullptr, Team team = Team()) {
    using NG = NetGeom<G, PW>;
    static_assert(NG::NT % Team::PARTS == 0, "whole tiles per team member");
    constexpr int W = NG::W, CIN = NG::CIN, A = NG::A, HW = NG::HW, SLOTS = NG::SLOTS, CP = NG::CP,
                  NT = NG::NT / Team::PARTS, T0 = Team::PART * NT, STEPS0 = NG::STEPS0, ACT = NG::ACT, PLANE = NG::PLANE;
    const int lane = threadIdx.x & 63;
    const int j = lane >> 4, nn = lane & 15;
    float *actA = wlds;
    float *actB = actA + ACT;
    float *inp = actB + ACT;
    auto OI = [&](int pos) { return slot_list ? slot_list[pos] : pos; };
#ifdef BB_STAMPS_NET
    long long _ns = clock64();
#endif

    // ---- issue every global load of the prologue first, then zero LDS while they are in flight ---------
    // (one packed board per lane, the first conv's weights and epilogue constants)
    const int my_pp = lane % PW;
    const typename G::State my_state = planes ? G::initial() : states[OI(pos0 + my_pp < n ? pos0 + my_pp : pos0)];
    float w0r[STEPS0];
#pragma unroll
    for (int s = 0; s < STEPS0; s++) w0r[s] = nd.w0[s * 64 + lane];
    const f32x4 bias0 = *(const f32x4 *)(nd.epi + 4 * j), scale0 = *(const f32x4 *)(nd.epi + 16 + 4 * j),
                shift0 = *(const f32x4 *)(nd.epi + 32 + 4 * j);
    // ---- zero this wave's LDS (halo pixels must read as 0 forever) --------------------------
    if (zero_lds && Team::PART == 0) { // a persistent caller zeroes once: halos are never written, interiors are always rewritten
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 *p = (f32x4 *)actA;
        for (int i = lane; i < NG::WAVE_FLOATS / 4; i += 64) p[i] = z;
    }
    // the PW boards go through LDS so that every lane can decode any cell of any position
    typename G::State *sst = (typename G::State *)(inp + PW * SLOTS * CP);
    wave_lds_handover(); // (the zero fill above and the data written below touch the same words from different lanes)
    if (!planes && lane < PW && Team::PART == 0) sst[lane] = my_state;
    wave_lds_handover();
    // ---- input planes -> inp[pos][slot][CP] ---------------------------------------------------
    for (int q = lane; Team::PART == 0 && q < PW * HW; q += 64) {
        int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
        int pos = pos0 + pp;
        if (pos >= n) continue;
        float *dst = inp + (pp * SLOTS + (y + 1) * (W + 1) + (x + 1)) * CP;
        if (planes) {
            const int8_t *src = planes + ((size_t)pos * HW + cell) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; c++) dst[c] = (float)src[c];
        } else {
            int8_t v[CIN];
            G::encode_cell(sst[pp], y, x, v);
            if constexpr (CIN > 4) { // wide input (DragonChess: 17 planes): whole float4 stores (CP is a multiple of 4; the pad planes are 0)
#pragma unroll
                for (int c4 = 0; c4 < CP / 4; c4++) {
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (4 * c4 + r < CIN) ? (float)v[4 * c4 + r] : 0.f;
                    *(f32x4 *)(dst + 4 * c4) = o;
                }
            } else {
#pragma unroll
                for (int c = 0; c < CIN; c++) dst[c] = (float)v[c];
            }
        }
    }
    // ---- per-tile addressing: lane (j, nn) <-> pixel nn of the tile, channels 4j..4j+3 ---------
    // The lanes past the last pixel (42 pixels occupy 48 MFMA columns) work on the LAST pixel once more: they compute
    // and store the values its own lane stores, to the same addresses, so nothing below needs a lane mask.
    // Offsets are biased by the most negative tap displacement: every tap of the 3x3 window is then a non-negative
    // IMMEDIATE offset of the ds_read (no address arithmetic per operand read), and the tile's own pixel sits at +CTR.
    constexpr int TAP0 = (W + 1) + 1;            // slots between the window's top-left tap and its centre
    constexpr int CTR = TAP0 * 4;                // ... in floats of an activation plane
    int aoffb[NT];  // float offset of (channel group j, pos, slot - TAP0) inside an activation buffer
    int ioffb[NT];  // float offset of (pos, slot - TAP0) inside inp
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int q = (T0 + t) * 16 + nn;
        int qq = q < PW * HW ? q : PW * HW - 1;
        int pp = qq / HW, cell = qq % HW, y = cell / W, x = cell % W;
        int slot = (y + 1) * (W + 1) + (x + 1);
        aoffb[t] = j * PLANE + (pp * SLOTS + slot - TAP0) * 4;
        ioffb[t] = (pp * SLOTS + slot - TAP0) * CP;
    }
    wave_lds_handover(); // input planes (and the staged boards) written above are read by other lanes below
    team.sync();
    NSTAMP(0);
    f32x4 acc[NT];
    // ---- first conv: K = 9*CIN in natural (tap, c) order, 4 k per MFMA ---------------------------
    {
        const f32x4 bias = bias0, scale = scale0, shift = shift0;
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = bias;
        auto step_toff = [&](int s) __attribute__((always_inline)) {
            int k = 4 * s + j;
            int kk = k < 9 * CIN ? k : 9 * CIN - 1; // padded k: weight is 0, any readable address will do
            int tap = kk / CIN, c = kk % CIN;
            return ((tap / 3) * (W + 1) + (tap % 3)) * CP + c;
        };
        if constexpr (WMODE == 1 && NT <= 6) {
            // a lone wave per SIMD: the pixel operands of k-step s + 1 are requested before the MFMAs of step s are issued
            float b[NT], bn[NT];
            {
                const int toff = step_toff(0);
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = inp[ioffb[t] + toff];
            }
#pragma unroll
            for (int s = 0; s < STEPS0; s++) {
                if (s + 1 < STEPS0) {
                    const int toffn = step_toff(s + 1);
#pragma unroll
                    for (int t = 0; t < NT; t++) bn[t] = inp[ioffb[t] + toffn];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0r[s], b[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = bn[t];
            }
        } else {
#pragma unroll
            for (int s = 0; s < STEPS0; s++) {
                const int toff = step_toff(s);
                float a = w0r[s];
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    float b = inp[ioffb[t] + toff];
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = fmaxf(__builtin_fmaf(acc[t][r], scale[r], shift[r]), 0.f);
            *(f32x4 *)(actA + aoffb[t] + CTR) = y;
        }
    }
    wave_lds_handover();
    team.sync();
    NSTAMP(1);
    // ---- residual tower: 2R convs, K order (tap, r, j) with channel c = 4j + r ---------------------
    // One layer = 9 taps x 4 k-steps x NT tiles of MFMAs + the epilogue; written once (conv_layer) and instantiated for
    // the first conv of a block (actA -> actB) and the second (actB -> actA, + the block's input before the ReLU), so
    // that buffers and the skip connection are compile-time facts of each copy.  How the operands arrive (WMODE):
    //   0  read where they are used (batch kernels: other waves of the SIMD cover the latency)
    //   1  weights stream from L2: the NEXT layer's 9 weight vectors are requested while this layer's MFMAs run
    //      (36 more VGPRs); with few tiles the pixel operands of tap + 1 are requested ahead of tap's MFMAs as well
    //   2  weights in LDS (persistent kernel): weight vector and pixel operands of tap + 1 are requested ahead of tap's MFMAs
    const int R_eff = ND_DBG(2) ? 0 : nd.R;
    const int L = 2 * R_eff;
    f32x4 wnext[WMODE == 1 ? 9 : 1];
    f32x4 enext[WMODE == 1 ? 3 : 1];
    if constexpr (WMODE == 1) {
        if (L > 0) {
#pragma unroll
            for (int tap = 0; tap < 9; tap++) wnext[tap] = nd.wt[(size_t)tap * 64 + lane];
            const float *ep = nd.epi + 48;
            enext[0] = *(const f32x4 *)(ep + 4 * j);
            enext[1] = *(const f32x4 *)(ep + 16 + 4 * j);
            enext[2] = *(const f32x4 *)(ep + 32 + 4 * j);
        }
    }
    auto conv_layer = [&](const int l, const float *in, float *out, auto skip_tag) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip_tag)::value; // tf.add(batch_norm_2, block input) before the ReLU
        const float *ep = nd.epi + (size_t)(1 + l) * 48;
        f32x4 bias, scale, shift;
        if constexpr (WMODE == 1) {
            f32x4 w[9];
#pragma unroll
            for (int tap = 0; tap < 9; tap++) w[tap] = wnext[tap];
            bias = enext[0];
            scale = enext[1];
            shift = enext[2];
            if (l + 1 < L) { // request layer l + 1 now; the scheduling barrier keeps the requests up here, ahead of this layer's MFMAs
#pragma unroll
                for (int tap = 0; tap < 9; tap++) wnext[tap] = nd.wt[((size_t)(l + 1) * 9 + tap) * 64 + lane];
                enext[0] = *(const f32x4 *)(ep + 48 + 4 * j);
                enext[1] = *(const f32x4 *)(ep + 48 + 16 + 4 * j);
                enext[2] = *(const f32x4 *)(ep + 48 + 32 + 4 * j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = bias;
            if constexpr (NT <= 6) {
                // a lone wave per SIMD (one-wave-per-game kernel, small batches) has nobody to hide its LDS round trips
                f32x4 b[NT], bn[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = *(const f32x4 *)(in + aoffb[t]);
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    if (tap + 1 < 9) {
                        const int toffn = (((tap + 1) / 3) * (W + 1) + ((tap + 1) % 3)) * 4;
#pragma unroll
                        for (int t = 0; t < NT; t++) bn[t] = *(const f32x4 *)(in + aoffb[t] + toffn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][r], b[t][r], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NT; t++) b[t] = bn[t];
                }
            } else {
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int toff = ((tap / 3) * (W + 1) + (tap % 3)) * 4;
                    f32x4 b[NT];
#pragma unroll
                    for (int t = 0; t < NT; t++) b[t] = *(const f32x4 *)(in + aoffb[t] + toff);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][r], b[t][r], acc[t], 0, 0, 0);
                }
            }
        } else {
            bias = *(const f32x4 *)(ep + 4 * j);
            scale = *(const f32x4 *)(ep + 16 + 4 * j);
            shift = *(const f32x4 *)(ep + 32 + 4 * j);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = bias;
            if constexpr (WMODE == 2) {
                f32x4 b[NT], bn[NT], wc, wn;
                wc = nd.wt[((size_t)l * 9) * 64 + lane];
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = *(const f32x4 *)(in + aoffb[t]);
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    if (tap + 1 < 9) {
                        const int toffn = (((tap + 1) / 3) * (W + 1) + ((tap + 1) % 3)) * 4;
                        wn = nd.wt[((size_t)l * 9 + tap + 1) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < NT; t++) bn[t] = *(const f32x4 *)(in + aoffb[t] + toffn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[r], b[t][r], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    wc = wn;
#pragma unroll
                    for (int t = 0; t < NT; t++) b[t] = bn[t];
                }
            } else {
                f32x4 w[9];
#pragma unroll
                for (int tap = 0; tap < 9; tap++) w[tap] = nd.wt[((size_t)l * 9 + tap) * 64 + lane];
#pragma unroll
                for (int tap = 0; tap < 9; tap++) {
                    const int toff = ((tap / 3) * (W + 1) + (tap % 3)) * 4;
                    f32x4 b[NT];
#pragma unroll
                    for (int t = 0; t < NT; t++) b[t] = *(const f32x4 *)(in + aoffb[t] + toff);
#pragma unroll
                    for (int r = 0; r < 4; r++)
#pragma unroll
                        for (int t = 0; t < NT; t++)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[tap][r], b[t][r], acc[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 y;
            f32x4 sk = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SKIP) sk = *(const f32x4 *)(out + aoffb[t] + CTR);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = __builtin_fmaf(acc[t][r], scale[r], shift[r]);
                if constexpr (SKIP) v = v + sk[r];
                y[r] = fmaxf(v, 0.f);
            }
            *(f32x4 *)(out + aoffb[t] + CTR) = y;
        }
        wave_lds_handover(); // this layer's pixels are the next layer's (and the heads') operands, across lanes
        team.sync();
    };
    for (int blk = 0; blk < R_eff; blk++) {
        conv_layer(2 * blk, actA, actB, std::false_type{});
        conv_layer(2 * blk + 1, actB, actA, std::true_type{});
    }
    if constexpr (Team::PART != 0) return; // the heads are part 0's
    if (ND_DBG(1)) {
        if (value_out && lane == 0) value_out[OI(pos0)] = acc[0][0];
        return;
    }
    NSTAMP(2);
    // ---- heads (tower output is in actA; actB and inp are scratch now) --------------------------
    const float *hp = nd.head;
    const float *vk = hp + nd.off_vk, *v3 = hp + nd.off_v3, *pk = hp + nd.off_pk, *p6 = hp + nd.off_p6;
    auto head_convs = [&](int q, float &o, float &o0, float &o1) __attribute__((always_inline)) { // 1x1 convs + BN + ReLU of pixel q
        int pp = q / HW, cell = q % HW, y = cell / W, x = cell % W;
        const float *xp = actA + (pp * SLOTS + (y + 1) * (W + 1) + (x + 1)) * 4;
        float av = v3[0], a0 = p6[0], a1 = p6[1];
#pragma unroll
        for (int c4 = 0; c4 < 4; c4++) {
            f32x4 xv = *(const f32x4 *)(xp + c4 * PLANE);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int c = 4 * c4 + r;
                av = __builtin_fmaf(xv[r], vk[c], av);
                a0 = __builtin_fmaf(xv[r], pk[2 * c], a0);
                a1 = __builtin_fmaf(xv[r], pk[2 * c + 1], a1);
            }
        }
        o = fmaxf(__builtin_fmaf(av, v3[1], v3[2]), 0.f);
        o0 = fmaxf(__builtin_fmaf(a0, p6[2], p6[4]), 0.f);
        o1 = fmaxf(__builtin_fmaf(a1, p6[3], p6[5]), 0.f);
    };
    if constexpr (PW == 1) {
        // one position per wave: pixel p's head activations stay in lane p's registers, pooled by butterflies -- no LDS scratch
        static_assert(HW <= 64, "one lane per pixel");
        float x = 0.f, x0 = 0.f, x1 = 0.f;
        if (lane < HW) head_convs(lane, x, x0, x1);
        NSTAMP(3);
        const bool live = pos0 < n;
        head_one<G>(nd, pooled_sum(x), pooled_sum(x0), pooled_sum(x1), live ? OI(pos0) : 0, live, game_id, serial, noise, value_out,
                    logits_out, policy_out, pstride, compact);
        NSTAMP(4);
    } else {
        float *rv = actB;               // [PW*HW] value-conv output
        float *rp = actB + PW * HW;     // [PW*HW][2] policy-conv output
        for (int q = lane; q < PW * HW; q += 64) {
            float x, x0, x1;
            head_convs(q, x, x0, x1);
            rv[q] = x;
            rp[2 * q] = x0;
            rp[2 * q + 1] = x1;
        }
        wave_lds_handover();
        NSTAMP(3);
        net_head_tail<G, PW>(nd, n, pos0, slot_list, rv, rp, game_id, serial, noise, value_out, logits_out, policy_out, pstride,
                             compact);
        NSTAMP(4);
        if (!zero_lds) { // persistent caller: the head scratch overlaid actB's halo slots -- restore the zeros
            wave_lds_handover();
            for (int i = lane; i < 3 * PW * HW; i += 64) actB[i] = 0.f;
            wave_lds_handover();
        }
    }
}

// fixed PW positions per wave, batch entry i == output index i (bb_net_eval, lock-step search)
template <class G, int PW>
__global__ void __launch_bounds__(256, 1)
k_net_fused16(NetDev nd, int n, const typename G::State *states, const int8_t *planes,
              const uint32_t *game_id, const int32_t *serial, int noise, float *value_out, float *logits_out,
              float *policy_out, int pstride) {
    using NG = NetGeom<G, PW>;
    __shared__ __attribute__((aligned(16))) float lds[4 * NG::WAVE_FLOATS];
    const int wave = threadIdx.x >> 6;
    const int pos0 = (blockIdx.x * 4 + wave) * PW;
    if (pos0 >= n) return; // whole wave idle (no block-level sync anywhere)
    net_body<G, PW, BB_FUSED_WMODE>(nd, n, pos0, nullptr, lds + wave * NG::WAVE_FLOATS, states, planes, game_id, serial, noise, value_out,
                          logits_out, policy_out, pstride);
}

// Compacted batch: *n_ptr leaves were posted this round (slots listed in slot_list).  The grid is always
// 256 workgroups x 4 waves; the leaves are dealt evenly, pw = ceil(n / 1024) per wave, so a round with
// fewer fresh leaves runs fewer MFMA tiles per wave instead of leaving CUs idle.
template <class G, int PWMAX>
__global__ void __launch_bounds__(256, 1)
k_net_compact(NetDev nd, const int *n_ptr, const int *slot_list, const typename G::State *states,
              const uint32_t *game_id, const int32_t *serial, int noise, float *value_out, float *policy_out,
              int pstride) {
    __shared__ __attribute__((aligned(16))) float lds[4 * NetGeom<G, PWMAX>::WAVE_FLOATS];
    const int n = *n_ptr;
    const int wave = threadIdx.x >> 6, gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    int pw = (n + nw - 1) / nw;
    if (pw > PWMAX) pw = PWMAX; // host guarantees n <= nw * PWMAX
    const int pos0 = gw * pw;
    if (pos0 >= n) return;
    float *wl = lds + wave * NetGeom<G, PWMAX>::WAVE_FLOATS;
    switch (pw) {
    case 1: net_body<G, 1>(nd, n, pos0, slot_list, wl, states, nullptr, game_id, serial, noise, value_out, nullptr, policy_out, pstride); break;
    case 2: net_body<G, 2>(nd, n, pos0, slot_list, wl, states, nullptr, game_id, serial, noise, value_out, nullptr, policy_out, pstride); break;
    case 3: net_body<G, 3>(nd, n, pos0, slot_list, wl, states, nullptr, game_id, serial, noise, value_out, nullptr, policy_out, pstride); break;
    default: net_body<G, PWMAX>(nd, n, pos0, slot_list, wl, states, nullptr, game_id, serial, noise, value_out, nullptr, policy_out, pstride); break;
    }
}
