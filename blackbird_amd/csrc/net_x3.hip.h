// net_x3.hip.h -- the 16-filter residual tower on the bf16 matrix pipe, float32 results by a three-way operand split.
//
// Why: on gfx950 the f32 MFMA (v_mfma_f32_16x16x4_f32, 64 FLOP/cycle/SIMD) runs on the SIMD's vector ALUs -- while one
// wave issues it back to back, every other wave's vector instruction on that SIMD waits ~70 cycles (tools/micro/coissue:
// x10 for a dependent v_fma chain, x7.5 for a ds_read chain; wave priority changes nothing).  The tree waves and the
// heads / epilogues of the persistent self-play kernel therefore PAY for every f32 MFMA cycle: matrix and vector time add
// up, and the kernel sat at ~87 % of the SIMDs' combined issue time.  v_mfma_f32_16x16x32_bf16 has its own pipe (16
// cycles for 8x the K, vector issue held for 8 of them; the same probes slow down x1.1 - x1.4 beside it).
//
// How: a float32 value is the exact sum of three bf16 values (8 + 8 + 8 significant bits):
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// and  w * x = sum_{i + j <= 4} w_i x_j  up to terms below 2^-24 |w x|  (w2 x3, w3 x2, w3 x3), each product exact in the
// f32 accumulator.  Six bf16 MFMAs (w1x1, w1x2, w2x1, w1x3, w2x2, w3x1) at 16 cycles for K = 32 replace eight f32 MFMAs
// at 32 cycles for K = 4 each: 96 against 256 cycles per 32-deep slice of a 16 x 16 tile.  The accumulation order inside
// the MFMA differs from the oracle's k-ordered fmaf chain, so logits agree with the oracle to ~1e-6 instead of bit for
// bit (north star: 1e-5; tests/test_gpu_net.py states the bound).  The tree arithmetic is unaffected: the oracle's
// search is compared on the evaluations the engine itself produced (bb_net_eval_keyed, tests/test_gpu_noise_parity.py).
//
// Layout (one wave = one position, its activations never leave LDS):
//   X    [slot][plane 0..2][16 ch] bf16 = 96 B per pixel slot (zero halo: one extra slot per board row and two extra rows, as
//        in net.hip.h); the three planes of a pixel's 8 channels are one ds_read_b128 each.  The layout is designed against
//        the LDS banking of MI355X_MICROARCH.md (a ds_read_b128 is served in the four lane groups {0-3, 12-15, 20-27},
//        {4-11, 16-19, 28-31}, ..., 64 banks; a ds_write_b64 in four groups of 16 consecutive lanes, 32 banks) with
//        tools/model/lds_banks.py:
//          * an MFMA tile is 16 CONSECUTIVE slots -- two board rows of Connect4 with their halo column (the 16 lanes' 96-byte
//            strides tile the 64 banks exactly); DragonChess (9 slots per row) gives lanes {0-3, 12-15} the first row of a
//            tile and lanes 4-11 the second.  Columns that are halo slots compute garbage and are not stored.
//          * 8-slot rows: the two 16-byte halves of a plane swap places in slots 4..7 of every 8 (X3Geom::swz), which makes
//            the 8-byte epilogue stores 2-way instead of 4-way conflicted and costs the readers nothing (the half is part of
//            a per-lane address that exists anyway).
//        Round 2's layout (112-byte slots, 14-pixel tiles) ran every operand read as a 2-way conflict in one of its four lane
//        groups: 306 LDS cycles of activation reads per layer where 168 suffice (PMC: 38 % of the LDS cycles were conflicts).
//        Written in place (a layer's reads are complete before its epilogue starts, the block input for the skip connection
//        stays in registers) -- or into a second buffer (PP).
//   the input planes (small integers: exact in bf16, no split) sit in the first bytes of their pixel's slot (8 B, DragonChess'
//   17 planes 64 B); the first conv's outputs overwrite them after its reads.
//   the last layer leaves float32 [slot][16 ch] in the first 64 B of each interior slot for the heads.
//   K order of a tower layer: taps (0,1), (3,4), (6,7), (2,5) as four K = 32 slices (lane group g = lane >> 4 holds
//   channels 8 (g & 1) .. +7 of the slice's tap g >> 1), then tap 8 -- K = 16 per plane pair -- as THREE K = 32 products by
//   concatenating planes along K:  [w1|w1].[x1;x2] + [w2|w2].[x1;x2] + [w1|w3].[x3;x1]  (lane groups 0,1 hold the first
//   plane of each pair, groups 2,3 the second): 27 MFMAs per tile and layer instead of 30 with zero-extended K = 16 operands.
//   Products are accumulated in ONE order in every variant (default / LEAN / PP): per slice w3 x1, w2 x1, w1 x1, w2 x2, w1 x2,
//   w1 x3, then the tap-8 products in the order above reversed -- so all launch structures give the same bits.
//   Weights: pre-split and pre-swizzled on the host into the A-operand lane order (engine.hip: pack_x3); planes 1 and 2 may
//   live in the caller's LDS, plane 3 always streams from L2 a layer ahead.
//   Callers: k_net_x3 (bb_net_eval, lock-step and asynchronous-round search), k_selfplay_queue (mega2.hip.h),
//   k_dc_selfplay_fused (mega_dc.hip.h); the per-layer launches of wider networks use the same split in gnet_x3.hip.h.
#pragma once
#include "net.hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <class G>
struct X3Geom {
    static constexpr int H = G::H, W = G::W, CIN = G::C, A = G::A, HW = H * W;
    static_assert(CIN <= 32, "the first conv packs at most 32 input planes per tap");
    static_assert(HW <= 64, "one lane per pixel in the heads");
    static constexpr bool WIDE_IN = CIN > 4;                        // DragonChess (17 planes): one K = 32 slice per tap
    static constexpr int RS = W + 1;                                // slots per board row: the pixels + one zero halo column
    static constexpr int S0 = RS + 1;                               // slot of pixel (0, 0)
    // MFMA tile = 16 pixel columns.  CONTIG (8 or 4 slots per row): 16 consecutive slots starting at S0 + 16 t, halo columns
    // included (their results are not stored).  Otherwise (DragonChess, 9 slots per row): two board rows, lanes {0-3, 12-15}
    // the first and lanes 4-11 the second -- the lane groups of a ds_read_b128 then see 16 different bank quads either way.
    static constexpr bool CONTIG = (16 % RS) == 0;
    static_assert(CONTIG || (W == 8 && H % 2 == 0), "tile shapes exist for 4-, 8- and 9-slot rows");
    static constexpr int NT = CONTIG ? (H * RS + 15) / 16 : H / 2;
    static constexpr bool SWZ = RS == 8;                            // see swz()
    static constexpr int SLOT_B = 96;                               // 3 planes x 16 channels x bf16
    static constexpr int SLOTS_BOARD = (H + 2) * RS + 2;
    static constexpr int SLOTS_TILES = CONTIG ? S0 + 16 * NT + RS + 2 : 0; // the last tile's halo columns read this far
    static constexpr int SLOTS = SLOTS_BOARD > SLOTS_TILES ? SLOTS_BOARD : SLOTS_TILES;
    static constexpr int X_B = SLOTS * SLOT_B;
    static constexpr int STATE_B = ((int)sizeof(typename G::State) + 15) / 16 * 16;
    static constexpr int WAVE_BYTES = X_B + STATE_B;
    static constexpr int WAVE_BYTES_PP = 2 * X_B + STATE_B; // two activation buffers (net_body_x3 PP: DragonChess kernel)
    // slot of column nn of tile t
    __host__ __device__ static constexpr int tile_slot(int t, int nn) {
        if (CONTIG) return S0 + 16 * t + nn;
        const int second = (nn >= 4 && nn < 12) ? 1 : 0, col = nn < 4 ? nn : (nn < 12 ? nn - 4 : nn - 8);
        return (2 * t + second + 1) * RS + col + 1;
    }
    // ... is a pixel of the board (else a halo slot: computed, never stored)
    __host__ __device__ static constexpr bool tile_valid(int t, int nn) {
        if (!CONTIG) return true;
        const int rel = 16 * t + nn;
        return rel % RS < W && rel / RS < H;
    }
    // 1: the two 16-byte halves of every plane of this slot are stored swapped (8-slot rows; invariant under row shifts)
    __host__ __device__ static constexpr int swz(int slot) { return SWZ ? (slot >> 2) & 1 : 0; }
    // packed weights (bytes).  A tower layer's first two planes (4 slices x 2 planes x 64 lanes x 16 B + tap 8: 2 planes x
    // [channel half][filter] x 16 B) are what the persistent kernel keeps in LDS; the third plane (one product in six reads it)
    // is a separate array that every kernel streams from L2, a layer ahead -- with all three planes in LDS only 5 network
    // waves fit a CU.  Its tap-8 entry is the whole A operand [w1|w3] of the third tap-8 product (64 lanes x 16 B).
    static constexpr int LAYER12_B = 4 * 2 * 64 * 16 + 2 * 32 * 16; // 9 216
    static constexpr int LAYER3_B = 4 * 64 * 16 + 64 * 16;          // 5 120
    // first conv, 3 planes: narrow input -- taps 0..7 (K = 32) x 3 planes + tap 8's three planes side by side in ONE K = 32
    // operand; wide input -- 9 taps x (K = 32: 32 planes) x 3 planes
    static constexpr int W0_B = WIDE_IN ? 9 * 3 * 64 * 16 : 3 * 64 * 16 + 64 * 16;
};

struct NetX3 {             // device pointers of the packed operands (nullptr: this network has no x3 form)
    const unsigned char *w0;   // X3Geom::W0_B
    const unsigned char *wt12; // [2R] x LAYER12_B   planes 1 and 2 of the tower weights
    const unsigned char *wt3;  // [2R] x LAYER3_B    plane 3 (+ the [w1|w3] tap-8 operand), always read from global memory
    const unsigned char *wt8;  // [2R] x 3 KB        tap 8's three A operands [w1|w1], [w2|w2], [w1|w3] as 64-lane images: PP form
    const unsigned char *wh;   // 3 KB               the two 1x1 head convolutions (value: 1 filter, policy: 2) as the same three operands
};

__device__ __forceinline__ unsigned bf16_bits(float v) { // round to nearest even, as v_cvt_pk_bf16_f32 does for finite values
    __bf16 h = (__bf16)v;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

// A compiler hazard met on the way to the tap-8 form above (ROCm 7.2, gfx950; tools/micro/mfma_mixk.hip is the repro):
// v_mfma_f32_16x16x32_bf16 directly followed by v_mfma_f32_16x16x16_bf16 on the same accumulator -- the second read its SrcC
// before the first had written it and a whole MFMA's contribution was lost, no wait states are inserted
// (tests/test_gpu_net.py: ..._every_tap_and_plane_contributes).  The tower therefore uses ONE MFMA kind throughout:
// nothing in this file (or gnet_x3.hip.h) may call __builtin_amdgcn_mfma_f32_16x16x16_bf16.
#pragma GCC poison __builtin_amdgcn_mfma_f32_16x16x16_bf16

// two 8-byte halves -> one K = 32 operand
__device__ __forceinline__ bf16x8 x3_cat(u32x2 lo, u32x2 hi) {
    const u32x4 both = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, both);
}

// (gnet_x3.hip.h, tap 8 of a channel block: a K = 16 operand zero-extended to the K = 32 instruction)
__device__ __forceinline__ bf16x8 x3_k16(s16x4 v) {
    const u32x2 lo = __builtin_bit_cast(u32x2, v);
    const u32x4 both = {lo[0], lo[1], 0u, 0u};
    return __builtin_bit_cast(bf16x8, both);
}

// y[0..3] -> three bf16 planes, each packed as 2 dwords (4 x bf16).  Pairs go through v_cvt_pk_bf16_f32 (two roundings per
// instruction, the pair already packed) and come back as floats by a shift / a mask: 18 vector instructions per four values
// where converting and packing element by element took 35 -- the epilogues are half of a network wave's vector instructions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) { // round to nearest even, as bf16_bits does
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void x3_split4(const f32x4 y, u32x2 &p1, u32x2 &p2, u32x2 &p3) {
    unsigned a[2], b[2], c[2];
    float r1[4], r2[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        a[h] = cvt_pk_bf16(y[2 * h], y[2 * h + 1]);
        r1[2 * h] = y[2 * h] - __uint_as_float(a[h] << 16);
        r1[2 * h + 1] = y[2 * h + 1] - __uint_as_float(a[h] & 0xffff0000u);
        b[h] = cvt_pk_bf16(r1[2 * h], r1[2 * h + 1]);
        r2[2 * h] = r1[2 * h] - __uint_as_float(b[h] << 16);
        r2[2 * h + 1] = r1[2 * h + 1] - __uint_as_float(b[h] & 0xffff0000u);
        c[h] = cvt_pk_bf16(r2[2 * h], r2[2 * h + 1]);
    }
    p1 = u32x2{a[0], a[1]};
    p2 = u32x2{b[0], b[1]};
    p3 = u32x2{c[0], c[1]};
}

// WLDS: the packed weights (x3.w0 / x3.wt, nd.epi, nd.head) are in LDS (persistent kernel) -- else global (L2-resident).
// LEAN: two network waves share each SIMD (12-wave persistent kernel, 168 VGPRs): the partner's MFMAs cover this wave's LDS
// round trips, so operands are read tile by tile right where they are used instead of a phase ahead -- a third of the
// operand registers.
// PP (with !WLDS; the caller provides X3Geom::WAVE_BYTES_PP): two activation buffers.  A lone wave per SIMD cannot hide its
// epilogue (float32 -> three planes, ~35 vector instructions and three writes per tile) behind its own MFMAs while it
// rewrites the buffer it reads; with a second buffer the layer runs tile by tile and tile t - 1's epilogue issues while
// tile t's 27 MFMAs execute (a bf16 MFMA leaves 8 of its 16 cycles to the wave's vector instructions).
// HEADS_OUT (persistent kernel): the wave stops after the three pooled head activations (R, R0, R1) and hands them to the
// caller's `pooled_out` -- the value / policy tails (head_one) then run on the tree wave that picks the result up.
// 16 bytes of packed weights from device memory through the global address space.  In a kernel that reaches its weights
// through a descriptor copied to LDS (mega_dc.hip.h) the pointers are generic, the loads flat: a flat load counts on lgkmcnt
// too and may return out of order with LDS reads, so with one in flight every wait for an LDS operand becomes a wait for
// everything -- a layer's first MFMA then waits for the NEXT layer's weights, requested a moment earlier.
__device__ __forceinline__ bf16x8 x3_ldg(const unsigned char *p) {
    return *(const __attribute__((address_space(1))) bf16x8 *)(const void *)p;
}
#ifdef BB_STAMPS_NET_SUB
#define NSUB(i) NSTAMP(i) // prologue split (diagnostic): 5 = loads issued + zero fill, 6 = state to LDS, 7 = planes; 0 = the rest
#else
#define NSUB(i) do {} while (0)
#endif
#define X3_DBG(bit) ND_DBG(bit) // ablation switches of the diagnostic build (bb_timing_net; results are wrong when set)
template <class G, bool WLDS, bool LEAN = false, bool PP = false, bool HEADS_OUT = false>
__device__ __forceinline__ void net_body_x3(const NetDev &nd, const NetX3 &x3, int n, int pos0, const int *slot_list,
                                            unsigned char *wl, const typename G::State *states, const int8_t *planes,
                                            const uint32_t *game_id, const int32_t *serial, int noise, float *value_out,
                                            float *logits_out, float *policy_out, int pstride, bool zero_lds,
                                            WideHead *compact = nullptr, const float *noise_ready = nullptr,
                                            float *pooled_out = nullptr) {
    using XG = X3Geom<G>;
    constexpr int W = XG::W, CIN = XG::CIN, HW = XG::HW, NT = XG::NT, SB = XG::SLOT_B, RS = XG::RS;
    constexpr bool WIDE_IN = XG::WIDE_IN;
    // (opaque to the optimiser: inside a persistent kernel's loop everything derived from the lane number is loop-invariant --
    // the compiler hoists the head / prologue / first-conv addresses out of the loop, holds them through the tower and spills
    // them; each reload is a scratch round trip with a full vmcnt wait, ~4 k cycles per evaluation in the heads alone)
    int lane_ = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_));
    const int lane = lane_;
    const int g = lane >> 4, nn = lane & 15, gh = g >> 1, gl = g & 1;
    static_assert(!PP || !WLDS, "the two-buffer form keeps its weights in registers, a layer ahead");
    unsigned char *X = wl;
    unsigned char *X2 = wl + XG::X_B; // (PP only)
    typename G::State *sst = (typename G::State *)(wl + (PP ? 2 : 1) * XG::X_B);
    auto OI = [&](int pos) { return slot_list ? slot_list[pos] : pos; };
    const bool live = pos0 < n;
#ifdef BB_STAMPS_NET
    long long _ns = clock64();
#endif

    // ---- prologue: the board, the first conv's operands, zero fill -------------------------------------------------
    // (PP = the one-wave-per-game kernel: its leaf positions already live in LDS (DCShadow), read them where they are)
    const typename G::State *sst_in = sst;
    typename G::State my_state;
    if constexpr (PP) sst_in = as_lds(&states[OI(live ? pos0 : 0)]);
    else my_state = planes ? G::initial() : states[OI(live ? pos0 : 0)];
    const unsigned char *w0p = x3.w0;
    bf16x8 w0a[WIDE_IN ? 27 : 3]; // wide input: [tap][plane], all requested now (L2), consumed tap by tap
    bf16x8 w0b;                   // narrow input: tap 8's three weight planes side by side in one K = 32 operand
    if constexpr (WIDE_IN) {
#pragma unroll
        for (int i = 0; i < 27; i++) w0a[i] = WLDS ? *(const bf16x8 *)(w0p + (i * 64 + lane) * 16) : x3_ldg(w0p + (i * 64 + lane) * 16);
    } else {
#pragma unroll
        for (int q = 0; q < 3; q++) w0a[q] = *(const bf16x8 *)(w0p + (q * 64 + lane) * 16);
        w0b = *(const bf16x8 *)(w0p + (3 * 64 + lane) * 16);
    }
    const float *epi = PP ? as_lds(nd.epi) : nd.epi; // (the one-wave-per-game kernel keeps the constants in its LDS)
    const f32x4 bias0 = *(const f32x4 *)(epi + 4 * g), scale0 = *(const f32x4 *)(epi + 16 + 4 * g),
                shift0 = *(const f32x4 *)(epi + 32 + 4 * g);
    NSUB(5);
    if (zero_lds) {
        u32x4 z = {0u, 0u, 0u, 0u};
        if constexpr (PP) {
            // every pixel slot is rewritten by each evaluation (input planes, then the first conv's three planes, every layer
            // after it); only the halo must read as zero: row 0, the first slot of rows 1 .. H, everything from row H + 1 on --
            // in both buffers, 16 bytes per lane and store
            constexpr int NHALO = RS + XG::H + (XG::SLOTS - (XG::H + 1) * RS), CH = SB / 16;
            constexpr int NCH = 2 * NHALO * CH;
#pragma unroll
            for (int k = 0; k < (NCH + 63) / 64; k++) { // (straight-line: a counted loop paid two branches and its index arithmetic per pass)
                const int i = lane + 64 * k;
                const int buf = i >= NHALO * CH, j = i - buf * NHALO * CH, hs = j / CH, c = j - hs * CH;
                const int slot = hs < RS ? hs : hs < RS + XG::H ? (hs - RS + 1) * RS : (XG::H + 1) * RS + (hs - RS - XG::H);
                if (64 * k + 63 < NCH || i < NCH) *(u32x4 *)(wl + buf * XG::X_B + slot * SB + c * 16) = z;
            }
        } else {
            for (int i = lane; i < XG::WAVE_BYTES / 16; i += 64) ((u32x4 *)wl)[i] = z;
        }
    }
    wave_lds_handover();
    NSUB(6);
    if constexpr (!PP && WIDE_IN) { // (a 16-byte board is decoded from registers below; an 80-byte one through LDS)
        if (!planes && lane == 0) *sst = my_state;
        wave_lds_handover();
    }
    if constexpr (WIDE_IN) {
        if (lane < HW && live) { // input planes of pixel `lane`: 32 x bf16 in the first 64 B of its X slot
            const int y = lane / W, x = lane % W;
            uint32_t pk[16]; // 32 bf16 values, two per word
            bool packed = false;
            if constexpr (G::CELL_BF16) {
                if (!planes) {
                    G::encode_cell_bf16(*sst_in, y, x, pk);
                    packed = true;
                }
            }
            if (!packed) {
                int8_t v[CIN];
                if (planes) {
                    const int8_t *src = planes + ((size_t)pos0 * HW + lane) * CIN;
#pragma unroll
                    for (int c = 0; c < CIN; c++) v[c] = src[c];
                } else {
                    G::encode_cell(*sst_in, y, x, v);
                }
                unsigned b[32];
#pragma unroll
                for (int c = 0; c < 32; c++) b[c] = c < CIN ? __float_as_uint((float)v[c]) >> 16 : 0u;
#pragma unroll
                for (int k = 0; k < 16; k++) pk[k] = b[2 * k] | (b[2 * k + 1] << 16);
            }
            unsigned char *dst = X + ((y + 1) * RS + (x + 1)) * SB;
#pragma unroll
            for (int k = 0; k < 4; k++) *(u32x4 *)(dst + 16 * k) = u32x4{pk[4 * k], pk[4 * k + 1], pk[4 * k + 2], pk[4 * k + 3]};
        }
    } else
    if (lane < HW && live) { // input planes of pixel `lane`: 4 x bf16 (the int8 plane values are exact in bf16) in the first 8 B of its slot
        const int y = lane / W, x = lane % W;
        int8_t v[4] = {0, 0, 0, 0};
        if (planes) {
            const int8_t *src = planes + ((size_t)pos0 * HW + lane) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; c++) v[c] = src[c];
        } else {
            int8_t e[CIN];
            if constexpr (PP) G::encode_cell(*sst_in, y, x, e);
            else G::encode_cell(my_state, y, x, e); // (every lane holds the board: two 64-bit words)
#pragma unroll
            for (int c = 0; c < CIN; c++) v[c] = e[c];
        }
        unsigned b[4];
#pragma unroll
        for (int c = 0; c < 4; c++) b[c] = __float_as_uint((float)v[c]) >> 16;
        *(u32x2 *)(X + ((y + 1) * RS + (x + 1)) * SB) = u32x2{b[0] | (b[1] << 16), b[2] | (b[3] << 16)};
    }
    NSUB(7);
    // ---- per-tile addressing (bytes; X3Geom: tile_slot, swz).  Offsets are biased by the window's top-left tap, so every
    // tap of a row slice is a non-negative immediate on one address register.
    // A tile's slots are the previous tile's + TS bytes (the swizzle repeats every 8 slots), so one address register per
    // operand kind serves every tile: the tile and tap displacements are immediates of the ds instructions.
    constexpr int TS = (XG::CONTIG ? 16 : 2 * RS) * SB;
    static_assert(XG::tile_slot(NT - 1, 5) * SB == XG::tile_slot(0, 5) * SB + (NT - 1) * TS && (!XG::SWZ || XG::swz(XG::tile_slot(1, 9)) == XG::swz(XG::tile_slot(0, 9))), "tile stride");
    auto tapoff = [](int tap) { return (tap / 3) * RS + (tap % 3); };
    const int t0off = tapoff(2 * g) * SB, t1off = tapoff(2 * g + 1) * SB; // first conv: lane group g holds taps 2g, 2g + 1
    const int s0 = XG::tile_slot(0, nn);
    const int aA0 = (s0 - RS - 1 + gh) * SB + ((gl ^ XG::swz(s0 - 1 + gh)) << 4);   // slices (3r, 3r + 1): + r RS SB
    const int aB0 = (s0 + (gh - 1) * RS + 1) * SB + ((gl ^ XG::swz(s0 + 1)) << 4);  // slice (2, 5)
    const int aC0 = (s0 + RS + 1) * SB + ((gl ^ XG::swz(s0 + 1)) << 4);             // tap 8: [x1;x2] at + c81, [x3;x1] at + c83
    const int aC1 = aC0 + gh * 32, aC3 = aC0 + (1 - gh) * 64;
    const int aO0 = s0 * SB + ((g ^ (XG::swz(s0) << 1)) << 3);                      // this lane's 4 output channels of its pixel, plane 0
    const int aH0 = s0 * SB + ((gl ^ XG::swz(s0)) << 4);                            // the pixel itself (head convolutions): planes as in tap 8
    const int aH1 = aH0 + gh * 32, aH3 = aH0 + (1 - gh) * 64;
    const int iA0 = (s0 - RS - 1) * SB;
    bool wv[NT]; // column nn of tile t is a pixel (else a halo slot: computed, not stored)
#pragma unroll
    for (int t = 0; t < NT; t++) wv[t] = XG::tile_valid(t, nn);
#define aA(t) (aA0 + (t) * TS)
#define aB(t) (aB0 + (t) * TS)
#define aC1(t) (aC1 + (t) * TS)
#define aC3(t) (aC3 + (t) * TS)
#define aO(t) (aO0 + (t) * TS)
#define aH1(t) (aH1 + (t) * TS)
#define aH3(t) (aH3 + (t) * TS)
#define iA(t) (iA0 + (t) * TS)
    wave_lds_handover();
    NSTAMP(0);
    f32x4 acc[NT], sk[NT];
    // ---- first conv: K = (tap, 4 planes); inputs in one bf16 plane, weights in three ---------------------------------
    {
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = bias0;
        if constexpr (WIDE_IN) { // one K = 32 slice per tap: lane group g holds input planes 8g .. 8g + 7 of the tap's pixel
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                bf16x8 b[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) b[t] = *(const bf16x8 *)(X + iA(t) + g * 16 + tapoff(tap) * SB);
#pragma unroll
                for (int q = 2; q >= 0; q--)
#pragma unroll
                    for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0a[tap * 3 + q], b[t], acc[t], 0, 0, 0);
            }
        } else {
        bf16x8 b[NT], b8[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const u32x2 lo = *(const u32x2 *)(X + iA(t) + t0off), hi = *(const u32x2 *)(X + iA(t) + t1off);
            const u32x2 e8 = *(const u32x2 *)(X + iA(t) + tapoff(8) * SB);
            b[t] = x3_cat(lo, hi);
            b8[t] = x3_cat(e8, e8); // lane group 0: [w1 | w2] . [x ; x], group 1: [w3 | 0] . [x ; x], groups 2, 3: zero weights
        }
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0b, b8[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int q = 2; q >= 0; q--) // small terms first
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0a[q], b[t], acc[t], 0, 0, 0);
        }
        wave_lds_handover(); // the inputs are read (their values feed the MFMAs above) before the output planes overwrite them
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) y[r] = fmaxf(__builtin_fmaf(acc[t][r], scale0[r], shift0[r]), 0.f);
            sk[t] = y;
            u32x2 p1, p2, p3;
            x3_split4(y, p1, p2, p3);
            if (wv[t]) {
                *(u32x2 *)(X + aO(t)) = p1;
                *(u32x2 *)(X + aO(t) + 32) = p2;
                *(u32x2 *)(X + aO(t) + 64) = p3;
            }
        }
    }
    wave_lds_handover();
    NSTAMP(1);
    // ---- residual tower ------------------------------------------------------------------------------------------------
    const int R_eff = X3_DBG(2) ? 0 : nd.R;
    const int L = 2 * R_eff;
    // plane 3 of the tower weights comes from L2: the five operands of a layer are requested a layer ahead.  So are planes 1
    // and 2 when the caller does not keep them in LDS (!WLDS: batch kernels and the DragonChess wave-per-game kernel, whose
    // LDS holds the 4032-wide head): 56 registers of weights in flight, which a 256-thread workgroup can afford.
    bf16x8 w3c[5], w12c[WLDS ? 1 : 5][2];
    const int w8lane = (gl * 16 + nn) * 16; // tap 8, planes 1 and 2: [channel half][filter] x 16 B (both lane-group halves read the same)
    auto request_layer = [&](int l) __attribute__((always_inline)) {
        const unsigned char *g3 = x3.wt3 + (size_t)l * XG::LAYER3_B;
#pragma unroll
        for (int sl = 0; sl < 5; sl++) w3c[sl] = *(const bf16x8 *)(g3 + (sl * 64 + lane) * 16); // (sl 4: the [w1|w3] operand of tap 8)
        if constexpr (!WLDS) {
            const unsigned char *g12 = x3.wt12 + (size_t)l * XG::LAYER12_B;
#pragma unroll
            for (int sl = 0; sl < 4; sl++)
#pragma unroll
                for (int q = 0; q < 2; q++) w12c[sl][q] = *(const bf16x8 *)(g12 + ((sl * 2 + q) * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < 2; q++) w12c[4][q] = *(const bf16x8 *)(g12 + 4 * 2 * 64 * 16 + q * 512 + w8lane);
        }
    };
    if (L > 0 && !PP) request_layer(0);
    auto conv_layer = [&](const int l, auto skip_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip_tag)::value, LAST = decltype(last_tag)::value;
        const float *ep = epi + (size_t)(1 + l) * 48;
        const unsigned char *wp = x3.wt12 + (size_t)l * XG::LAYER12_B;
        {
            const f32x4 bias = *(const f32x4 *)(ep + 4 * g);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = bias;
        }
        // slice s: 0,1,2 = taps (3s, 3s + 1) at row s; 3 = taps (2, 5); then tap 8 (three plane-concatenated products).
        // Operands roll through the registers plane by plane: while the 3 NT MFMAs that use the pixels' first plane run
        // (w3 x1, w2 x1, w1 x1 for every tile), the second plane arrives; during its 2 NT MFMAs the third plane, the next
        // slice's weights and its first plane arrive -- a third of the registers of a whole-slice double buffer (which
        // measured slower: spills).
        auto xoff = [&](int s, int t) __attribute__((always_inline)) { return s < 3 ? aA(t) + s * RS * SB : aB(t); };
        if constexpr (LEAN) {
            static_assert(!LEAN || WLDS, "the in-place schedule reads its weights from LDS");
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const bf16x8 wa = *(const bf16x8 *)(wp + ((s * 2 + 0) * 64 + lane) * 16), wb = *(const bf16x8 *)(wp + ((s * 2 + 1) * 64 + lane) * 16);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const bf16x8 xa = *(const bf16x8 *)(X + xoff(s, t)), xb = *(const bf16x8 *)(X + xoff(s, t) + 32),
                                 xc_ = *(const bf16x8 *)(X + xoff(s, t) + 64);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3c[s], xa, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xa, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xb, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xc_, acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const bf16x8 a1 = *(const bf16x8 *)(wp + 4 * 2 * 64 * 16 + w8lane), a2 = *(const bf16x8 *)(wp + 4 * 2 * 64 * 16 + 512 + w8lane);
            const bf16x8 a3 = w3c[4];
            bf16x8 y1[NT], y3[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) {
                y1[t] = *(const bf16x8 *)(X + aC1(t));
                y3[t] = *(const bf16x8 *)(X + aC3(t));
            }
            if (l + 1 < L) request_layer(l + 1);
#pragma unroll
            for (int t = 0; t < NT; t++) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, y3[t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, y1[t], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, y1[t], acc[t], 0, 0, 0);
            }
        } else {
        bf16x8 wc[2], wn[2], x0[NT], x1[NT], x2[NT];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if constexpr (WLDS) wc[q] = *(const bf16x8 *)(wp + ((0 * 2 + q) * 64 + lane) * 16);
            else wc[q] = w12c[0][q];
        }
#pragma unroll
        for (int t = 0; t < NT; t++) x0[t] = *(const bf16x8 *)(X + xoff(0, t));
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < NT; t++) x1[t] = *(const bf16x8 *)(X + xoff(s, t) + 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3c[s], x0[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int q = 1; q >= 0; q--)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[q], x0[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; t++) x2[t] = *(const bf16x8 *)(X + xoff(s, t) + 64);
            if (s + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if constexpr (WLDS) wn[q] = *(const bf16x8 *)(wp + (((s + 1) * 2 + q) * 64 + lane) * 16);
                    else wn[q] = w12c[WLDS ? 0 : s + 1][q];
                }
#pragma unroll
                for (int t = 0; t < NT; t++) x0[t] = *(const bf16x8 *)(X + xoff(s + 1, t));
            } else { // tap 8: [x3;x1] into the first-plane registers, [w1|w1] / [w2|w2] into the next-slice weight registers
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    if constexpr (WLDS) wn[q] = *(const bf16x8 *)(wp + 4 * 2 * 64 * 16 + q * 512 + w8lane);
                    else wn[q] = w12c[WLDS ? 0 : 4][q];
                }
#pragma unroll
                for (int t = 0; t < NT; t++) x0[t] = *(const bf16x8 *)(X + aC3(t));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 1; q >= 0; q--)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[q], x1[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[0], x2[t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 2; q++) wc[q] = wn[q];
        }
        // tap 8: [w1|w3].[x3;x1], then [w2|w2].[x1;x2] and [w1|w1].[x1;x2] on one operand read
#pragma unroll
        for (int t = 0; t < NT; t++) x1[t] = *(const bf16x8 *)(X + aC1(t));
        const bf16x8 a3 = w3c[4];
        if (l + 1 < L) request_layer(l + 1); // (this layer's slice registers were read for the last time above; a3 is a copy)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, x0[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int q = 1; q >= 0; q--)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[q], x1[t], acc[t], 0, 0, 0);
        }
        // (the batch-norm constants are read here, not at the top of the layer: 8 registers less through the slices)
        const f32x4 scale = *(const f32x4 *)(ep + 16 + 4 * g), shift = *(const f32x4 *)(ep + 32 + 4 * g);
        wave_lds_handover(); // every lane's reads of X are done (their results feed the MFMAs above) before X is rewritten
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = __builtin_fmaf(acc[t][r], scale[r], shift[r]);
                if constexpr (SKIP) v = v + sk[t][r];
                y[r] = fmaxf(v, 0.f);
            }
            if constexpr (SKIP) sk[t] = y;
            {
                (void)LAST; // (the last layer writes its planes like every other: the head convolutions are MFMAs too)
                u32x2 p1, p2, p3;
                x3_split4(y, p1, p2, p3);
                if (wv[t]) {
                    *(u32x2 *)(X + aO(t)) = p1;
                    *(u32x2 *)(X + aO(t) + 32) = p2;
                    *(u32x2 *)(X + aO(t) + 64) = p3;
                }
            }
        }
        wave_lds_handover();
    };
    // ---- the same layer, tile by tile, from buffer `src` into buffer `dst` (PP) -------------------------------------------------
    // Two named sets of layer weights (15 operands each: 5 slices x 3 planes; slice 4 = tap 8's three plane-concatenated A
    // operands): a layer computes from one set while the next layer's requests fill the other -- the block loop below
    // alternates them, so no register copies are needed.
    struct WSet { bf16x8 w1[5], w2[5], w3[5]; };
    auto request_set = [&](WSet &ws, int l) __attribute__((always_inline)) {
        const unsigned char *g12 = x3.wt12 + (size_t)l * XG::LAYER12_B, *g3 = x3.wt3 + (size_t)l * XG::LAYER3_B, *g8 = x3.wt8 + (size_t)l * 3 * 1024;
#pragma unroll
        for (int sl = 0; sl < 4; sl++) {
            ws.w1[sl] = x3_ldg(g12 + ((sl * 2 + 0) * 64 + lane) * 16);
            ws.w2[sl] = x3_ldg(g12 + ((sl * 2 + 1) * 64 + lane) * 16);
            ws.w3[sl] = x3_ldg(g3 + (sl * 64 + lane) * 16);
        }
        ws.w1[4] = x3_ldg(g8 + (0 * 64 + lane) * 16); // [w1|w1]
        ws.w2[4] = x3_ldg(g8 + (1 * 64 + lane) * 16); // [w2|w2]
        ws.w3[4] = x3_ldg(g8 + (2 * 64 + lane) * 16); // [w1|w3]
    };
    auto conv_layer_pp = [&](const int l, const WSet &ws, const unsigned char *src, unsigned char *dst, auto skip_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr bool SKIP = decltype(skip_tag)::value, LAST = decltype(last_tag)::value;
        const float *ep = epi + (size_t)(1 + l) * 48;
        const f32x4 bias = *(const f32x4 *)(ep + 4 * g), scale = *(const f32x4 *)(ep + 16 + 4 * g), shift = *(const f32x4 *)(ep + 32 + 4 * g);
        // operand q of slice s: slices 0..3 = pixel planes 1, 2, 3; slice 4 (tap 8) = [x1;x2], [x3;x1], none
        auto xaddr = [&](int s, int t, int q) __attribute__((always_inline)) {
            return s < 3 ? aA(t) + s * RS * SB + q * 32 : (s == 3 ? aB(t) + q * 32 : (q == 0 ? aC1(t) : aC3(t)));
        };
        auto nops = [](int s) { return s == 4 ? 2 : 3; };
        // operands two slices ahead: a slice is 6 MFMAs = 96 cycles, an LDS round trip of three 1 KB reads is longer
        bf16x8 xc[3], xn[3], xnn[3];
#pragma unroll
        for (int q = 0; q < 3; q++) {
            xc[q] = *(const bf16x8 *)(src + xaddr(0, 0, q));
            xn[q] = *(const bf16x8 *)(src + xaddr(1, 0, q));
        }
        auto epilogue = [&](int t) __attribute__((always_inline)) {
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = __builtin_fmaf(acc[t][r], scale[r], shift[r]);
                if constexpr (SKIP) v = v + sk[t][r];
                y[r] = fmaxf(v, 0.f);
            }
            if constexpr (SKIP) sk[t] = y;
            {
                (void)LAST;
                u32x2 p1, p2, p3;
                x3_split4(y, p1, p2, p3);
                if (wv[t]) {
                    *(u32x2 *)(dst + aO(t)) = p1;
                    *(u32x2 *)(dst + aO(t) + 32) = p2;
                    *(u32x2 *)(dst + aO(t) + 64) = p3;
                }
            }
        };
        // Tile t - 1's epilogue, cut into pieces of one or two vector instructions and placed by hand: piece m runs right after
        // MFMA m of tile t (a bf16 MFMA leaves 8 of its 16 cycles to the wave's vector instructions), each MFMA + piece pair fenced
        // so that the order stands.  (Left to the scheduler -- sched_group_barrier patterns -- the pieces ended up in two
        // clumps, the larger one of ~25 instructions BETWEEN two tiles with the matrix pipe idle: ~150 cycles per tile.)
        float ev[4];
        unsigned ea[2], eb[2], ec[2];
        float er1[4], er2[4];
        auto ep_piece = [&](int t1, int m) __attribute__((always_inline)) {
            switch (m) {
            case 0: case 1: case 2: case 3: {
                float v = __builtin_fmaf(acc[t1][m], scale[m], shift[m]);
                if constexpr (SKIP) v = v + sk[t1][m];
                ev[m] = v;
                break;
            }
            case 4: ev[0] = fmaxf(ev[0], 0.f); ev[1] = fmaxf(ev[1], 0.f); break;
            case 5: ev[2] = fmaxf(ev[2], 0.f); ev[3] = fmaxf(ev[3], 0.f); if constexpr (SKIP) sk[t1] = f32x4{ev[0], ev[1], ev[2], ev[3]}; break;
            case 6: ea[0] = cvt_pk_bf16(ev[0], ev[1]); break;
            case 7: ea[1] = cvt_pk_bf16(ev[2], ev[3]); break;
            case 8: er1[0] = ev[0] - __uint_as_float(ea[0] << 16); break;
            case 9: er1[1] = ev[1] - __uint_as_float(ea[0] & 0xffff0000u); break;
            case 10: er1[2] = ev[2] - __uint_as_float(ea[1] << 16); break;
            case 11: er1[3] = ev[3] - __uint_as_float(ea[1] & 0xffff0000u); break;
            case 12: eb[0] = cvt_pk_bf16(er1[0], er1[1]); break;
            case 13: eb[1] = cvt_pk_bf16(er1[2], er1[3]); break;
            case 14: er2[0] = er1[0] - __uint_as_float(eb[0] << 16); break;
            case 15: er2[1] = er1[1] - __uint_as_float(eb[0] & 0xffff0000u); break;
            case 16: er2[2] = er1[2] - __uint_as_float(eb[1] << 16); break;
            case 17: er2[3] = er1[3] - __uint_as_float(eb[1] & 0xffff0000u); break;
            case 18: ec[0] = cvt_pk_bf16(er2[0], er2[1]); break;
            case 19: ec[1] = cvt_pk_bf16(er2[2], er2[3]); break;
            case 20:
                if (wv[t1]) {
                    *(u32x2 *)(dst + aO(t1)) = u32x2{ea[0], ea[1]};
                    *(u32x2 *)(dst + aO(t1) + 32) = u32x2{eb[0], eb[1]};
                    *(u32x2 *)(dst + aO(t1) + 64) = u32x2{ec[0], ec[1]};
                }
                break;
            default: break;
            }
            // (pin the piece here: pure arithmetic is otherwise sunk to its first user, the stores of piece 20)
            if (m < 4) asm volatile("" : "+v"(ev[m]));
            else if (m == 4) asm volatile("" : "+v"(ev[0]), "+v"(ev[1]));
            else if (m == 5) asm volatile("" : "+v"(ev[2]), "+v"(ev[3]));
            else if (m == 6 || m == 7) asm volatile("" : "+v"(ea[m - 6]));
            else if (m >= 8 && m < 12) asm volatile("" : "+v"(er1[m - 8]));
            else if (m == 12 || m == 13) asm volatile("" : "+v"(eb[m - 12]));
            else if (m >= 14 && m < 18) asm volatile("" : "+v"(er2[m - 14]));
            else if (m == 18 || m == 19) asm volatile("" : "+v"(ec[m - 18]));
        };
        (void)LAST;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            acc[t] = bias;
            __builtin_amdgcn_sched_barrier(0);
            int m = 0;
            auto MF = [&](const bf16x8 &w, const bf16x8 &x) __attribute__((always_inline)) {
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[t], 0, 0, 0);
                if (t > 0) ep_piece(t - 1, m);
                m++;
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int s = 0; s < 5; s++) {
                const int sn = (s + 2) % 5, tn = s + 2 < 5 ? t : t + 1;
                if (tn < NT) {
#pragma unroll
                    for (int q = 0; q < 3; q++)
                        if (q < nops(sn)) xnn[q] = *(const bf16x8 *)(src + xaddr(sn, tn, q));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s < 4) {
                    MF(ws.w3[s], xc[0]);
                    MF(ws.w2[s], xc[0]);
                    MF(ws.w1[s], xc[0]);
                    MF(ws.w2[s], xc[1]);
                    MF(ws.w1[s], xc[1]);
                    MF(ws.w1[s], xc[2]);
                } else {
                    MF(ws.w3[4], xc[1]);
                    MF(ws.w2[4], xc[0]);
                    MF(ws.w1[4], xc[0]);
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    xc[q] = xn[q];
                    xn[q] = xnn[q];
                }
            }
        }
        (void)epilogue;
        epilogue(NT - 1);
        wave_lds_handover();
    };
    if constexpr (PP) {
        // (the first conv wrote its planes into X; layers alternate X -> X2 -> X; the last one is odd and ends in X, where the heads read)
        WSet wsa, wsb;
        if (L > 0) request_set(wsa, 0);
        for (int blk = 0; blk < R_eff; blk++) {
            request_set(wsb, 2 * blk + 1);
            __builtin_amdgcn_sched_barrier(0);
            conv_layer_pp(2 * blk, wsa, X, X2, std::false_type{}, std::false_type{});
            if (blk + 1 < R_eff) {
                request_set(wsa, 2 * blk + 2);
                __builtin_amdgcn_sched_barrier(0);
                conv_layer_pp(2 * blk + 1, wsb, X2, X, std::true_type{}, std::false_type{});
            } else {
                conv_layer_pp(2 * blk + 1, wsb, X2, X, std::true_type{}, std::true_type{});
            }
        }
    } else
    for (int blk = 0; blk < R_eff; blk++) {
        conv_layer(2 * blk, std::false_type{}, std::false_type{});
        if (blk + 1 < R_eff) conv_layer(2 * blk + 1, std::true_type{}, std::false_type{});
        else conv_layer(2 * blk + 1, std::true_type{}, std::true_type{});
    }
    (void)L;
    NSTAMP(2);
    // ---- heads: the value (1 filter) and policy (2 filters) 1x1 convolutions are one more K = 16 "tap" on the matrix pipe --
    // rows 0..2 of a 16-filter operand, the three plane-concatenated products of tap 8 at the pixel itself -- then BN + ReLU and
    // the spatial sums: lane (g = 0, nn) holds the three head activations of column nn of every tile.  (Round 2 formed them
    // on the vector ALUs from a float32 copy of the last layer: 48 fmas and four 16-byte reads per pixel, 2.0 k cycles.)
    const float *hp = PP ? as_lds(nd.head) : nd.head; // (the one-wave-per-game kernel keeps the head parameters in its LDS)
    // Filter rows 0, 4 and 8 of the operand are the value conv and the two policy convs (engine.hip pack_x3), so lane group g
    // (< 3) finds the activation of head g in the FIRST result register: one accumulator per lane, and the three spatial sums
    // are ONE row-wise reduction -- rows 0, 1, 2 of the wave -- instead of three wave reductions.  (Same bits as three
    // pooled_sum calls on a value that is zero outside row 0: the other rows only ever added exact zeros.)
    float R, R0, R1;
    {
        const float *v3p = hp + nd.off_v3, *p6p = hp + nd.off_p6;
        // this lane group's constants, read once and by every lane: conv bias, batch-norm scale and shift
        // (by address, not by a chain of selects: v3 = {bias, scale, shift}, p6 = {bias0, bias1, scale0, scale1, shift0, shift1};
        // the fourth lane group reads group 2's and is masked out of the sum)
        const float *cp = g == 0 ? v3p : p6p + (g == 1 ? 0 : 1);
        const int cs = g == 0 ? 1 : 2;
        const float hbias = cp[0], hscale = cp[cs], hshift = cp[2 * cs];
        const unsigned char *whp = x3.wh;
        const bf16x8 h1 = PP ? x3_ldg(whp + (0 * 64 + lane) * 16) : *(const bf16x8 *)(whp + (0 * 64 + lane) * 16),
                     h2 = PP ? x3_ldg(whp + (1 * 64 + lane) * 16) : *(const bf16x8 *)(whp + (1 * 64 + lane) * 16),
                     h3 = PP ? x3_ldg(whp + (2 * 64 + lane) * 16) : *(const bf16x8 *)(whp + (2 * 64 + lane) * 16);
        const f32x4 hb = {hbias, 0.f, 0.f, 0.f};
        bf16x8 y1[NT], y3[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            y1[t] = *(const bf16x8 *)(X + aH1(t));
            y3[t] = *(const bf16x8 *)(X + aH3(t));
        }
        float xs = 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            f32x4 a = hb;
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h3, y3[t], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h2, y1[t], a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h1, y1[t], a, 0, 0, 0);
            const float v = fmaxf(__builtin_fmaf(a[0], hscale, hshift), 0.f);
            xs += (g < 3 && wv[t]) ? v : 0.f; // (halo columns and the fourth lane group contribute exact zeros)
        }
        xs += dpp_f32(xs, 0);
        xs += dpp_f32(xs, 1);
        xs += dpp_f32(xs, 2);
        xs += dpp_f32(xs, 3);
        R = lane_f32(xs, 0);
        R0 = lane_f32(xs, 16);
        R1 = lane_f32(xs, 32);
    }
    NSTAMP(3);
    if constexpr (HEADS_OUT) {
        if (lane == 0) {
            pooled_out[0] = R;
            pooled_out[1] = R0;
            pooled_out[2] = R1;
        }
    } else {
    head_one<G>(nd, R, R0, R1, live ? OI(pos0) : 0, live, game_id, serial, noise, value_out,
                logits_out, policy_out, pstride, compact, noise_ready);
    }
    NSTAMP(4);
}

#undef aA
#undef aB
#undef aC1
#undef aC3
#undef aO
#undef aH1
#undef aH3
#undef iA
// bb_net_eval / lock-step and asynchronous-round search: one position per wave, four waves per workgroup, the packed
// weights streamed from L2
template <class G>
__global__ void __launch_bounds__(256) k_net_x3(NetDev nd, NetX3 x3, int n, const int *n_ptr, const int *slot_list, const typename G::State *states,
                                                const int8_t *planes, const uint32_t *game_id, const int32_t *serial, int noise,
                                                float *value_out, float *logits_out, float *policy_out, int pstride) {
    using XG = X3Geom<G>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * XG::WAVE_BYTES];
    const int wave = threadIdx.x >> 6;
    const int pos0 = blockIdx.x * 4 + wave;
    if (n_ptr) n = *n_ptr; // compacted batch of an asynchronous round: the leaves posted this round, slots in slot_list
    if (pos0 >= n) return;
    net_body_x3<G, false>(nd, x3, n, pos0, slot_list, lds + wave * XG::WAVE_BYTES, states, planes, game_id, serial, noise, value_out,
                          logits_out, policy_out, pstride, true);
}
