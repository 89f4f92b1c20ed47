// tree_dc.hip.h -- node-pool MCTS for DragonChess (wide action space, compact child lists).
//
// A=4032 but a position has <= 137 legal moves, so a node owns a contiguous run of "edges"
// (action id, N, Q, W, child, c_puct*prior) in its slot's edge pool instead of a dense row.  One
// 64-lane wave serves one game: lane = from-square during move generation, lane = edge index (mod 64)
// during selection.  Semantics are those of tree.hip.h (same reference lines); what differs:
//   * priors: np.sum over the dense 4032-vector is numpy's pairwise tree (32 blocks of 126 with 8
//     partial sums each); reproduced over an LDS image of the dense vector so illegal zeros sit
//     where numpy sees them;
//   * prior noise is drawn for the legal moves only, at expansion: the reference's normalisation
//     over all 4032 actions (NetworkFactory.py:182) cancels in GetPriors' renormalisation
//     (Blackbird.py:386-387), so the distribution of the priors is the same.
#pragma once
#include "net.hip.h"
#include "tree.hip.h"

struct alignas(128) DCNode {
    int32_t flags;    // NODE_EXPANDED | NODE_TERMINAL | Player << 4 | (winner+1) << 8
    int32_t n_edges;  // sum(LegalActions)
    int32_t edge_off; // first edge in the slot's edge pool
    int32_t all;      // sum(ChildPlays)
    double sq;        // sqrt(1.0 + all)
    int32_t serial, pad;
    DCState st;
};
static_assert(sizeof(DCNode) == 128, "DragonChess node row is one 128-byte line");

// One edge = one 32-byte record.  The lanes of a wave read the edges of a node as one contiguous run (2 x dwordx4
// per lane), and a level of the descent touches ONE page of the game's edge pool instead of one page in each of six
// per-field pools -- on these sparsely touched multi-GB pools every extra page is an address-translation miss.
struct alignas(16) DCEdge {
    int32_t N;      // child.Plays
    float Q;        // child.Value / child.Plays (float32), 0 when unvisited
    float W;        // child.Value
    int32_t child;  // node index | CHILD_TERM_BIT, or CHILD_NONE
    double cP;      // c_puct * prior
    uint16_t act;   // action id
    // Where the child's own edges are (0, 0 until it has been expanded -- a node is expanded once, so the pair never changes
    // afterwards): a copy of child.n_edges / child.edge_off.  The descent requests the child's edges together with the
    // child's node row instead of after it -- one memory round trip per tree level instead of two.  Only a hint: the node
    // row stays the authority (dc_phase_select checks the pair against it and otherwise loads the edges the ordinary way).
    uint16_t c_edges;
    uint32_t c_off;
};
static_assert(sizeof(DCEdge) == 32, "DragonChess edge record is half a cache line");

struct DCEdges { // per-slot edge pool, stride edge_cap
    DCEdge *e;
    int32_t *used; // [n_slots] allocation cursor
    uint32_t *path_edge; // [n_slots][MAXPATH] absolute edge index | player << 30
    int edge_cap;
    int noise_on;
    float alpha, eps;
};

// The wave's LDS scratch: the dense float32 image of the masked policy (4032 entries; the values ARE float32 -- the
// float64 of the reference's `policy * LegalActions` is formed when they are summed) + the compact move list.
#ifdef BB_STAMPS
// diagnostic build: cycle stamps accumulate in the wave's LDS scratch (16 words past the policy image) and are flushed
// once when the kernel ends, spread over 64 copies -- atomics inside the timed sections would sit in front of every
// later s_waitcnt and a thousand waves on one address serialise in L2
#define DC_STAMP_OFF 5136 // past the policy image and its 112 floats of sum scratch AND past the network's activations, which share the scratch in mega_dc.hip.h
#define DC_LDS_FLOATS (DC_STAMP_OFF + 32)
#define DST(i, v) do { if (lane == 0) ((unsigned long long *)(lds + DC_STAMP_OFF))[i] += (unsigned long long)(v); } while (0)
#else
#define DC_LDS_FLOATS (4032 + 112) // the policy image + dc_np_sum's scratch (8 + 48 doubles)
#define DST(i, v) do {} while (0)
#endif

// What the one-wave-per-game kernel hands from its network phase to its tree phase instead of the evaluator mailbox in
// global memory (mega_dc.hip.h): the evaluation reduced to five numbers (net.hip.h: WideHead) and the policy-head weights
// in LDS, from which the expansion computes the probabilities of the legal moves only.
struct DCHeadLocal {
    WideHead h;
    const __attribute__((address_space(3))) float *pdk, *pdb; // policy/policy kernel [2][4032] and bias [4032]
};

// ---- wave (64 lanes) collectives ----------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes in DPP moves only (the scan LLVM's atomic optimiser builds): row_shr 1, 2, 4, 8 inside
// the rows of 16 (lanes without a source add 0), then row_bcast:15 into rows 1 / 3 and row_bcast:31 into rows 2 / 3.
__device__ __forceinline__ int wave_incl_scan_i(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_i(v), 63); }
__device__ __forceinline__ int wave_excl_scan_i(int v, int /*lane*/) { return wave_incl_scan_i(v) - v; }
// first maximum (lowest idx on ties); idx < 0 marks "no candidate".  Two stages instead of a six-step butterfly over
// (value, index, two payloads) -- 30 cross-lane reads: the wave maximum of the score (DPP inside the rows of 16, two
// ds_bpermute steps across rows), the smallest index among the lanes that hold it, then two v_readlane for the payloads.
// Wave-wide max / min without the LDS crossbar: butterflies inside the rows of 16 (quad_perm, row_half_mirror, row_mirror: every
// lane of a row ends with the row's result), then the gfx9 DPP row broadcasts -- row_bcast:15 hands lane 15 of rows 0 / 2 to
// rows 1 / 3, row_bcast:31 hands lane 31 to rows 2 and 3 -- leave the wave's result in row 3, and one v_readlane of lane 63
// makes it a scalar.  (The two cross-row steps were ds_bpermute round trips before: ~250 cycles per reduction for a lone wave,
// two reductions per tree level.)
__device__ __forceinline__ int dpp_bcast15_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false); }
__device__ __forceinline__ int dpp_bcast31_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false); }
__device__ __forceinline__ double dpp_bcast15_d(double v) {
    long long b = __double_as_longlong(v);
    int lo = dpp_bcast15_i((int)(b & 0xffffffffll)), hi = dpp_bcast15_i((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double dpp_bcast31_d(double v) {
    long long b = __double_as_longlong(v);
    int lo = dpp_bcast31_i((int)(b & 0xffffffffll)), hi = dpp_bcast31_i((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = __builtin_fmax(v, dpp_step_d<0>(v));
    v = __builtin_fmax(v, dpp_step_d<1>(v));
    v = __builtin_fmax(v, dpp_step_d<2>(v));
    v = __builtin_fmax(v, dpp_step_d<3>(v));
    v = __builtin_fmax(v, dpp_bcast15_d(v));
    v = __builtin_fmax(v, dpp_bcast31_d(v));
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_step_i<0>(v));
    v = min(v, dpp_step_i<1>(v));
    v = min(v, dpp_step_i<2>(v));
    v = min(v, dpp_step_i<3>(v));
    v = min(v, dpp_bcast15_i(v));
    v = min(v, dpp_bcast31_i(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void wave_argmax(double &u, int &idx, int &p0, int &p1) {
    const double m = wave_max_f64(idx >= 0 ? u : -2.0); // PUCT scores are >= -1
    const bool tie = idx >= 0 && u == m;
    const int kmin = wave_min_i32(tie ? idx : 0x7fffffff);
    const unsigned long long who = __ballot(tie && idx == kmin);
    if (who) { // uniform
        const int w = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
        p0 = __builtin_amdgcn_readlane(p0, w);
        p1 = __builtin_amdgcn_readlane(p1, w);
        u = m;
        idx = kmin;
    } else {
        idx = -1;
    }
}

// numpy pairwise add.reduce over the dense image a[4032] in LDS (float32 values, summed as float64).  numpy's recursion
// (n > 128: n2 = n/2 rounded down to a multiple of 8; pairwise(a, n2) + pairwise(a+n2, n-n2)) cuts 4032
// into 16 runs of 252 = leaf(120) + (leaf(64) + leaf(68)); a leaf keeps 8 strided partials, folds them
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and adds its tail sequentially.
//
// A position has a few dozen legal moves, so most of the 48 leaves hold nothing but zeros and sum to exactly 0.0.  `leaves`
// = bit L set for every leaf that holds a nonzero (uniform; the caller ORs it together from its moves).  Only those are
// summed, EIGHT LANES PER LEAF -- lane j of a group owns partial r_j, 15 (or 8) sequential adds instead of one lane's
// 120 -- eight leaves per round; the fold of the partials is the xor butterfly 1, 2, 4 inside the group, which is exactly
// the bracket above.  `scr`: 8 + 48 doubles of scratch (leaf list, leaf sums).
__device__ __forceinline__ int dc_leaf_of(int a) { // leaf index 3 * run + part of action id a
    const int run = a / 252, pos = a - run * 252;
    return 3 * run + (pos >= 184 ? 2 : (pos >= 120 ? 1 : 0));
}
__device__ __forceinline__ double dc_np_sum(const float *a, unsigned long long leaves, double *scr, int lane) {
    unsigned char *list = (unsigned char *)scr; // [48] indices of the non-empty leaves, ascending
    double *leafsum = scr + 8;                 // [48]
    if (lane < 48) {
        leafsum[lane] = 0.0;
        if ((leaves >> lane) & 1ull) list[bb_popc64(leaves & ((1ull << lane) - 1ull))] = (unsigned char)lane;
    }
    __threadfence_block();
    const int n_leaves = bb_popc64(leaves);
    const int grp = lane >> 3, j = lane & 7;
    for (int base = 0; base < n_leaves; base += 8) { // (uniform trip count)
        const bool on = base + grp < n_leaves;
        const int L = on ? (int)list[base + grp] : 0;
        const int run = L / 3, part = L - 3 * run;
        const int off = run * 252 + (part == 0 ? 0 : part == 1 ? 120 : 184);
        const int steps = part == 0 ? 15 : 8; // main part: 120 / 8 or 64 / 8 strided elements per partial
        const float *b = a + off + j;
        double r = 0.0; // (0.0 + x == x: starting from zero is starting from the first element)
        if (on) {
#pragma unroll
            for (int i = 0; i < 8; i++) r += (double)b[8 * i];
            if (steps == 15) {
#pragma unroll
                for (int i = 8; i < 15; i++) r += (double)b[8 * i];
            }
        }
        r += dpp_step_d<0>(r); // (r0+r1) ...                                  quad_perm [1,0,3,2]
        r += dpp_step_d<1>(r); // ((r0+r1)+(r2+r3)) ...                        quad_perm [2,3,0,1]
        r += dpp_step_d<2>(r); // ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))          row_half_mirror: lane i <-> 7 - i of the group
        if (on && j == 0) {
            if (part == 2)     // leaf(68): the 4 elements past the 64 strided ones, in order
                for (int i = 64; i < 68; i++) r += (double)a[off + i];
            leafsum[L] = r;
        }
    }
    __threadfence_block();
    double res = lane < 48 ? leafsum[lane] : 0.0;
    double nxt = __shfl_down(res, 1, 64);
    double r12 = res + nxt;                      // meaningful on part-1 lanes: leaf(64) + leaf(68)
    double nxt12 = __shfl_down(r12, 1, 64);
    double run_sum = res + nxt12;                // meaningful on part-0 lanes: leaf(120) + (...)
    double v = __shfl(run_sum, (3 * lane) & 63, 64); // lanes 0..15 <- runs 0..15
    if (lane >= 16) v = 0.0;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        double right = __shfl_down(v, o, 64);
        if ((lane & (2 * o - 1)) == 0) v = v + right;
    }
    return __shfl(v, 0, 64);
}

// ---- a position spread over the wave: lane k holds square k, the scalars are uniform ---------------------------------------
// A DCState held in registers is 20 VGPRs that the game rules index DYNAMICALLY (b[sq1], b[sq2], ray walks): the compiler
// answers with scratch memory, and the descent paid several scratch round trips per created node.  With one square per
// lane a move is two selects around a v_readlane, the castle flags are six readlanes and "is a king missing" is a ballot.
struct DCLane {
    int sq;                   // piece code on this lane's square (DragonChess.py:12-19)
    int player, prev, castle; // uniform; castle = bit i <-> castle[i] (wK wQ bK bQ)
};
__device__ __forceinline__ DCLane dc_lane_load(const DCState *p, int lane) {
    DCLane l;
    l.sq = p->b[lane];
    const uint32_t w0 = *(const uint32_t *)&p->player, w1 = *(const uint16_t *)&p->castle[2]; // player, prev, castle[0..1] | castle[2..3]
    l.player = (int)(int8_t)(w0 & 0xff);
    l.prev = (int)(int8_t)((w0 >> 8) & 0xff);
    l.castle = (int)(((w0 >> 16) & 1u) | (((w0 >> 24) & 1u) << 1) | ((w1 & 1u) << 2) | (((w1 >> 8) & 1u) << 3));
    return l;
}
__device__ __forceinline__ void dc_lane_store(DCState *p, const DCLane &l, int lane) {
    p->b[lane] = (int8_t)l.sq;
    if (lane == 0) { // player, prev, castle[4], 10 pad bytes (zero, as every DCState in the pools has them)
        uint32_t *w = (uint32_t *)&p->player;
        w[0] = (uint32_t)(l.player & 0xff) | ((uint32_t)(l.prev & 0xff) << 8) | ((uint32_t)(l.castle & 1) << 16) | ((uint32_t)((l.castle >> 1) & 1) << 24);
        w[1] = (uint32_t)((l.castle >> 2) & 1) | ((uint32_t)((l.castle >> 3) & 1) << 8);
        w[2] = 0;
        w[3] = 0;
    }
}
// ApplyAction (DragonChess.py:127-159 + Move :172-214) for a move KNOWN to be legal (it comes from the node's edge list),
// then Winner() (:161-167): -1 none, 1, 2.  `action` is wave-uniform.
__device__ __forceinline__ int dc_lane_apply(DCLane &l, int action, int lane) {
    int sq1, sq2;
    DragonChess::action_squares(__builtin_amdgcn_readfirstlane(action), sq1, sq2);
    const int moved = __builtin_amdgcn_readlane(l.sq, sq1);
    l.sq = lane == sq2 ? moved : (lane == sq1 ? 0 : l.sq);
    if (l.prev == 1 && l.player == 1) { // :192-197: W, W, B, W, W, B ...
        l.player = 2;
        l.prev = 1;
    } else {
        l.prev = l.player;
        l.player = 1;
    }
    const int b4 = __builtin_amdgcn_readlane(l.sq, 4), b7 = __builtin_amdgcn_readlane(l.sq, 7), b0 = __builtin_amdgcn_readlane(l.sq, 0);
    const int b60 = __builtin_amdgcn_readlane(l.sq, 60), b63 = __builtin_amdgcn_readlane(l.sq, 63), b56 = __builtin_amdgcn_readlane(l.sq, 56);
    int c = l.castle; // :199-212
    if (b4 != 1) c &= ~3;
    else if (b7 != 5) c &= ~1;
    if (b0 != 5) c &= ~2;
    if (b60 != -1) c &= ~12;
    else if (b63 != -5) c &= ~4;
    if (b56 != -5) c &= ~8;
    l.castle = c;
    const bool bk = __ballot(l.sq == -1) != 0, wk = __ballot(l.sq == 1) != 0;
    return !bk ? 1 : (!wk ? 2 : -1);
}

// AddChildren for the node `node` (state st): move generation (lane = from-square), priors, edge rows.
// policy == nullptr and hl == nullptr -> MCTS.GetPriors default (ones).  lds: DC_LDS_FLOATS floats of scratch owned by this wave.
// board_mem: the 64 board bytes of `st` in memory (one coalesced load instead of dynamic indexing into registers)
// node_idx / node_flags / used: the node's index (== its serial), its flags before the expansion and the slot's edge
// cursor, loaded by the caller along with its other first-round loads (used is advanced on success).
__device__ bool dc_expand(const TreeDev &d, const DCEdges &E, int g, DCNode *node, const int st_player,
                          const int8_t *board_mem, const float *policy, const DCHeadLocal *hl, uint32_t gid, int lane,
                          float *lds, int node_idx, int node_flags, int &used) {
    const bool priors = policy != nullptr || hl != nullptr; // evaluator priors (dense row in memory, or the compact head)
#ifdef BB_STAMPS
    long long x0 = clock64();
#endif
    // move generation: lane = from-square; occupancy by two ballots, targets by bit operations (games.hip.h)
    const int piece = board_mem[lane];
    const uint64_t white = __ballot(piece > 0), black = __ballot(piece < 0);
    uint64_t m = DragonChess::targets_bits(piece, st_player, lane, white, black);
    int cnt = bb_popc64(m);
    int pre = wave_excl_scan_i(cnt, lane);
    int total = wave_sum_i(cnt);
    int off = used;
    if (off + total > E.edge_cap) return false;
#ifdef BB_STAMPS
    long long e0 = clock64(), e1 = e0, e2 = e0;
    if (priors) DST(9, e0 - x0);
#endif
    // The moves are dealt evenly over the lanes (move j of the from-square-major enumeration -> lane j & 63): one
    // policy load, one Beta draw and one edge per lane instead of a serial loop over the busiest square's moves.
    constexpr int MPL = 4; // moves per lane: up to 256 legal moves
    if (total > 64 * MPL) return false;
    uint16_t *mlist = (uint16_t *)lds; // compact action list, parked in the scratch until every lane has its entries
    {
        uint64_t mm = m;
        int k = pre;
        while (mm) {
            int sq2 = bb_ctz64(mm);
            mm &= mm - 1;
            mlist[k++] = (uint16_t)DragonChess::action_id(lane, sq2);
        }
    }
    __threadfence_block();
    int mya[MPL];
#pragma unroll
    for (int i = 0; i < MPL; i++) mya[i] = (lane + 64 * i < total) ? (int)mlist[lane + 64 * i] : -1;
    __threadfence_block();
    double tot = 1.0;
    float myp[MPL];
    if (priors) {
        {
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            f32x4 *img = (f32x4 *)lds;
            for (int i = lane; i < 4032 / 4; i += 64) img[i] = z;
        }
        __threadfence_block();
        unsigned long long leaves = 0ull; // which leaves of numpy's pairwise sum hold one of this lane's moves
#pragma unroll
        for (int i = 0; i < MPL; i++) {
            myp[i] = 0.f;
            if (mya[i] >= 0) {
                leaves |= 1ull << dc_leaf_of(mya[i]);
                // getPolicy()[a]: from the dense row the network kernel wrote, or computed here from the compact head --
                // the same operations on the same inputs, so the same bits (net.hip.h: wide_prob)
                float p = hl ? wide_prob<64>(hl->h, hl->pdk[mya[i]], hl->pdk[4032 + mya[i]], hl->pdb[mya[i]]) : policy[mya[i]];
                if (E.noise_on)
                    p = (1.0f - E.eps) * p + E.eps * bb_beta_noise(d.seed, gid, (uint32_t)node_idx, (uint32_t)mya[i], E.alpha);
                myp[i] = p;
                lds[mya[i]] = p; // float32 * float64 legal mask (1.0): widened when summed
            }
        }
#ifdef BB_STAMPS
        e1 = clock64();
#endif
        __threadfence_block();
        {   // OR the lanes' leaf bits together (DPP inside the rows, row broadcasts across them, lane 63 has the wave's)
            unsigned lo = (unsigned)leaves, hi = (unsigned)(leaves >> 32);
#define BB_OR_STEP(f) lo |= (unsigned)f((int)lo); hi |= (unsigned)f((int)hi);
            BB_OR_STEP(dpp_step_i<0>) BB_OR_STEP(dpp_step_i<1>) BB_OR_STEP(dpp_step_i<2>) BB_OR_STEP(dpp_step_i<3>)
            BB_OR_STEP(dpp_bcast15_i) BB_OR_STEP(dpp_bcast31_i)
#undef BB_OR_STEP
            leaves = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
        }
        tot = dc_np_sum(lds, leaves, (double *)(lds + 4032), lane);
#ifdef BB_STAMPS
        e2 = clock64();
#endif
    }
    size_t base = (size_t)(g + d.pool_g0) * E.edge_cap + off;
#pragma unroll
    for (int i = 0; i < MPL; i++)
        if (mya[i] >= 0) {
            size_t e = base + lane + 64 * i;
            E.e[e].act = (uint16_t)mya[i];
            E.e[e].N = 0;
            E.e[e].Q = 0.f;
            E.e[e].W = 0.f;
            E.e[e].child = CHILD_NONE;
            E.e[e].cP = priors ? d.c_puct * __ddiv_rn((double)myp[i], tot) : d.c_puct * 1.0;
            E.e[e].c_edges = 0;
            E.e[e].c_off = 0u;
        }
    if (lane == 0) {
        node->flags = node_flags | NODE_EXPANDED;
        node->n_edges = total;
        node->edge_off = off;
        node->all = 0;
        node->sq = 1.0;
        E.used[g] = off + total;
    }
    used = off + total;
#ifdef BB_STAMPS
    if (priors) {
        DST(5, e1 - e0);
        DST(6, e2 - e1);
        DST(7, clock64() - e2);
        DST(8, 1);
    }
#endif
    return true;
}

// SH = the per-game control state (TreeDev's [n_slots] arrays, the paths, DCEdges::used) lives in the caller's LDS
// (mega_dc.hip.h: DCShadow) -- tell the compiler, or every access is a flat instruction (games.hip.h: as_lds)
template <bool SH, class T>
__device__ __forceinline__ T *dc_ctl(T *p) {
    if constexpr (SH) return as_lds(p);
    else return p;
}
template <bool SH = false>
__device__ void dc_phase_apply(const TreeDev &d, const DCEdges &E, int g, int lane, float *lds, const DCHeadLocal *hl = nullptr) {
    // One wave per game and one wave per SIMD: this step is a chain of dependent HBM round trips (2-3 us each on
    // these sparsely touched pools), so every word that does not depend on another load is requested up front, and
    // the statistics the backup will update are fetched BEFORE the expansion, whose work then hides their latency.
#ifdef BB_STAMPS
    const long long a_in = clock64();
#endif
    const int leaf = dc_ctl<SH>(d.pend_leaf)[g];
    const int pend_exp = dc_ctl<SH>(d.pend_expand)[g];
    const float v = hl ? hl->h.value : dc_ctl<SH>(d.eval_value)[g];
    const int plen = dc_ctl<SH>(d.path_len)[g];
    const int lid = dc_ctl<SH>(d.game_lid)[g];
    const uint32_t *pn = dc_ctl<SH>(d.path) + (size_t)g * DragonChess::MAXPATH;
    const uint32_t *pe = dc_ctl<SH>(E.path_edge) + (size_t)g * DragonChess::MAXPATH;
    const uint32_t pn0 = pn[lane], pe0 = pe[lane]; // lane < 64 <= MAXPATH: in bounds whatever the path length
    int root_n = 0, pp = 0;
    float root_w = 0.f;
    if (lane == 0) {
        root_n = dc_ctl<SH>(d.root_N)[g];
        pp = dc_ctl<SH>(d.root_pp)[g];
        root_w = dc_ctl<SH>(d.root_W)[g];
    }
    int used = dc_ctl<SH>(E.used)[g];
    if (leaf < 0) return;
#ifdef BB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long a_r1 = clock64();
#endif
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    DCNode *node = pool + leaf;
    const DCState *leaf_st = dc_ctl<SH>((const DCState *)d.leaf_state) + g;
    const int st_player = leaf_st->player, st_prev = leaf_st->prev;
    // the path's statistics (edges and nodes above the leaf: the expansion below touches none of them)
    const bool on_path = lane < plen;
    DCNode *my_nd = pool + (on_path ? pn0 : 0u);
    const size_t my_e = (size_t)(g + d.pool_g0) * E.edge_cap + (on_path ? (pe0 & 0x3FFFFFFFu) : 0u);
    int my_n = 0, my_all = 0;
    float my_w = 0.f;
    if (on_path) {
        my_n = E.e[my_e].N;
        my_w = E.e[my_e].W;
        my_all = my_nd->all;
    }
    const int leaf_flags = node->flags;
#ifdef BB_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long a_r2 = clock64();
        DST(3, a_r2 - a_r1);  // second round of loads
        DST(10, a_r1 - a_in); // entry -> first round complete (tools/dc_stamps.py)
    }
#endif
    if (pend_exp) {
        uint32_t gid = d.first_game_id + (uint32_t)lid;
        const int used0 = used;
        if (!dc_expand(d, E, g, node, st_player, leaf_st->b, hl ? nullptr : d.eval_policy + (size_t)g * 4032, hl, gid,
                       lane, lds, leaf, leaf_flags, used)) {
            if (lane == 0) dc_ctl<SH>(d.ctr)[(size_t)g * 8 + 6] += 1;
        } else if (plen > 0 && plen <= 64 && lane == plen - 1 && used - used0 <= 0xFFFF) {
            E.e[my_e].c_edges = (uint16_t)(used - used0); // the edge that leads to the leaf (DCEdge::c_edges)
            E.e[my_e].c_off = (uint32_t)used0;
        }
    }
    int player = st_player, prev = st_prev;
    float v01;
    if (d.evaluator == 2) {
        v01 = v;
    } else {
        v01 = (v + 1.0f) * 0.5f;
        if (player != prev) v01 = 1.0f - v01;
    }
    float vflip = 1.0f - v01;
#ifdef BB_STAMPS
    long long b0 = clock64();
#endif
    if (on_path) {
        int pl = (int)(pe0 >> 30);
        int n = my_n + 1, all = my_all + 1;
        float w = my_w + ((pl == prev) ? v01 : vflip);
        E.e[my_e].N = n;
        E.e[my_e].W = w;
        E.e[my_e].Q = __fdiv_rn(w, (float)n);
        my_nd->all = all;
        my_nd->sq = __dsqrt_rn(1.0 + (double)all);
    }
    for (int k = 64 + lane; k < plen; k += 64) { // paths beyond 64 edges (not seen at 400 simulations)
        DCNode *nd = pool + pn[k];
        uint32_t ew = pe[k];
        int pl = (int)(ew >> 30);
        size_t e = (size_t)(g + d.pool_g0) * E.edge_cap + (ew & 0x3FFFFFFFu);
        int n = E.e[e].N + 1, all = nd->all + 1;
        float w = E.e[e].W + ((pl == prev) ? v01 : vflip);
        E.e[e].N = n;
        E.e[e].W = w;
        E.e[e].Q = __fdiv_rn(w, (float)n);
        nd->all = all;
        nd->sq = __dsqrt_rn(1.0 + (double)all);
    }
    if (lane == 0) {
        dc_ctl<SH>(d.root_N)[g] = root_n + 1;
        if (pp) dc_ctl<SH>(d.root_W)[g] = root_w + ((pp == prev) ? v01 : vflip);
        dc_ctl<SH>(d.pend_leaf)[g] = -1;
    }
#ifdef BB_STAMPS
    (void)b0;
#endif
}

// create the child reached by edge e (absolute) of parent; lane 0 writes
__device__ __forceinline__ int dc_create_child(const TreeDev &d, const DCEdges &E, int g, DCNode *pool, const DCState &pst,
                                               int action, size_t e, int lane, int &nn, DCState &st2, bool &terminal) {
    int idx = nn;
    st2 = pst;
    DragonChess::apply(st2, action);
    int w = DragonChess::winner(st2, action);
    terminal = w >= 0;
    if (idx >= d.node_cap) return CHILD_NONE;
    int word = idx | (terminal ? CHILD_TERM_BIT : 0);
    nn = idx + 1;
    if (lane == 0) {
        DCNode *c = pool + idx;
        c->st = st2;
        c->flags = (terminal ? (NODE_TERMINAL | ((w + 1) << 8)) : 0) | ((int)st2.player << 4);
        c->n_edges = 0;
        c->edge_off = 0;
        c->all = 0;
        c->serial = idx;
        E.e[e].child = word;
        // (n_nodes[g] and the node counter are written once by the caller's tail: nn carries the new count)
    }
    return word;
}

// A node row's head and an edge record through the global address space, as two 16-byte loads each
struct DCNodeHead {
    int flags, n_edges, edge_off;
    double sq;
};
__device__ __forceinline__ DCNodeHead dc_head_load(const DCNode *p) {
    typedef const __attribute__((address_space(1))) u32x4 *GP;
    const GP q = (GP)(const void *)p;
    const u32x4 a = q[0], b = q[1];
    DCNodeHead h;
    h.flags = (int)a[0];
    h.n_edges = (int)a[1];
    h.edge_off = (int)a[2];
    h.sq = __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | b[0]));
    return h;
}
__device__ __forceinline__ DCEdge dc_edge_load(const DCEdge *p) {
    typedef const __attribute__((address_space(1))) u32x4 *GP;
    const GP q = (GP)(const void *)p;
    const u32x4 a = q[0], b = q[1];
    DCEdge r;
    r.N = (int)a[0];
    r.Q = __uint_as_float(a[1]);
    r.W = __uint_as_float(a[2]);
    r.child = (int)a[3];
    r.cP = __longlong_as_double((long long)(((unsigned long long)b[1] << 32) | b[0]));
    r.act = (uint16_t)(b[2] & 0xffffu);
    r.c_edges = (uint16_t)(b[2] >> 16);
    r.c_off = b[3];
    return r;
}
static_assert(offsetof(DCNode, sq) == 16 && offsetof(DCEdge, cP) == 16 && offsetof(DCEdge, act) == 24 &&
              offsetof(DCEdge, c_edges) == 26 && offsetof(DCEdge, c_off) == 28, "dc_head_load / dc_edge_load read these layouts");

template <bool SH = false>
__device__ void dc_phase_select(const TreeDev &d, const DCEdges &E, int g, int lane, float *lds) {
    const int lid = dc_ctl<SH>(d.game_lid)[g], sims_left = dc_ctl<SH>(d.sims_left)[g];
    int cur = dc_ctl<SH>(d.root)[g];
    int nn = dc_ctl<SH>(d.n_nodes)[g];
    const int nn0 = nn;
    int used = dc_ctl<SH>(E.used)[g];
    // the counters the tail updates, requested with the first round of loads instead of after the descent
    int t_serial = 0;
    uint64_t t_evals = 0, t_c0 = 0, t_c1 = 0, t_c2 = 0, t_c3 = 0, t_c6 = 0;
    uint64_t *ctr = dc_ctl<SH>(d.ctr) + (size_t)g * 8;
    if (lane == 0) {
        t_serial = dc_ctl<SH>(d.sim_serial)[g];
        t_evals = dc_ctl<SH>(d.evals)[g];
        t_c0 = ctr[0];
        t_c1 = ctr[1];
        t_c2 = ctr[2];
        t_c3 = ctr[3];
        t_c6 = ctr[6];
    }
    if (lid < 0 || sims_left <= 0) return;
#ifdef BB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long q0 = clock64(), q_pre = 0, q_hdr = 0, q_edge = 0, q_cmp = 0, q_rest = 0, q_lv = 0;
#define QS(acc) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); long long _q = clock64(); acc += _q - q0; q0 = _q; } while (0)
#else
#define QS(acc) do {} while (0)
#endif
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    uint32_t *pn = dc_ctl<SH>(d.path) + (size_t)g * DragonChess::MAXPATH;
    uint32_t *pe = dc_ctl<SH>(E.path_edge) + (size_t)g * DragonChess::MAXPATH;
    int depth = 0, expand = 0, overflow = 0, term_leaf = 0;
    const bool inline_expand = d.priors_ones != 0;
    const bool fixed = d.kind == 1, rollout = d.evaluator == 2;
    DCLane st = {0, 0, 0, 0}; // the leaf's position, one square per lane
    int flags = 0;
    bool have = false;
    uint32_t my_pn = 0, my_pe = 0;
    int hint_n = 0;         // DCEdge::c_edges / c_off of the edge just taken: where this node's edges are, if it has any yet
    uint32_t hint_off = 0u;
    const size_t ebase = (size_t)(g + d.pool_g0) * E.edge_cap;
    for (int it = 0;; it++) {
        DCNode *node = pool + cur;
        // This lane's edge of the first pass, requested TOGETHER with the node row (the hint says where the edges are): both
        // loads are unconditional -- lanes past the hint read its last edge, no hint reads the pool's first edge -- because a
        // load inside a divergent branch is waited for at the end of that branch, which would put the two round trips back in
        // sequence.  (Global-address-space loads: a flat load also ties up lgkmcnt.)
        const int hl = min(lane, max(hint_n, 1) - 1);
        DCEdge pre = dc_edge_load(E.e + ebase + hint_off + hl);
        const DCNodeHead hd = dc_head_load(node);
        if (!have) flags = hd.flags;
        int n_edges = hd.n_edges, edge_off = hd.edge_off;
        double sq = hd.sq;
        const bool pre_ok = hint_n > 0 && hint_n == n_edges && hint_off == (uint32_t)edge_off;
#ifdef BB_STAMPS
        asm volatile("" ::"v"(n_edges), "v"(edge_off), "v"(sq), "v"(flags));
        if (it == 0) QS(q_pre); else QS(q_rest);
        q_lv++;
        QS(q_hdr);
#endif
        bool have_st = have;
        have = false;
        if (fixed && it >= d.max_depth) {
            if (!have_st) st = dc_lane_load(&node->st, lane);
            break;
        }
        if (!(flags & NODE_EXPANDED)) {
            if (!have_st) st = dc_lane_load(&node->st, lane);
            if (flags & NODE_TERMINAL) { term_leaf = 1; break; }
            if (!inline_expand) { expand = 1; break; }
            if (!dc_expand(d, E, g, node, st.player, node->st.b, nullptr, nullptr, 0u, lane, lds, cur, flags, used)) { overflow = 1; break; }
            __threadfence_block();
            if (!fixed) break;
            n_edges = node->n_edges;
            edge_off = node->edge_off;
            sq = 1.0;
        }
        if (n_edges == 0) {
            if (!have_st) st = dc_lane_load(&node->st, lane);
            break;
        }
        // PUCT over the node's edges, 64 per pass
        double bu = -1.0;
        int bi = -1, bchild = CHILD_NONE, bact = 0;
        size_t base = ebase + edge_off;
        if (!pre_ok) pre = dc_edge_load(E.e + base + min(lane, max(n_edges, 1) - 1)); // (root, a node whose incoming edge carries no hint)
#ifdef BB_STAMPS
        asm volatile("" ::"v"(pre.N), "v"(pre.cP));
        QS(q_edge);
#endif
        int bhint = 0; // the winner's c_edges, c_off
        uint32_t bhoff = 0u;
        bool picked = false;
#if BB_PUCT_FILTER
        if (!rollout) { // float32 scores first (tree.hip.h grp_argmax_puct: bound and argument); a lane keeps its best and its runner-up
            float b1 = -1.0f, b2 = -1.0f;
            for (int k = lane; k < n_edges; k += 64) {
                const DCEdge r = k < 64 ? pre : E.e[base + k];
                const float uf = __builtin_fmaf((float)(r.cP * sq), __builtin_amdgcn_rcpf((float)(r.N + 1)), r.Q);
                if (uf > b1) {
                    b2 = b1;
                    b1 = uf;
                    bi = k;
                    bchild = r.child;
                    bact = r.act;
                    bhint = r.c_edges;
                    bhoff = r.c_off;
                } else {
                    b2 = __builtin_fmaxf(b2, uf);
                }
            }
            const float mf = wave_max_f32(b1), thr = mf - mf * 0x1p-18f;
            const unsigned long long cand = __ballot(bi >= 0 && b1 >= thr), amb = __ballot(bi >= 0 && b2 >= thr);
            if (amb == 0ull && cand != 0ull && (cand & (cand - 1ull)) == 0ull) { // one edge of the node can hold the float64 maximum
                const int w = __builtin_amdgcn_readfirstlane(__ffsll((long long)cand) - 1);
                bi = __builtin_amdgcn_readlane(bi, w);
                bchild = __builtin_amdgcn_readlane(bchild, w);
                bact = __builtin_amdgcn_readlane(bact, w);
                bhint = __builtin_amdgcn_readlane(bhint, w);
                bhoff = (uint32_t)__builtin_amdgcn_readlane((int)bhoff, w);
                picked = true;
            } else {
                bi = -1;
            }
        }
#endif
        if (!picked) {
            for (int k = lane; k < n_edges; k += 64) {
                const DCEdge r = k < 64 ? pre : E.e[base + k];
                int Ni = r.N;
                double q = child_q(d, r.Q, rollout ? r.W : 0.f, Ni);
                double u = puct_score(q, r.cP, sq, Ni, true);
                if (bi < 0 || u > bu) {
                    bu = u;
                    bi = k;
                    bchild = r.child;
                    bact = r.act;
                }
            }
            wave_argmax(bu, bi, bchild, bact);
            bhint = 0; // (rare path: the next level loads its edges after its node row)
        }
        hint_n = bhint;
        hint_off = bhoff;
        QS(q_cmp);
        if (depth >= DragonChess::MAXPATH) { overflow = 1; if (!have_st) st = dc_lane_load(&node->st, lane); break; }
        int child = bchild;
        { // the path stays in registers (lane k <-> depth k) until the descent is over: a store per level would put
          // a write acknowledgement in front of every following load (vmcnt counts both, in order)
            const uint32_t pnv = (uint32_t)cur, pev = (uint32_t)(edge_off + bi) | ((uint32_t)((flags >> 4) & 3) << 30);
            if (depth < 64) {
                if (lane == depth) {
                    my_pn = pnv;
                    my_pe = pev;
                }
            } else if (lane == 0) {
                pn[depth] = pnv;
                pe[depth] = pev;
            }
        }
        if (child == CHILD_NONE) { // materialise the child: _applyAction on the parent's position, Winner(lastAction)
            hint_n = 0;
            st = dc_lane_load(&node->st, lane);
            if (nn >= d.node_cap) { overflow = 1; break; } // pool exhausted: the parent's position stands in as the leaf
            const int w = dc_lane_apply(st, bact, lane);
            const bool terminal = w >= 0;
            const int idx = nn++;
            child = idx | (terminal ? CHILD_TERM_BIT : 0);
            {
                DCNode *c = pool + idx;
                dc_lane_store(&c->st, st, lane);
                if (lane == 0) {
                    c->flags = (terminal ? (NODE_TERMINAL | ((w + 1) << 8)) : 0) | (st.player << 4);
                    c->n_edges = 0;
                    c->edge_off = 0;
                    c->all = 0;
                    c->serial = idx;
                    E.e[base + bi].child = child;
                    // (n_nodes[g] and the node counter are written once by the tail: nn carries the new count)
                }
            }
            flags = (terminal ? NODE_TERMINAL : 0) | (st.player << 4);
            have = true;
            __threadfence_block();
            if (!inline_expand && !fixed) { // the node just created is this descent's leaf: no need to go round once more to read it back
                depth++;
                cur = child & ~CHILD_TERM_BIT;
                if (terminal) term_leaf = 1;
                else expand = 1;
                break;
            }
        }
        depth++;
        cur = child & ~CHILD_TERM_BIT;
    }
    if (lane < depth) { // (depth <= 64 here; deeper entries were stored as they came)
        pn[lane] = my_pn;
        pe[lane] = my_pe;
    }
    dc_lane_store(dc_ctl<SH>((DCState *)d.leaf_state) + g, st, lane);
    if (lane == 0) {
        dc_ctl<SH>(d.leaf_game_id)[g] = d.first_game_id + (uint32_t)lid;
        dc_ctl<SH>(d.leaf_serial)[g] = cur;
        dc_ctl<SH>(d.pend_leaf)[g] = cur;
        dc_ctl<SH>(d.pend_expand)[g] = expand;
        dc_ctl<SH>(d.path_len)[g] = depth;
        dc_ctl<SH>(d.sims_left)[g] = sims_left - 1;
        dc_ctl<SH>(d.sim_serial)[g] = t_serial + 1;
        dc_ctl<SH>(d.evals)[g] = t_evals + 1;
        ctr[0] = t_c0 + 1;
        ctr[1] = t_c1 + (uint64_t)depth;
        if (nn != nn0) {
            dc_ctl<SH>(d.n_nodes)[g] = nn;
            ctr[2] = t_c2 + (uint64_t)(nn - nn0);
        }
        ctr[3] = t_c3 + (uint64_t)term_leaf;
        ctr[6] = t_c6 + (uint64_t)overflow;
    }
#ifdef BB_STAMPS
    QS(q_rest);
    DST(11, q_pre);
    DST(12, q_hdr);
    DST(13, q_edge);
    DST(14, q_cmp);
    DST(15, q_rest);
    DST(1, q_lv);
#endif
#undef QS
}

__global__ void __launch_bounds__(256) k_dc_tree_step(TreeDev d, DCEdges E) {
    __shared__ __attribute__((aligned(16))) float lds_all[4][DC_LDS_FLOATS];
    int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (g >= d.n_slots) return;
    float *lds = lds_all[wv];
#ifdef BB_STAMPS
    if (lane < 16) ((unsigned long long *)(lds + DC_STAMP_OFF))[lane] = 0;
    __threadfence_block();
    long long t0 = clock64();
#endif
    dc_phase_apply(d, E, g, lane, lds);
#ifdef BB_STAMPS
    long long t1 = clock64();
#endif
    __threadfence_block();
    dc_phase_select(d, E, g, lane, lds);
#ifdef BB_STAMPS
    long long t2 = clock64();
    if (lane == 0 && d.stamps) {
        DST(0, t1 - t0);
        DST(2, t2 - t1);
        DST(4, 1);
        for (int i = 0; i < 16; i++) atomicAdd(&d.stamps[(size_t)(g & 63) * 16 + i], ((unsigned long long *)(lds + DC_STAMP_OFF))[i]);
    }
#endif
}

__global__ void __launch_bounds__(256) k_dc_tree_apply(TreeDev d, DCEdges E) {
    __shared__ __attribute__((aligned(16))) float lds[4][DC_LDS_FLOATS];
    int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (g >= d.n_slots) return;
    dc_phase_apply(d, E, g, lane, lds[wv]);
}

// ---- root statistics / move choice ------------------------------------------------------------------------
// lane 0 walks the (<=144) edges: the np.random.choice law needs the sequential cumsum
__device__ int dc_choose_move(const TreeDev &d, const DCEdges &E, int g, int lane, double temp, double u, int &total,
                              int &edge_out) {
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    DCNode *node = pool + d.root[g];
    int act = -3, tot = 0, eo = -1;
    if (lane == 0 && (node->flags & NODE_EXPANDED)) {
        int n = node->n_edges;
        size_t base = (size_t)(g + d.pool_g0) * E.edge_cap + node->edge_off;
        for (int k = 0; k < n; k++) tot += E.e[base + k].N;
        if (temp == 0.0) {
            double bu = -1.0;
            for (int k = 0; k < n; k++) {
                int Ni = E.e[base + k].N;
                double uu = puct_score(child_q(d, E.e[base + k].Q, E.e[base + k].W, Ni), E.e[base + k].cP, node->sq, Ni, true);
                if (eo < 0 || uu > bu) {
                    bu = uu;
                    eo = k;
                }
            }
            act = n > 0 ? (int)E.e[base + eo].act : -4;
        } else {
            double it = 1.0 / temp, allp = 0.0;
            for (int k = 0; k < n; k++) {
                double c = (double)E.e[base + k].N;
                allp += (it == 1.0) ? c : pow(c, it);
            }
            if (!(allp > 0.0)) {
                act = -4;
            } else {
                double last = 0.0;
                for (int k = 0; k < n; k++) {
                    double c = (double)E.e[base + k].N;
                    last += __ddiv_rn((it == 1.0) ? c : pow(c, it), allp);
                }
                double run = 0.0;
                eo = n - 1;
                for (int k = 0; k < n; k++) {
                    double c = (double)E.e[base + k].N;
                    run += __ddiv_rn((it == 1.0) ? c : pow(c, it), allp);
                    if (__ddiv_rn(run, last) > u) {
                        eo = k;
                        break;
                    }
                }
                act = (int)E.e[base + eo].act;
            }
        }
    }
    total = __shfl(tot, 0, 64);
    edge_out = __shfl(eo, 0, 64);
    return __shfl(act, 0, 64);
}

__global__ void __launch_bounds__(256) k_dc_sample(TreeDev d, DCEdges E, double temp, int32_t *out_child_action) {
    int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (g >= d.n_slots) return;
    constexpr int S = DragonChess::S;
    for (int k = lane; k < S; k += 64) {
        if (d.out_child_plays) d.out_child_plays[(size_t)g * S + k] = 0;
        if (d.out_child_value) d.out_child_value[(size_t)g * S + k] = 0.f;
        if (out_child_action) out_child_action[(size_t)g * S + k] = -1;
    }
    if (d.game_lid[g] < 0) {
        if (lane == 0 && d.out_action) d.out_action[g] = -3;
        return;
    }
    double u = d.in_u ? d.in_u[g] : bb_u53(d.seed, d.first_game_id + (uint32_t)d.game_lid[g], (uint32_t)d.ply[g]);
    int total, eo;
    int act = dc_choose_move(d, E, g, lane, temp, u, total, eo);
    DCNode *node = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap + d.root[g];
    if (node->flags & NODE_EXPANDED) {
        size_t base = (size_t)(g + d.pool_g0) * E.edge_cap + node->edge_off;
        for (int k = lane; k < node->n_edges; k += 64) {
            if (d.out_child_plays) d.out_child_plays[(size_t)g * S + k] = E.e[base + k].N;
            if (d.out_child_value) d.out_child_value[(size_t)g * S + k] = E.e[base + k].W;
            if (out_child_action) out_child_action[(size_t)g * S + k] = E.e[base + k].act;
        }
    }
    if (lane == 0) {
        if (d.out_action) d.out_action[g] = act;
        if (d.out_root_plays) d.out_root_plays[g] = d.root_N[g];
        if (d.out_root_winrate) {
            int n = d.root_N[g];
            float w = d.root_W[g];
            d.out_root_winrate[g] = n > 0 ? ((d.evaluator == 2) ? (float)((double)w / (double)n) : __fdiv_rn(w, (float)n)) : 0.f;
        }
    }
}

__device__ __forceinline__ void dc_reset_slot(const TreeDev &d, const DCEdges &E, int g, int lid, const DCState &st) {
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    pool->st = st;
    pool->flags = (int)st.player << 4;
    pool->n_edges = 0;
    pool->edge_off = 0;
    pool->all = 0;
    pool->serial = 0;
    d.n_nodes[g] = 1;
    E.used[g] = 0;
    d.root[g] = 0;
    d.root_N[g] = 0;
    d.root_W[g] = 0.f;
    d.root_pp[g] = 0;
    d.ply[g] = 0;
    d.pend_leaf[g] = -1;
    d.pend_expand[g] = 0;
    d.path_len[g] = 0;
    d.sim_serial[g] = 0;
    d.game_lid[g] = lid;
}

// _moveRoot by action id (all lanes call; lane 0 writes)
__device__ void dc_advance_root(const TreeDev &d, const DCEdges &E, int g, int lane, int action, DCState &new_st) {
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    DCNode *node = pool + d.root[g];
    DCState st = node->st;
    if (!(node->flags & NODE_EXPANDED)) {
        new_st = st;
        DragonChess::apply(new_st, action);
        if (lane == 0) dc_reset_slot(d, E, g, d.game_lid[g], new_st);
        return;
    }
    int n = node->n_edges;
    size_t base = (size_t)(g + d.pool_g0) * E.edge_cap + node->edge_off;
    int k = -1;
    for (int i = lane; i < n; i += 64)
        if ((int)E.e[base + i].act == action) k = i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) k = max(k, __shfl_xor(k, o, 64));
    if (k < 0) { // not a legal move of the root: leave the tree alone (the next FindMove asserts)
        new_st = st;
        return;
    }
    size_t e = base + k;
    int child = E.e[e].child, cn = E.e[e].N;
    float cw = E.e[e].W;
    if (child == CHILD_NONE) {
        bool terminal;
        int nn = d.n_nodes[g];
        child = dc_create_child(d, E, g, pool, st, action, e, lane, nn, new_st, terminal);
        if (child == CHILD_NONE) {
            if (lane == 0) {
                int lid = d.game_lid[g], ply = d.ply[g];
                dc_reset_slot(d, E, g, lid, new_st);
                d.ply[g] = ply;
                d.ctr[(size_t)g * 8 + 6] += 1;
            }
            return;
        }
        if (lane == 0) {
            d.n_nodes[g] = nn;
            d.ctr[(size_t)g * 8 + 2] += 1;
        }
    } else {
        new_st = pool[child & ~CHILD_TERM_BIT].st;
    }
    if (lane == 0) {
        d.root[g] = child & ~CHILD_TERM_BIT;
        d.root_N[g] = cn;
        d.root_W[g] = cw;
        d.root_pp[g] = st.player;
    }
}

__global__ void __launch_bounds__(256) k_dc_move_roots(TreeDev d, DCEdges E, const int32_t *actions) {
    int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (g >= d.n_slots) return;
    int a = actions[g];
    if (a < 0 || a >= DragonChess::A || d.game_lid[g] < 0) return;
    DCState ns;
    dc_advance_root(d, E, g, lane, a, ns);
    if (lane == 0) d.ply[g] += 1;
}

__global__ void __launch_bounds__(256) k_dc_set_roots(TreeDev d, DCEdges E, int n, const int32_t *slots, const DCState *states,
                                                      const uint32_t *game_ids) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int g = slots ? slots[i] : i;
    if (g < 0 || g >= d.n_slots) return;
    dc_reset_slot(d, E, g, game_ids ? (int)game_ids[i] : g, states[i]);
    d.sims_left[g] = 0;
}

__global__ void __launch_bounds__(256) k_dc_get_roots(TreeDev d, DCState *out) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    out[g] = ((DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap + d.root[g])->st;
}

__global__ void __launch_bounds__(256) k_dc_selfplay_begin(TreeDev d, DCEdges E) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    if (g < d.n_games_target) {
        dc_reset_slot(d, E, g, g, DragonChess::initial());
        d.sims_left[g] = d.sims_per_move;
    } else {
        dc_reset_slot(d, E, g, -1, DragonChess::initial());
        d.sims_left[g] = 0;
    }
}

// example record: ExampleHdr, DCState, u32 visits[S], u16 action[S]
__device__ void dc_write_example(const TreeDev &d, const DCEdges &E, int lid, int ply, const DCNode *node, const DCState &st,
                                 int g, int lane, int total, uint32_t gid, bool terminal_example) {
    constexpr int S = DragonChess::S;
    uint8_t *p = d.examples + ((size_t)lid * (d.max_plies + 1) + ply) * d.example_bytes;
    uint32_t *vis = (uint32_t *)(p + sizeof(ExampleHdr) + sizeof(DCState));
    uint16_t *act = (uint16_t *)(p + sizeof(ExampleHdr) + sizeof(DCState) + 4 * S);
    int n = terminal_example ? 0 : node->n_edges;
    size_t base = terminal_example ? 0 : (size_t)(g + d.pool_g0) * E.edge_cap + node->edge_off;
    for (int k = lane; k < S; k += 64) {
        vis[k] = k < n ? (uint32_t)E.e[base + k].N : 0u;
        act[k] = k < n ? E.e[base + k].act : (uint16_t)0xFFFF;
    }
    if (lane == 0) {
        ExampleHdr h;
        h.game_id = gid;
        h.ply = (uint16_t)ply;
        h.player = (uint8_t)st.player;
        h.z = 0;
        h.total = (uint32_t)total;
        h.n_children = (uint32_t)n;
        *(ExampleHdr *)p = h;
        *(DCState *)(p + sizeof(ExampleHdr)) = st;
    }
}

// GenerateTrainingSamples' loop body for one game (one wave): last result applied, move sampled, example written,
// root advanced, game finished / slot handed to the next game (Blackbird.py:240-268)
__device__ void dc_selfplay_move_body(const TreeDev &d, const DCEdges &E, int g, int lane, float *lds, const DCHeadLocal *hl = nullptr) {
    dc_phase_apply(d, E, g, lane, lds, hl);
    __threadfence_block();
    int lid = d.game_lid[g];
    if (lid < 0) return;
    DCNode *pool = (DCNode *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    DCNode *root = pool + d.root[g];
    DCState st = root->st;
    uint32_t gid = d.first_game_id + (uint32_t)lid;
    int ply = d.ply[g];
    double u = bb_u53(d.seed, gid, (uint32_t)ply);
    int total, eo;
    int act = dc_choose_move(d, E, g, lane, d.temp, u, total, eo);
    if (act < 0) {
        if (lane == 0) {
            d.game_lid[g] = -1;
            d.ctr[(size_t)g * 8 + 6] += 1;
        }
        return;
    }
    dc_write_example(d, E, lid, ply, root, st, g, lane, total, gid, false);
    DCState ns;
    dc_advance_root(d, E, g, lane, act, ns);
    __threadfence_block();
    ply += 1;
    int w = DragonChess::winner(ns, -1);
    bool over = w >= 0 || ply >= d.max_plies;
    if (!over) {
        if (lane == 0) {
            d.ply[g] = ply;
            d.sims_left[g] = d.sims_per_move;
            d.ctr[(size_t)g * 8 + 5] += 1;
        }
        return;
    }
    dc_write_example(d, E, lid, ply, nullptr, ns, g, lane, 0, gid, true);
    __threadfence_block();
    for (int k = lane; k <= ply; k += 64) {
        ExampleHdr *h = (ExampleHdr *)(d.examples + ((size_t)lid * (d.max_plies + 1) + k) * d.example_bytes);
        h->z = (w <= 0) ? 0 : (h->player == w ? 1 : -1);
    }
    __threadfence(); // (records before the `done` word, device-wide: see selfplay_move_body in tree.hip.h)
    if (lane == 0) {
        int32_t *gh = d.game_hdr + (size_t)lid * 4;
        gh[0] = ply + 1;
        gh[1] = w;
        gh[2] = ply;
        __threadfence();
        gh[3] = 1;
        uint64_t *c = d.ctr + (size_t)g * 8;
        c[4] += 1;
        c[5] += 1;
        c[7] += (uint64_t)(ply + 1);
        int next = lid + d.lid_stride;
        if (next < d.n_games_target) {
            dc_reset_slot(d, E, g, next, DragonChess::initial());
            d.sims_left[g] = d.sims_per_move;
        } else {
            d.game_lid[g] = -1;
            d.sims_left[g] = 0;
        }
    }
}

__global__ void __launch_bounds__(256) k_dc_selfplay_move(TreeDev d, DCEdges E) {
    __shared__ __attribute__((aligned(16))) float lds[4][DC_LDS_FLOATS];
    int g = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (g >= d.n_slots) return;
    dc_selfplay_move_body(d, E, g, lane, lds[wv]);
}

// MCTS.SampleValue rollouts for DragonChess: one wave per leaf (lane = from-square), capped at 2048 plies
__global__ void __launch_bounds__(256) k_dc_rollout(int n, const DCState *st, const uint32_t *game_id, const int32_t *sim_serial,
                                                    const int32_t *pend_leaf, uint64_t seed, float *value) {
    int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (i >= n) return;
    if (pend_leaf && pend_leaf[i] < 0) return;
    DCState s = st[i];
    int player = s.prev;
    int w = DragonChess::winner(s, -1);
    uint32_t serial = (uint32_t)(sim_serial[i] - 1);
    for (uint32_t step = 0; w < 0 && step < 2048; step++) {
        uint64_t m = DragonChess::targets(s, lane);
        int cnt = bb_popc64(m), pre = wave_excl_scan_i(cnt, lane), total = wave_sum_i(cnt);
        if (total == 0) break;
        Philox4 r = philox4x32_10(seed, game_id[i], serial, BB_TAG_ROLL, step);
        int pick = (int)(((uint64_t)r.x[0] * (uint64_t)total) >> 32);
        int a = -1;
        if (pick >= pre && pick < pre + cnt) {
            uint64_t mm = m;
            for (int j = pick - pre; j > 0; j--) mm &= mm - 1;
            a = DragonChess::action_id(lane, bb_ctz64(mm));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = max(a, __shfl_xor(a, o, 64));
        DragonChess::apply(s, a);
        w = DragonChess::winner(s, a);
    }
    if (lane == 0) value[i] = w <= 0 ? 0.5f : (player == w ? 1.0f : 0.0f);
}
