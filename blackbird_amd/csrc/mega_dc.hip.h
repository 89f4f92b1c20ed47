// mega_dc.hip.h -- DragonChess self-play with one wave per game for a whole launch.
//
// In the launch-per-simulation structure every game waits twice per simulation for the slowest of the 1024 waves
// (the tree step lasts as long as the deepest descent: 63 us for a mean of 31 us of work per game; the network
// kernel likewise), and every launch starts its code cold.  The games share nothing, so here a wave simply keeps
// its game: apply -> select -> the network for its own leaf (the same net_body, one position, weights streamed from
// L2) -> ..., then the move, for `plies` plies.  No wave ever waits for another one, so there is nothing to spin on:
// every loop is bounded by plies x sims.  Per game this is exactly the lock-step sequence of operations, so the
// results are identical (tests/test_gpu_mcts.py compares both with the oracle).
#pragma once
#include "net.hip.h"
#include "net_x3.hip.h"
#include "tree_dc.hip.h"

#define DC_HEAD_FLOATS 12288 // packed head parameters (2 x 4032 policy kernel + 4032 bias + the small ones) kept in LDS: 48 KB
#define DC_RMAX 8            // residual blocks whose bias / batch-norm constants (NetDev::epi, 48 floats per layer) fit the LDS copy
#define DC_EPI_FLOATS (48 * (1 + 2 * DC_RMAX))

// The three phases are separate functions on purpose: inlined into one loop body the compiler keeps every phase's
// address arithmetic alive across the others (256 VGPRs + 402 spilled, slower than the launches it replaces).
__device__ __attribute__((noinline)) void dc_fused_tree(const TreeDev &d_, const DCEdges &E_, int g, int lane, float *tl, const DCHeadLocal *hl) {
    const TreeDev &d = *as_lds(&d_); // everything the kernel hands over lives in its LDS (see as_lds)
    const DCEdges &E = *as_lds(&E_);
    tl = as_lds(tl);
    hl = as_lds(hl);
    dc_phase_apply<true>(d, E, g, lane, tl, hl);
    __threadfence_block();
    dc_phase_select<true>(d, E, g, lane, tl);
    __threadfence_block();
}
// The network for the wave's own leaf.  Nothing 4032-wide leaves the wave: the policy head is reduced to (R0, R1, max,
// 1 / sum exp) in `hl` (net.hip.h: WideHead) and the expansion of the next tree phase computes the probabilities of its
// legal moves from those and the head weights in LDS -- instead of a 16 KB policy row written to and gathered from
// memory, and 189 L2 loads per lane for the head weights, per evaluation.  (The prior noise of a wide game is mixed in
// at expansion, over the legal moves only: dc_expand.)
__device__ __attribute__((noinline)) void dc_fused_net(const NetDev &nd_, const TreeDev &d_, const int *slot, float *nl, DCHeadLocal *hl) {
    const NetDev &nd = *as_lds(&nd_);
    const TreeDev &d = *as_lds(&d_);
    slot = as_lds(slot);
    nl = as_lds(nl);
    hl = as_lds(hl);
    net_body<DragonChess, 1, 1>(nd, 1, 0, slot, nl, (const DCState *)d.leaf_state, nullptr, d.leaf_game_id, d.leaf_serial, 0,
                             nullptr, nullptr, nullptr, DragonChess::A, true, &hl->h);
    __threadfence_block();
}
// The evaluated leaf alone (what dc_fused_tree does first): the last thing a wave does when the launch's pool of
// simulations is dry, so that nothing of the evaluation has to outlive the launch (its policy summary `hl` is in LDS).
__device__ __attribute__((noinline)) void dc_fused_apply(const TreeDev &d_, const DCEdges &E_, int g, int lane, float *tl, const DCHeadLocal *hl) {
    const TreeDev &d = *as_lds(&d_);
    const DCEdges &E = *as_lds(&E_);
    tl = as_lds(tl);
    hl = as_lds(hl);
    dc_phase_apply<true>(d, E, g, lane, tl, hl);
    __threadfence_block();
}
// The same on the bf16 matrix pipe (net_x3.hip.h): every operand plane of the tower streams from L2 a layer ahead (this
// kernel's LDS holds the four waves' scratch and the 4032-wide head).
__device__ __forceinline__ void dc_fused_net_x3(const NetDev &nd_, const NetX3 &x3_, const TreeDev &d_, const int *slot, float *nl,
                                                          DCHeadLocal *hl) {
    const NetDev &nd = *as_lds(&nd_);
    const NetX3 &x3 = *as_lds(&x3_);
    const TreeDev &d = *as_lds(&d_);
    slot = as_lds(slot);
    nl = as_lds(nl);
    hl = as_lds(hl);
    net_body_x3<DragonChess, false, false, true>(nd, x3, 1, 0, slot, (unsigned char *)nl, (const DCState *)d.leaf_state, nullptr, d.leaf_game_id,
                                    d.leaf_serial, 0, nullptr, nullptr, nullptr, DragonChess::A, true, &hl->h);
    __threadfence_block();
}
__device__ __attribute__((noinline)) void dc_fused_move(const TreeDev &d_, const DCEdges &E_, int g, int lane, float *tl, const DCHeadLocal *hl) {
    const TreeDev &d = *as_lds(&d_);
    const DCEdges &E = *as_lds(&E_);
    tl = as_lds(tl);
    hl = as_lds(hl);
    dc_selfplay_move_body(d, E, g, lane, tl, hl);
    __threadfence_block();
}

// Per-game control state of the four games of a workgroup, kept in LDS for the launch (as the Connect4 work-queue kernel
// does, mega2.hip.h): every scalar the tree phases read first and write last, the recorded path and the leaf mailbox.
// A lone wave per SIMD pays a full memory round trip for each of these otherwise (the phases open with a dozen dependent
// loads and close with as many stores).  The tree code indexes these arrays with the wave's number in the workgroup;
// TreeDev::pool_g0 carries the workgroup's first slot for the node and edge pools, which stay in HBM.
struct DCShadow {
    static constexpr int GW = 4, MP = DragonChess::MAXPATH;
    int32_t root[GW], root_N[GW], n_nodes[GW], ply[GW], sims_left[GW], pend_leaf[GW], pend_expand[GW], path_len[GW],
        game_lid[GW], sim_serial[GW], leaf_serial[GW], used[GW];
    uint32_t leaf_game_id[GW];
    float root_W[GW], eval_value[GW];
    uint64_t evals[GW];
    uint64_t ctr[GW * 8];
    DCState leaf_state[GW];
    uint32_t path[GW * MP], path_edge[GW * MP];
    int8_t root_pp[GW];

#define BB_DC_SHADOW_ARRAYS(X)                                                                                       \
    X(d, root, 1) X(d, root_N, 1) X(d, n_nodes, 1) X(d, ply, 1) X(d, sims_left, 1) X(d, pend_leaf, 1) X(d, pend_expand, 1) \
    X(d, path_len, 1) X(d, game_lid, 1) X(d, sim_serial, 1) X(d, leaf_serial, 1) X(d, leaf_game_id, 1) X(d, root_W, 1)   \
    X(d, eval_value, 1) X(d, evals, 1) X(d, ctr, 8) X(d, path, MP) X(d, root_pp, 1) X(E, used, 1) X(E, path_edge, MP)

    __device__ __forceinline__ void load(const TreeDev &d, const DCEdges &E, int g0, int n, int nthreads) {
#define X(o, f, per) for (int i = threadIdx.x; i < n * (per); i += nthreads) f[i] = o.f[(size_t)g0 * (per) + i];
        BB_DC_SHADOW_ARRAYS(X)
#undef X
        for (int i = threadIdx.x; i < n; i += nthreads) leaf_state[i] = ((const DCState *)d.leaf_state)[g0 + i];
    }
    __device__ __forceinline__ void store(const TreeDev &d, const DCEdges &E, int g0, int n, int nthreads) {
#define X(o, f, per) for (int i = threadIdx.x; i < n * (per); i += nthreads) o.f[(size_t)g0 * (per) + i] = f[i];
        BB_DC_SHADOW_ARRAYS(X)
#undef X
        for (int i = threadIdx.x; i < n; i += nthreads) ((DCState *)d.leaf_state)[g0 + i] = leaf_state[i];
    }
    __device__ __forceinline__ void point(TreeDev &d, DCEdges &E, int g0) { // d, E: copies of the kernel arguments
#define X(o, f, per) o.f = f;
        BB_DC_SHADOW_ARRAYS(X)
#undef X
        d.leaf_state = leaf_state;
        d.pool_g0 = g0;
    }
};

#define DC_SIM_CHUNK 20 // simulations a wave draws from the launch's pool at a time (~1 ms of work)

__global__ void __launch_bounds__(256) k_dc_selfplay_fused(TreeDev d_arg, DCEdges E_arg, NetDev nd_arg, NetX3 x3_arg, int noise_on, int own_sims) {
    using NG = NetGeom<DragonChess, 1>;
    // the tree's scratch (the 4032-float policy image) and the network's activations are never live together
    constexpr int TREE_BYTES = DC_LDS_FLOATS * 4;
    constexpr int NET_BYTES = NG::WAVE_FLOATS * 4 > X3Geom<DragonChess>::WAVE_BYTES_PP ? NG::WAVE_FLOATS * 4 : X3Geom<DragonChess>::WAVE_BYTES_PP;
    constexpr int WAVE_BYTES = ((TREE_BYTES > NET_BYTES ? TREE_BYTES : NET_BYTES) + 15) / 16 * 16;
    static_assert(4 * WAVE_BYTES + DC_HEAD_FLOATS * 4 + 1024 <= 163840, "four waves' scratch and the head weights must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4][WAVE_BYTES];
    __shared__ __attribute__((aligned(16))) float s_head[DC_HEAD_FLOATS];
    // a layer's bias feeds its first MFMA: from L2 that is a full round trip at the top of every layer, and the wait for it also
    // waits for the NEXT layer's weights, requested just before (vmcnt counts in order)
    __shared__ __attribute__((aligned(16))) float s_epi[DC_EPI_FLOATS];
    __shared__ int myslot[4];
    __shared__ DCHeadLocal s_hl[4];
    // the phase functions take the three descriptor structs by reference: from LDS copies (made once) rather than from a
    // per-lane scratch copy of the kernel arguments -- a lone wave feels every round trip of its ~100 field loads per call
    __shared__ TreeDev s_d;
    __shared__ DCEdges s_E;
    __shared__ NetDev s_nd;
    __shared__ NetX3 s_x3;
    __shared__ DCShadow shadow;
    const int g0 = blockIdx.x * 4;
    const int n_mine = d_arg.n_slots - g0 < 4 ? d_arg.n_slots - g0 : 4;
    shadow.load(d_arg, E_arg, g0, n_mine, blockDim.x);
    for (int i = threadIdx.x; i < nd_arg.head_floats; i += blockDim.x) s_head[i] = nd_arg.head[i]; // (host: head_floats <= DC_HEAD_FLOATS)
    for (int i = threadIdx.x; i < 48 * (1 + 2 * nd_arg.R); i += blockDim.x) s_epi[i] = nd_arg.epi[i]; // (host: R <= DC_RMAX)
    if (threadIdx.x == 0) {
        TreeDev dl = d_arg;
        DCEdges El = E_arg;
        shadow.point(dl, El, g0);
        s_d = dl;
        s_E = El;
        s_nd = nd_arg;
        s_nd.head = s_head;
        s_nd.epi = s_epi;
        s_x3 = x3_arg;
    }
    if (threadIdx.x < 4) {
        using LP = const __attribute__((address_space(3))) float *;
        s_hl[threadIdx.x].pdk = (LP)s_head + nd_arg.off_pdk;
        s_hl[threadIdx.x].pdb = (LP)s_head + nd_arg.off_pdb;
        s_hl[threadIdx.x].h = WideHead{0.f, 0.f, 0.f, 0.f, 0.f};
    }
#ifdef BB_STAMPS_NET
    if (threadIdx.x < 8) s_net_stamps[threadIdx.x] = 0;
#endif
    __syncthreads();
    (void)noise_on; // E.noise_on carries it to the expansion
    const TreeDev &d = s_d;
    const DCEdges &E = s_E;
    const NetDev &nd = s_nd;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = wv; // the wave's game, by its number in the workgroup: index of the LDS copies (pools: + pool_g0)
    const bool mine = g0 + wv < d.n_slots;
    float *tl = (float *)lds_all[wv];
    float *nl = (float *)lds_all[wv];
    DCHeadLocal *hl = &s_hl[wv];
    if (lane == 0) myslot[wv] = g;
#ifdef BB_STAMPS
    float *lds = tl; // DST's accumulator sits past the policy image and the network's activations
    static_assert(DC_STAMP_OFF >= NG::WAVE_FLOATS && DC_STAMP_OFF * 4 >= X3Geom<DragonChess>::WAVE_BYTES_PP, "stamp words must survive the network's zeroing of its scratch");
    if (lane < 16) ((unsigned long long *)(lds + DC_STAMP_OFF))[lane] = 0;
#endif
    __threadfence_block();
    // A wave starts with its own share of the launch's simulations (`own_sims`: 7/8 of plies x simulations per move -- so that
    // every game advances even when more workgroups are launched than the chip holds at once); the last eighth is one pool
    // (d.visit_pool) that the waves then draw from in small chunks until it is dry: every wave stops within a chunk of the
    // others, wherever its game is in its move (sims_left carries over), instead of the launch waiting for the game with the
    // slowest `plies` moves.
    int chunk = own_sims;
    while (mine) {
        if (d.game_lid[g] < 0) break; // this slot has played its last game
        if (d.sims_left[g] <= 0) {    // MCTS.FindMove's tail and the self-play loop body (applies the last leaf first)
            dc_fused_move(d, E, g, lane, tl, hl);
            continue;
        }
        if (chunk == 0) {
            int got = 0;
            if (lane == 0) {
                int old = atomicSub(d.visit_pool, DC_SIM_CHUNK);
                got = old < 0 ? 0 : old < DC_SIM_CHUNK ? old : DC_SIM_CHUNK;
            }
            chunk = __builtin_amdgcn_readfirstlane(got);
            if (chunk == 0) {
                if (d.pend_leaf[g] >= 0) dc_fused_apply(d, E, g, lane, tl, hl);
                break;
            }
        }
        chunk--;
#ifdef BB_STAMPS
        long long c0 = clock64();
#endif
        dc_fused_tree(d, E, g, lane, tl, hl);
#ifdef BB_STAMPS
        long long c1 = clock64();
#endif
        if (d.pend_leaf[g] >= 0) { // (uniform) a leaf was posted: evaluate it right here
            if (s_x3.w0) dc_fused_net_x3(nd, s_x3, d, &myslot[wv], nl, hl);
            else dc_fused_net(nd, d, &myslot[wv], nl, hl);
        }
#ifdef BB_STAMPS
        DST(0, c1 - c0);          // tree phases (tools/dc_stamps.py)
        DST(2, clock64() - c1);   // network
        DST(4, 1);
#endif
    }
#ifdef BB_STAMPS
    if (mine && lane == 0 && d.stamps)
        for (int i = 0; i < 16; i++) atomicAdd(&d.stamps[(size_t)((g0 + g) & 63) * 16 + i], ((unsigned long long *)(lds + DC_STAMP_OFF))[i]);
#endif
    __syncthreads();
#ifdef BB_STAMPS_NET
    if (threadIdx.x < 8) atomicAdd(&g_net_stamps[threadIdx.x], s_net_stamps[threadIdx.x]);
#endif
    shadow.store(d_arg, E_arg, g0, n_mine, blockDim.x); // hand the per-game state back to HBM
}
