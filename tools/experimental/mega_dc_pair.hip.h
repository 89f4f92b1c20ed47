// mega_dc_pair.hip.h -- EXPERIMENT (round 2), not part of the product build: DragonChess self-play with TWO waves per game.
// To rebuild it: include this file from engine.hip after mega_dc.hip.h and launch k_dc_selfplay_pair<<<(n_slots + 3) / 4, 512>>>
// in bb_selfplay_step's DragonChess branch.  Byte-identical to the one-wave kernel (tests/test_gpu_mcts.py passed with it),
// but slower: 20.3 ms per ply against 18.4 -- whichever SIMD the helper sits on (HW_ID: waves w and w + 4 of a workgroup share a
// SIMD; helper shift 0 / 1 / 2 / 3 -> 20.4 / 20.9 / 20.3 / 21.2 ms).  Every SIMD still executes one game's worth of MFMAs per
// simulation cycle (half of its own game's, half of another's), the evaluations of the two games overlap most of the time, and
// the ten pair hand-overs per evaluation plus MFMA work landing next to another game's tree phase cost more than the halved
// tile count saves.
#pragma once
#include "../../blackbird_amd/csrc/mega_dc.hip.h"

// ---- two waves per game ------------------------------------------------------------------------------------------------
// One wave per game leaves the MFMA pipes two thirds idle and nothing to overlap them with: a game IS its chain of
// tree step -> evaluation -> tree step.  The only parallelism left is inside the evaluation: here every game has a second
// wave (the "helper", on another SIMD when the hardware deals the waves of a workgroup round-robin) that computes half of the
// 16-pixel tiles of every conv layer (net_body's Team), in the same LDS region.  The tree phases, the prologue and the heads
// stay with the game's first wave.  The two meet at a pair barrier in LDS after the prologue and after each of the nine conv
// layers (ten hand-overs per evaluation); the helper spends the rest of its life parked at the first of them.  Results are
// those of the one-wave kernel bit for bit (same operations per tile; tests/test_gpu_mcts.py).
//
// Every wait is bounded by wall-clock time (abort flag -> both waves leave -> bb_counters.overflow), and a primary always
// releases its helper before it leaves, so the grid drains.
struct PairSync {
    volatile int epoch[2]; // [part]: hand-overs that wave has reached
    volatile int stop;     // set by part 0 when the game's launch is over (or by either on a timeout)
};
template <int PART_>
struct PairTeam {
    static constexpr int PARTS = 2, PART = PART_;
    PairSync *ps;
    int *phase;            // this wave's count of hand-overs (register copy lives in the caller)
    long long t_start, t_limit;
    __device__ __forceinline__ void sync() const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // my LDS writes are complete before my epoch says so
        const int p = ++*phase;
        if ((threadIdx.x & 63) == 0) ps->epoch[PART] = p;
        int spins = 0;
        while (__builtin_amdgcn_readfirstlane(ps->epoch[PART ^ 1]) < p) {
            if (__builtin_amdgcn_readfirstlane(ps->stop)) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023) == 0 && wall_clock64() - t_start > t_limit) {
                ps->stop = 2; // timed out: reported through the overflow counter by the kernel's tail
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

__device__ __attribute__((noinline)) void dc_pair_net0(const NetDev &nd_, const TreeDev &d_, const int *slot, float *nl, DCHeadLocal *hl,
                                                       PairSync *ps, int *phase, long long t_start, long long t_limit) {
    const NetDev &nd = *as_lds(&nd_);
    const TreeDev &d = *as_lds(&d_);
    slot = as_lds(slot);
    nl = as_lds(nl);
    hl = as_lds(hl);
    ps = as_lds(ps);
    PairTeam<0> team{ps, phase, t_start, t_limit};
    net_body<DragonChess, 1, 1, PairTeam<0>>(nd, 1, 0, slot, nl, (const DCState *)d.leaf_state, nullptr, d.leaf_game_id, d.leaf_serial, 0,
                                             nullptr, nullptr, nullptr, DragonChess::A, true, &hl->h, team);
    __threadfence_block();
}
__device__ __attribute__((noinline)) void dc_pair_net1(const NetDev &nd_, const TreeDev &d_, const int *slot, float *nl, PairSync *ps,
                                                       int *phase, long long t_start, long long t_limit) {
    const NetDev &nd = *as_lds(&nd_);
    const TreeDev &d = *as_lds(&d_);
    slot = as_lds(slot);
    nl = as_lds(nl);
    ps = as_lds(ps);
    PairTeam<1> team{ps, phase, t_start, t_limit};
    net_body<DragonChess, 1, 1, PairTeam<1>>(nd, 1, 0, slot, nl, (const DCState *)d.leaf_state, nullptr, d.leaf_game_id, d.leaf_serial, 0,
                                             nullptr, nullptr, nullptr, DragonChess::A, true, nullptr, team);
}

#ifndef BB_DC_HELPER_SHIFT
#define BB_DC_HELPER_SHIFT 2 // helper wave 4 + ((game + SHIFT) & 3): another SIMD than the game's first wave under round-robin placement
#endif

__global__ void __launch_bounds__(512) k_dc_selfplay_pair(TreeDev d_arg, DCEdges E_arg, NetDev nd_arg, int plies, int sims, int limit_s) {
    using NG = NetGeom<DragonChess, 1>;
    constexpr int TREE_BYTES = DC_LDS_FLOATS * 4, NET_BYTES = NG::WAVE_FLOATS * 4;
    constexpr int WAVE_BYTES = ((TREE_BYTES > NET_BYTES ? TREE_BYTES : NET_BYTES) + 15) / 16 * 16;
    static_assert(4 * WAVE_BYTES + DC_HEAD_FLOATS * 4 + 8192 <= 163840, "four games' scratch and the head weights must fit the 160 KiB LDS");
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[4][WAVE_BYTES];
    __shared__ __attribute__((aligned(16))) float s_head[DC_HEAD_FLOATS];
    __shared__ int myslot[4];
    __shared__ DCHeadLocal s_hl[4];
    __shared__ PairSync s_ps[4];
    __shared__ TreeDev s_d;
    __shared__ DCEdges s_E;
    __shared__ NetDev s_nd;
    __shared__ DCShadow shadow;
    const int g0 = blockIdx.x * 4;
    const int n_mine = d_arg.n_slots - g0 < 4 ? d_arg.n_slots - g0 : 4;
    shadow.load(d_arg, E_arg, g0, n_mine, blockDim.x);
    for (int i = threadIdx.x; i < nd_arg.head_floats; i += blockDim.x) s_head[i] = nd_arg.head[i];
    if (threadIdx.x == 0) {
        TreeDev dl = d_arg;
        DCEdges El = E_arg;
        shadow.point(dl, El, g0);
        s_d = dl;
        s_E = El;
        s_nd = nd_arg;
        s_nd.head = s_head;
    }
    if (threadIdx.x < 4) {
        using LP = const __attribute__((address_space(3))) float *;
        s_hl[threadIdx.x].pdk = (LP)s_head + nd_arg.off_pdk;
        s_hl[threadIdx.x].pdb = (LP)s_head + nd_arg.off_pdb;
        s_hl[threadIdx.x].h = WideHead{0.f, 0.f, 0.f, 0.f, 0.f};
        s_ps[threadIdx.x].epoch[0] = 0;
        s_ps[threadIdx.x].epoch[1] = 0;
        s_ps[threadIdx.x].stop = 0;
        myslot[threadIdx.x] = threadIdx.x;
    }
    __syncthreads();
    const TreeDev &d = s_d;
    const DCEdges &E = s_E;
    const NetDev &nd = s_nd;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool helper = wv >= 4;
    const int g = helper ? ((wv - 4 + 4 - BB_DC_HELPER_SHIFT) & 3) : wv; // helper wave 4 + ((g + SHIFT) & 3) serves game g
    const bool mine = g0 + g < d.n_slots;
    float *tl = (float *)lds_all[g];
    float *nl = (float *)lds_all[g];
    DCHeadLocal *hl = &s_hl[g];
    PairSync *ps = &s_ps[g];
    const long long t_start = wall_clock64(), t_limit = 100000000ll * limit_s;
    int phase = 0;
#ifdef BB_STAMPS
    if (blockIdx.x == 3 && lane == 0 && d_arg.stamps) // where the hardware put the eight waves of a workgroup (tools/dc_pair_where.py)
        d_arg.stamps[wv] = 0x100 | (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xffff);
#endif
    if (mine && !helper) {
        __builtin_amdgcn_s_setprio(2); // the game's chain runs through this wave; helpers of other games share its SIMD
        for (int p = 0; p < plies; p++) {
            if (d.game_lid[g] < 0 || ps->stop) break;
            for (int s = 0; s < sims && !ps->stop; s++) {
                dc_fused_tree(d, E, g, lane, tl, hl);
                if (d.pend_leaf[g] >= 0) dc_pair_net0(nd, d, &myslot[g], nl, hl, ps, &phase, t_start, t_limit);
            }
            if (ps->stop) break;
            dc_fused_move(d, E, g, lane, tl, hl);
        }
        // release the helper: it is parked at the first hand-over of an evaluation that will not come
        if (lane == 0 && ps->stop == 0) ps->stop = 1;
    } else if (mine) {
        for (;;) {
            dc_pair_net1(nd, d, &myslot[g], nl, ps, &phase, t_start, t_limit);
            if (__builtin_amdgcn_readfirstlane(ps->stop)) break;
        }
    }
    __syncthreads();
    if (threadIdx.x < 4 && s_ps[threadIdx.x].stop == 2) d.ctr[threadIdx.x * 8 + 6] += 1; // a hand-over timed out
    __syncthreads();
    shadow.store(d_arg, E_arg, g0, n_mine, blockDim.x);
}
