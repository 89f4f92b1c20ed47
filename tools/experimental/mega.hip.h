// mega.hip.h -- persistent self-play kernel: tree search and network evaluation of the SAME 16 games
// share one workgroup (one CU) and hand leaves / evaluations to each other through that CU alone.
//
// Why: with one launch per phase (k_tree_async, k_net_compact) every round waits for the deepest
// descent among all 4096 games, and the MFMA units idle while the latency-bound tree kernel runs.
// Here a workgroup = 4 network waves + 4 tree waves and owns 16 games split into two sets of 8:
//
//     phase p :   tree waves  -> apply / move / descend for set (p & 1), post its leaves
//                 net  waves  -> evaluate the leaves set ((p+1) & 1) posted in phase p-1   (PW = 2 per wave)
//                 __syncthreads()
//
// so tree latency hides under MFMA work of the other set, a deep game delays only its own workgroup,
// there are no grid-wide barriers and no launch gaps.  Synchronisation is workgroup-scope only
// (s_barrier; global mailboxes are visible through the CU's own L1), every wave runs the same number of
// barriers (phase count is a kernel argument), so the kernel cannot dead-lock on placement.
#pragma once
#include "net.hip.h"
#include "tree.hip.h"

#ifndef BB_MEGA_NO_PRIO
#define BB_MEGA_NO_PRIO 0
#endif
#ifndef MEGA_PW
#define MEGA_PW 2 // positions per network wave (2 -> 4 network waves; 1 -> 8 network waves, measured equal: profiles/README.md)
#endif
#define MEGA_THREADS ((8 / MEGA_PW + 4) * 64)
#ifndef MEGA_APPLY_IN_NET
#define MEGA_APPLY_IN_NET 0 // 1: network waves expand + back up the leaves they evaluated (measured equal)
#endif
#define MEGA_RMAX 4       // residual blocks whose weights fit the 160 KiB LDS next to the activations
#define MEGA_HEAD_FLOATS 256

template <class G>
__global__ void __launch_bounds__(MEGA_THREADS) k_selfplay_mega(TreeDev d, NetDev nd, int phases, int noise_on) {
    constexpr int S = G::S, GW = 16, SET = 8, PW = MEGA_PW, NETW = SET / PW; // NETW network waves, then 4 tree waves
    using NG = NetGeom<G, PW>;
    constexpr int RMAX = MEGA_RMAX, STEPS0 = NG::STEPS0;
    constexpr int WT_F = 2 * RMAX * 9 * 64 * 4, W0_F = STEPS0 * 64, EPI_F = (1 + 2 * RMAX) * 48, HEAD_F = MEGA_HEAD_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[NETW * NG::WAVE_FLOATS];
    __shared__ __attribute__((aligned(16))) float wlds[WT_F + W0_F + EPI_F + HEAD_F]; // the whole network, once per launch
    __shared__ int post_list[4][SET];
    __shared__ int post_n[4];
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int g0 = blockIdx.x * GW;
    if (threadIdx.x < 4) post_n[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < NETW * NG::WAVE_FLOATS; i += MEGA_THREADS) lds[i] = 0.f; // halo zeros, once per launch
    // the whole network (83 KB at R4/F16) lives in LDS for the launch; the host only picks this kernel when
    // R <= MEGA_RMAX and the head parameters fit (bb_selfplay_step), so the pointers below are LDS-only
    NetDev ndl = nd;
    {
        const float *gwt = (const float *)nd.wt;
        for (int i = threadIdx.x; i < 2 * nd.R * 9 * 64 * 4; i += MEGA_THREADS) wlds[i] = gwt[i];
        for (int i = threadIdx.x; i < W0_F; i += MEGA_THREADS) wlds[WT_F + i] = nd.w0[i];
        for (int i = threadIdx.x; i < (1 + 2 * nd.R) * 48; i += MEGA_THREADS) wlds[WT_F + W0_F + i] = nd.epi[i];
        for (int i = threadIdx.x; i < nd.head_floats; i += MEGA_THREADS) wlds[WT_F + W0_F + EPI_F + i] = nd.head[i];
        ndl.wt = (const f32x4 *)wlds;
        ndl.w0 = wlds + WT_F;
        ndl.epi = wlds + WT_F + W0_F;
        ndl.head = wlds + WT_F + W0_F + EPI_F;
    }
    __syncthreads();
    const typename G::State *ls = (const typename G::State *)d.leaf_state;
#ifdef BB_STAMPS
    long long t_work = 0, t_all0 = clock64();
#endif
    // Two role-specific loops with the same number of workgroup barriers (phases + 1 each): keeping them
    // apart keeps the tree code's registers out of the MFMA loop's allocation.
    if (wave >= NETW) { // ---- tree waves: 2 games per wave, S lanes per game
        const int tw = wave - NETW;
        // the tree waves are the latency-critical, issue-light partner of an MFMA wave on the same SIMD: let them win issue arbitration
        if (d.level_budget > 0 && !BB_MEGA_NO_PRIO) __builtin_amdgcn_s_setprio(3);
        const int li = tw * 2 + l64 / S, lane = l64 % S;
        for (int p = 0; p <= phases; p++) {
#ifdef BB_STAMPS
            long long ts = clock64();
#endif
            if (p < phases) {
                int g = g0 + (p & 1) * SET + li;
                bool live = l64 < 2 * S && g < d.n_slots;
                bool posted = live ? async_game<G>(d, g, lane) : false;
                if (posted && lane == 0) {
                    int idx = atomicAdd(&post_n[p & 3], 1);
                    post_list[p & 3][idx] = g;
                }
                if (threadIdx.x == NETW * 64) post_n[(p + 2) & 3] = 0; // free during this phase
            }
#ifdef BB_STAMPS
            t_work += clock64() - ts;
#endif
            __syncthreads();
        }
    } else { // ---- network waves: the leaves posted in the previous phase, PW = 2 per wave
        for (int p = 0; p <= phases; p++) {
#ifdef BB_STAMPS
            long long ts = clock64();
#endif
            if (p > 0) {
                int buf = (p - 1) & 3;
                int n = post_n[buf];
                int pos0 = wave * PW;
                if (pos0 < n) {
                    net_body<G, PW>(ndl, n, pos0, post_list[buf], lds + wave * NG::WAVE_FLOATS, ls, nullptr, d.leaf_game_id,
                                    d.leaf_serial, noise_on, d.eval_value, nullptr, d.eval_policy, S, false, d.eval_noise);
                    // The network wave has slack, the tree waves are the critical path: expand + back up the
                    // evaluated leaves right here (S lanes per position), so a tree wave starts its phase descending.
                    __threadfence_block();
                    int pp = l64 / S;
                    if (MEGA_APPLY_IN_NET && l64 < PW * S && pos0 + pp < n) {
                        int g = post_list[buf][pos0 + pp];
                        phase_apply<G>(d, g, l64 % S);
                        if (l64 % S == 0) d.sims_left[g] -= 1;
                    }
                }
            }
#ifdef BB_STAMPS
            t_work += clock64() - ts;
#endif
            __syncthreads();
        }
    }
#ifdef BB_STAMPS
    if (l64 == 0 && d.stamps) { // [0] net work, [1] net total, [2] tree work, [3] tree total, [4]/[5] wave counts
        long long tot = clock64() - t_all0;
        atomicAdd(&d.stamps[wave < NETW ? 0 : 2], (unsigned long long)t_work);
        atomicAdd(&d.stamps[wave < NETW ? 1 : 3], (unsigned long long)tot);
        atomicAdd(&d.stamps[wave < NETW ? 4 : 5], 1ull);
    }
#endif
}
