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

template <class G>
__global__ void __launch_bounds__(512) k_selfplay_mega(TreeDev d, NetDev nd, int phases, int noise_on) {
    constexpr int S = G::S, GW = 16, SET = 8, PW = 2;
    using NG = NetGeom<G, PW>;
    __shared__ __attribute__((aligned(16))) float lds[4 * NG::WAVE_FLOATS];
    __shared__ int post_list[4][SET];
    __shared__ int post_n[4];
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int g0 = blockIdx.x * GW;
    if (threadIdx.x < 4) post_n[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < 4 * NG::WAVE_FLOATS; i += 512) lds[i] = 0.f; // halo zeros, once per launch
    __syncthreads();
    const typename G::State *ls = (const typename G::State *)d.leaf_state;
    for (int p = 0; p <= phases; p++) {
        if (wave >= 4) {
            if (p < phases) { // ---- tree waves: 2 games per wave, S lanes per game
                int tw = wave - 4, set = p & 1;
                int li = tw * 2 + l64 / S, lane = l64 % S;
                int g = g0 + set * SET + li;
                bool live = l64 < 2 * S && g < d.n_slots;
                bool posted = live ? async_game<G>(d, g, lane) : false;
                if (posted && lane == 0) {
                    int idx = atomicAdd(&post_n[p & 3], 1);
                    post_list[p & 3][idx] = g;
                }
                if (threadIdx.x == 256) post_n[(p + 2) & 3] = 0; // free during this phase
            }
        } else if (p > 0) { // ---- network waves: the leaves posted in phase p-1
            int buf = (p - 1) & 3;
            int n = post_n[buf];
            int pos0 = wave * PW;
            if (pos0 < n)
                net_body<G, PW>(nd, n, pos0, post_list[buf], lds + wave * NG::WAVE_FLOATS, ls, nullptr, d.leaf_game_id,
                                d.leaf_serial, noise_on, d.eval_value, nullptr, d.eval_policy, S, false);
        }
        __syncthreads();
    }
}
