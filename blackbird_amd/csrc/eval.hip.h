// eval.hip.h -- non-network evaluators and the batched game-op kernels (dense grid games).
#pragma once
#include "games.hip.h"
#include "rng.hip.h"

// ---- batched GameState kernels: one thread per board, packed states are 16 B (one dwordx4) ----
template <class G>
__global__ void __launch_bounds__(256) k_game_legal(int n, const typename G::State *st, uint8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t m = G::legal_mask(st[i]);
    for (int a = 0; a < G::A; a++) out[(size_t)i * G::A + a] = (m >> a) & 1u;
}

template <class G>
__global__ void __launch_bounds__(256) k_game_apply(int n, typename G::State *st, const int32_t *actions,
                                                    int32_t *status) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typename G::State s = st[i];
    bool ok = G::apply(s, actions[i]);
    if (ok) st[i] = s;
    if (status) status[i] = ok ? 0 : -1;
}

template <class G>
__global__ void __launch_bounds__(256) k_game_winner(int n, const typename G::State *st, const int32_t *prev,
                                                     int8_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int p = prev ? prev[i] : -1;
    if (p >= G::A) p = -1;
    out[i] = (int8_t)G::winner(st[i], p);
}

// AsInputArray: thread per output cell so that the int8 stores of a wave are contiguous
template <class G>
__global__ void __launch_bounds__(256) k_game_encode(int n, const typename G::State *st, int8_t *out) {
    constexpr int CELLS = G::H * G::W;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * CELLS) return;
    int i = (int)(t / CELLS), c = (int)(t % CELLS);
    int8_t v[3];
    G::encode_cell(st[i], c / G::W, c % G::W, v);
    int8_t *o = out + t * 3;
    o[0] = v[0];
    o[1] = v[1];
    o[2] = v[2];
}

// ---- validation evaluator: integer hash of the AsInputArray bytes (spec: rng.hip.h) -------------
template <class G>
__device__ __forceinline__ uint64_t hash_state(const typename G::State &s, uint64_t salt) {
    HashAcc h(salt);
    for (int r = 0; r < G::H; r++)
        for (int c = 0; c < G::W; c++) {
            int8_t v[3];
            G::encode_cell(s, r, c, v);
            h.byte(v[0]);
            h.byte(v[1]);
            h.byte(v[2]);
        }
    return h.final();
}

template <class G>
__global__ void __launch_bounds__(256) k_hash_eval(int n, const typename G::State *st, const uint32_t *game_id,
                                                   uint64_t salt, int salt_per_game, uint32_t first_game_id,
                                                   float *value, float *policy, int pstride) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t sl = salt + ((salt_per_game && game_id) ? (uint64_t)(game_id[i] - first_game_id) : 0ull);
    uint64_t z = hash_state<G>(st[i], sl);
    if (value) value[i] = bb_hash_value(z);
    if (policy)
        for (int a = 0; a < G::A; a++) policy[(size_t)i * pstride + a] = bb_hash_policy(z, a);
}

// ---- MCTS.SampleValue rollouts (MCTS.py:360-383): one thread per leaf --------------------------
template <class G>
__global__ void __launch_bounds__(256) k_rollout(int n, const typename G::State *st, const uint32_t *game_id,
                                                 const int32_t *sim_serial, const int32_t *pend_leaf, uint64_t seed,
                                                 float *value) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (pend_leaf && pend_leaf[i] < 0) return;
    typename G::State s = st[i];
    int player = gs_prev(s); // value is for leaf.State.PreviousPlayer (MCTS.py:302)
    int w = G::winner(s, -1);
    uint32_t step = 0;
    uint32_t serial = (uint32_t)(sim_serial[i] - 1); // sim_serial was advanced when the leaf was posted
    while (w < 0) {
        uint32_t m = G::legal_mask(s);
        int cnt = __popc(m);
        Philox4 r = philox4x32_10(seed, game_id[i], serial, BB_TAG_ROLL, step++);
        int pick = (int)(((uint64_t)r.x[0] * (uint64_t)cnt) >> 32);
        int a = 0;
        for (int k = 0; k < G::A; k++)
            if ((m >> k) & 1u) {
                if (pick == 0) { a = k; break; }
                pick--;
            }
        G::apply(s, a);
        w = G::winner(s, a);
    }
    value[i] = w == 0 ? 0.5f : (player == w ? 1.0f : 0.0f);
}

// ---- DragonChess batched kernels ---------------------------------------------------------------------
// LegalActions (DragonChess.py:78-106): one wave per board, lane = from-square; each lane emits its 63
// contiguous mask bytes (action id = sq1*63 + sq2 - (sq2 > sq1)).
__global__ void __launch_bounds__(256) k_dc_legal(int n, const DCState *st, uint8_t *out) {
    int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= n) return;
    const int piece = st[w].b[lane];
    const uint64_t white = __ballot(piece > 0), black = __ballot(piece < 0);
    uint64_t m = DragonChess::targets_bits(piece, st[w].player, lane, white, black);
    uint8_t *o = out + (size_t)w * 4032 + lane * 63;
    for (int sq2 = 0, k = 0; sq2 < 64; sq2++) {
        if (sq2 == lane) continue;
        o[k++] = (uint8_t)((m >> sq2) & 1);
    }
}

__global__ void __launch_bounds__(256) k_dc_encode(int n, const DCState *st, int8_t *out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)n * 64) return;
    int i = (int)(t >> 6), c = (int)(t & 63);
    int8_t v[17];
    DragonChess::encode_cell(st[i], c >> 3, c & 7, v);
    int8_t *o = out + t * 17;
    for (int k = 0; k < 17; k++) o[k] = v[k];
}

// validation evaluator for DragonChess: one wave per leaf (FNV chain on lane 0, 4032 policy hashes spread over lanes)
__global__ void __launch_bounds__(256) k_dc_hash_eval(int n, const DCState *st, const uint32_t *game_id, uint64_t salt,
                                                      int salt_per_game, uint32_t first_game_id, float *value,
                                                      float *policy, int pstride) {
    int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (w >= n) return;
    uint64_t z = 0;
    if (lane == 0) {
        DCState s = st[w];
        uint64_t sl = salt + ((salt_per_game && game_id) ? (uint64_t)(game_id[w] - first_game_id) : 0ull);
        HashAcc h(sl);
        for (int c = 0; c < 64; c++) {
            int8_t v[17];
            DragonChess::encode_cell(s, c >> 3, c & 7, v);
            for (int k = 0; k < 17; k++) h.byte(v[k]);
        }
        z = h.final();
        if (value) value[w] = bb_hash_value(z);
    }
    z = __shfl(z, 0, 64);
    if (policy)
        for (int a = lane; a < 4032; a += 64) policy[(size_t)w * pstride + a] = bb_hash_policy(z, a);
}
