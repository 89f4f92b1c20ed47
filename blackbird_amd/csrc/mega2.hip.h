// mega2.hip.h -- persistent self-play kernel: tree search and network evaluation of the SAME 16 games share one
// workgroup (one CU) and hand leaves / evaluations to each other through that CU's LDS alone.
//
// With one launch per phase (k_tree_async, k_net_compact) every round waits for the deepest descent among all 4096
// games and the MFMA units idle while the latency-bound tree kernel runs.  Here the 16 games of a workgroup circulate
// freely between
//     4 tree waves   (4 games each, S lanes per game): apply result -> [move] -> descend -> post leaf
//     8 network waves (2 per SIMD, one position each): pop leaf -> tower + heads -> publish result
// through an LDS ring of game indices and a per-game state word.  A network wave's head/softmax VALU work
// overlaps its SIMD partner's MFMAs (the partners drift apart by themselves), and nobody waits for the
// slowest game of a set.
//
// Synchronisation is workgroup-scope only (LDS atomics + __threadfence_block, all waves on one CU share L1),
// so it is placement-independent.  Every wait loop is bounded by wall-clock time: a protocol bug ends the
// launch with the abort word set (reported through the overflow counter) instead of hanging the GPU.
#pragma once
#include "net.hip.h"
#include "net_x3.hip.h"
#include "tree.hip.h"

#define MEGA_RMAX 4          // residual blocks whose weights fit the 160 KiB LDS next to the activations
#define MEGA_HEAD_FLOATS 192 // packed head parameters that fit next to them (16 filters, D <= 24, <= 16 actions: 136 at the defaults)

// Waves of the workgroup = NETW network waves + the tree waves.  float32-MFMA network: 12 waves (8 + 4) at 168 VGPRs;
// bf16-pipe network (X3): 8 waves at 256 VGPRs -- 5 + 3 for Connect4, 4 + 4 for TicTacToe (4 games of 16 lanes per tree wave).
#define MEGA2_QCAP 32
#ifndef BB_TREE_IDLE_SLEEP
#define BB_TREE_IDLE_SLEEP 4 // s_sleep argument (x 64 cycles) of a tree wave that finds none of its games ready
#endif
#ifndef BB_NET_IDLE_SLEEP
#define BB_NET_IDLE_SLEEP 4  // ... of a network wave that finds the queue empty
#endif
#ifndef BB_NET_APPLIES
#define BB_NET_APPLIES -1 // tuning override of NET_APPLIES (k_selfplay_queue): 1 network waves apply their result, 0 tree waves do
#endif
#ifndef BB_TREE_HEADS
#define BB_TREE_HEADS 0 // bf16-pipe network: 1 = the value / policy tails of an evaluation run on the tree wave that picks it up (net.hip.h head_tree)
#endif
#ifndef BB_TREE_NOISE
#define BB_TREE_NOISE 0 // 0 = the prior noise of a leaf is drawn by the network wave that evaluates it (head_one), 1 = by the tree wave that posted it, right behind the queue entry (also forced by BB_TREE_HEADS).  Round 2 chose 1: the tree waves had 20 % slack then; since the tree calls carry the posts (early_post) and the heads are short, the tree waves are the 93 % busy side: 0 measures +1.5 % (113.3 vs 111.6 M evaluations/s, eight alternations)
#endif
// Tree-wave schedule of the bf16-pipe kernel.  0 (default): a tree wave makes one async_game call at a time for those of its games
// that are ready (apply -> descend to a leaf -> post).  1: level-stepped -- every iteration advances every descending game of the
// wave by one tree level and a game whose evaluation has arrived joins at once.  Measured (MI355X, Connect4 @800, 4096 games):
// 1 takes the result pick-up wait from 8.2 to 1.1 us and the tree waves from 87 to 62 % busy, but a game then comes back to the
// network waves sooner than they can take it (90 % busy, queue wait 1.4 -> 2.8 us): 112.0 against 113.9 M evaluations/s.
// Both give the same bits (tests/test_gpu_noise_parity.py passes with either build).
#ifndef BB_TREE_STEP
#define BB_TREE_STEP 0
#endif
#ifndef BB_X3_LEAN
#define BB_X3_LEAN 0 // net_x3.hip.h operand schedule of the persistent kernel: 0 = a phase ahead (38 spilled registers at 168, still faster), 1 = in place
#endif
#ifndef BB_QUEUE_WMODE
#define BB_QUEUE_WMODE 2 // net.hip.h conv_layer: weights in LDS, next tap's operands requested ahead of this tap's MFMAs
#endif

__device__ __forceinline__ int lds_load(volatile int *p) { return *p; }
// Release for a flag that lives in LDS while the data lives in global memory: the workgroup-scope fence
// the compiler emits waits for LDS traffic only, so a flag store could pass this wave's global stores.
__device__ __forceinline__ void release_global_then_lds() {
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


// ---- per-game control state shadowed in LDS for the lifetime of the launch ----------------------------------
// async_game touches a dozen per-slot scalars, the evaluator mailbox and the recorded path of a game on every
// call, each a dependent L2 round trip.  The persistent kernel owns its 16 games for the whole launch, so it
// copies that state into LDS once, points a private TreeDev at the copies and writes everything back before the
// launch ends.  The tree code indexes these arrays with the workgroup-LOCAL game number (TreeDev::pool_g0 carries the
// workgroup's first slot for the node pools, which stay in HBM), so the pointers are the plain addresses of __shared__
// arrays: the compiler sees the LDS address space and emits ds_read / ds_write with immediate offsets from one base
// register, instead of flat_* instructions through twenty 64-bit generic addresses held in VGPR pairs.
template <class G, int GW, bool REC>
struct GameShadow {
    static constexpr int S = G::S, MP = G::MAXPATH;
    int32_t root[GW], root_N[GW], n_nodes[GW], ply[GW], sims_left[GW], pend_leaf[GW], pend_expand[GW], path_len[GW],
        game_lid[GW], sim_serial[GW], leaf_serial[GW], resume_cur[GW], resume_depth[GW], leaf_flags[GW];
    int32_t path_N[REC ? GW * MP : 1], path_all[REC ? GW * MP : 1]; // (REC: the level-stepped tree waves' recorded path statistics)
    float path_W[REC ? GW * MP : 1];
    uint32_t leaf_game_id[GW];
    float root_W[GW], eval_value[GW];
    float eval_policy[GW * S];
    uint64_t evals[GW];
    uint64_t ctr[GW * 8];
    typename G::State leaf_state[GW];
    uint32_t path[GW * MP];
    int8_t root_pp[GW];

#define BB_SHADOW_ARRAYS(X)                                                                                           \
    X(root, 1) X(root_N, 1) X(n_nodes, 1) X(ply, 1) X(sims_left, 1) X(pend_leaf, 1) X(pend_expand, 1) X(path_len, 1)  \
    X(game_lid, 1) X(sim_serial, 1) X(leaf_serial, 1) X(resume_cur, 1) X(resume_depth, 1) X(leaf_game_id, 1)          \
    X(root_W, 1) X(eval_value, 1) X(eval_policy, S) X(evals, 1) X(ctr, 8) X(path, MP) X(root_pp, 1) X(leaf_flags, 1)
#define BB_SHADOW_REC_ARRAYS(X) X(path_N, MP) X(path_all, MP) X(path_W, MP)

    // all threads of the workgroup; n = games of this workgroup that exist (g0 + i < n_slots)
    __device__ __forceinline__ void load(const TreeDev &d, int g0, int n, int nthreads) {
#define X(f, per) for (int i = threadIdx.x; i < n * (per); i += nthreads) f[i] = d.f[(size_t)g0 * (per) + i];
        BB_SHADOW_ARRAYS(X)
        if constexpr (REC) { BB_SHADOW_REC_ARRAYS(X) }
#undef X
        for (int i = threadIdx.x; i < n; i += nthreads) leaf_state[i] = ((const typename G::State *)d.leaf_state)[g0 + i];
    }
    __device__ __forceinline__ void store(const TreeDev &d, int g0, int n, int nthreads) {
#define X(f, per) for (int i = threadIdx.x; i < n * (per); i += nthreads) d.f[(size_t)g0 * (per) + i] = f[i];
        BB_SHADOW_ARRAYS(X)
        if constexpr (REC) { BB_SHADOW_REC_ARRAYS(X) }
#undef X
        for (int i = threadIdx.x; i < n; i += nthreads) ((typename G::State *)d.leaf_state)[g0 + i] = leaf_state[i];
    }
    __device__ __forceinline__ TreeDev local(const TreeDev &d, int g0) {
        TreeDev r = d;
#define X(f, per) r.f = f;
        BB_SHADOW_ARRAYS(X)
        if constexpr (REC) { BB_SHADOW_REC_ARRAYS(X) }
#undef X
        r.leaf_state = leaf_state;
        r.pool_g0 = g0;
        return r;
    }
};

struct QueueCtl {
    int q[MEGA2_QCAP];
    int head, tail, tree_done, abort_flag;
};

// Pop one queued game for a network wave: >= 0 game index, -1 nothing queued right now, -2 finished (or aborted).
// Deliberately NOT inlined: with the lane-0 pop inlined into the wave's loop the compiler threads the other 63
// lanes past it, and the wave then runs the network with lane 0 split off from the rest.
__device__ __attribute__((noinline)) int queue_pop(QueueCtl *c, long long t_start, long long t_limit, int n_tree) {
    int li = -1;
    if ((threadIdx.x & 63) == 0) {
        int h = lds_load(&c->head), t = lds_load(&c->tail);
        if (h < t) {
            if (atomicCAS(&c->head, h, h + 1) == h) { // compare-and-swap so that an empty queue is never over-popped
                volatile int *slot = &c->q[h & (MEGA2_QCAP - 1)];
                for (int spin = 0; spin < (1 << 20) && (li = *slot) < 0; spin++) __builtin_amdgcn_s_sleep(1);
                *slot = -1; // at most 16 entries are ever outstanding, so the slot is not reused before this
                if (li < 0) {
                    c->abort_flag = 1;
                    li = -2;
                }
            }
        } else if (lds_load(&c->tree_done) >= n_tree) {
            li = -2;
        } else if (wall_clock64() - t_start > t_limit || lds_load(&c->abort_flag)) {
            c->abort_flag = 1;
            li = -2;
        }
    }
    return __builtin_amdgcn_readfirstlane(li);
}

// Tree-wave side: publish this call's outcome for every game leader lane (`leader`) of the wave.  Not inlined for the
// same reason as queue_pop: the lanes of the wave must come back from here together.
// GLOBAL_TOO: the consumer of the queue entry also reads what this wave wrote to global memory (the network wave applies
// the result to the tree itself: float32 form); otherwise only the mailbox in LDS has to be visible.
template <bool GLOBAL_TOO>
__device__ __attribute__((noinline)) void queue_push(QueueCtl *c, uint8_t *state_byte, bool leader, bool posted, int li) {
    if (GLOBAL_TOO) release_global_then_lds(); // mailbox + tree writes before the queue entry
    else __threadfence_block();
    if (leader) {
        if (posted) {
            *(volatile uint8_t *)state_byte = 1;
            int idx = atomicAdd(&c->tail, 1); // the entry becomes valid when its slot turns non-negative
            *(volatile int *)&c->q[idx & (MEGA2_QCAP - 1)] = li;
        } else {
            *(volatile uint8_t *)state_byte = 0;
        }
    }
}

// X3: the network runs on the bf16 matrix pipe (net_x3.hip.h); its packed operands are 1.5x the float32 ones, so fewer
// network waves fit next to them (5 at 4 residual blocks) -- they need far fewer matrix cycles per evaluation.
template <class G, int NETW, bool X3 = false, int WAVES = 12>
__global__ void __launch_bounds__(WAVES * 64) k_selfplay_queue(TreeDev dg, NetDev nd, NetX3 x3, int noise_on, int limit_s, int own_visits) {
    constexpr int MEGA2_THREADS = WAVES * 64;
    // The wave that holds an evaluation also expands the leaf and backs the value up when the tree waves are the busier side
    // (float32 network: they share their SIMD's vector ALUs with the f32 MFMAs); beside the bf16-pipe network the network
    // waves are the busy side and the tree waves take the result back.
    constexpr bool NET_APPLIES = BB_NET_APPLIES >= 0 ? BB_NET_APPLIES != 0 : !X3;
    constexpr int S = G::S, GW = 16, TREEW = WAVES - NETW, GPT = (GW + TREEW - 1) / TREEW; // games per workgroup, tree waves, games per tree wave
    static_assert(GPT * S <= 64, "a tree wave holds at most 64 / S games");
    using NG = NetGeom<G, 1>;
    constexpr int RMAX = MEGA_RMAX, STEPS0 = NG::STEPS0;
    using XG = X3Geom<G>;
    constexpr int EPI_F = (1 + 2 * RMAX) * 48, HEAD_F = MEGA_HEAD_FLOATS;
    // operand floats: float32 form [wt][w0], x3 form [w0 bytes][wt bytes]; then epilogue constants and head parameters
    constexpr int WT_F = X3 ? 2 * RMAX * XG::LAYER12_B / 4 : 2 * RMAX * 9 * 64 * 4, W0_F = X3 ? XG::W0_B / 4 : STEPS0 * 64;
    constexpr int WAVE_F = X3 ? XG::WAVE_BYTES / 4 : NG::WAVE_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[NETW * WAVE_F];
    constexpr int WH_F = X3 ? 3 * 64 * 16 / 4 : 0; // (x3: the head convolutions' three MFMA operands)
    __shared__ __attribute__((aligned(16))) float wlds[WT_F + W0_F + EPI_F + HEAD_F + WH_F];
    __shared__ QueueCtl qc;
    __shared__ GameShadow<G, GW, X3> shadow;
    // Per-game state: 0 owned by its tree wave, 1 leaf queued / being evaluated, 2 result published.  One byte per game, the
    // games of a tree wave next to each other (GSI): the wave sees all of its games' states in ONE 8-byte LDS read.
    static_assert(GPT <= 8, "a tree wave's game states are one 8-byte word");
    __shared__ __attribute__((aligned(8))) uint8_t gstate[TREEW * 8];
#define GSI(li) ((((li) % TREEW) << 3) + (li) / TREEW)
    __shared__ int myslot[NETW];
    // Visits are not dealt per game.  A workgroup starts with its own share (`own_visits` per slot: 7/8 of the launch's visits,
    // so that every slot advances even when more workgroups are launched than the chip holds at once -- the resident ones
    // cannot drain what belongs to the later ones) and every game of the workgroup searches until that is used up; the last
    // eighth is ONE launch-wide pool (dg.visit_pool) that the workgroups then draw from in chunks until it is dry.  With a
    // fixed count per game the pipeline of a workgroup ran empty game by game at the end of every launch and the launch
    // waited for its slowest workgroup (~35 ms of a 16-step launch); a game's results do not depend on when its visits happen.
    __shared__ int wg_pool, wg_dry, wg_refill;
    // X3: the prior noise of a posted leaf is drawn by its tree wave (20 % slack) before the leaf is queued, not by the network
    // wave at the end of the evaluation (the busy side): the same Philox trials in the same order, so the same values.
    __shared__ float s_noise[X3 ? GW * S : 1];
    (void)s_noise;
    constexpr bool TREE_HEADS = X3 && !NET_APPLIES && (BB_TREE_HEADS != 0);
    constexpr bool TREE_STEP = X3 && !NET_APPLIES && (BB_TREE_STEP != 0);
    constexpr bool TREE_NOISE = X3 && (TREE_HEADS || BB_TREE_NOISE != 0);
    __shared__ float s_pooled[TREE_HEADS ? GW * 4 : 1]; // R, R0, R1 of a finished evaluation (the tree wave forms value and priors from them)
    constexpr int CHUNK = GW * 32;
#ifdef BB_STAMPS
    __shared__ long long ts_post[GW], ts_done[GW];
#endif
    const int wave = threadIdx.x >> 6, l64 = threadIdx.x & 63;
    const int g0 = blockIdx.x * GW;
    const int n_mine = dg.n_slots - g0 < GW ? dg.n_slots - g0 : GW;
    shadow.load(dg, g0, n_mine, MEGA2_THREADS);
    const TreeDev d = shadow.local(dg, g0); // everything below works on the LDS copies, indexed by the local game number
    if (threadIdx.x == 0) {
        qc.head = 0;
        qc.tail = 0;
        qc.tree_done = 0;
        qc.abort_flag = 0;
        wg_pool = n_mine * own_visits;
        wg_dry = 0;
        wg_refill = 0;
    }
    if (threadIdx.x < TREEW * 8) gstate[threadIdx.x] = 0;
#ifdef BB_STAMPS_NET
    if (threadIdx.x < 8) s_net_stamps[threadIdx.x] = 0;
#endif
    if (threadIdx.x < MEGA2_QCAP) qc.q[threadIdx.x] = -1;
    for (int i = threadIdx.x; i < NETW * WAVE_F; i += MEGA2_THREADS) lds[i] = 0.f;
    if (X3) for (int i = threadIdx.x; i < GW * S; i += MEGA2_THREADS) s_noise[i] = 0.f; // (the spare slot of a row is its ready flag)
    NetDev ndl = nd;
    NetX3 x3l = x3;
    {
        const float *gwt = X3 ? (const float *)x3.wt12 : (const float *)nd.wt;
        const float *gw0 = X3 ? (const float *)x3.w0 : nd.w0;
        const int wt_used = X3 ? 2 * nd.R * XG::LAYER12_B / 4 : 2 * nd.R * 9 * 64 * 4;
        for (int i = threadIdx.x; i < wt_used; i += MEGA2_THREADS) wlds[i] = gwt[i];
        for (int i = threadIdx.x; i < W0_F; i += MEGA2_THREADS) wlds[WT_F + i] = gw0[i];
        x3l.wt12 = (const unsigned char *)wlds; // (x3l.wt3 stays in global memory)
        x3l.w0 = (const unsigned char *)(wlds + WT_F);
        for (int i = threadIdx.x; i < (1 + 2 * nd.R) * 48; i += MEGA2_THREADS) wlds[WT_F + W0_F + i] = nd.epi[i];
        for (int i = threadIdx.x; i < nd.head_floats; i += MEGA2_THREADS) wlds[WT_F + W0_F + EPI_F + i] = nd.head[i];
        if constexpr (X3) {
            for (int i = threadIdx.x; i < WH_F; i += MEGA2_THREADS) wlds[WT_F + W0_F + EPI_F + HEAD_F + i] = ((const float *)x3.wh)[i];
            x3l.wh = (const unsigned char *)(wlds + WT_F + W0_F + EPI_F + HEAD_F);
        }
        ndl.wt = (const f32x4 *)wlds;
        ndl.w0 = wlds + WT_F;
        ndl.epi = wlds + WT_F + W0_F;
        ndl.head = wlds + WT_F + W0_F + EPI_F;
    }
    __syncthreads();
    const long long t_start = wall_clock64();
    const long long t_limit = 100000000ll * limit_s; // wall clock runs at 100 MHz

    if (wave >= NETW) { // ---------------- tree waves ----------------
        const int tw = wave - NETW;
        // the tree waves are the latency chain of every game: let them win issue arbitration against the
        // throughput-bound network waves of their SIMD (+8 % games/s)
        __builtin_amdgcn_s_setprio(3);
        const int li = (l64 / S) * TREEW + tw, lane = l64 % S; // games are dealt round-robin to the tree waves
        const bool mine = l64 < GPT * S && li < GW && g0 + li < d.n_slots;
        const int g = li; // local game number: index of the LDS copies
        bool alive = mine && d.game_lid[mine ? g : 0] >= 0; // false once the slot has run out of games (or never had one)
#ifdef BB_STAMPS
        long long t_work = 0, t_all0 = clock64(), n_calls = 0, n_lanes = 0, t_pick = 0, n_pick = 0;
#endif
        // TREE_HEADS: value + priors of a finished evaluation from the pooled head activations its network wave left in s_pooled
        auto finish_eval = [&](bool doit) __attribute__((always_inline)) {
            if (doit) {
                const float *hp = as_lds(ndl.head);
                // the leaf's prior noise: the same Philox trials in the same order as head_one makes.  Drawn here and not when the
                // leaf is posted: in the level-stepped loop this work runs under the other games' row loads
                float nz = 0.f;
                if (noise_on && lane < G::A) {
                    if constexpr (TREE_STEP) nz = bb_beta_noise(ndl.seed, d.leaf_game_id[g], (uint32_t)d.leaf_serial[g], (uint32_t)lane, ndl.alpha);
                    else nz = s_noise[li * S + lane];
                }
                float prior;
                const float value = head_tree<G>(ndl, hp, s_pooled[li * 4], s_pooled[li * 4 + 1], s_pooled[li * 4 + 2], lane, l64 - lane,
                                                 noise_on != 0, nz, &prior);
                if (lane == 0) d.eval_value[g] = value;
                if (lane < G::A) d.eval_policy[g * S + lane] = prior;
            }
            wave_lds_handover();
        };
        // A leaf goes to the network waves the moment its mailbox is written (async_game's on_post hook) -- a call of this wave
        // lasts as long as the slowest of its games' descents -- and its prior noise (~1 k cycles of Philox + transcendentals) is
        // drawn right behind the queue entry, while the evaluation already runs: the network wave needs the draws only in its
        // tail, ~30 k cycles later, and waits there for the flag in the game's spare noise slot (head_one).
        static_assert(!TREE_NOISE || G::A < S, "the noise flag sits in the lane group's spare slot");
        auto early_post = [&](int gg, int ln) __attribute__((always_inline)) {
            if (NET_APPLIES) release_global_then_lds(); // mailbox (+ tree writes) before the queue entry
            else __threadfence_block();
            if (ln == 0) {
#ifdef BB_STAMPS
                ts_post[li] = wall_clock64();
#endif
                *(volatile uint8_t *)&gstate[GSI(li)] = 1;
                const int idx = atomicAdd(&qc.tail, 1); // the entry becomes valid when its slot turns non-negative
                *(volatile int *)&qc.q[idx & (MEGA2_QCAP - 1)] = li;
            }
            if constexpr (TREE_NOISE) {
                if (noise_on) {
                    if (ln < G::A) s_noise[li * S + ln] = bb_beta_noise(ndl.seed, d.leaf_game_id[gg], (uint32_t)d.leaf_serial[gg], (uint32_t)ln, ndl.alpha);
                    __threadfence_block();
                    if (ln == 0) __hip_atomic_store(&s_noise[li * S + S - 1], 1.0f, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        (void)early_post;
        if constexpr (TREE_STEP) {
        // ---- level-stepped tree wave -----------------------------------------------------------------------------------------
        // The games of a wave do not take turns in whole async_game calls (apply -> descend to a leaf -> post; a result that
        // arrived meanwhile waited ~8 us for the call in progress, and a call moved at the pace of its slowest game): every
        // iteration of this loop advances EVERY descending game of the wave by one tree level, and a game whose evaluation has
        // arrived joins in the same iteration (value / priors from the pooled activations, expand + backup, [move], root).
        // The descending games' row loads are issued first, so the intake work runs under their latency and the dependent
        // loads of all the wave's games overlap.  Per game the sequence of operations is async_game's (tree.hip.h) step for
        // step -- same functions, same order -- so the examples are the same bits (tests/test_gpu_noise_parity.py).
        using Node = DenseNode<G>;
        constexpr int A = G::A;
        enum { PH_WAIT = 0, PH_DESC = 1, PH_DEAD = 2, PH_START = 3 };
        enum { F_LEAF = 4, F_TERM = 8, F_EXPAND = 16, F_OVERFLOW = 32 };
        int ph = (mine && d.game_lid[mine ? g : 0] >= 0) ? PH_WAIT : PH_DEAD;
        int cur = 0, depth = 0, nn = 0, sims_done = 0, depth_sum = 0, term_hits = 0;
        Node *const npool = (Node *)d.nodes + (size_t)((mine ? g : 0) + d.pool_g0) * d.node_cap;
        const int pb = (mine ? g : 0) * G::MAXPATH; // this game's part of the recorded-path arrays (LDS: indices, not pointers)
        typename G::State st = G::initial();
        unsigned iter = 0;
        // the next simulation of a game that is owned by this wave: FindMove's tail when the move's simulations are done, then the root
        auto start_sim = [&]() __attribute__((always_inline)) {
            if (d.sims_left[g] <= 0) {
                selfplay_move_body<G>(d, g, lane);
                __threadfence_block();
                if (d.game_lid[g] < 0) {
                    ph = PH_DEAD;
                    return;
                }
            }
            cur = d.resume_cur[g];
            depth = 0;
            if (cur >= 0) {
                depth = d.resume_depth[g];
                if (lane == 0) d.resume_cur[g] = -1;
            } else {
                cur = d.root[g];
            }
            nn = d.n_nodes[g];
            ph = PH_DESC;
        };
        auto flush_counters = [&]() __attribute__((always_inline)) {
            if (lane == 0 && sims_done) {
                uint64_t *c = d.ctr + (size_t)g * 8;
                c[0] += (uint64_t)sims_done;
                c[1] += (uint64_t)depth_sum;
                c[3] += (uint64_t)term_hits;
            }
            sims_done = depth_sum = term_hits = 0;
        };
        struct Row { // what a level needs of a node's row: one round trip
            typename G::State st;
            int flags, Ni, ci, all;
            uint32_t mask;
            float Qi, Wi;
            double sq, cPi, cached;
        };

        auto load_row = [&](Row &r, const Node *node) __attribute__((always_inline)) {
            r.st = node->st;
            r.flags = node->flags;
            r.mask = node->legal_mask;
            r.sq = node->sq;
            r.Ni = node->N[lane];
            r.Qi = node->Q[lane];
            r.Wi = node->W[lane];
            r.all = node->all;
            r.cPi = node->cP[lane];
            r.ci = node->child[lane];
            r.cached = node->pad0;
        };
        for (;;) {
            // (1) the descending games' rows: requested before anything else
            Row row;
            if (ph == PH_DESC) load_row(row, npool + cur);
            // (2) the workgroup's share of the launch's visits
            const int poolv = lds_load(&wg_pool), dry = lds_load(&wg_dry);
            if (poolv < CHUNK / 4 && !dry) { // (wave-uniform) top the workgroup's share up before it runs out
                if (l64 == 0 && atomicCAS(&wg_refill, 0, 1) == 0) {
                    int old = atomicSub(d.visit_pool, CHUNK);
                    int got = old < 0 ? 0 : old < CHUNK ? old : CHUNK;
                    if (got) atomicAdd(&wg_pool, got);
                    else *(volatile int *)&wg_dry = 1;
                    __threadfence_block();
                    *(volatile int *)&wg_refill = 0;
                }
            }
            // (3) intake: a game whose evaluation has arrived (or that this wave owns without a leaf in flight: launch start)
            const int stt = ph == PH_WAIT ? (int)*(volatile uint8_t *)&gstate[GSI(li)] : 1;
            const bool intake = ph == PH_WAIT && stt != 1 && poolv > 0;
            const bool was_desc = ph == PH_DESC;
            const bool busy = ph == PH_DESC || (ph == PH_WAIT && (stt == 1 || poolv > 0 || !dry)); // descending, a leaf in flight, or visits left to draw
            if (!__any(busy)) break;
#ifdef BB_STAMPS
            const long long ts = clock64();
#endif
            if (__any(intake)) {
                __threadfence_block(); // acquire: the network wave's results for state 2
                {
                    const int used = __popcll(__ballot(intake && lane == 0));
                    if (l64 == 0) atomicSub(&wg_pool, used);
                }
                if (intake) {
#ifdef BB_STAMPS
                    if (lane == 0 && stt == 2) {
                        t_pick += wall_clock64() - ts_done[li];
                        n_pick++;
                    }
#endif
                    if constexpr (TREE_HEADS) finish_eval(stt == 2);
                    if (lane == 0) *(volatile uint8_t *)&gstate[GSI(li)] = 0;
                    if (d.pend_leaf[g] >= 0) {
                        if (d.leaf_flags[g] & LEAF_RECORDED) phase_apply_rec<G>(d, g, lane);
                        else phase_apply<G>(d, g, lane); // (a leaf that another kernel posted)
                        if (lane == 0) d.sims_left[g] -= 1;
                        __threadfence_block();
                    }
                    ph = PH_START;
                }
            }
            // (4) one level for every descending game
            bool posting = false;
            if (was_desc) {
                asm volatile("" ::"v"(row.flags), "v"(row.mask), "v"(row.sq), "v"(row.Ni), "v"(row.Qi), "v"(row.cPi), "v"(row.ci)); // one wait for the whole row
                st = row.st;
                int flags = row.flags, fl = 0;
                if (!(flags & NODE_EXPANDED)) {
                    fl |= F_LEAF;
                    if (flags & NODE_TERMINAL) {
                        fl |= F_TERM;
                        if (flags & NODE_CACHED) { // value already known (the reference's lru_cache on SampleValue): finish this simulation here
                            const float v01 = (float)row.cached;
                            __threadfence_block(); // path stores of this descent
                            backup_path_rec<G>(d, g, lane, npool, depth, v01, gs_prev(st));
                            if (lane == 0) d.sims_left[g] -= 1;
                            __threadfence_block();
                            sims_done++;
                            depth_sum += depth;
                            term_hits++;
                            fl = 0; // nothing to post: the next simulation (or the move) starts
                            if (lane == 0) d.sim_serial[g] += 1;
                            ph = PH_START;
                        }
                    } else {
                        fl |= F_EXPAND;
                    }
                } else {
                    static_assert(G::MAXPATH >= G::H * G::W + 2, "a descent places at most H*W stones");
                    const double u = puct_score(child_q(d, row.Qi, 0.f, row.Ni), row.cPi, row.sq, row.Ni, lane < A && ((row.mask >> lane) & 1u));
                    int child = row.ci;
                    const int a = grp_argmax<S>(u, lane, child);
                    if (lane == 0) {
                        d.path[pb + depth] = ((uint32_t)cur << 6) | ((uint32_t)((flags >> 4) & 3) << 4) | (uint32_t)a;
                        d.path_all[pb + depth] = row.all;
                    }
                    if (lane == a) { // what the backup of this edge will start from
                        d.path_N[pb + depth] = row.Ni;
                        d.path_W[pb + depth] = row.Wi;
                    }
                    if (child == CHILD_NONE) {
                        typename G::State st2;
                        bool terminal;
                        child = create_child<G>(d, g, npool, npool + cur, st, a, lane, nn, st2, terminal, &flags);
                        if (child == CHILD_NONE) {
                            fl |= F_OVERFLOW | F_LEAF;
                        } else { // the node just created is the leaf of this descent (never expanded, never cached)
                            st = st2;
                            depth++;
                            cur = child & ~CHILD_TERM_BIT;
                            fl |= F_LEAF | (terminal ? F_TERM : F_EXPAND);
                        }
                    } else {
                        depth++;
                        cur = child & ~CHILD_TERM_BIT;
                    }
                }
                if (fl & F_LEAF) { // post the leaf for the evaluator
                    const uint32_t gid = d.first_game_id + (uint32_t)d.game_lid[g];
                    if (lane == 0) {
                        ((typename G::State *)d.leaf_state)[g] = st;
                        d.leaf_game_id[g] = gid;
                        d.leaf_serial[g] = cur;
                        d.pend_leaf[g] = cur;
                        d.leaf_flags[g] = flags | LEAF_RECORDED;
                        d.pend_expand[g] = (fl & F_EXPAND) ? 1 : 0;
                        d.path_len[g] = depth;
                        d.sim_serial[g] += 1;
                        d.evals[g] += 1;
                        d.ctr[(size_t)g * 8 + 6] += (uint64_t)((fl & F_OVERFLOW) ? 1 : 0);
                    }
                    sims_done++;
                    depth_sum += depth;
                    term_hits += (fl & F_TERM) ? 1 : 0;
                    flush_counters();
                    // the leaf's prior noise, drawn here (the same Philox trials in the same order as the network wave would make)
                    if constexpr (TREE_NOISE && !TREE_HEADS) // (with TREE_HEADS the draw is made when the evaluation is picked up: finish_eval)
                        if (noise_on && lane < A) s_noise[li * S + lane] = bb_beta_noise(ndl.seed, gid, (uint32_t)cur, (uint32_t)lane, ndl.alpha);
                        if (noise_on && lane == 0) s_noise[li * S + S - 1] = 1.0f; // ready flag (head_one waits for it; here the draw precedes the post)
                    posting = true;
                    ph = PH_WAIT;
#ifdef BB_STAMPS
                    if (lane == 0) ts_post[li] = wall_clock64();
#endif
                }
            }
            // (5) the next simulation of the games that finished one above (ONE call site: the move body stays out of line only once)
            if (ph == PH_START) {
                start_sim();
                if (ph == PH_DEAD) flush_counters();
            }
            const unsigned long long act = __ballot(ph == PH_DESC || posting || intake);
            if (__any(posting)) queue_push<false>(&qc, &gstate[GSI(li < GW ? li : 0)], posting && lane == 0, true, li);
            if (!act || (++iter & 1023) == 0) { // nothing of this wave's could move: every game waits for its evaluation (or for visits)
                if (!act) __builtin_amdgcn_s_sleep(BB_TREE_IDLE_SLEEP);
                int late = wall_clock64() - t_start > t_limit || lds_load(&qc.abort_flag); // (every wait of the kernel is bounded by wall-clock time)
                if (__builtin_amdgcn_readfirstlane(late)) {
                    qc.abort_flag = 1;
                    break;
                }
            }
#ifdef BB_STAMPS
            if (act) {
                n_calls++;
                n_lanes += __popcll(act) / S;
                t_work += clock64() - ts;
            }
#endif
        }
        flush_counters();
        } else
        for (;;) {
            int stt = mine ? (int)*(volatile uint8_t *)&gstate[GSI(li)] : 1;
            const int pool = lds_load(&wg_pool), dry = lds_load(&wg_dry);
            if (pool < CHUNK / 4 && !dry && __any(alive)) { // (wave-uniform) top the workgroup's share up before it runs out
                if (l64 == 0 && atomicCAS(&wg_refill, 0, 1) == 0) {
                    int old = atomicSub(d.visit_pool, CHUNK);
                    int got = old < 0 ? 0 : old < CHUNK ? old : CHUNK;
                    if (got) atomicAdd(&wg_pool, got);
                    else *(volatile int *)&wg_dry = 1;
                    __threadfence_block();
                    *(volatile int *)&wg_refill = 0;
                }
                if (pool <= 0) { // another wave's refill is on its way: wait for it like every other wait, bounded by wall-clock time
                    __builtin_amdgcn_s_sleep(1);
                    int late = wall_clock64() - t_start > t_limit || lds_load(&qc.abort_flag);
                    if (__builtin_amdgcn_readfirstlane(late)) {
                        qc.abort_flag = 1;
                        break;
                    }
                    continue;
                }
            }
            bool ready = mine && alive && pool > 0 && stt != 1;
            bool busy = mine && ((alive && pool > 0) || (alive && !dry) || stt == 1); // visits left to draw, or a leaf of mine is in flight
            if (!__any(busy)) break;
            if (__any(ready)) {
                __threadfence_block(); // acquire: the network wave's results for state 2
                bool posted = false;
                if constexpr (TREE_HEADS) finish_eval(ready && stt == 2);
#ifdef BB_STAMPS
                long long ts = clock64();
                n_calls++;
                n_lanes += __popcll(__ballot(ready)) / S;
                if (ready && lane == 0 && stt == 2) {
                    t_pick += wall_clock64() - ts_done[li];
                    n_pick++;
                }
#endif
#ifdef BB_STAMPS_LIGHT
                int ll = 0, lv = 0, lld = 0, lpu = 0;
                if (ready) {
                    posted = async_game<G, X3>(d, g, lane, ll, lv, lld, lpu, early_post);
                    if (d.game_lid[g] < 0) alive = false;
                }
                { // the game with the most levels ran the whole length of the call: its loop time per level is undiluted
                    int m = lv;
                    for (int o = 32; o; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
                    unsigned long long who = __ballot(ready && lane == 0 && lv == m);
                    if (who && l64 == (int)__builtin_ctzll(who) && d.stamps) {
                        atomicAdd(&d.stamps[6], (unsigned long long)ll);
                        atomicAdd(&d.stamps[7], (unsigned long long)lv);
                        atomicAdd(&d.stamps[11], (unsigned long long)lld);   // (LIGHT: 11 / 12 carry the split, not the call totals)
                        atomicAdd(&d.stamps[12], (unsigned long long)lpu);
                    }
                }
#else
                if (ready) {
                    posted = async_game<G, X3>(d, g, lane, early_post);
                    if (d.game_lid[g] < 0) alive = false; // slot ran out of games
                }
#endif
                {
                    int used = __popcll(__ballot(ready && lane == 0));
                    if (l64 == 0) atomicSub(&wg_pool, used);
                }
                // (posted leaves went to the network waves inside the call: early_post; the others go back to their tree wave)
                queue_push<NET_APPLIES>(&qc, &gstate[GSI(li < GW ? li : 0)], ready && lane == 0 && !posted, false, li);
#ifdef BB_STAMPS
                t_work += clock64() - ts;
#endif
            } else {
                __builtin_amdgcn_s_sleep(BB_TREE_IDLE_SLEEP);
                int late = wall_clock64() - t_start > t_limit || lds_load(&qc.abort_flag);
                if (__builtin_amdgcn_readfirstlane(late)) {
                    qc.abort_flag = 1;
                    break;
                }
            }
        }
        if constexpr (TREE_HEADS) { // an evaluation that nobody picked up any more goes to the next launch through the game's mailbox
            __threadfence_block();
            finish_eval(mine && *(volatile uint8_t *)&gstate[GSI(li)] == 2);
        }
        if (l64 == 0) atomicAdd(&qc.tree_done, 1);
#ifdef BB_STAMPS
        if (l64 == 0 && d.stamps) {
            atomicAdd(&d.stamps[2], (unsigned long long)t_work);
            atomicAdd(&d.stamps[3], (unsigned long long)(clock64() - t_all0));
            atomicAdd(&d.stamps[5], 1ull);
            atomicAdd(&d.stamps[13], (unsigned long long)n_calls);
            atomicAdd(&d.stamps[14], (unsigned long long)n_lanes);
        }
        if (lane == 0 && d.stamps && n_pick) {
            atomicAdd(&d.stamps[8], (unsigned long long)t_pick);
            atomicAdd(&d.stamps[9], (unsigned long long)n_pick);
        }
#endif
    } else { // ---------------- network waves ----------------
        float *wl = lds + wave * WAVE_F;
#ifdef BB_STAMPS
        long long t_work = 0, t_all0 = clock64(), n_evals = 0, t_qwait = 0;
#endif
        for (;;) {
            const int li = __builtin_amdgcn_readfirstlane(queue_pop(&qc, t_start, t_limit, TREEW)); // scalar: uniform branches below
            if (li == -2) break;
            if (li < 0) {
                __builtin_amdgcn_s_sleep(BB_NET_IDLE_SLEEP);
                continue;
            }
            __threadfence_block(); // acquire: the tree wave's mailbox writes
#ifdef BB_STAMPS
            long long ts = clock64();
            n_evals++;
            t_qwait += wall_clock64() - ts_post[li];
#endif
            if (l64 == 0) myslot[wave] = li;
#ifdef Q_HASH
            {
                int slot = li;
                uint64_t sl = d.salt + (d.salt_per_game ? (uint64_t)(d.leaf_game_id[slot] - d.first_game_id) : 0ull);
                uint64_t z = hash_state<G>(((const typename G::State *)d.leaf_state)[slot], sl);
                if (l64 == 0) d.eval_value[slot] = bb_hash_value(z);
                if (l64 < G::A) d.eval_policy[(size_t)slot * S + l64] = bb_hash_policy(z, l64);
            }
#else
            if constexpr (X3)
                net_body_x3<G, true, (BB_X3_LEAN != 0), false, TREE_HEADS>(ndl, x3l, 1, 0, &myslot[wave], (unsigned char *)wl, (const typename G::State *)d.leaf_state, nullptr,
                                     d.leaf_game_id, d.leaf_serial, noise_on, d.eval_value, nullptr, d.eval_policy, S, false, nullptr,
                                     (noise_on && TREE_NOISE) ? s_noise + li * S : nullptr, TREE_HEADS ? s_pooled + li * 4 : nullptr);
            else
                net_body<G, 1, BB_QUEUE_WMODE>(ndl, 1, 0, &myslot[wave], wl, (const typename G::State *)d.leaf_state, nullptr,
                                               d.leaf_game_id, d.leaf_serial, noise_on, d.eval_value, nullptr, d.eval_policy, S, false);
#endif
            if constexpr (NET_APPLIES) {
            // The evaluated leaf is expanded and its value backed up right here, by the wave that holds the result, instead
            // of waiting until the game's tree wave comes round (the tree waves are every game's latency chain and the
            // busier side of the queue: 91 % against 67 %; a result used to wait ~10 us to be picked up).  Same function,
            // same lane layout (lane i <-> child slot i on lanes 0..S-1), so the same bits.
            __threadfence_block(); // the mailbox writes of net_body (other lanes) before phase_apply reads them
#ifdef BB_STAMPS_NET
            long long _na = clock64();
#endif
            if (l64 < S) {
                phase_apply<G>(d, li, l64);
                if (l64 == 0) d.sims_left[li] -= 1;
            }
#ifdef BB_STAMPS_NET
            if (l64 == 0) atomicAdd(&s_net_stamps[5], (unsigned long long)(clock64() - _na));
#endif
            }
            release_global_then_lds(); // value / policy (and the tree rows) before the state word
#ifdef BB_STAMPS
            if (l64 == 0) ts_done[li] = wall_clock64();
#endif
            if (l64 == 0) *(volatile uint8_t *)&gstate[GSI(li)] = 2;
#ifdef BB_STAMPS
            t_work += clock64() - ts;
#endif
        }
#ifdef BB_STAMPS
        if (l64 == 0 && d.stamps) {
            atomicAdd(&d.stamps[0], (unsigned long long)t_work);
            atomicAdd(&d.stamps[1], (unsigned long long)(clock64() - t_all0));
            atomicAdd(&d.stamps[4], 1ull);
            atomicAdd(&d.stamps[15], (unsigned long long)n_evals);
            atomicAdd(&d.stamps[10], (unsigned long long)t_qwait);
        }
#endif
    }
    __syncthreads();
#ifdef BB_STAMPS_NET
    if (threadIdx.x < 8) atomicAdd(&g_net_stamps[threadIdx.x], s_net_stamps[threadIdx.x]);
#endif
    if (qc.abort_flag) { // a wait ran into the wall-clock limit: surfaces as bb_counters.overflow
        if (threadIdx.x == 0) d.ctr[6] += 1;
        // a leaf that was queued but never evaluated must not be applied by the next launch (its mailbox holds the
        // previous evaluation): drop it, the simulation is redone from the root
        if ((int)threadIdx.x < n_mine && gstate[GSI(threadIdx.x)] == 1) shadow.pend_leaf[threadIdx.x] = -1;
    }
    __syncthreads();
    shadow.store(dg, g0, n_mine, MEGA2_THREADS); // hand the per-game state back to HBM
}
