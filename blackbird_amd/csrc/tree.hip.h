// tree.hip.h -- flat node-pool MCTS for dense-action games (Connect4, TicTacToe) on gfx950.
//
// One group of G::S lanes owns one game slot; lane i owns child slot i of the node row being
// looked at (N/W/P/child live side by side in one node record, so a level costs one round trip).
// Semantics follow the reference exactly (citations relative to /root/reference/src):
//   select   MCTS._selectAction  MCTS.py:327-334  (float64 PUCT, first maximum wins)
//            DynamicMCTS._findLeaf DynamicMCTS.py:14-34 / FixedMCTS._findLeaf FixedMCTS.py:21-34
//   expand   MCTS.AddChildren MCTS.py:122-139 + Model.GetPriors Blackbird.py:372-389
//   backup   MCTS._backProp MCTS.py:238-258 + Model.SampleValue Blackbird.py:350-370
//   move     MCTS._selectAction(exploring=False) :335-338, _moveRoot :260-282,
//            Blackbird.GenerateTrainingSamples Blackbird.py:238-268
// Arithmetic types are the ones numpy>=2 gives the reference: float32 value/Value/WinRate,
// float64 priors and PUCT.  (Rollout evaluator: Python floats -> float64 WinRate.)
//
// A node is created when a simulation first reaches it and is expanded in the same simulation
// (the reference creates child Node objects eagerly in AddChildren but reads a child's priors
// only once that child has children of its own, so this is observationally identical).
#pragma once
#include "games.hip.h"
#include "rng.hip.h"

// Measured on MI355X (4096 games, 800 sims): touching all 7 children's rows per level makes the
// descent bandwidth-bound (46 MB/step) and is not faster than the plain dependent row load.
#ifndef BB_PREFETCH_CHILDREN
#ifndef BB_PREFETCH_BEST
#define BB_PREFETCH_BEST 0
#endif
#define BB_PREFETCH_CHILDREN 0
#endif
#ifndef BB_MOVE_BODY_ATTR
#define BB_MOVE_BODY_ATTR __forceinline__ // (an out-of-line callee takes TreeDev by reference: the struct then lives in scratch and every field use is a scratch load)
#endif
#define NODE_EXPANDED 1
#define NODE_TERMINAL 2
#define NODE_CACHED 4 // terminal node whose evaluator value is stored in pad0 (Model.SampleValue's lru_cache)
#define LEAF_RECORDED 0x40000000
#define CHILD_NONE (-1)
#define CHILD_TERM_BIT 0x40000000

// One node row.  Everything a descent needs from a node sits in the first 128-byte line plus cP
// (second line); the statistics are kept in the form the PUCT expression consumes them:
//   Q[i]  = child.WinRate()  (float32 division, refreshed by the backup that changes it)
//   sq    = sqrt(1.0 + sum(ChildPlays)) (float64, refreshed by the backup)
//   cP[i] = ExplorationRate * Priors[i] (float64, fixed at expansion)
// so that one level of _selectAction costs one row load, one float64 multiply, one divide, one add.
template <class G>
struct alignas(128) DenseNode {
    int32_t flags;       // NODE_EXPANDED | NODE_TERMINAL | Player << 4 | (winner+1) << 8
    uint32_t legal_mask;
    int32_t all;         // sum(ChildPlays)
    int32_t serial;
    double sq;           // sqrt(1.0 + all)
    double pad0;
    int32_t N[G::S];     // child.Plays
    float Q[G::S];       // child.Value / child.Plays (float32), 0 when unvisited
    int32_t child[G::S]; // node index | CHILD_TERM_BIT, or CHILD_NONE
    alignas(128) double cP[G::S]; // c_puct * Node.Priors
    float W[G::S];       // child.Value
    typename G::State st;
};

struct ExampleHdr { // 16 bytes, then packed state, then u32 visits[S]
    uint32_t game_id;
    uint16_t ply;
    uint8_t player;
    int8_t z;
    uint32_t total;
    uint32_t n_children;
};

struct TreeDev {
    // configuration
    int n_slots, node_cap, sims_per_move, max_plies, kind, max_depth, evaluator, priors_ones;
    int salt_per_game, max_games;
    int gpw; // games per 64-lane wave in k_tree_step (latency tuning: fewer games per wave = less SIMT divergence)
    double c_puct;
    uint64_t seed, salt;
    uint32_t first_game_id;
    // per-slot tree state
    int32_t *root, *root_N, *n_nodes, *ply, *sims_left, *pend_leaf, *pend_expand, *path_len;
    int32_t *game_lid; // local game index played in this slot, -1 idle
    int32_t *sim_serial;
    float *root_W;
    int8_t *root_pp; // Player of the root's parent state, 0 = root has no parent
    uint32_t *path;  // [n_slots][MAXPATH]  node<<6 | player<<4 | action
    // what the descent saw of every edge of the recorded path (persistent kernel, level-stepped tree waves): the backup is
    // then stores only -- N + 1, W + v, Q, sum + 1, sqrt -- instead of a read-modify-write round trip per simulation
    int32_t *path_N, *path_all; // [n_slots][MAXPATH]  child.Plays of the chosen edge, sum(ChildPlays) of its node
    float *path_W;              // [n_slots][MAXPATH]  child.Value of the chosen edge
    int32_t *leaf_flags;        // [n_slots] flags word of the posted leaf | LEAF_RECORDED (0: posted by a kernel that records nothing)
    // MCTS.ResetRoot (MCTS.py:214-225) walks Root back to its top-most ancestor, and _backProp (:238-258) recurses through every
    // ancestor above the current root, so the statistics found there include every later simulation.  With track_anc (the
    // FindMove front end's engines) a slot keeps the chain of (node, player, action) edges from the first root down to the
    // current one, every backup also walks it, and bb_reset_roots puts the root back at its top.
    int track_anc;
    uint32_t *anc;      // [n_slots][max_plies + 2]  node<<6 | player<<4 | action, top-most first
    int32_t *anc_len;   // [n_slots]
    int32_t *top_N;     // [n_slots] Plays of the top-most ancestor (it has no parent edge to keep them in)
    // evaluator mailboxes
    void *leaf_state;       // [n_slots] packed state of the pending leaf
    uint32_t *leaf_game_id; // [n_slots]
    int32_t *leaf_serial;   // [n_slots]
    float *eval_value;      // [n_slots]
    float *eval_policy;     // [n_slots][S]
    uint64_t *evals;        // [n_slots] leaves handed to the evaluator
    uint64_t *ctr;          // [n_slots][8]: sims, sum_depth, nodes, terminal, games, plies, overflow, examples
    void *nodes;            // [n_slots][node_cap]
    // self-play
    int n_games_target;
    double temp;
    uint8_t *examples; // [max_games][max_plies+1][example_bytes]
    int32_t *game_hdr; // [max_games][4]: n_examples, winner, plies, done
    int example_bytes;
    // bb_sample_moves outputs (device staging)
    int32_t *out_action, *out_root_plays, *out_child_plays;
    float *out_root_winrate, *out_child_value;
    const double *in_u; // optional uniforms
    // asynchronous self-play (k_tree_async): resumable descents + compacted leaf list
    int32_t *resume_cur, *resume_depth; // [n_slots] descent parked by the level budget (-1: none)
    int *post_count;                    // [4] leaves posted in round r -> post_count[r & 3]
    int *post_slot;                     // [n_slots] slots of the posted leaves, in arrival order
    int level_budget;                   // tree levels one launch may descend per game
    // a TreeDev may be a VIEW of a slot range (pointers pre-offset): pipelined self-play runs two views on two streams
    int slot_offset;                    // first slot of this view in the engine
    int lid_stride;                     // engine-wide slot count: a slot's next game id is lid + lid_stride
    // The persistent kernel keeps its workgroup's per-slot arrays (everything above except the node pool) in LDS and
    // indexes them with the LOCAL slot number; the node pool stays in HBM and is indexed with pool_g0 + local number.
    // Everywhere else pool_g0 is 0 and the slot number is the engine's.
    int pool_g0;
    // visits (async_game calls) the running persistent launch may still hand out; its workgroups draw them in chunks
    int *visit_pool;
    float noise_alpha;
    unsigned long long *stamps; // diagnostic build only (BB_STAMPS): [apply, fence, select, levels, waves]
};

// ---- group (G::S lanes) collectives ---------------------------------------------------------
// Butterfly steps over an 8- or 16-lane group as DPP moves (no LDS round trip):
// quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror (i <-> 7-i), row_mirror (i <-> 15-i).
template <int STEP>
__device__ __forceinline__ int dpp_step_i(int v) {
    if (STEP == 0) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);
    if (STEP == 1) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);
    if (STEP == 2) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false);
    return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false);
}
template <int STEP>
__device__ __forceinline__ double dpp_step_d(double v) {
    long long b = __double_as_longlong(v);
    int lo = dpp_step_i<STEP>((int)(b & 0xffffffffll)), hi = dpp_step_i<STEP>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int S>
__device__ __forceinline__ int grp_sum_i(int v) {
    v += dpp_step_i<0>(v);
    v += dpp_step_i<1>(v);
    v += dpp_step_i<2>(v);
    if (S == 16) v += dpp_step_i<3>(v);
    return v;
}

// numpy add.reduce over A float64 values held one per lane (A < 8: sequential; 8 <= A <= 128: 8 partials)
template <int A, int S>
__device__ __forceinline__ double grp_np_sum(double x) {
    double a[A];
#pragma unroll
    for (int k = 0; k < A; k++) a[k] = __shfl(x, k, S);
    if (A < 8) {
        double res = 0.;
#pragma unroll
        for (int k = 0; k < A; k++) res += a[k];
        return res;
    } else {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i = 8;
        for (; i < A - (A % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < A; i++) res += a[i];
        return res;
    }
}

// PUCT argmax over the group's lanes (first maximum wins; u < 0 marks lanes that may not be chosen).
// `payload` rides along, so every lane ends up with the winner's index AND its child word.
template <int S, int STEP>
__device__ __forceinline__ void argmax_step(double &u, int &idx, int &payload) {
    double ou = dpp_step_d<STEP>(u);
    int oi = dpp_step_i<STEP>(idx), op = dpp_step_i<STEP>(payload);
    if (ou > u || (ou == u && oi < idx)) {
        u = ou;
        idx = oi;
        payload = op;
    }
}
template <int S>
__device__ __forceinline__ int grp_argmax(double u, int idx, int &payload) {
    // Two stages instead of a compare-and-select butterfly (whose value/index/payload selects compile to a chain of
    // VALU->SALU exec-mask round trips): the group maximum by v_max_f64 butterflies, then the first lane that holds
    // it (np.argmax: first maximum wins) from one wave-wide equality ballot, then one cross-lane read of the payload.
    // Legal scores are >= -1 and never NaN (illegal lanes are exactly -1.0), so max/== see ordinary numbers.
    (void)idx;
    double m = u;
    m = __builtin_fmax(m, dpp_step_d<0>(m));
    m = __builtin_fmax(m, dpp_step_d<1>(m));
    m = __builtin_fmax(m, dpp_step_d<2>(m));
    if (S == 16) m = __builtin_fmax(m, dpp_step_d<3>(m));
    const unsigned long long eq = __ballot(u == m);
    const int l64 = (int)__lane_id();
    const unsigned grp = (unsigned)(eq >> (l64 & ~(S - 1))) & ((1u << S) - 1u);
    const int win = __ffs(grp) - 1;
    payload = __shfl(payload, win, S);
    return win;
}

// U_i = ChildWinRates_i + (ExplorationRate * Priors_i * sqrt(1 + allPlays)) / (1 + ChildPlays_i)   (MCTS.py:327-332)
__device__ __forceinline__ double puct_score(double q, double cPi, double sq, int Ni, bool legal) {
    double u = q + __ddiv_rn(cPi * sq, 1.0 + (double)Ni);
    return legal ? u : -1.0;
}
// The same argmax without the float64 arithmetic wherever float32 can already tell: BB_PUCT_FILTER.
//   uf = fl32(q + fl32(cP * sq) * rcp32(1 + N)) differs from the float64 score u by at most 2^-21 * u (q, cP, sq >= 0: one rounding
//   of the product, one ulp of v_rcp_f32, one rounding of the fma), and every u is <= the group's true maximum <= mf + that.
//   A lane whose uf lies more than 2^-18 * mf under the float32 maximum mf therefore cannot hold the float64 maximum; if exactly
//   one lane of every group of the wave survives, it is the float64 argmax (first-maximum rule included: there is no tie).
//   Otherwise -- near ties, exact ties, all-zero scores -- the whole wave takes the float64 path.  The float64 divide, the
//   v_max_f64 butterflies and the 64-bit DPP moves (8- to 16-cycle instructions) leave the common path of a tree level.
#ifndef BB_PUCT_FILTER
#define BB_PUCT_FILTER 1
#endif
template <int S>
__device__ __forceinline__ int grp_argmax_puct(float qf, double cPi, double sq, int Ni, bool legal, int &payload) {
#if BB_PUCT_FILTER
    const float num = (float)(cPi * sq);
    const float uf = legal ? __builtin_fmaf(num, __builtin_amdgcn_rcpf((float)(Ni + 1)), qf) : -1.0f;
    float mf = uf;
    mf = __builtin_fmaxf(mf, __int_as_float(dpp_step_i<0>(__float_as_int(mf))));
    mf = __builtin_fmaxf(mf, __int_as_float(dpp_step_i<1>(__float_as_int(mf))));
    mf = __builtin_fmaxf(mf, __int_as_float(dpp_step_i<2>(__float_as_int(mf))));
    if (S == 16) mf = __builtin_fmaxf(mf, __int_as_float(dpp_step_i<3>(__float_as_int(mf))));
    const bool mine = uf >= mf - mf * 0x1p-18f && uf >= 0.0f;
    const unsigned long long cand = __ballot(mine);
    const int l64 = (int)__lane_id();
    const unsigned grp = (unsigned)(cand >> (l64 & ~(S - 1))) & ((1u << S) - 1u);
    if (__ballot((grp & (grp - 1u)) != 0u || grp == 0u) == 0ull) { // one candidate in every group of the wave
        const int win = __ffs(grp) - 1;
        // the winner's payload to every lane of its group: an OR butterfly over the group's DPP steps (one lane contributes)
        // instead of a cross-lane read through LDS (ds_bpermute: a round trip on the critical path of every tree level)
        int p = mine ? payload : 0;
        p |= dpp_step_i<0>(p);
        p |= dpp_step_i<1>(p);
        p |= dpp_step_i<2>(p);
        if (S == 16) p |= dpp_step_i<3>(p);
        payload = p;
        return win;
    }
#endif
    return grp_argmax<S>(puct_score((double)qf, cPi, sq, Ni, legal), 0, payload);
}
// Node.WinRate() of a child: float32 division for float32 evaluators, Python-float division for rollouts
__device__ __forceinline__ double child_q(const TreeDev &d, float Qi, float Wi, int Ni) {
    if (d.evaluator != 2) return (double)Qi;
    return Ni > 0 ? (double)Wi / (double)Ni : 0.0;
}

// MCTS._backProp (MCTS.py:238-258): one (parent, action) edge of the recorded path per lane; v01 is the
// value for `prev` (= leaf.State.PreviousPlayer).  Refreshes Q and sqrt(1+sum N) of the rows it touches.
template <class G>
__device__ __forceinline__ void backup_path(const TreeDev &d, int g, int lane, DenseNode<G> *pool, int plen, float v01,
                                            int prev) {
    constexpr int S = G::S;
    float vflip = 1.0f - v01;
    const uint32_t *path = d.path + (size_t)g * G::MAXPATH;
    auto edge = [&](uint32_t e) __attribute__((always_inline)) {
        int a = e & 15, pl = (e >> 4) & 3;
        DenseNode<G> *pn = pool + (e >> 6);
        int n = pn->N[a] + 1, all = pn->all + 1;
        float w = pn->W[a] + ((pl == prev) ? v01 : vflip);
        pn->N[a] = n;
        pn->W[a] = w;
        pn->Q[a] = __fdiv_rn(w, (float)n);
        pn->all = all;
        pn->sq = __dsqrt_rn(1.0 + (double)all);
    };
    for (int k = lane; k < plen; k += S) edge(path[k]);
    if (d.track_anc) { // the reference's recursion does not stop at the current root: the edges above it (MCTS.py:252-258)
        const uint32_t *anc = d.anc + (size_t)g * (d.max_plies + 2);
        const int na = d.anc_len[g];
        for (int k = lane; k < na; k += S) edge(anc[k]);
        if (lane == 0) d.top_N[g] += 1;
    }
    if (lane == 0) {
        d.root_N[g] += 1;
        int pp = d.root_pp[g];
        if (pp) d.root_W[g] += (pp == prev) ? v01 : vflip;
    }
}

// ---- phase A: apply the evaluator's answer for the pending leaf (expand + backup) -------------
template <class G>
__device__ __forceinline__ void phase_apply(const TreeDev &d, int g, int lane) {
    using Node = DenseNode<G>;
    constexpr int S = G::S, A = G::A;
    int leaf = d.pend_leaf[g];
    if (leaf < 0) return;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    Node *node = pool + leaf;
    typename G::State st = ((const typename G::State *)d.leaf_state)[g]; // posted by phase_select
    float v = d.eval_value[g];
    if (d.pend_expand[g]) { // AddChildren with evaluator priors
        uint32_t mask = G::legal_mask(st);
        double x = 0.0;
        if (lane < A) x = (double)d.eval_policy[(size_t)g * S + lane] * (double)((mask >> lane) & 1u);
        double tot = grp_np_sum<A, S>(x);
        node->N[lane] = 0;
        node->Q[lane] = 0.f;
        node->W[lane] = 0.f;
        node->child[lane] = CHILD_NONE;
        node->cP[lane] = (lane < A) ? d.c_puct * __ddiv_rn(x, tot) : 0.0;
        if (lane == 0) {
            node->flags |= NODE_EXPANDED;
            node->legal_mask = mask;
            node->all = 0;
            node->sq = 1.0;
        }
    }
    int player = gs_player(st), prev = gs_prev(st);
    float v01;
    if (d.evaluator == 2) {
        v01 = v; // rollout kernel already returns the value for `prev`
    } else {
        v01 = (v + 1.0f) * 0.5f; // Model.SampleValue: float32 arithmetic
        if (player != prev) v01 = 1.0f - v01;
    }
    if (d.evaluator != 2 && !d.pend_expand[g] && (node->flags & NODE_TERMINAL) && lane == 0) { // remember a terminal leaf's value
        node->pad0 = (double)v01;
        node->flags |= NODE_CACHED;
    }
    backup_path<G>(d, g, lane, pool, d.path_len[g], v01, prev);
    if (lane == 0) d.pend_leaf[g] = -1;
}

// The same backup from what the descent recorded of every edge (TreeDev::path_N / path_W / path_all): stores only.  Nothing else
// touches a game's tree between the descent of a simulation and its backup, so the recorded numbers ARE the memory's.
template <class G>
__device__ __forceinline__ void backup_path_rec(const TreeDev &d, int g, int lane, DenseNode<G> *pool, int plen, float v01, int prev) {
    constexpr int S = G::S;
    const float vflip = 1.0f - v01;
    const uint32_t *path = d.path + (size_t)g * G::MAXPATH;
    const int32_t *pN = d.path_N + (size_t)g * G::MAXPATH, *pA = d.path_all + (size_t)g * G::MAXPATH;
    const float *pW = d.path_W + (size_t)g * G::MAXPATH;
    for (int k = lane; k < plen; k += S) {
        const uint32_t e = path[k];
        const int a = e & 15, pl = (e >> 4) & 3;
        DenseNode<G> *pn = pool + (e >> 6);
        const int n = pN[k] + 1, all = pA[k] + 1;
        const float w = pW[k] + ((pl == prev) ? v01 : vflip);
        pn->N[a] = n;
        pn->W[a] = w;
        pn->Q[a] = __fdiv_rn(w, (float)n);
        pn->all = all;
        pn->sq = __dsqrt_rn(1.0 + (double)all);
    }
    if (lane == 0) {
        d.root_N[g] += 1;
        int pp = d.root_pp[g];
        if (pp) d.root_W[g] += (pp == prev) ? v01 : vflip;
    }
}

// phase_apply for a leaf posted with its flags word and path statistics recorded (LEAF_RECORDED): no loads from the tree at all.
template <class G>
__device__ __forceinline__ void phase_apply_rec(const TreeDev &d, int g, int lane) {
    using Node = DenseNode<G>;
    constexpr int S = G::S, A = G::A;
    const int leaf = d.pend_leaf[g];
    if (leaf < 0) return;
    const int lf = d.leaf_flags[g] & ~LEAF_RECORDED;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    Node *node = pool + leaf;
    const typename G::State st = ((const typename G::State *)d.leaf_state)[g];
    const float v = d.eval_value[g];
    const int expand = d.pend_expand[g];
    if (expand) { // AddChildren with evaluator priors
        const uint32_t mask = G::legal_mask(st);
        double x = 0.0;
        if (lane < A) x = (double)d.eval_policy[(size_t)g * S + lane] * (double)((mask >> lane) & 1u);
        const double tot = grp_np_sum<A, S>(x);
        node->N[lane] = 0;
        node->Q[lane] = 0.f;
        node->W[lane] = 0.f;
        node->child[lane] = CHILD_NONE;
        node->cP[lane] = (lane < A) ? d.c_puct * __ddiv_rn(x, tot) : 0.0;
        if (lane == 0) {
            node->flags = lf | NODE_EXPANDED;
            node->legal_mask = mask;
            node->all = 0;
            node->sq = 1.0;
        }
    }
    const int player = gs_player(st), prev = gs_prev(st);
    float v01 = (v + 1.0f) * 0.5f; // Model.SampleValue: float32 arithmetic
    if (player != prev) v01 = 1.0f - v01;
    if (!expand && (lf & NODE_TERMINAL) && lane == 0) { // remember a terminal leaf's value
        node->pad0 = (double)v01;
        node->flags = lf | NODE_CACHED;
    }
    backup_path_rec<G>(d, g, lane, pool, d.path_len[g], v01, prev);
    if (lane == 0) d.pend_leaf[g] = -1;
}

// allocate + initialise a child node (lane 0 writes).  nn = the slot's allocation cursor (register copy).
// Returns the child word (index | TERM bit) or CHILD_NONE when the pool is exhausted.
template <class G>
__device__ __forceinline__ int create_child(const TreeDev &d, int g, DenseNode<G> *pool, DenseNode<G> *parent,
                                            const typename G::State &pst, int a, int lane, int &nn,
                                            typename G::State &st2, bool &terminal, int *flags_out = nullptr) {
    int idx = nn;
    st2 = pst;
    G::apply(st2, a);
    int w = G::winner(st2, a);
    terminal = w >= 0;
    if (idx >= d.node_cap) return CHILD_NONE;
    int word = idx | (terminal ? CHILD_TERM_BIT : 0);
    nn = idx + 1;
    const int cflags = (terminal ? (NODE_TERMINAL | ((w + 1) << 8)) : 0) | (gs_player(st2) << 4);
    if (flags_out) *flags_out = cflags;
    if (lane == 0) {
        DenseNode<G> *c = pool + idx;
        c->st = st2;
        c->flags = cflags;
        c->legal_mask = 0;
        c->all = 0;
        c->serial = idx;
        parent->child[a] = word;
        d.n_nodes[g] = idx + 1;
        d.ctr[(size_t)g * 8 + 2] += 1;
    }
    return word;
}

// MCTS.GetPriors default: ones * LegalActions (MCTS.py:39,346-358).  Returns the legal mask.
template <class G>
__device__ __forceinline__ uint32_t expand_ones(const TreeDev &d, DenseNode<G> *node, const typename G::State &st,
                                                int lane, double &cPi) {
    uint32_t mask = G::legal_mask(st);
    cPi = (lane < G::A && ((mask >> lane) & 1u)) ? d.c_puct * 1.0 : 0.0;
    node->N[lane] = 0;
    node->Q[lane] = 0.f;
    node->W[lane] = 0.f;
    node->child[lane] = CHILD_NONE;
    node->cP[lane] = cPi;
    if (lane == 0) {
        node->flags |= NODE_EXPANDED;
        node->legal_mask = mask;
        node->all = 0;
        node->sq = 1.0;
    }
    return mask;
}

// ---- phase B: one _findLeaf descent; leaves the leaf in the evaluator mailbox ------------------
template <class G>
__device__ void phase_select(const TreeDev &d, int g, int lane) {
    using Node = DenseNode<G>;
    constexpr int S = G::S, A = G::A;
    if (d.game_lid[g] < 0 || d.sims_left[g] <= 0) return;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    uint32_t *path = d.path + (size_t)g * G::MAXPATH;
    int cur = d.root[g];
    int nn = d.n_nodes[g];
    int depth = 0, expand = 0, overflow = 0, term_leaf = 0;
    const bool inline_expand = d.priors_ones != 0;
    const bool fixed = d.kind == 1, rollout = d.evaluator == 2;
    typename G::State st;
    int flags = 0;
    bool have = false; // st/flags of `cur` already in registers (node created this simulation)
    int pf0 = 0, pf1 = 0, spec = 0; // speculative touches of the children's rows (see below)
#ifdef BB_STAMPS
    long long acc_load = 0, acc_iter = 0;
#endif
    for (int it = 0;; it++) {
#ifdef BB_STAMPS
        long long ts0 = clock64();
        acc_iter++;
#endif
        Node *node = pool + cur;
        // ONE round trip per level: issue every load of the row together, then wait once.
        typename G::State st_l = node->st;
        int flags_l = node->flags;
        uint32_t mask = node->legal_mask;
        double sq = node->sq;
        int Ni = node->N[lane];
        float Qi = node->Q[lane];
        float Wi = rollout ? node->W[lane] : 0.f;
        double cPi = node->cP[lane];
        int ci = node->child[lane];
        spec ^= pf0 ^ pf1; // previous level's touches are older than the loads above: no extra stall
        asm volatile("" ::"v"(flags_l), "v"(mask), "v"(sq), "v"(Ni), "v"(Qi), "v"(cPi), "v"(ci)); // keep them ahead of the branches
#ifdef BB_STAMPS
        acc_load += clock64() - ts0;
#endif
        if (!have) {
            st = st_l;
            flags = flags_l;
        }
        have = false;
        if (fixed && it >= d.max_depth) break; // FixedMCTS: range(MaxDepth) exhausted
        if (!(flags & NODE_EXPANDED)) {
            if (flags & NODE_TERMINAL) { term_leaf = 1; break; } // Winner(lastAction) is not None
            if (!inline_expand) { expand = 1; break; }            // AddChildren once the evaluator has answered
            mask = expand_ones<G>(d, node, st, lane, cPi);
            if (!fixed) break; // DynamicMCTS: AddChildren(node); break
            Ni = 0;
            Qi = 0.f;
            Wi = 0.f;
            sq = 1.0;
            ci = CHILD_NONE;
        }
        if (mask == 0) break; // np.sum(LegalActions) == 0
        // While the PUCT arithmetic runs, lane i pulls its own child's row towards this CU, so the
        // next level's (dependent) row load hits cache instead of paying an HBM round trip.
        pf0 = 0;
        pf1 = 0;
        if (BB_PREFETCH_CHILDREN && lane < A && ci >= 0 && !(ci & CHILD_TERM_BIT)) {
            const int *pc = (const int *)(pool + ci);
            pf0 = pc[0];
            pf1 = pc[32];
        }
        double u = puct_score(child_q(d, Qi, Wi, Ni), cPi, sq, Ni, lane < A && ((mask >> lane) & 1u));
        int child = ci;
        int a = grp_argmax<S>(u, lane, child);
        if (depth >= G::MAXPATH) { overflow = 1; break; }
        if (child == CHILD_NONE) {
            typename G::State st2;
            bool terminal;
            child = create_child<G>(d, g, pool, node, st, a, lane, nn, st2, terminal);
            if (child == CHILD_NONE) { overflow = 1; break; }
            if (lane == 0) path[depth] = ((uint32_t)cur << 6) | ((uint32_t)((flags >> 4) & 3) << 4) | (uint32_t)a;
            st = st2;
            flags = (terminal ? NODE_TERMINAL : 0) | (gs_player(st2) << 4);
            have = true;
        } else {
            if (lane == 0) path[depth] = ((uint32_t)cur << 6) | ((uint32_t)((flags >> 4) & 3) << 4) | (uint32_t)a;
        }
        depth++;
        cur = child & ~CHILD_TERM_BIT;
    }
#ifdef BB_STAMPS
    if (d.stamps && lane == 0) {
        atomicAdd(&d.stamps[5], (unsigned long long)acc_load);
        atomicAdd(&d.stamps[6], (unsigned long long)acc_iter);
        atomicAdd(&d.stamps[7], 1ull);
    }
#endif
    spec ^= pf0 ^ pf1;
    if (spec == 0x5bd1e995 && depth == -7) d.ctr[(size_t)g * 8 + 6] += 1; // never true: keeps the touches alive
    if (lane == 0) {
        ((typename G::State *)d.leaf_state)[g] = st;
        d.leaf_game_id[g] = d.first_game_id + (uint32_t)d.game_lid[g];
        d.leaf_serial[g] = cur;
        d.pend_leaf[g] = cur;
        if (d.leaf_flags) d.leaf_flags[g] = 0;
        d.pend_expand[g] = expand;
        d.path_len[g] = depth;
        d.sims_left[g] -= 1;
        d.sim_serial[g] += 1;
        d.evals[g] += 1;
        uint64_t *c = d.ctr + (size_t)g * 8;
        c[0] += 1;
        c[1] += (uint64_t)depth;
        c[3] += (uint64_t)term_leaf;
        c[6] += (uint64_t)overflow;
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_tree_step(TreeDev d) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int wv = t >> 6, l64 = t & 63;
    if (l64 >= d.gpw * S) return;
    int g = wv * d.gpw + l64 / S, lane = l64 % S;
    if (g >= d.n_slots) return;
#ifdef BB_STAMPS
    long long t0 = clock64();
#endif
    phase_apply<G>(d, g, lane);
#ifdef BB_STAMPS
    long long t1 = clock64();
#endif
    __threadfence_block();
#ifdef BB_STAMPS
    long long t2 = clock64();
#endif
    phase_select<G>(d, g, lane);
#ifdef BB_STAMPS
    long long t3 = clock64();
    if ((threadIdx.x & 63) == 0 && d.stamps) {
        atomicAdd(&d.stamps[0], (unsigned long long)(t1 - t0));
        atomicAdd(&d.stamps[1], (unsigned long long)(t2 - t1));
        atomicAdd(&d.stamps[2], (unsigned long long)(t3 - t2));
        atomicAdd(&d.stamps[4], 1ull);
    }
#endif
}

template <class G>
__global__ void __launch_bounds__(256) k_tree_apply(TreeDev d) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = t / S, lane = t % S;
    if (g >= d.n_slots) return;
    phase_apply<G>(d, g, lane);
}

// ---- root statistics + move choice -------------------------------------------------------------
// returns the chosen action (same on every lane) or -4 (NaN probabilities) / -3 (no tree)
template <class G>
__device__ __forceinline__ int choose_move(const TreeDev &d, int g, int lane, double temp, double u, int &total, int &Ni_out,
                           float &Wi_out) {
    using Node = DenseNode<G>;
    constexpr int S = G::S, A = G::A;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    Node *node = pool + d.root[g];
    if (!(node->flags & NODE_EXPANDED)) {
        total = 0;
        Ni_out = 0;
        Wi_out = 0.f;
        return -3;
    }
    uint32_t mask = node->legal_mask;
    bool legal = lane < A && ((mask >> lane) & 1u);
    int Ni = legal ? node->N[lane] : 0;
    float Wi = legal ? node->W[lane] : 0.f;
    Ni_out = Ni;
    Wi_out = Wi;
    int all = grp_sum_i<S>(Ni);
    total = all;
    if (temp == 0.0) { // `exploring or temp == 0` -> PUCT argmax (MCTS.py:327-334)
        double uu = puct_score(child_q(d, node->Q[lane], Wi, Ni), node->cP[lane], node->sq, Ni, legal);
        int dummy = 0;
        return grp_argmax<S>(uu, lane, dummy);
    }
    // p_i = N_i^(1/temp) / sum; np.random.choice: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right')
    double it = 1.0 / temp;
    double w = (it == 1.0) ? (double)Ni : pow((double)Ni, it);
    double ws[A];
#pragma unroll
    for (int k = 0; k < A; k++) ws[k] = __shfl(w, k, S);
    double allp = 0.0;
#pragma unroll
    for (int k = 0; k < A; k++) allp += ws[k];
    if (!(allp > 0.0)) return -4;
    double last = 0.0;
#pragma unroll
    for (int k = 0; k < A; k++) last += __ddiv_rn(ws[k], allp);
    double run = 0.0;
    int act = A - 1;
    bool found = false;
#pragma unroll
    for (int k = 0; k < A; k++) {
        run += __ddiv_rn(ws[k], allp);
        if (!found && __ddiv_rn(run, last) > u) {
            act = k;
            found = true;
        }
    }
    return act;
}

template <class G>
__global__ void __launch_bounds__(256) k_sample(TreeDev d, double temp) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = t / S, lane = t % S;
    if (g >= d.n_slots) return;
    if (d.game_lid[g] < 0) {
        if (lane == 0 && d.out_action) d.out_action[g] = -3;
        return;
    }
    double u = d.in_u ? d.in_u[g] : bb_u53(d.seed, d.first_game_id + (uint32_t)d.game_lid[g], (uint32_t)d.ply[g]);
    int total, Ni;
    float Wi;
    int act = choose_move<G>(d, g, lane, temp, u, total, Ni, Wi);
    if (d.out_child_plays) d.out_child_plays[(size_t)g * S + lane] = Ni;
    if (d.out_child_value) d.out_child_value[(size_t)g * S + lane] = Wi;
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

// _moveRoot by action.  Lane-0 state updates; all lanes of the group call it.
template <class G>
__device__ __forceinline__ void advance_root(const TreeDev &d, int g, int lane, int a, typename G::State &new_st) {
    using Node = DenseNode<G>;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    Node *node = pool + d.root[g];
    typename G::State st = node->st;
    if (!(node->flags & NODE_EXPANDED)) { // `Root.Children is None` -> Root = None; re-prime with the new state
        new_st = st;
        G::apply(new_st, a);
        if (lane == 0) {
            Node *r = pool;
            r->st = new_st;
            r->flags = gs_player(new_st) << 4;
            r->serial = 0;
            d.n_nodes[g] = 1;
            d.root[g] = 0;
            d.root_N[g] = 0;
            d.root_W[g] = 0.f;
            d.root_pp[g] = 0;
            if (d.track_anc) { // a new root without a parent
                d.anc_len[g] = 0;
                d.top_N[g] = 0;
            }
        }
        return;
    }
    int child = node->child[a];
    int cn = node->N[a];
    float cw = node->W[a];
    if (child == CHILD_NONE) {
        bool terminal;
        int nn = d.n_nodes[g];
        child = create_child<G>(d, g, pool, node, st, a, lane, nn, new_st, terminal);
        if (child == CHILD_NONE) { // pool exhausted: restart the tree at the new state
            child = 0;
            cn = 0;
            cw = 0.f;
            if (lane == 0) {
                pool->st = new_st;
                pool->flags = gs_player(new_st) << 4;
                d.n_nodes[g] = 1;
                d.ctr[(size_t)g * 8 + 6] += 1;
            }
        }
    } else {
        new_st = pool[child & ~CHILD_TERM_BIT].st;
    }
    if (lane == 0) {
        if (d.track_anc) {
            if (child == 0 && cn == 0 && d.n_nodes[g] == 1) { // (the pool was exhausted above: the tree restarted)
                d.anc_len[g] = 0;
                d.top_N[g] = 0;
            } else if (d.anc_len[g] < d.max_plies + 2) {
                d.anc[(size_t)g * (d.max_plies + 2) + d.anc_len[g]] = ((uint32_t)d.root[g] << 6) | ((uint32_t)gs_player(st) << 4) | (uint32_t)a;
                d.anc_len[g] += 1;
            }
        }
        d.root[g] = child & ~CHILD_TERM_BIT;
        d.root_N[g] = cn;
        d.root_W[g] = cw;
        d.root_pp[g] = (int8_t)gs_player(st);
    }
}

// MCTS.ResetRoot (MCTS.py:214-225): the root goes back to its top-most ancestor; nothing is forgotten.
template <class G>
__global__ void __launch_bounds__(256) k_reset_roots(TreeDev d) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots || !d.track_anc || d.game_lid[g] < 0 || d.anc_len[g] <= 0) return;
    d.root[g] = (int)(d.anc[(size_t)g * (d.max_plies + 2)] >> 6);
    d.root_N[g] = d.top_N[g];
    d.root_W[g] = 0.f;
    d.root_pp[g] = 0;
    d.anc_len[g] = 0;
}

// One node of a slot's tree for the host's Node view (MCTS.py:7-98): node < 0 = the root.  Per child slot i: the child's pool
// index (CHILD_NONE where AddChildren would have made a Node that no simulation has reached yet), Plays, Value; the node's state,
// flags and legal mask.
template <class G>
__global__ void __launch_bounds__(64) k_node_view(TreeDev d, int g, int node, int32_t *child_out, int32_t *plays_out, float *value_out,
                                                   typename G::State *state_out, int32_t *info_out) {
    using Node = DenseNode<G>;
    const int lane = threadIdx.x;
    if (lane >= G::S) return;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    if (node < 0) node = d.root[g];
    const Node *n = pool + node;
    const bool exp = (n->flags & NODE_EXPANDED) != 0;
    child_out[lane] = exp ? n->child[lane] : CHILD_NONE;
    plays_out[lane] = exp ? n->N[lane] : 0;
    value_out[lane] = exp ? n->W[lane] : 0.f;
    if (lane == 0) {
        *state_out = n->st;
        info_out[0] = n->flags;
        info_out[1] = (int32_t)(exp ? n->legal_mask : 0u);
        info_out[2] = node;
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_move_roots(TreeDev d, const int32_t *actions) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = t / S, lane = t % S;
    if (g >= d.n_slots) return;
    int a = actions[g];
    if (a < 0 || a >= G::A || d.game_lid[g] < 0) return;
    typename G::State ns;
    advance_root<G>(d, g, lane, a, ns);
    if (lane == 0) d.ply[g] += 1;
}

template <class G>
__device__ __forceinline__ void reset_slot(const TreeDev &d, int g, int lid, const typename G::State &st) {
    using Node = DenseNode<G>;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    pool->st = st;
    pool->flags = gs_player(st) << 4;
    pool->legal_mask = 0;
    pool->all = 0;
    pool->serial = 0;
    d.n_nodes[g] = 1;
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
    if (d.leaf_flags) d.leaf_flags[g] = 0;
    if (d.track_anc) {
        d.anc_len[g] = 0;
        d.top_N[g] = 0;
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_set_roots(TreeDev d, int n, const int32_t *slots,
                                                   const typename G::State *states, const uint32_t *game_ids) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int g = slots ? slots[i] : i;
    if (g < 0 || g >= d.n_slots) return;
    reset_slot<G>(d, g, game_ids ? (int)game_ids[i] : g, states[i]);
    d.sims_left[g] = 0;
}

template <class G>
__global__ void __launch_bounds__(256) k_add_sims(TreeDev d, int sims, const uint8_t *mask = nullptr) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    if (d.game_lid[g] >= 0 && (!mask || mask[g])) d.sims_left[g] = sims; // (slots outside the mask keep their trees untouched)
}

template <class G>
__global__ void __launch_bounds__(256) k_get_roots(TreeDev d, typename G::State *out) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    out[g] = ((DenseNode<G> *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap + d.root[g])->st;
}

// ---- self-play: begin / one move for every slot --------------------------------------------------
template <class G>
__global__ void __launch_bounds__(256) k_selfplay_begin(TreeDev d) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.n_slots) return;
    if (g + d.slot_offset < d.n_games_target) {
        reset_slot<G>(d, g, g + d.slot_offset, G::initial());
        d.sims_left[g] = d.sims_per_move;
    } else {
        reset_slot<G>(d, g, -1, G::initial());
        d.sims_left[g] = 0;
    }
}

template <class G>
__device__ __forceinline__ uint8_t *example_ptr(const TreeDev &d, int lid, int ply) {
    return d.examples + ((size_t)lid * (d.max_plies + 1) + ply) * d.example_bytes;
}

template <class G>
__device__ __forceinline__ void write_example(const TreeDev &d, int lid, int ply, const typename G::State &st, int lane, int Ni,
                              int total, uint32_t gid) {
    uint8_t *p = example_ptr<G>(d, lid, ply);
    if (lane == 0) {
        ExampleHdr h;
        h.game_id = gid;
        h.ply = (uint16_t)ply;
        h.player = (uint8_t)gs_player(st);
        h.z = 0;
        h.total = (uint32_t)total;
        h.n_children = G::A;
        *(ExampleHdr *)p = h;
        *(typename G::State *)(p + sizeof(ExampleHdr)) = st;
    }
    ((uint32_t *)(p + sizeof(ExampleHdr) + sizeof(typename G::State)))[lane] = (uint32_t)Ni;
}

// apply the last pending leaf, then FindMove's tail + the body of GenerateTrainingSamples' while loop
// FindMove's tail + the body of GenerateTrainingSamples' while loop for one slot (all lanes of the group call)
template <class G>
__device__ BB_MOVE_BODY_ATTR void selfplay_move_body(const TreeDev &d, int g, int lane) {
    using Node = DenseNode<G>;
    constexpr int S = G::S;
    int lid = d.game_lid[g];
    if (lid < 0) return;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
    typename G::State st = pool[d.root[g]].st;
    uint32_t gid = d.first_game_id + (uint32_t)lid;
    int ply = d.ply[g];
    double u = bb_u53(d.seed, gid, (uint32_t)ply);
    int total, Ni;
    float Wi;
    int act = choose_move<G>(d, g, lane, d.temp, u, total, Ni, Wi);
    if (act < 0) { // cannot happen with sims_per_move >= 2; park the slot
        if (lane == 0) {
            d.game_lid[g] = -1;
            d.ctr[(size_t)g * 8 + 6] += 1;
        }
        return;
    }
    write_example<G>(d, lid, ply, st, lane, Ni, total, gid); // (s, pi, player) of the state before the move
    typename G::State ns;
    advance_root<G>(d, g, lane, act, ns);
    __threadfence_block();
    ply += 1;
    int w = G::winner(ns, -1); // state.Winner(lastAction=None): full scan (Blackbird.py:253)
    bool over = w >= 0 || ply >= d.max_plies;
    if (!over) {
        if (lane == 0) {
            d.ply[g] = ply;
            d.sims_left[g] = d.sims_per_move;
            d.ctr[(size_t)g * 8 + 5] += 1;
        }
        return;
    }
    // terminal example with pi = zeros (Blackbird.py:256-258), then z for every example (:260-264)
    write_example<G>(d, lid, ply, ns, lane, 0, 0, gid);
    __threadfence_block();
    for (int k = lane; k <= ply; k += S) {
        ExampleHdr *h = (ExampleHdr *)example_ptr<G>(d, lid, k);
        h->z = (w <= 0) ? 0 : (h->player == w ? 1 : -1);
    }
    __threadfence(); // the game's records (every lane's z above) are visible device-wide before its `done` word: a host copy or a
                     // collective may read finished games while other games are still being played
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
        int next = lid + d.lid_stride; // this slot's next game id
        if (next < d.n_games_target) {
            reset_slot<G>(d, g, next, G::initial());
            d.sims_left[g] = d.sims_per_move;
        } else {
            d.game_lid[g] = -1;
            d.sims_left[g] = 0;
        }
    }
}

template <class G>
__global__ void __launch_bounds__(256) k_selfplay_move(TreeDev d) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int g = t / S, lane = t % S;
    if (g >= d.n_slots) return;
    phase_apply<G>(d, g, lane);
    __threadfence_block();
    selfplay_move_body<G>(d, g, lane);
}

// ---- asynchronous self-play round ------------------------------------------------------------------------
// Games no longer move in lock-step.  One launch gives every game a budget of tree levels:
//   apply the evaluated leaf (if any) -> [800 simulations done? sample + move + maybe next game] ->
//   descend; a terminal leaf whose value is already known (the reference's lru_cache on SampleValue,
//   Blackbird.py:350) is backed up on the spot and the next simulation starts; the first leaf that needs
//   the evaluator is posted to the compacted leaf list and the game waits for the network kernel;
//   a descent that outlives the budget is parked (resume_cur/resume_depth) and continues next launch.
// Per game the sequence of simulations is exactly the sequential one, so results do not depend on the
// schedule (tests/test_gpu_mcts.py: self-play == oracle, example by example).
// REC (persistent kernel with the recorded-path arrays in LDS): the descent notes N, W and sum(N) of every edge it takes and the
// backups are stores only (backup_path_rec / phase_apply_rec).
// on_post(g, lane): called by every lane of the game's group right after the leaf's mailbox has been written -- the persistent
// kernel hands the leaf to its network waves THERE (mega2.hip.h), not when the call returns: a wave's call lasts as long as
// the slowest of its games' descents.
struct NoPostHook {
    __device__ __forceinline__ void operator()(int, int) const {}
};
template <class G, bool REC = false, class OnPost = NoPostHook>
__device__ bool async_game(const TreeDev &d, int g, int lane
#ifdef BB_STAMPS_LIGHT
                           , int &g_light_loop, int &g_light_levels, int &g_light_load, int &g_light_puct
#endif
                           , OnPost on_post = OnPost()) {
    using Node = DenseNode<G>;
    constexpr int S = G::S, A = G::A;
    if (d.game_lid[g] < 0) return false;
    Node *pool = (Node *)d.nodes + (size_t)(g + d.pool_g0) * d.node_cap;
#ifdef BB_STAMPS_DEEP
    long long sa0 = clock64(), s_apply = 0, s_level = 0, s_backup = 0, s_move = 0;
#endif
#ifdef BB_STAMPS_LIGHT
    int lt_in = (int)clock64(), lt_loop = 0, ln_levels = 0, lt_create = 0, lt_puct = 0;
#endif
    if (d.pend_leaf[g] >= 0) {
        if (REC && (d.leaf_flags[g] & LEAF_RECORDED)) phase_apply_rec<G>(d, g, lane);
        else phase_apply<G>(d, g, lane);
        if (lane == 0) d.sims_left[g] -= 1;
        __threadfence_block();
    }
#ifdef BB_STAMPS_DEEP
    s_apply = clock64() - sa0;
#endif
    int budget = d.level_budget;
    uint32_t *path = d.path + (size_t)g * G::MAXPATH;
    int sims_done = 0, depth_sum = 0, term_hits = 0;
    bool posted = false;
#ifdef BB_STAMPS_DEEP
    long long st_load = 0, st_levels = 0, st_t0 = clock64();
#endif
    for (;;) {
#ifdef BB_STAMPS_DEEP
        long long sm0 = clock64();
#endif
        if (d.sims_left[g] <= 0) { // MCTS.FindMove's tail, GenerateTrainingSamples' loop body
            selfplay_move_body<G>(d, g, lane);
            __threadfence_block();
            if (d.game_lid[g] < 0) break;
        }
#ifdef BB_STAMPS_DEEP
        s_move += clock64() - sm0;
#endif
        int cur = d.resume_cur[g], depth = 0;
        if (cur >= 0) depth = d.resume_depth[g];
        else cur = d.root[g];
        int nn = d.n_nodes[g];
        typename G::State st;
        int flags = 0;
        // Outcome bits of the descent, one integer per lane on purpose: as separate bools they live in scalar lane masks and
        // every exit of the loop below pays a three-instruction merge for each of them on every iteration
        enum { F_PARKED = 2, F_LEAF = 4, F_TERM = 8, F_EXPAND = 16, F_OVERFLOW = 32 };
        int fl = 0;
        int pf_touch = 0; // speculative touch of the likeliest child's row (BB_PREFETCH_BEST)
#ifdef BB_STAMPS_LIGHT
        int lt_e = (int)clock64();
#endif
        for (;;) {
            if (budget <= 0) { fl |= F_PARKED; break; }
            budget--;
#ifdef BB_STAMPS_LIGHT
            ln_levels++;
#ifndef BB_STAMPS_LIGHT2
            int lt_a = (int)clock64();
#endif
#endif
#ifdef BB_STAMPS_DEEP
            st_levels++;
#ifdef BB_STAMPS_PERLEVEL
            long long ts0 = clock64();
#endif
#endif
            Node *node = pool + cur;
            typename G::State st_l = node->st;
            int flags_l = node->flags;
            uint32_t mask = node->legal_mask;
            double sq = node->sq;
            int Ni = node->N[lane];
            float Qi = node->Q[lane];
            double cPi = node->cP[lane];
            int ci = node->child[lane];
            double cached = node->pad0;
            float Wi = 0.f;
            int all_l = 0;
            if (REC) {
                Wi = node->W[lane];
                all_l = node->all;
            }
            asm volatile("" ::"v"(flags_l), "v"(mask), "v"(sq), "v"(Ni), "v"(Qi), "v"(cPi), "v"(ci), "v"(pf_touch)); // (the previous level's touch is older than these loads)
#if defined(BB_STAMPS_DEEP) && defined(BB_STAMPS_PERLEVEL)
            st_load += clock64() - ts0;
#endif
#if defined(BB_STAMPS_LIGHT) && !defined(BB_STAMPS_LIGHT2)
            int lt_b = (int)clock64(); // the row has arrived (the asm above made the loads' results live)
            lt_create += lt_b - lt_a;
#endif
            st = st_l;
            flags = flags_l;
            if (!(flags & NODE_EXPANDED)) {
                fl |= F_LEAF;
                if (flags & NODE_TERMINAL) {
                    fl |= F_TERM;
                    if (flags & NODE_CACHED) { // value already known: finish this simulation here
#ifdef BB_STAMPS_DEEP
                        long long sb0 = clock64();
#endif
                        float v01 = (float)cached;
#ifdef BB_STAMPS_LIGHT2
                        int lt_k = (int)clock64();
#endif
                        __threadfence_block(); // path stores of this descent
                        if (REC) backup_path_rec<G>(d, g, lane, pool, depth, v01, gs_prev(st));
                        else backup_path<G>(d, g, lane, pool, depth, v01, gs_prev(st));
                        if (lane == 0) d.sims_left[g] -= 1;
                        __threadfence_block();
#ifdef BB_STAMPS_LIGHT2
                        lt_puct += (int)clock64() - lt_k;
#endif
                        sims_done++;
                        depth_sum += depth;
                        term_hits++;
#ifdef BB_STAMPS_DEEP
                        s_backup += clock64() - sb0;
#endif
                        fl &= ~F_LEAF; // nothing to post; start the next simulation
                    }
                } else {
                    fl |= F_EXPAND;
                }
                break;
            }
            // (DynamicMCTS.py:29 also stops at an expanded node without legal actions.  On these boards a node is expanded
            // only if it is not terminal, and a non-terminal board has an empty cell, so the case cannot arise; likewise a
            // path cannot outgrow MAXPATH = H*W + 2 edges, one move each.  Two loop exits less in the hottest loop.)
            static_assert(G::MAXPATH >= G::H * G::W + 2, "a descent places at most H*W stones");
#if BB_PREFETCH_BEST
            { // the most visited child is the likeliest next step: pull its row towards this CU while the PUCT arithmetic runs
                int key = (ci >= 0 && lane < A) ? ((Ni << 4) | lane) : -1;
                int best = key;
                best = max(best, dpp_step_i<0>(best));
                best = max(best, dpp_step_i<1>(best));
                best = max(best, dpp_step_i<2>(best));
                if (S == 16) best = max(best, dpp_step_i<3>(best));
                int bchild = __shfl(ci, best & 15, S);
                if (best >= 0) pf_touch = ((const int *)(pool + (bchild & ~CHILD_TERM_BIT)))[lane * 8];
            }
#endif
            // (float32 evaluators only reach this kernel: a child's WinRate is its float32 Q, child_q)
            int child = ci;
            int a = grp_argmax_puct<S>(Qi, cPi, sq, Ni, lane < A && ((mask >> lane) & 1u), child);
#if defined(BB_STAMPS_LIGHT) && !defined(BB_STAMPS_LIGHT2)
            asm volatile("" ::"v"(a), "v"(child));
            lt_puct += (int)clock64() - lt_b;
#endif
            if (lane == 0) path[depth] = ((uint32_t)cur << 6) | ((uint32_t)((flags >> 4) & 3) << 4) | (uint32_t)a;
            if (REC) { // what the backup of this edge will start from
                if (lane == 0) d.path_all[(size_t)g * G::MAXPATH + depth] = all_l;
                if (lane == a) {
                    d.path_N[(size_t)g * G::MAXPATH + depth] = Ni;
                    d.path_W[(size_t)g * G::MAXPATH + depth] = Wi;
                }
            }
            if (child == CHILD_NONE) {
                typename G::State st2;
                bool terminal;
#ifdef BB_STAMPS_LIGHT2
                int lt_c = (int)clock64();
#endif
                child = create_child<G>(d, g, pool, node, st, a, lane, nn, st2, terminal, &flags);
#ifdef BB_STAMPS_LIGHT2
                asm volatile("" ::"v"(child));
                lt_create += (int)clock64() - lt_c;
#endif
                if (child == CHILD_NONE) { fl |= F_OVERFLOW | F_LEAF; break; }
                // the node just created is the leaf of this descent (never expanded, never cached): finish here instead of
                // going round the loop once more to read back the row that was written a moment ago
                st = st2; // (flags = the word create_child stored for the new node)
                depth++;
                cur = child & ~CHILD_TERM_BIT;
                fl |= F_LEAF | (terminal ? F_TERM : F_EXPAND);
                break;
            }
            depth++;
            cur = child & ~CHILD_TERM_BIT;
        }
#ifdef BB_STAMPS_LIGHT
        lt_loop += (int)clock64() - lt_e;
#endif
        if (lane == 0) {
            d.resume_cur[g] = (fl & F_PARKED) ? cur : -1;
            d.resume_depth[g] = (fl & F_PARKED) ? depth : 0;
        }
        if (fl & F_PARKED) break;
        if (!(fl & F_LEAF)) {           // cached terminal: loop on to the next simulation (or the move)
            if (lane == 0) d.sim_serial[g] += 1;
            continue;
        }
        if (lane == 0) {             // post the leaf for the evaluator
            ((typename G::State *)d.leaf_state)[g] = st;
            d.leaf_game_id[g] = d.first_game_id + (uint32_t)d.game_lid[g];
            d.leaf_serial[g] = cur;
            d.pend_leaf[g] = cur;
            if (d.leaf_flags) d.leaf_flags[g] = REC ? (flags | LEAF_RECORDED) : 0;
            d.pend_expand[g] = (fl & F_EXPAND) ? 1 : 0;
            d.path_len[g] = depth;
            d.sim_serial[g] += 1;
            d.evals[g] += 1;
            d.ctr[(size_t)g * 8 + 6] += (uint64_t)((fl & F_OVERFLOW) ? 1 : 0);
        }
        on_post(g, lane);
        sims_done++;
        depth_sum += depth;
        term_hits += (fl & F_TERM) ? 1 : 0;
        posted = true;
        break;
    }
#ifdef BB_STAMPS_DEEP
    if (lane == 0 && d.stamps) {
        atomicAdd(&d.stamps[6], (unsigned long long)st_load);
        atomicAdd(&d.stamps[7], (unsigned long long)st_levels);
        atomicAdd(&d.stamps[8], (unsigned long long)s_apply);
        atomicAdd(&d.stamps[9], (unsigned long long)s_backup);
        atomicAdd(&d.stamps[10], (unsigned long long)s_move);
        atomicAdd(&d.stamps[11], (unsigned long long)(clock64() - st_t0));
        atomicAdd(&d.stamps[12], 1ull);
    }
#endif
#ifdef BB_STAMPS_LIGHT
    g_light_loop = lt_loop;
    g_light_load = lt_create;
    g_light_puct = lt_puct;      // read back by the caller, which reduces over the wave (mega2.hip.h)
    g_light_levels = ln_levels;
    (void)lt_in;
#endif
    if (lane == 0 && sims_done) {
        uint64_t *c = d.ctr + (size_t)g * 8;
        c[0] += (uint64_t)sims_done;
        c[1] += (uint64_t)depth_sum;
        c[3] += (uint64_t)term_hits;
    }
    return posted;
}

template <class G>
__global__ void __launch_bounds__(256) k_tree_async(TreeDev d, int round) {
    constexpr int S = G::S;
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int wv = t >> 6, l64 = t & 63;
    int g = wv * d.gpw + l64 / S, lane = l64 % S;
    bool live = l64 < d.gpw * S && g < d.n_slots;
#ifdef BB_STAMPS_LIGHT
    int ll = 0, lv = 0, lld = 0, lpu = 0;
    bool posted = live ? async_game<G>(d, g, lane, ll, lv, lld, lpu) : false;
    {
        int m = lv;
        for (int o = 32; o; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
        unsigned long long who = __ballot(live && lane == 0 && lv == m);
        if (who && l64 == (int)__builtin_ctzll(who) && d.stamps) {
            atomicAdd(&d.stamps[6], (unsigned long long)ll);
            atomicAdd(&d.stamps[7], (unsigned long long)lv);
            atomicAdd(&d.stamps[11], (unsigned long long)lld);
            atomicAdd(&d.stamps[12], (unsigned long long)lpu);
        }
    }
#else
    bool posted = live ? async_game<G>(d, g, lane) : false;
#endif
    // wave-aggregated append to the round's compacted leaf list
    unsigned long long m = __ballot(posted && lane == 0);
    if (t == 0) d.post_count[(round + 2) & 3] = 0; // recycled two rounds from now
    if (m) {
        int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (l64 == leader) base = atomicAdd(&d.post_count[round & 3], __popcll(m));
        base = __shfl(base, leader, 64);
        if (posted && lane == 0) d.post_slot[base + __popcll(m & ((1ull << l64) - 1ull))] = g;
    }
}
