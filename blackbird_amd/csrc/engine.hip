// engine.hip -- host side of libblackbird_hip.so: the C ABI of include/blackbird_hip.h.
// Owns all device memory (hipMalloc), one HIP stream per engine, and the launch sequences:
//   simulation  = k_tree_step (apply previous leaf: expand+backup; select next leaf)  ->  evaluator kernel
//   move        = k_selfplay_move (apply last leaf; sample; record example; re-root; game over -> next game)
#include "../../include/blackbird_hip.h"
#include "eval.hip.h"
#include "net.hip.h"
#include "net_x3.hip.h"
#include "gnet.hip.h"
#include "gnet_x3.hip.h"
#include "tree.hip.h"
#include "tree_dc.hip.h"
#include "mega2.hip.h"
#include "mega_dc.hip.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(x)                                                                                      \
    do {                                                                                               \
        hipError_t _e = (x);                                                                           \
        if (_e != hipSuccess) return fail(BB_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), \
                                          __FILE__, __LINE__);                                         \
    } while (0)

extern "C" const char *bb_last_error(void) { return g_err.c_str(); }

extern "C" int bb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

template <class G>
static void fill_info(bb_game_info *o) {
    o->H = G::H;
    o->W = G::W;
    o->C = G::C;
    o->A = G::A;
    o->S = G::S;
    o->state_bytes = (int)sizeof(typename G::State);
    o->dense = 1;
    o->example_bytes = (int)(sizeof(ExampleHdr) + sizeof(typename G::State) + 4 * G::S);
}

extern "C" int bb_game_info_get(int game, bb_game_info *out) {
    if (!out) return fail(BB_ERR_ARG, "null out");
    switch (game) {
    case BB_GAME_CONNECT4: fill_info<Connect4>(out); return BB_OK;
    case BB_GAME_TICTACTOE: fill_info<TicTacToe>(out); return BB_OK;
    case BB_GAME_DRAGONCHESS:
        fill_info<DragonChess>(out);
        out->dense = 0;
        out->example_bytes = (int)(sizeof(ExampleHdr) + sizeof(DCState) + 4 * DragonChess::S + 2 * DragonChess::S);
        return BB_OK;
    default: return fail(BB_ERR_ARG, "unknown or unsupported game %d", game);
    }
}

// ---- device scratch with automatic release ---------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        return e == hipSuccess ? 0 : fail(BB_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
};

static inline int nblk(size_t n, int per = 256) { return (int)((n + per - 1) / per); }

// ---- stateless batched game ops ------------------------------------------------------------------
template <class G>
static int game_legal(int n, const void *states, uint8_t *out) {
    DevBuf ds, dout;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || dout.alloc((size_t)n * G::A)) return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if constexpr (G::GID == BB_GAME_DRAGONCHESS)
        k_dc_legal<<<nblk((size_t)n * 64), 256>>>(n, (const DCState *)ds.p, (uint8_t *)dout.p);
    else
        k_game_legal<G><<<nblk(n), 256>>>(n, (const typename G::State *)ds.p, (uint8_t *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout.p, (size_t)n * G::A, hipMemcpyDefault));
    return BB_OK;
}
template <class G>
static int game_apply(int n, void *states, const int32_t *actions, int32_t *status) {
    DevBuf ds, da, dst;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || da.alloc((size_t)n * 4) || dst.alloc((size_t)n * 4))
        return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    HIPCHK(hipMemcpy(da.p, actions, (size_t)n * 4, hipMemcpyDefault));
    k_game_apply<G><<<nblk(n), 256>>>(n, (typename G::State *)ds.p, (const int32_t *)da.p, (int32_t *)dst.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(states, ds.p, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if (status) HIPCHK(hipMemcpy(status, dst.p, (size_t)n * 4, hipMemcpyDefault));
    return BB_OK;
}
template <class G>
static int game_winner(int n, const void *states, const int32_t *prev, int8_t *out) {
    DevBuf ds, dp, dout;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || dp.alloc((size_t)n * 4) || dout.alloc((size_t)n))
        return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if (prev) HIPCHK(hipMemcpy(dp.p, prev, (size_t)n * 4, hipMemcpyDefault));
    k_game_winner<G><<<nblk(n), 256>>>(n, (const typename G::State *)ds.p, prev ? (const int32_t *)dp.p : nullptr,
                                       (int8_t *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout.p, (size_t)n, hipMemcpyDefault));
    return BB_OK;
}
template <class G>
static int game_encode(int n, const void *states, int8_t *out) {
    size_t ob = (size_t)n * G::H * G::W * G::C;
    DevBuf ds, dout;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || dout.alloc(ob)) return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if constexpr (G::GID == BB_GAME_DRAGONCHESS)
        k_dc_encode<<<nblk((size_t)n * 64), 256>>>(n, (const DCState *)ds.p, (int8_t *)dout.p);
    else
        k_game_encode<G><<<nblk((size_t)n * G::H * G::W), 256>>>(n, (const typename G::State *)ds.p, (int8_t *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(out, dout.p, ob, hipMemcpyDefault));
    return BB_OK;
}

#define GAME_SWITCH(game, ...)                                                   \
    switch (game) {                                                              \
    case BB_GAME_CONNECT4: { using G = Connect4; __VA_ARGS__; }                  \
    case BB_GAME_TICTACTOE: { using G = TicTacToe; __VA_ARGS__; }                \
    case BB_GAME_DRAGONCHESS: { using G = DragonChess; __VA_ARGS__; }            \
    default: return fail(BB_ERR_ARG, "unknown game %d", game);                   \
    }
#define GAME_SWITCH_ALL GAME_SWITCH

extern "C" int bb_game_legal(int game, int n, const void *states, uint8_t *legal_out) {
    if (n == 0) return BB_OK; // an empty batch is a no-op
    if (n < 0 || !states || !legal_out) return fail(BB_ERR_ARG, "bad arguments");
    GAME_SWITCH_ALL(game, return game_legal<G>(n, states, legal_out));
}
extern "C" int bb_game_apply(int game, int n, void *states, const int32_t *actions, int32_t *status_out) {
    if (n == 0) return BB_OK;
    if (n < 0 || !states || !actions) return fail(BB_ERR_ARG, "bad arguments");
    GAME_SWITCH_ALL(game, return game_apply<G>(n, states, actions, status_out));
}
extern "C" int bb_game_winner(int game, int n, const void *states, const int32_t *prev, int8_t *winner_out) {
    if (n == 0) return BB_OK;
    if (n < 0 || !states || !winner_out) return fail(BB_ERR_ARG, "bad arguments");
    GAME_SWITCH_ALL(game, return game_winner<G>(n, states, prev, winner_out));
}
extern "C" int bb_game_encode(int game, int n, const void *states, int8_t *planes_out) {
    if (n == 0) return BB_OK;
    if (n < 0 || !states || !planes_out) return fail(BB_ERR_ARG, "bad arguments");
    GAME_SWITCH_ALL(game, return game_encode<G>(n, states, planes_out));
}
extern "C" int bb_game_initial(int game, void *state_out) {
    if (!state_out) return fail(BB_ERR_ARG, "null out");
    GAME_SWITCH_ALL(game, {
        typename G::State s = G::initial();
        memcpy(state_out, &s, sizeof s);
        return BB_OK;
    });
}

// ---- engine ----------------------------------------------------------------------------------------
struct bb_engine {
    bb_config cfg;
    bb_game_info info;
    TreeDev dev;
    DCEdges edges; // DragonChess only
    int32_t *d_child_action = nullptr;
    NetDev net;
    NetX3 x3 = {nullptr, nullptr, nullptr, nullptr, nullptr}; // 16-filter network of a dense game on the bf16 matrix pipe (net_x3.hip.h); null: float32 MFMA path
    size_t x3_bytes = 0;
    bool has_weights = false;
    int net_F = 0, net_C = 0;
    bool general_net = false; // F != 16 (or BB_GNET=1): one implicit-GEMM launch per conv layer (gnet.hip.h)
    GNetDev gnet = {};
    GNetX3 gx3 = {nullptr, {nullptr, nullptr}}; // tower layers of the general-filter network on the bf16 matrix pipe (gnet_x3.hip.h)
    size_t gx3_bytes = 0;
    int gnet_C = 0;
    size_t net_sizes[4] = {0, 0, 0, 0};
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    int n_games_target = 0;
    int sims_now = 0;
    size_t node_bytes = 0;
    double *d_u = nullptr;
    int32_t *d_actions = nullptr;
    // optional HIP-event timing of the evaluator launches
    // pipelined asynchronous self-play: slot-range views of `dev`, one HIP stream each, so that one
    // group's (latency-bound) tree kernel runs underneath the other group's (MFMA-bound) network kernel
    int n_views = 1;
    TreeDev view[2];
    hipStream_t vstream[2] = {nullptr, nullptr};
    int vround[2] = {0, 0};
    // kernel-tuning knobs, read from the environment ONCE in bb_create (never on the step path); not part of the API
    struct {
        int launch_steps = 64, queue_limit_s = 30, queue_netw = 0, queue_waves = 12;
        bool level_budget_set = false;
    } tune;
    bool mega = false; // persistent per-CU self-play kernel with an LDS work queue (mega2.hip.h)
    bool async_selfplay = false; // dense games, DynamicMCTS, deterministic evaluators: k_tree_async rounds
    bool dc_fused = false;       // DragonChess, DynamicMCTS, 16-filter network: one wave keeps its game for a whole launch (mega_dc.hip.h)
    int round = 0;
    int time_every = 0;
    uint64_t eval_launches = 0;
    std::vector<hipEvent_t> ev_pool; // pairs (start, stop)
    size_t ev_used = 0;
};

static hipError_t sync_all(bb_engine *e) {
    if (e->vstream[1]) {
        hipError_t r = hipStreamSynchronize(e->vstream[1]);
        if (r != hipSuccess) return r;
    }
    return hipStreamSynchronize(e->stream);
}

template <class T>
static int dalloc(bb_engine *e, T *&p, size_t count, bool zero = true) {
    void *q = nullptr;
    size_t bytes = count * sizeof(T);
    hipError_t err = hipMalloc(&q, bytes ? bytes : 16);
    if (err != hipSuccess) return fail(BB_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(err));
    e->allocs.push_back(q);
    if (zero && bytes) {
        err = hipMemsetAsync(q, 0, bytes, e->stream);
        if (err != hipSuccess) return fail(BB_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(err));
    }
    p = (T *)q;
    return 0;
}

template <class G>
static int engine_alloc(bb_engine *e) {
    TreeDev &d = e->dev;
    const bb_config &c = e->cfg;
    size_t n = (size_t)c.n_slots;
    constexpr bool DC = G::GID == BB_GAME_DRAGONCHESS;
    constexpr size_t NODE_BYTES = DC ? sizeof(DCNode) : sizeof(DenseNode<typename std::conditional<DC, Connect4, G>::type>);
    constexpr size_t PSTRIDE = DC ? (size_t)G::A : (size_t)G::S;
    e->node_bytes = NODE_BYTES;
    if (dalloc(e, d.root, n) || dalloc(e, d.root_N, n) || dalloc(e, d.n_nodes, n) || dalloc(e, d.ply, n) ||
        dalloc(e, d.sims_left, n) || dalloc(e, d.pend_leaf, n) || dalloc(e, d.pend_expand, n) ||
        dalloc(e, d.path_len, n) || dalloc(e, d.game_lid, n) || dalloc(e, d.sim_serial, n) ||
        dalloc(e, d.root_W, n) || dalloc(e, d.root_pp, n) || dalloc(e, d.path, n * G::MAXPATH) ||
        dalloc(e, d.anc, n * (size_t)(c.max_plies + 2)) || dalloc(e, d.anc_len, n) || dalloc(e, d.top_N, n) ||
        dalloc(e, d.path_N, n * G::MAXPATH) || dalloc(e, d.path_all, n * G::MAXPATH) || dalloc(e, d.path_W, n * G::MAXPATH) || dalloc(e, d.leaf_flags, n) ||
        dalloc(e, d.leaf_game_id, n) || dalloc(e, d.leaf_serial, n) || dalloc(e, d.eval_value, n) ||
        dalloc(e, d.eval_policy, n * PSTRIDE) || dalloc(e, d.ctr, n * 8) || dalloc(e, d.evals, n) || dalloc(e, d.out_action, n) ||
        dalloc(e, d.out_root_plays, n) || dalloc(e, d.out_child_plays, n * G::S) ||
        dalloc(e, d.out_root_winrate, n) || dalloc(e, d.out_child_value, n * G::S) || dalloc(e, e->d_u, n) ||
        dalloc(e, e->d_actions, n) || dalloc(e, d.stamps, 16 * 64) || dalloc(e, d.visit_pool, 16) || dalloc(e, d.resume_cur, n) ||
        dalloc(e, d.resume_depth, n) || dalloc(e, d.post_count, 8) || dalloc(e, d.post_slot, n) ||
        false)
        return BB_ERR_HIP;
    typename G::State *ls;
    if (dalloc(e, ls, n)) return BB_ERR_HIP;
    d.leaf_state = ls;
    uint8_t *nodes;
    if (dalloc(e, nodes, n * (size_t)d.node_cap * NODE_BYTES, false)) return BB_ERR_HIP;
    d.nodes = nodes;
    if (dalloc(e, e->d_child_action, n * G::S)) return BB_ERR_HIP;
    if constexpr (DC) {
        DCEdges &E = e->edges;
        E.edge_cap = d.node_cap * 24; // ~15-25 legal moves per position; overflow is counted, never silent
        size_t ne = n * (size_t)E.edge_cap;
        if (dalloc(e, E.e, ne, false) || dalloc(e, E.used, n) || dalloc(e, E.path_edge, n * G::MAXPATH))
            return BB_ERR_HIP;
        E.noise_on = c.noise_on;
        E.alpha = c.alpha;
        E.eps = c.epsilon;
    }
    size_t ng = (size_t)c.max_games;
    if (dalloc(e, d.examples, ng * (size_t)(c.max_plies + 1) * (size_t)e->info.example_bytes, false) ||
        dalloc(e, d.game_hdr, ng * 4))
        return BB_ERR_HIP;
    // every slot idle until roots are set / self-play begins
    HIPCHK(hipMemsetAsync(d.game_lid, 0xFF, n * 4, e->stream));
    HIPCHK(hipMemsetAsync(d.pend_leaf, 0xFF, n * 4, e->stream));
    HIPCHK(hipMemsetAsync(d.resume_cur, 0xFF, n * 4, e->stream));
    HIPCHK(sync_all(e));
    return BB_OK;
}

// Slot-range views for pipelined self-play: every per-slot pointer is advanced by the view's first slot.
template <class G>
static void make_views(bb_engine *e) {
    const TreeDev &d = e->dev;
    int n = d.n_slots;
    for (int v = 0; v < e->n_views; v++) {
        TreeDev w = d;
        int off = v == 0 ? 0 : n / 2;
        int cnt = e->n_views == 1 ? n : (v == 0 ? n / 2 : n - n / 2);
        w.n_slots = cnt;
        w.slot_offset = off;
        w.root += off; w.root_N += off; w.n_nodes += off; w.ply += off; w.sims_left += off; w.pend_leaf += off;
        w.pend_expand += off; w.path_len += off; w.game_lid += off; w.sim_serial += off; w.root_W += off;
        w.root_pp += off; w.path += (size_t)off * G::MAXPATH;
        w.anc += (size_t)off * (d.max_plies + 2); w.anc_len += off; w.top_N += off;
        w.path_N += (size_t)off * G::MAXPATH; w.path_all += (size_t)off * G::MAXPATH; w.path_W += (size_t)off * G::MAXPATH; w.leaf_flags += off;
        w.leaf_state = (char *)d.leaf_state + (size_t)off * sizeof(typename G::State);
        w.leaf_game_id += off; w.leaf_serial += off; w.eval_value += off;
        w.eval_policy += (size_t)off * (G::GID == BB_GAME_DRAGONCHESS ? G::A : G::S);
        w.evals += off; w.ctr += (size_t)off * 8;
        w.nodes = (char *)d.nodes + (size_t)off * d.node_cap * e->node_bytes;
        w.resume_cur += off; w.resume_depth += off; w.post_slot += off; w.post_count += 4 * v;
        e->view[v] = w;
    }
}

// ---- pool sizing ------------------------------------------------------------------------------------------------
static long node_capacity_of(const bb_config *cfg) {
    long cap = cfg->node_capacity > 0 ? cfg->node_capacity : (long)cfg->sims_per_move * cfg->max_plies + 2;
    if (cfg->mcts_kind == BB_MCTS_FIXED && cfg->node_capacity <= 0) cap = cap * cfg->max_depth;
    return cap;
}

// device bytes engine_alloc<G> asks for: per slot (node pool, DragonChess edge pool, mailboxes, paths) and per engine
// (example store of max_games games)
template <class G>
static void pool_bytes(const bb_config *cfg, size_t *per_slot, size_t *fixed) {
    constexpr bool DC = G::GID == BB_GAME_DRAGONCHESS;
    constexpr size_t NODE_BYTES = DC ? sizeof(DCNode) : sizeof(DenseNode<typename std::conditional<DC, Connect4, G>::type>);
    const size_t cap = (size_t)node_capacity_of(cfg);
    size_t ps = cap * NODE_BYTES;
    if (DC) ps += cap * 24 * sizeof(DCEdge) + (size_t)G::MAXPATH * 4;
    ps += (size_t)G::MAXPATH * 4 + (DC ? (size_t)G::A : (size_t)G::S) * 4 + (size_t)G::S * 16 + sizeof(typename G::State) + 256;
    bb_game_info gi;
    bb_game_info_get(cfg->game, &gi);
    const size_t ng = (size_t)(cfg->max_games > 0 ? cfg->max_games : cfg->n_slots);
    *per_slot = ps;
    *fixed = ng * ((size_t)(cfg->max_plies + 1) * (size_t)gi.example_bytes + 16) + (64u << 20); // + weights, scratch, runtime slack
}

static int check_config(const bb_config *cfg) {
    if (!cfg) return fail(BB_ERR_ARG, "null argument");
    if (cfg->n_slots <= 0) return fail(BB_ERR_ARG, "n_slots must be positive");
    if (cfg->mcts_kind == BB_MCTS_FIXED && cfg->max_depth <= 0)
        return fail(BB_ERR_ARG, "MaxDepth for MCTS must be > 0."); // FixedMCTS.py:15-16
    if (cfg->sims_per_move < 0 || cfg->max_plies <= 0) return fail(BB_ERR_ARG, "bad sims_per_move/max_plies");
    int ndev = bb_device_count();
    if (ndev <= 0) return fail(BB_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(BB_ERR_ARG, "device %d out of range", cfg->device);
    if (node_capacity_of(cfg) >= (1 << 26)) return fail(BB_ERR_ARG, "node capacity too large");
    return BB_OK;
}

extern "C" int bb_fit_slots(const bb_config *cfg, int *n_slots_out, uint64_t *bytes_per_slot_out) {
    int rc = check_config(cfg);
    if (rc) return rc;
    size_t per_slot = 0, fixed = 0;
    GAME_SWITCH(cfg->game, pool_bytes<G>(cfg, &per_slot, &fixed); break);
    HIPCHK(hipSetDevice(cfg->device));
    size_t free_b = 0, total_b = 0;
    HIPCHK(hipMemGetInfo(&free_b, &total_b));
    const size_t usable = free_b - free_b / 16; // leave 1/16 of what is free to the runtime and to other engines' scratch
    long fit = usable > fixed ? (long)((usable - fixed) / per_slot) : 0;
    if (fit > cfg->n_slots) fit = cfg->n_slots;
    if (n_slots_out) *n_slots_out = (int)fit;
    if (bytes_per_slot_out) *bytes_per_slot_out = (uint64_t)per_slot;
    if (fit <= 0)
        return fail(BB_ERR_CAPACITY, "not even one game slot (%zu bytes) + the example store (%zu bytes) fits the %zu free bytes of device %d",
                    per_slot, fixed, free_b, cfg->device);
    return BB_OK;
}

extern "C" int bb_create(const bb_config *cfg, bb_engine **out) {
    if (!cfg || !out) return fail(BB_ERR_ARG, "null argument");
    {
        int rc = check_config(cfg);
        if (rc) return rc;
        int fit = 0;
        uint64_t per_slot = 0;
        rc = bb_fit_slots(cfg, &fit, &per_slot);
        if (rc) return rc;
        if (fit < cfg->n_slots)
            return fail(BB_ERR_CAPACITY, "%d game slots of %llu bytes each do not fit the free memory of device %d (%d would; "
                        "fewer slots play the same games one after another)", cfg->n_slots, (unsigned long long)per_slot, cfg->device, fit);
    }
    bb_engine *e = new bb_engine();
    e->cfg = *cfg;
    e->sims_now = cfg->sims_per_move;
    int rc = bb_game_info_get(cfg->game, &e->info);
    if (rc) {
        delete e;
        return rc;
    }
    HIPCHK(hipSetDevice(cfg->device));
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    TreeDev &d = e->dev;
    memset(&d, 0, sizeof d);
    d.n_slots = cfg->n_slots;
    d.sims_per_move = cfg->sims_per_move;
    d.max_plies = cfg->max_plies;
    d.kind = cfg->mcts_kind;
    d.max_depth = cfg->max_depth;
    d.evaluator = cfg->evaluator;
    d.priors_ones = (cfg->mcts_kind == BB_MCTS_FIXED || cfg->evaluator == BB_EVAL_ROLLOUT) ? 1 : 0;
    d.salt_per_game = cfg->salt_per_game;
    d.track_anc = (cfg->track_ancestors != 0 && cfg->game != BB_GAME_DRAGONCHESS) ? 1 : 0;
    d.max_games = cfg->max_games > 0 ? cfg->max_games : cfg->n_slots;
    e->cfg.max_games = d.max_games;
    d.c_puct = cfg->c_puct;
    d.seed = cfg->seed;
    d.salt = cfg->hash_salt;
    d.first_game_id = cfg->first_game_id;
    d.node_cap = (int)node_capacity_of(cfg);
    d.example_bytes = e->info.example_bytes;
    d.gpw = 64 / e->info.S;
    d.level_budget = 16; // tree levels per call of the launch-per-round structures (20 x 256 in-search: 12 / 16 / 24 -> 93 / 105 / 106 k evaluations/s); the persistent kernel uses 12, below
    d.slot_offset = 0;
    d.pool_g0 = 0;
    d.noise_alpha = cfg->alpha;
    d.lid_stride = cfg->n_slots;
    auto env_int = [](const char *name, int dflt) {
        const char *v = getenv(name);
        return v ? atoi(v) : dflt;
    };
    if (int v = env_int("BB_LEVEL_BUDGET", 0); v >= 1) {
        d.level_budget = v;
        e->tune.level_budget_set = true;
    }
    e->tune.launch_steps = env_int("BB_LAUNCH_STEPS", 64);
    e->tune.queue_limit_s = env_int("BB_QUEUE_LIMIT_S", 30);
    e->tune.queue_netw = env_int("BB_QUEUE_NETW", 0);
    e->tune.queue_waves = env_int("BB_QUEUE_WAVES", 12);
    if (cfg->launch < 0 || cfg->launch > BB_LAUNCH_ROUNDS || cfg->net_form < 0 || cfg->net_form > BB_NET_FORM_SPLIT) {
        delete e;
        return fail(BB_ERR_ARG, "bad bb_config.launch / net_form");
    }
    e->async_selfplay = cfg->game != BB_GAME_DRAGONCHESS && cfg->mcts_kind == BB_MCTS_DYNAMIC &&
                        cfg->evaluator != BB_EVAL_ROLLOUT && cfg->launch != BB_LAUNCH_LOCKSTEP;
    e->mega = e->async_selfplay && cfg->evaluator == BB_EVAL_NET && cfg->launch == BB_LAUNCH_AUTO;
    e->dc_fused = cfg->game == BB_GAME_DRAGONCHESS && cfg->mcts_kind == BB_MCTS_DYNAMIC && cfg->evaluator == BB_EVAL_NET &&
                  cfg->launch == BB_LAUNCH_AUTO;
    if (int v = env_int("BB_TREE_GPW", 0); v >= 1 && v <= 64 / e->info.S) d.gpw = v;
    d.temp = 1.0;
    GAME_SWITCH(cfg->game, rc = engine_alloc<G>(e); break);
    if (rc) {
        bb_destroy(e);
        return rc;
    }
    e->n_views = 1;
    if (e->async_selfplay && !e->mega && cfg->evaluator == BB_EVAL_NET && cfg->n_slots >= 512) e->n_views = 2;
    if (const char *env = getenv("BB_GROUPS")) e->n_views = (atoi(env) == 2 && e->async_selfplay && cfg->n_slots >= 2) ? 2 : 1; // (tuning)
    e->vstream[0] = e->stream;
    if (e->n_views == 2) HIPCHK(hipStreamCreateWithFlags(&e->vstream[1], hipStreamNonBlocking));
    GAME_SWITCH(cfg->game, make_views<G>(e); break);
    *out = e;
    return BB_OK;
}

extern "C" int bb_destroy(bb_engine *e) {
    if (!e) return BB_OK;
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (void *p : e->allocs) (void)hipFree(p);
    for (auto ev : e->ev_pool) (void)hipEventDestroy(ev);
    if (e->vstream[1]) {
        (void)hipStreamSynchronize(e->vstream[1]);
        (void)hipStreamDestroy(e->vstream[1]);
    }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return BB_OK;
}

extern "C" int bb_timing_enable(bb_engine *e, int every_n) {
    if (!e || every_n < 0) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    if (every_n > 0 && e->ev_pool.empty()) {
        e->ev_pool.resize(2 * 512);
        for (auto &ev : e->ev_pool) HIPCHK(hipEventCreate(&ev));
    }
    e->time_every = every_n;
    e->eval_launches = 0;
    e->ev_used = 0;
    return BB_OK;
}

extern "C" int bb_timing_read(bb_engine *e, double *mean_ms_out, double *min_ms_out, int *count_out) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    double sum = 0.0, mn = 1e30;
    int cnt = 0;
    for (size_t i = 0; i + 1 < e->ev_used; i += 2) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[i], e->ev_pool[i + 1]));
        sum += ms;
        if (ms < mn) mn = ms;
        cnt++;
    }
    if (mean_ms_out) *mean_ms_out = cnt ? sum / cnt : 0.0;
    if (min_ms_out) *min_ms_out = cnt ? mn : 0.0;
    if (count_out) *count_out = cnt;
    e->ev_used = 0;
    e->eval_launches = 0;
    return BB_OK;
}

extern "C" int bb_net_form(bb_engine *e) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    if (!e->has_weights) return fail(BB_ERR_WEIGHTS, "bb_load_weights has not been called");
    if (e->general_net) return e->gx3.wt ? 3 : 1;
    return e->x3.w0 ? 2 : 0;
}

extern "C" int bb_selfplay_mode(bb_engine *e) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    // the persistent kernels carry a 16-filter network of at most MEGA_RMAX blocks in LDS; anything else runs as rounds
    const bool fits = !e->has_weights || (!e->general_net && e->net.R <= MEGA_RMAX && e->net.head_floats <= MEGA_HEAD_FLOATS);
    if (e->mega && fits) return 3;
    if (e->dc_fused && (!e->has_weights || (!e->general_net && e->net.head_floats <= DC_HEAD_FLOATS && e->net.R <= DC_RMAX))) return 5;
    return e->async_selfplay ? 1 : 0;
}

extern "C" int bb_synchronize(bb_engine *e) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    if (e->vstream[1]) HIPCHK(hipStreamSynchronize(e->vstream[1]));
    HIPCHK(sync_all(e));
    return BB_OK;
}

// ---- weights: fold BN, swizzle into MFMA operand order, upload ---------------------------------------
static void bn_fold(const float *bn, int F, float *scale, float *shift) {
    for (int f = 0; f < F; f++) {
        float g = bn[0 * F + f], b = bn[1 * F + f], m = bn[2 * F + f], v = bn[3 * F + f];
        float s = g / sqrtf(v + 1e-3f); // tf.layers.batch_normalization default epsilon
        float t = m * s;
        scale[f] = s;
        shift[f] = b - t;
    }
}


// ---- general-F network (gnet.hip.h): operand layouts, buffers, launch sequence -----------------------------------
// ---- operands of net_x3.hip.h: every weight as three bf16 planes (w = w1 + w2 + w3 exactly), in A-operand lane order ----
static uint16_t bf16_rne(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_value(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static void bf16_split3(float v, uint16_t out[3]) {
    out[0] = bf16_rne(v);
    float r = v - bf16_value(out[0]);
    out[1] = bf16_rne(r);
    r = r - bf16_value(out[1]);
    out[2] = bf16_rne(r);
}
template <class G>
static int gnet_reserve(bb_engine *e, int n) {
    using GG = GNetGeom<G>;
    GNetDev &g = e->gnet;
    int cap = (n + GG::PPB - 1) / GG::PPB * GG::PPB;
    if (cap <= g.cap) return BB_OK;
    HIPCHK(sync_all(e));
    size_t pos_floats = (size_t)g.NCB * 4 * GG::PLANE;
    if (dalloc(e, g.inp, (size_t)cap * GG::SLOTS * GG::CP) || dalloc(e, g.act[0], (size_t)cap * pos_floats))
        return BB_ERR_HIP; // zero-filled: the halo ring of every position stays zero for the buffers' lifetime
    if (e->gx3.wt) { // bf16-pipe tower: two buffers of three bf16 planes; the float32 buffer above only carries the last layer
        const size_t pos_bytes = (size_t)g.NCB * GG::SLOTS * 96;
        if (dalloc(e, e->gx3.act3[0], (size_t)cap * pos_bytes) || dalloc(e, e->gx3.act3[1], (size_t)cap * pos_bytes)) return BB_ERR_HIP;
    } else if (dalloc(e, g.act[1], (size_t)cap * pos_floats)) {
        return BB_ERR_HIP;
    }
    HIPCHK(sync_all(e)); // the fills ran on the engine stream; the caller may launch on another one
    g.cap = cap;
    return BB_OK;
}

static int load_general_weights(bb_engine *e, const bb_net_weights *w) {
    const int F = w->F, C = w->C, R = w->R, NCB = F / 16;
    const int steps0 = (9 * C + 3) / 4;
    std::vector<float> w0((size_t)NCB * steps0 * 64), wt((size_t)2 * R * NCB * 9 * NCB * 64 * 4), epi((size_t)(1 + 2 * R) * NCB * 48);
    for (int fb = 0; fb < NCB; fb++)
        for (int s = 0; s < steps0; s++)
            for (int lane = 0; lane < 64; lane++) {
                int f = lane & 15, j = lane >> 4, k = 4 * s + j;
                w0[((size_t)fb * steps0 + s) * 64 + lane] = k < 9 * C ? w->conv0_k[(size_t)k * F + 16 * fb + f] : 0.f;
            }
    for (int l = 0; l < 2 * R; l++)
        for (int fb = 0; fb < NCB; fb++)
            for (int cb = 0; cb < NCB; cb++)
                for (int tap = 0; tap < 9; tap++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int r = 0; r < 4; r++) {
                            int f = lane & 15, j = lane >> 4, c = 16 * cb + 4 * j + r;
                            wt[((((((size_t)l * NCB + fb) * NCB + cb) * 9 + tap) * 64) + lane) * 4 + r] =
                                w->blk_k[(((size_t)l * 9 + tap) * F + c) * F + 16 * fb + f];
                        }
    std::vector<float> sc(F), sh(F);
    for (int l = 0; l < 1 + 2 * R; l++) {
        const float *b = l == 0 ? w->conv0_b : w->blk_b + (size_t)(l - 1) * F;
        const float *bn = l == 0 ? w->conv0_bn : w->blk_bn + (size_t)(l - 1) * 4 * F;
        bn_fold(bn, F, sc.data(), sh.data());
        for (int fb = 0; fb < NCB; fb++) {
            float *o = &epi[((size_t)l * NCB + fb) * 48];
            memcpy(o, b + 16 * fb, 64);
            memcpy(o + 16, sc.data() + 16 * fb, 64);
            memcpy(o + 32, sh.data() + 16 * fb, 64);
        }
    }
    GNetDev &g = e->gnet;
    // weights are reloaded after every training step: keep the device buffers (operands and activation scratch) when
    // the network shape is unchanged instead of allocating new ones each time
    const bool same = g.F == F && g.R == R && g.NCB == NCB && g.w0 && e->gnet_C == C;
    float *d_w0 = (float *)g.w0, *d_wt = (float *)g.wt, *d_epi = (float *)g.epi;
    if (!same) {
        g = GNetDev{};
        g.F = F;
        g.NCB = NCB;
        g.R = R;
        e->gnet_C = C;
        if (dalloc(e, d_w0, w0.size(), false) || dalloc(e, d_wt, wt.size(), false) || dalloc(e, d_epi, epi.size(), false))
            return BB_ERR_HIP;
    }
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpy(d_w0, w0.data(), w0.size() * 4, hipMemcpyHostToDevice));
    if (!wt.empty()) HIPCHK(hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_epi, epi.data(), epi.size() * 4, hipMemcpyHostToDevice));
    g.w0 = d_w0;
    g.wt = (const f32x4 *)d_wt;
    g.epi = d_epi;
    // the tower layers' operands as three bf16 planes (gnet_x3.hip.h); BB_NET_FORM_F32 keeps the float32-MFMA layers
    const bool want3 = R > 0 && e->cfg.net_form != BB_NET_FORM_F32;
    if (want3) {
        static const int slice_taps[4][2] = {{0, 1}, {3, 4}, {6, 7}, {2, 5}};
        const size_t per = GX3_PAIR_B / 2; // uint16 elements per (fb, cb) block
        std::vector<uint16_t> x((size_t)2 * R * NCB * NCB * per);
        uint16_t h[3];
        for (int l = 0; l < 2 * R; l++)
            for (int fb = 0; fb < NCB; fb++)
                for (int cb = 0; cb < NCB; cb++) {
                    uint16_t *o = x.data() + (((size_t)l * NCB + fb) * NCB + cb) * per;
                    for (int lane = 0; lane < 64; lane++) {
                        const int f = lane & 15, gg = lane >> 4;
                        for (int sl = 0; sl < 4; sl++)
                            for (int i = 0; i < 8; i++) {
                                int tap = slice_taps[sl][gg >> 1], c = 16 * cb + 8 * (gg & 1) + i;
                                bf16_split3(w->blk_k[(((size_t)l * 9 + tap) * F + c) * F + 16 * fb + f], h);
                                for (int q = 0; q < 3; q++) o[(((size_t)sl * 3 + q) * 64 + lane) * 8 + i] = h[q];
                            }
                        for (int i = 0; i < 4; i++) {
                            bf16_split3(w->blk_k[(((size_t)l * 9 + 8) * F + 16 * cb + 4 * gg + i) * F + 16 * fb + f], h);
                            for (int q = 0; q < 3; q++) o[(size_t)4 * 3 * 64 * 8 + ((size_t)q * 64 + lane) * 4 + i] = h[q];
                        }
                    }
                }
        unsigned char *d_x = (unsigned char *)e->gx3.wt;
        if (!d_x || x.size() * 2 != e->gx3_bytes) {
            if (dalloc(e, d_x, x.size() * 2 + 16, false)) return BB_ERR_HIP;
            e->gx3_bytes = x.size() * 2;
            g.cap = 0; // (the activation buffers of the other form are re-reserved on the next launch)
        }
        HIPCHK(hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice));
        e->gx3.wt = d_x;
    } else {
        if (e->gx3.wt) g.cap = 0;
        e->gx3.wt = nullptr;
    }
    return BB_OK;
}

// n_ptr != nullptr: the batch size lives in device memory (<= n_max); slot_list maps batch entries to mailbox slots
template <class G>
static int launch_gnet(bb_engine *e, int n_max, const int *n_ptr, const int *slot_list, const typename G::State *states,
                       const int8_t *planes, const uint32_t *game_id, const int32_t *serial, int noise, float *value,
                       float *logits, float *policy, int pstride, hipStream_t st, int buf_offset = 0) {
    // buf_offset: first position of the activation buffers this call may use (the two slot-range views of pipelined
    // rounds run on two streams at once and must not share scratch)
    using GG = GNetGeom<G>;
    buf_offset = (buf_offset + GG::PPB - 1) / GG::PPB * GG::PPB;
    int rc = gnet_reserve<G>(e, buf_offset + n_max);
    if (rc) return rc;
    GNetDev g = e->gnet;
    {
        const size_t pos_floats = (size_t)g.NCB * 4 * GG::PLANE;
        g.inp += (size_t)buf_offset * GG::SLOTS * GG::CP;
        g.act[0] += (size_t)buf_offset * pos_floats;
        if (g.act[1]) g.act[1] += (size_t)buf_offset * pos_floats; // (not allocated when the tower layers run in the split-operand form)
    }
    k_gnet_input<G><<<nblk((size_t)n_max * GG::HW), 256, 0, st>>>(g, n_max, n_ptr, slot_list, states, planes);
    if (e->gx3.wt) { // tower layers on the bf16 matrix pipe (gnet_x3.hip.h); first conv in float32 MFMA, writing the split form
        GNetX3 gx = e->gx3;
        const size_t pos_bytes = (size_t)g.NCB * GG::SLOTS * 96;
        gx.act3[0] += (size_t)buf_offset * pos_bytes;
        gx.act3[1] += (size_t)buf_offset * pos_bytes;
        const int L = 2 * g.R;
        if ((long)n_max * g.NCB <= 2048 && !n_ptr) { // small batch: one position x one filter block per wave
            const int ppw = g.NCB >= 4 ? 4 : (g.NCB >= 2 ? 2 : 1);
            dim3 grid((n_max + 4 / ppw - 1) / (4 / ppw), (g.NCB + ppw - 1) / ppw);
            k_gnet_conv<G, true, 1, 1><<<grid, 256, 0, st>>>(g, 0, n_max, n_ptr, nullptr, g.act[0], 0, ppw, gx.act3[0]);
            for (int l = 0; l < L; l++) {
                if (l + 1 < L) k_gnet_conv_x3<G, 1, 1, false><<<grid, 256, 0, st>>>(g, gx, 1 + l, n_max, n_ptr, gx.act3[l & 1], gx.act3[(l & 1) ^ 1], nullptr, l & 1, ppw);
                else k_gnet_conv_x3<G, 1, 1, true><<<grid, 256, 0, st>>>(g, gx, 1 + l, n_max, n_ptr, gx.act3[l & 1], gx.act3[(l & 1) ^ 1], g.act[0], l & 1, ppw);
            }
        } else {
            constexpr int FBW3 = 4;
            const int fgroups = (g.NCB + FBW3 - 1) / FBW3;               // groups of 4 filter blocks
            const int gpw = fgroups >= 4 ? 4 : (fgroups >= 2 ? 2 : 1);   // ... per workgroup
            const int groups = (n_max + GG::PPB - 1) / GG::PPB;          // groups of PPB positions
            dim3 grid((groups + 4 / gpw - 1) / (4 / gpw), (fgroups + gpw - 1) / gpw);
            {
                const int pairs = (g.NCB + GN_FBW - 1) / GN_FBW;
                const int ppw = pairs >= 4 ? 4 : (pairs >= 2 ? 2 : 1);
                dim3 grid0((groups + 4 / ppw - 1) / (4 / ppw), (pairs + ppw - 1) / ppw);
                k_gnet_conv<G, true><<<grid0, 256, 0, st>>>(g, 0, n_max, n_ptr, nullptr, g.act[0], 0, ppw, gx.act3[0]);
            }
            for (int l = 0; l < L; l++) {
                if (l + 1 < L) k_gnet_conv_x3<G, GG::PPB, FBW3, false><<<grid, 256, 0, st>>>(g, gx, 1 + l, n_max, n_ptr, gx.act3[l & 1], gx.act3[(l & 1) ^ 1], nullptr, l & 1, gpw);
                else k_gnet_conv_x3<G, GG::PPB, FBW3, true><<<grid, 256, 0, st>>>(g, gx, 1 + l, n_max, n_ptr, gx.act3[l & 1], gx.act3[(l & 1) ^ 1], g.act[0], l & 1, gpw);
            }
        }
    } else
    if ((long)n_max * g.NCB <= 2048 && !n_ptr) {
        // small batch: one position x one filter block per wave (latency of a lone evaluation: 40 layers x ~25 us
        // instead of x ~170 us at 256 filters)
        const int ppw = g.NCB >= 4 ? 4 : (g.NCB >= 2 ? 2 : 1);
        dim3 grid((n_max + 4 / ppw - 1) / (4 / ppw), (g.NCB + ppw - 1) / ppw);
        k_gnet_conv<G, true, 1, 1><<<grid, 256, 0, st>>>(g, 0, n_max, n_ptr, nullptr, g.act[0], 0, ppw);
        for (int l = 0; l < 2 * g.R; l++)
            k_gnet_conv<G, false, 1, 1><<<grid, 256, 0, st>>>(g, 1 + l, n_max, n_ptr, g.act[l & 1], g.act[(l & 1) ^ 1], l & 1, ppw);
    } else {
    const int pairs = (g.NCB + GN_FBW - 1) / GN_FBW;
    const int ppw = pairs >= 4 ? 4 : (pairs >= 2 ? 2 : 1); // filter-block pairs per workgroup
    const int groups = (n_max + GG::PPB - 1) / GG::PPB;    // groups of PPB positions
    dim3 grid((groups + 4 / ppw - 1) / (4 / ppw), (pairs + ppw - 1) / ppw);
    k_gnet_conv<G, true><<<grid, 256, 0, st>>>(g, 0, n_max, n_ptr, nullptr, g.act[0], 0, ppw);
    for (int l = 0; l < 2 * g.R; l++)
        k_gnet_conv<G, false><<<grid, 256, 0, st>>>(g, 1 + l, n_max, n_ptr, g.act[l & 1], g.act[(l & 1) ^ 1], l & 1, ppw);
    }
    k_gnet_heads<G><<<(n_max + 3) / 4, 256, 0, st>>>(g, e->net, n_max, n_ptr, slot_list, g.act[0], game_id, serial, noise, value,
                                                     logits, policy, pstride);
    HIPCHK(hipGetLastError());
    return BB_OK;
}

// w0:   narrow input: [plane][lane][8] (taps 2g, 2g + 1 x 4 input planes), then ONE operand [lane][8] for tap 8: lane group 0
//       = [w1 | w2] (4 input planes each), group 1 = [w3 | 0], groups 2, 3 zero -- against B = [x | x] that is all three planes
//       of the tap in one K = 32 product.  Wide input (DragonChess): [tap][plane][lane][8], lane group g = input planes 8g .. 8g + 7
// wt12: per layer [slice 0..3][plane 0..1][lane][8] (slices = taps (0,1), (3,4), (6,7), (2,5); lane group g: tap g >> 1 of the
//       slice, channels 8 (g & 1) .. + 7) then tap 8: [plane 0..1][channel half][filter][8] (what both halves of the lane groups read)
// wt3:  per layer [slice][lane][8] (the third plane alone) then [lane][8] = tap 8's operand [w1 | w3]: lane groups 0, 1 plane 1,
//       groups 2, 3 plane 3, channels 8 (g & 1) .. + 7
// wt8:  per layer [3][lane][8]: tap 8's three A operands [w1|w1], [w2|w2], [w1|w3] as 64-lane images (PP form)
// (tap 8 = three K = 32 products on plane-concatenated operands: [w1|w1].[x1;x2] + [w2|w2].[x1;x2] + [w1|w3].[x3;x1], net_x3.hip.h)
// wh:   [3][lane][8]: the head convolutions as a 16-filter K = 16 layer -- filter 0 = value conv, 4 and 8 = policy conv, the rest
//       zero (so that lane groups 0, 1, 2 of the result each hold ONE head's activation) -- in the three operands of tap 8
static void pack_x3(const bb_net_weights *w, std::vector<uint16_t> &w0, std::vector<uint16_t> &wt12, std::vector<uint16_t> &wt3,
                    std::vector<uint16_t> &wt8, std::vector<uint16_t> &wh) {
    const int F = 16, C = w->C, R = w->R;
    const bool wide = C > 4; // DragonChess: one K = 32 slice per tap, lane group g = input planes 8g .. 8g + 7
    w0.assign(wide ? (size_t)9 * 3 * 64 * 8 : (size_t)(3 * 64 * 8 + 64 * 8), 0);
    const size_t per12 = 4 * 2 * 64 * 8 + 2 * 32 * 8, per3 = 4 * 64 * 8 + 64 * 8;
    wt12.assign((size_t)2 * R * per12, 0);
    wt3.assign((size_t)2 * R * per3, 0);
    wt8.assign((size_t)2 * R * 3 * 64 * 8, 0);
    uint16_t h[3];
    for (int lane = 0; wide && lane < 64; lane++) {
        const int f = lane & 15, g = lane >> 4;
        for (int tap = 0; tap < 9; tap++)
            for (int i = 0; i < 8; i++) {
                int ch = 8 * g + i;
                bf16_split3(ch < C ? w->conv0_k[((size_t)tap * C + ch) * F + f] : 0.f, h);
                for (int q = 0; q < 3; q++) w0[(((size_t)tap * 3 + q) * 64 + lane) * 8 + i] = h[q];
            }
    }
    for (int lane = 0; !wide && lane < 64; lane++) {
        const int f = lane & 15, g = lane >> 4;
        for (int i = 0; i < 8; i++) {
            int tap = 2 * g + (i >> 2), ch = i & 3;
            bf16_split3(ch < C ? w->conv0_k[((size_t)tap * C + ch) * F + f] : 0.f, h);
            for (int q = 0; q < 3; q++) w0[((size_t)q * 64 + lane) * 8 + i] = h[q];
        }
        for (int i = 0; i < 8 && g < 2; i++) { // tap 8: k slot i of lane group g = plane 2g + (i >> 2) of input plane i & 3
            const int q = 2 * g + (i >> 2), ch = i & 3;
            if (q > 2) continue;
            bf16_split3(ch < C ? w->conv0_k[((size_t)8 * C + ch) * F + f] : 0.f, h);
            w0[(size_t)3 * 64 * 8 + (size_t)lane * 8 + i] = h[q];
        }
    }
    wh.assign((size_t)3 * 64 * 8, 0);
    for (int lane = 0; lane < 64; lane++) {
        const int f = lane & 15, g = lane >> 4;
        const int head = f == 0 ? 0 : f == 4 ? 1 : f == 8 ? 2 : -1; // filter rows 0, 4, 8: the first result register of lane groups 0, 1, 2
        for (int i = 0; i < 8 && head >= 0; i++) {
            const int ch = 8 * (g & 1) + i;
            bf16_split3(head == 0 ? w->v_conv_k[ch] : w->p_conv_k[(size_t)ch * 2 + (head - 1)], h);
            wh[((size_t)0 * 64 + lane) * 8 + i] = h[0];
            wh[((size_t)1 * 64 + lane) * 8 + i] = h[1];
            wh[((size_t)2 * 64 + lane) * 8 + i] = g < 2 ? h[0] : h[2];
        }
    }
    static const int slice_taps[4][2] = {{0, 1}, {3, 4}, {6, 7}, {2, 5}};
    for (int l = 0; l < 2 * R; l++) {
        uint16_t *o12 = wt12.data() + (size_t)l * per12, *o3 = wt3.data() + (size_t)l * per3;
        for (int lane = 0; lane < 64; lane++) {
            const int f = lane & 15, g = lane >> 4;
            for (int sl = 0; sl < 4; sl++)
                for (int i = 0; i < 8; i++) {
                    int tap = slice_taps[sl][g >> 1], ch = 8 * (g & 1) + i;
                    bf16_split3(w->blk_k[(((size_t)l * 9 + tap) * F + ch) * F + f], h);
                    for (int q = 0; q < 2; q++) o12[(((size_t)sl * 2 + q) * 64 + lane) * 8 + i] = h[q];
                    o3[((size_t)sl * 64 + lane) * 8 + i] = h[2];
                }
            for (int i = 0; i < 8; i++) { // tap 8, channels 8 (g & 1) .. + 7
                bf16_split3(w->blk_k[(((size_t)l * 9 + 8) * F + (8 * (g & 1) + i)) * F + f], h);
                if (g < 2)
                    for (int q = 0; q < 2; q++) o12[(size_t)4 * 2 * 64 * 8 + (((size_t)q * 2 + g) * 16 + f) * 8 + i] = h[q];
                const uint16_t a3 = g < 2 ? h[0] : h[2];
                o3[(size_t)4 * 64 * 8 + (size_t)lane * 8 + i] = a3;
                wt8[(((size_t)l * 3 + 0) * 64 + lane) * 8 + i] = h[0];
                wt8[(((size_t)l * 3 + 1) * 64 + lane) * 8 + i] = h[1];
                wt8[(((size_t)l * 3 + 2) * 64 + lane) * 8 + i] = a3;
            }
        }
    }
}

extern "C" int bb_load_weights(bb_engine *e, const bb_net_weights *w) {
    if (!e || !w) return fail(BB_ERR_ARG, "null argument");
    const bb_game_info &gi = e->info;
    if (w->H != gi.H || w->W != gi.W || w->C != gi.C || w->A != gi.A)
        return fail(BB_ERR_ARG, "weights are for a %dx%dx%d/%d network, game needs %dx%dx%d/%d", w->H, w->W, w->C,
                    w->A, gi.H, gi.W, gi.C, gi.A);
    if (w->F <= 0 || w->F % 16 != 0 || w->F > 1024)
        return fail(BB_ERR_ARG, "filters must be a multiple of 16 (MFMA tile), got %d", w->F);
    if (w->D <= 0 || w->D > 64 || w->R < 0) return fail(BB_ERR_ARG, "unsupported dense/blocks");
    HIPCHK(hipSetDevice(e->cfg.device));
    const int F = w->F, C = w->C, R = w->R, D = w->D, A = w->A;
    const int steps0 = (9 * C + 3) / 4;
    e->general_net = F != 16 || e->cfg.general_net != 0;
    if (e->general_net) {
        int rc = load_general_weights(e, w);
        if (rc) return rc;
    }
    std::vector<float> w0((size_t)steps0 * 64), wt((size_t)2 * R * 9 * 64 * 4), epi((size_t)(1 + 2 * R) * 48);
    if (F == 16) // operands of the fused single-wave tower (net.hip.h)
    for (int s = 0; s < steps0; s++)
        for (int lane = 0; lane < 64; lane++) {
            int f = lane & 15, j = lane >> 4, k = 4 * s + j;
            w0[(size_t)s * 64 + lane] = k < 9 * C ? w->conv0_k[(size_t)k * F + f] : 0.f;
        }
    if (F == 16)
    for (int l = 0; l < 2 * R; l++)
        for (int tap = 0; tap < 9; tap++)
            for (int lane = 0; lane < 64; lane++)
                for (int r = 0; r < 4; r++) {
                    int f = lane & 15, j = lane >> 4, c = 4 * j + r;
                    wt[(((size_t)l * 9 + tap) * 64 + lane) * 4 + r] = w->blk_k[(((size_t)l * 9 + tap) * F + c) * F + f];
                }
    for (int l = 0; F == 16 && l < 1 + 2 * R; l++) {
        const float *b = l == 0 ? w->conv0_b : w->blk_b + (size_t)(l - 1) * F;
        const float *bn = l == 0 ? w->conv0_bn : w->blk_bn + (size_t)(l - 1) * 4 * F;
        memcpy(&epi[(size_t)l * 48], b, F * sizeof(float));
        bn_fold(bn, F, &epi[(size_t)l * 48 + 16], &epi[(size_t)l * 48 + 32]);
    }
    std::vector<float> head;
    NetDev &nd = e->net;
    auto push = [&](const float *p, int n) {
        int off = (int)head.size();
        head.insert(head.end(), p, p + n);
        while (head.size() % 4) head.push_back(0.f);
        return off;
    };
    float v3[3], p6[6], s1, t1, s2[2], t2[2];
    bn_fold(w->v_bn, 1, &s1, &t1);
    bn_fold(w->p_bn, 2, s2, t2);
    v3[0] = w->v_conv_b[0]; v3[1] = s1; v3[2] = t1;
    p6[0] = w->p_conv_b[0]; p6[1] = w->p_conv_b[1]; p6[2] = s2[0]; p6[3] = s2[1]; p6[4] = t2[0]; p6[5] = t2[1];
    nd.off_vk = push(w->v_conv_k, F);
    nd.off_v3 = push(v3, 3);
    nd.off_d1k = push(w->v_d1_k, D);
    nd.off_d1b = push(w->v_d1_b, D);
    nd.off_d2k = push(w->v_d2_k, D);
    nd.off_d2b = push(w->v_d2_b, 1);
    nd.off_pk = push(w->p_conv_k, 2 * F);
    nd.off_p6 = push(p6, 6);
    nd.off_pdk = push(w->p_d_k, 2 * A);
    nd.off_pdb = push(w->p_d_b, A);
    nd.head_floats = (int)head.size();
    // (weights are reloaded after every training step: the operand buffers are reused while their sizes stay the same)
    float *d_w0 = (float *)nd.w0, *d_wt = (float *)nd.wt, *d_epi = (float *)nd.epi, *d_head = (float *)nd.head;
    const size_t sizes[4] = {w0.size(), wt.size(), epi.size(), head.size()};
    if (!d_w0 || memcmp(sizes, e->net_sizes, sizeof(sizes)) != 0) {
        if (dalloc(e, d_w0, w0.size(), false) || dalloc(e, d_wt, wt.size(), false) || dalloc(e, d_epi, epi.size(), false) ||
            dalloc(e, d_head, head.size(), false))
            return BB_ERR_HIP;
        memcpy(e->net_sizes, sizes, sizeof(sizes));
    }
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpy(d_w0, w0.data(), w0.size() * 4, hipMemcpyHostToDevice));
    if (!wt.empty()) HIPCHK(hipMemcpy(d_wt, wt.data(), wt.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_epi, epi.data(), epi.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_head, head.data(), head.size() * 4, hipMemcpyHostToDevice));
    // the bf16-pipe form of the same network (every game, 16 filters); BB_NET_FORM_F32 keeps the float32 MFMA path
    {
        const bool want = F == 16 && C <= 32 && !e->general_net && e->cfg.net_form != BB_NET_FORM_F32;
        if (want) {
            std::vector<uint16_t> xw0, xw12, xw3, xw8, xwh;
            pack_x3(w, xw0, xw12, xw3, xw8, xwh);
            const size_t b0 = xw0.size() * 2, b12 = xw12.size() * 2, b3 = xw3.size() * 2, b8 = xw8.size() * 2, bh = xwh.size() * 2,
                         bytes = b0 + b12 + b3 + b8 + bh;
            unsigned char *d_x = (unsigned char *)e->x3.w0;
            if (!d_x || bytes != e->x3_bytes) {
                if (dalloc(e, d_x, bytes + 16, false)) return BB_ERR_HIP;
                e->x3_bytes = bytes;
            }
            HIPCHK(hipMemcpy(d_x, xw0.data(), b0, hipMemcpyHostToDevice));
            if (b12) HIPCHK(hipMemcpy(d_x + b0, xw12.data(), b12, hipMemcpyHostToDevice));
            if (b3) HIPCHK(hipMemcpy(d_x + b0 + b12, xw3.data(), b3, hipMemcpyHostToDevice));
            if (b8) HIPCHK(hipMemcpy(d_x + b0 + b12 + b3, xw8.data(), b8, hipMemcpyHostToDevice));
            HIPCHK(hipMemcpy(d_x + b0 + b12 + b3 + b8, xwh.data(), bh, hipMemcpyHostToDevice));
            e->x3.w0 = d_x;
            e->x3.wt12 = d_x + b0;
            e->x3.wt3 = d_x + b0 + b12;
            e->x3.wt8 = d_x + b0 + b12 + b3;
            e->x3.wh = d_x + b0 + b12 + b3 + b8;
        } else {
            e->x3.w0 = e->x3.wt12 = e->x3.wt3 = e->x3.wt8 = e->x3.wh = nullptr;
        }
    }
    nd.R = R;
    nd.D = D;
    nd.A = A;
    nd.w0 = d_w0;
    nd.wt = (const f32x4 *)d_wt;
    nd.epi = d_epi;
    nd.head = d_head;
    nd.seed = e->cfg.seed;
    nd.alpha = e->cfg.alpha;
    nd.eps = e->cfg.epsilon;
    nd.inv_alpha = 1.0f / nd.alpha;
    nd.inv_beta = 1.0f / (1.0f - nd.alpha);
    nd.dbg = 0; // (ablation switches exist in diagnostic builds only: bb_timing_net, -DBB_DIAG)
    e->has_weights = true;
    e->net_F = F;
    e->net_C = C;
    return BB_OK;
}

// positions per wave of the fused tower, chosen so that 4 waves' activations fill the 160 KiB LDS
template <class G> struct NetPW;
template <> struct NetPW<Connect4> { static constexpr int v = 4; };
template <> struct NetPW<TicTacToe> { static constexpr int v = 12; };
template <> struct NetPW<DragonChess> { static constexpr int v = 1; }; // 64 pixels = 4 full tiles; 1024 games fill 256 CUs

template <class G>
static int launch_net(bb_engine *e, int n, const typename G::State *states, const int8_t *planes,
                      const uint32_t *game_id, const int32_t *serial, int noise, float *value, float *logits,
                      float *policy, int pstride, hipStream_t st) {
    if (e->general_net)
        return launch_gnet<G>(e, n, nullptr, nullptr, states, planes, game_id, serial, noise, value, logits, policy, pstride, st);
    // positions per wave: the fewest that still put a wave on every SIMD (1024 waves) -- a single FindMove position
    // must not pay for the 11 MFMA tiles of a 4-position wave
    constexpr int PW = NetPW<G>::v;
    {
        if (e->x3.w0) { // one position per wave on the bf16 pipe: the same arithmetic everywhere (bb_net_eval, search, self-play)
            k_net_x3<G><<<(n + 3) / 4, 256, 0, st>>>(e->net, e->x3, n, nullptr, nullptr, states, planes, game_id, serial, noise, value,
                                                     logits, policy, pstride);
            HIPCHK(hipGetLastError());
            return BB_OK;
        }
    }
    if (PW > 1 && n <= 1024) {
        k_net_fused16<G, 1><<<(n + 3) / 4, 256, 0, st>>>(e->net, n, states, planes, game_id, serial, noise, value, logits, policy,
                                                         pstride);
    } else if (PW > 2 && n <= 2048) {
        k_net_fused16<G, 2><<<(n + 7) / 8, 256, 0, st>>>(e->net, n, states, planes, game_id, serial, noise, value, logits, policy,
                                                         pstride);
    } else {
        int blocks = (n + 4 * PW - 1) / (4 * PW);
        k_net_fused16<G, PW><<<blocks, 256, 0, st>>>(e->net, n, states, planes, game_id, serial, noise, value, logits,
                                                      policy, pstride);
    }
    HIPCHK(hipGetLastError());
    return BB_OK;
}

template <class G>
static int net_eval(bb_engine *e, int n, const void *states, const int8_t *planes, float *value, float *logits,
                    float *policy, int noise, const uint32_t *game_ids = nullptr, const int32_t *serials = nullptr) {
    const int A = G::A;
    size_t pb = (size_t)n * G::H * G::W * G::C;
    DevBuf din, dv, dl, dp, dg, dser;
    if (din.alloc(states ? (size_t)n * sizeof(typename G::State) : pb) || dv.alloc((size_t)n * 4) ||
        dl.alloc((size_t)n * A * 4) || dp.alloc((size_t)n * A * 4) || dg.alloc((size_t)n * 4) || dser.alloc((size_t)n * 4))
        return BB_ERR_HIP;
    HIPCHK(hipMemcpy(din.p, states ? states : (const void *)planes, states ? (size_t)n * sizeof(typename G::State) : pb,
                     hipMemcpyDefault));
    if (game_ids) HIPCHK(hipMemcpy(dg.p, game_ids, (size_t)n * 4, hipMemcpyDefault));
    if (serials) HIPCHK(hipMemcpy(dser.p, serials, (size_t)n * 4, hipMemcpyDefault));
    int rc = launch_net<G>(e, n, states ? (const typename G::State *)din.p : nullptr,
                           states ? nullptr : (const int8_t *)din.p, game_ids ? (const uint32_t *)dg.p : nullptr,
                           serials ? (const int32_t *)dser.p : nullptr, noise, (float *)dv.p,
                           (float *)dl.p, (float *)dp.p, A, e->stream);
    if (rc) return rc;
    HIPCHK(sync_all(e));
    if (value) HIPCHK(hipMemcpy(value, dv.p, (size_t)n * 4, hipMemcpyDefault));
    if (logits) HIPCHK(hipMemcpy(logits, dl.p, (size_t)n * A * 4, hipMemcpyDefault));
    if (policy) HIPCHK(hipMemcpy(policy, dp.p, (size_t)n * A * 4, hipMemcpyDefault));
    return BB_OK;
}

extern "C" int bb_net_eval(bb_engine *e, int n, const void *states, const int8_t *planes, float *value_out,
                           float *logits_out, float *policy_out, int noise) {
    if (e && n == 0) return BB_OK; // an empty batch is a no-op
    if (!e || n < 0 || (!states == !planes)) return fail(BB_ERR_ARG, "bad arguments (exactly one of states/planes)");
    if (!e->has_weights) return fail(BB_ERR_WEIGHTS, "bb_load_weights has not been called");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, return net_eval<G>(e, n, states, planes, value_out, logits_out, policy_out, noise));
}

extern "C" int bb_net_eval_keyed(bb_engine *e, int n, const void *states, const int8_t *planes, const uint32_t *game_ids,
                                 const int32_t *node_serials, float *value_out, float *logits_out, float *policy_out) {
    if (e && n == 0) return BB_OK;
    if (!e || n < 0 || (!states == !planes) || !game_ids || !node_serials)
        return fail(BB_ERR_ARG, "bad arguments (exactly one of states/planes; game_ids and node_serials are required)");
    if (!e->has_weights) return fail(BB_ERR_WEIGHTS, "bb_load_weights has not been called");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, return net_eval<G>(e, n, states, planes, value_out, logits_out, policy_out, 1, game_ids, node_serials));
}

template <class G>
static int hash_eval(bb_engine *e, int n, const void *states, float *value, float *policy) {
    DevBuf ds, dv, dp;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || dv.alloc((size_t)n * 4) || dp.alloc((size_t)n * G::A * 4))
        return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if constexpr (G::GID == BB_GAME_DRAGONCHESS)
        k_dc_hash_eval<<<nblk((size_t)n * 64), 256, 0, e->stream>>>(n, (const DCState *)ds.p, nullptr, e->cfg.hash_salt, 0, 0,
                                                                      (float *)dv.p, (float *)dp.p, G::A);
    else
        k_hash_eval<G><<<nblk(n), 256, 0, e->stream>>>(n, (const typename G::State *)ds.p, nullptr, e->cfg.hash_salt, 0, 0,
                                                       (float *)dv.p, (float *)dp.p, G::A);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_all(e));
    if (value) HIPCHK(hipMemcpy(value, dv.p, (size_t)n * 4, hipMemcpyDefault));
    if (policy) HIPCHK(hipMemcpy(policy, dp.p, (size_t)n * G::A * 4, hipMemcpyDefault));
    return BB_OK;
}

extern "C" int bb_hash_eval(bb_engine *e, int n, const void *states, float *value_out, float *policy_out) {
    if (e && n == 0) return BB_OK;
    if (!e || n < 0 || !states) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, return hash_eval<G>(e, n, states, value_out, policy_out));
}

// ---- simulation loop ---------------------------------------------------------------------------------
template <class G>
static int launch_eval_inner(bb_engine *e);

template <class G>
static int launch_eval(bb_engine *e) {
    bool timed = e->time_every > 0 && (e->eval_launches++ % (uint64_t)e->time_every) == 0 &&
                 e->ev_used + 2 <= e->ev_pool.size();
    if (timed) HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], e->stream));
    int rc = launch_eval_inner<G>(e);
    if (rc) return rc;
    if (timed) {
        HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], e->stream));
        e->ev_used += 2;
    }
    return BB_OK;
}

template <class G>
static int launch_eval_inner(bb_engine *e) {
    TreeDev &d = e->dev;
    int n = d.n_slots;
    const typename G::State *ls = (const typename G::State *)d.leaf_state;
    switch (d.evaluator) {
    case BB_EVAL_HASH:
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_hash_eval<<<nblk((size_t)n * 64), 256, 0, e->stream>>>(n, ls, d.leaf_game_id, d.salt, d.salt_per_game,
                                                                          d.first_game_id, d.eval_value, d.eval_policy, G::A);
        else
            k_hash_eval<G><<<nblk(n), 256, 0, e->stream>>>(n, ls, d.leaf_game_id, d.salt, d.salt_per_game, d.first_game_id,
                                                           d.eval_value, d.eval_policy, G::S);
        break;
    case BB_EVAL_NET:
        // (a wide game's prior noise is mixed in at expansion, over the legal moves only: tree_dc.hip.h)
        return launch_net<G>(e, n, ls, nullptr, d.leaf_game_id, d.leaf_serial, (G::GID == BB_GAME_DRAGONCHESS) ? 0 : e->cfg.noise_on,
                             d.eval_value, nullptr, d.eval_policy, (G::GID == BB_GAME_DRAGONCHESS) ? G::A : G::S, e->stream);
    case BB_EVAL_ROLLOUT:
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_rollout<<<nblk((size_t)n * 64), 256, 0, e->stream>>>(n, ls, d.leaf_game_id, d.sim_serial, d.pend_leaf, d.seed,
                                                                        d.eval_value);
        else
            k_rollout<G><<<nblk(n), 256, 0, e->stream>>>(n, ls, d.leaf_game_id, d.sim_serial, d.pend_leaf, d.seed,
                                                         d.eval_value);
        break;
    default: return fail(BB_ERR_ARG, "unknown evaluator %d", d.evaluator);
    }
    HIPCHK(hipGetLastError());
    return BB_OK;
}

template <class G>
static int run_sims(bb_engine *e, int sims) {
    TreeDev &d = e->dev;
    int tb = nblk((size_t)((d.n_slots + d.gpw - 1) / d.gpw) * 64);
    for (int s = 0; s < sims; s++) {
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_tree_step<<<nblk((size_t)d.n_slots * 64), 256, 0, e->stream>>>(d, e->edges);
        else
            k_tree_step<G><<<tb, 256, 0, e->stream>>>(d);
        HIPCHK(hipGetLastError());
        int rc = launch_eval<G>(e);
        if (rc) return rc;
    }
    return BB_OK;
}

static int check_eval(bb_engine *e) {
    if (e->cfg.evaluator == BB_EVAL_NET && !e->has_weights)
        return fail(BB_ERR_WEIGHTS, "network evaluator needs bb_load_weights first");
    return BB_OK;
}

template <class G>
static int set_roots(bb_engine *e, int n, const int32_t *slots, const void *states, const uint32_t *gids) {
    DevBuf ds, dsl, dg;
    if (ds.alloc((size_t)n * sizeof(typename G::State)) || dsl.alloc((size_t)n * 4) || dg.alloc((size_t)n * 4))
        return BB_ERR_HIP;
    HIPCHK(hipMemcpy(ds.p, states, (size_t)n * sizeof(typename G::State), hipMemcpyDefault));
    if (slots) HIPCHK(hipMemcpy(dsl.p, slots, (size_t)n * 4, hipMemcpyDefault));
    if (gids) HIPCHK(hipMemcpy(dg.p, gids, (size_t)n * 4, hipMemcpyDefault));
    if constexpr (G::GID == BB_GAME_DRAGONCHESS)
        k_dc_set_roots<<<nblk(n), 256, 0, e->stream>>>(e->dev, e->edges, n, slots ? (const int32_t *)dsl.p : nullptr,
                                                         (const DCState *)ds.p, gids ? (const uint32_t *)dg.p : nullptr);
    else
        k_set_roots<G><<<nblk(n), 256, 0, e->stream>>>(e->dev, n, slots ? (const int32_t *)dsl.p : nullptr,
                                                       (const typename G::State *)ds.p, gids ? (const uint32_t *)dg.p : nullptr);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_all(e));
    return BB_OK;
}

extern "C" int bb_set_roots(bb_engine *e, int n, const int32_t *slots, const void *states, const uint32_t *game_ids) {
    if (!e || n <= 0 || n > e->cfg.n_slots || !states) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, return set_roots<G>(e, n, slots, states, game_ids));
}

static int run_sims_api(bb_engine *e, int sims, const uint8_t *mask) {
    if (!e || sims <= 0) return fail(BB_ERR_ARG, "Not enough information to decide a stop time."); // MCTS.py:181-182
    int rc = check_eval(e);
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->cfg.device));
    DevBuf dm;
    if (mask) {
        if (dm.alloc((size_t)e->dev.n_slots)) return BB_ERR_HIP;
        HIPCHK(hipMemcpyAsync(dm.p, mask, (size_t)e->dev.n_slots, hipMemcpyDefault, e->stream));
    }
    GAME_SWITCH(e->cfg.game, {
        k_add_sims<typename std::conditional<G::GID == BB_GAME_DRAGONCHESS, Connect4, G>::type><<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev, sims, mask ? (const uint8_t *)dm.p : nullptr);
        rc = run_sims<G>(e, sims);
        if (rc) return rc;
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_tree_apply<<<nblk((size_t)e->dev.n_slots * 64), 256, 0, e->stream>>>(e->dev, e->edges);
        else
            k_tree_apply<G><<<nblk((size_t)e->dev.n_slots * G::S), 256, 0, e->stream>>>(e->dev);
        HIPCHK(hipGetLastError());
        if (mask) HIPCHK(sync_all(e)); // the mask buffer is freed on return
        return BB_OK;
    });
}

extern "C" int bb_run_sims(bb_engine *e, int sims) { return run_sims_api(e, sims, nullptr); }

extern "C" int bb_run_sims_masked(bb_engine *e, int sims, const uint8_t *mask) {
    if (!mask) return fail(BB_ERR_ARG, "null mask");
    return run_sims_api(e, sims, mask);
}

template <class G>
static int sample_moves(bb_engine *e, double temp, const double *u, int32_t *action, float *wr, int32_t *rp,
                        int32_t *cact, int32_t *cplays, float *cval) {
    TreeDev d = e->dev;
    size_t n = (size_t)d.n_slots;
    if (u) {
        HIPCHK(hipMemcpyAsync(e->d_u, u, n * 8, hipMemcpyDefault, e->stream));
        d.in_u = e->d_u;
    } else {
        d.in_u = nullptr;
    }
    if constexpr (G::GID == BB_GAME_DRAGONCHESS)
        k_dc_sample<<<nblk(n * 64), 256, 0, e->stream>>>(d, e->edges, temp, e->d_child_action);
    else
        k_sample<G><<<nblk(n * G::S), 256, 0, e->stream>>>(d, temp);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_all(e));
    if (action) HIPCHK(hipMemcpy(action, d.out_action, n * 4, hipMemcpyDefault));
    if (wr) HIPCHK(hipMemcpy(wr, d.out_root_winrate, n * 4, hipMemcpyDefault));
    if (rp) HIPCHK(hipMemcpy(rp, d.out_root_plays, n * 4, hipMemcpyDefault));
    if (cplays) HIPCHK(hipMemcpy(cplays, d.out_child_plays, n * G::S * 4, hipMemcpyDefault));
    if (cval) HIPCHK(hipMemcpy(cval, d.out_child_value, n * G::S * 4, hipMemcpyDefault));
    if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
        if (cact) HIPCHK(hipMemcpy(cact, e->d_child_action, n * G::S * 4, hipMemcpyDefault));
    } else if (cact) { // dense games: slot i is action i
        std::vector<int32_t> ca(n * G::S);
        for (size_t g = 0; g < n; g++)
            for (int i = 0; i < G::S; i++) ca[g * G::S + i] = i < G::A ? i : -1;
        HIPCHK(hipMemcpy(cact, ca.data(), ca.size() * 4, hipMemcpyDefault));
    }
    return BB_OK;
}

extern "C" int bb_sample_moves(bb_engine *e, double temp, const double *u, int32_t *action_out,
                               float *root_winrate_out, int32_t *root_plays_out, int32_t *child_action_out,
                               int32_t *child_plays_out, float *child_value_out) {
    if (!e || temp < 0) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, return sample_moves<G>(e, temp, u, action_out, root_winrate_out, root_plays_out,
                                                    child_action_out, child_plays_out, child_value_out));
}

extern "C" int bb_move_roots(bb_engine *e, const int32_t *actions) {
    if (!e || !actions) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemcpyAsync(e->d_actions, actions, (size_t)e->dev.n_slots * 4, hipMemcpyDefault, e->stream));
    GAME_SWITCH(e->cfg.game, {
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_move_roots<<<nblk((size_t)e->dev.n_slots * 64), 256, 0, e->stream>>>(e->dev, e->edges, e->d_actions);
        else
            k_move_roots<G><<<nblk((size_t)e->dev.n_slots * G::S), 256, 0, e->stream>>>(e->dev, e->d_actions);
        HIPCHK(hipGetLastError());
        HIPCHK(sync_all(e));
        return BB_OK;
    });
}

extern "C" int bb_reset_roots(bb_engine *e) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    if (!e->dev.track_anc) return fail(BB_ERR_STATE, "this engine does not keep the ancestors of its roots (bb_config.track_ancestors)");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, {
        if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
            return fail(BB_ERR_ARG, "ancestors are kept for the dense-action games");
        } else {
            k_reset_roots<G><<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev);
            HIPCHK(hipGetLastError());
            HIPCHK(sync_all(e));
            return BB_OK;
        }
    });
}

extern "C" int bb_node_view(bb_engine *e, int slot, int node, int32_t *child_node_out, int32_t *child_plays_out, float *child_value_out,
                            void *state_out, int32_t *info_out) {
    if (!e || slot < 0 || slot >= e->cfg.n_slots || !child_node_out || !child_plays_out || !child_value_out || !state_out || !info_out)
        return fail(BB_ERR_ARG, "bad arguments");
    if (node >= e->dev.node_cap) return fail(BB_ERR_ARG, "node %d out of range", node);
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, {
        if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
            return fail(BB_ERR_ARG, "node views exist for the dense-action games");
        } else {
            DevBuf dc, dp, dv, ds, di;
            if (dc.alloc(G::S * 4) || dp.alloc(G::S * 4) || dv.alloc(G::S * 4) || ds.alloc(sizeof(typename G::State)) || di.alloc(16)) return BB_ERR_HIP;
            HIPCHK(sync_all(e));
            k_node_view<G><<<1, 64, 0, e->stream>>>(e->dev, slot, node, (int32_t *)dc.p, (int32_t *)dp.p, (float *)dv.p,
                                                    (typename G::State *)ds.p, (int32_t *)di.p);
            HIPCHK(hipGetLastError());
            HIPCHK(sync_all(e));
            HIPCHK(hipMemcpy(child_node_out, dc.p, G::S * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(child_plays_out, dp.p, G::S * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(child_value_out, dv.p, G::S * 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(state_out, ds.p, sizeof(typename G::State), hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(info_out, di.p, 12, hipMemcpyDeviceToHost));
            return BB_OK;
        }
    });
}

extern "C" int bb_get_root_states(bb_engine *e, void *states_out) {
    if (!e || !states_out) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, {
        DevBuf ds;
        size_t bytes = (size_t)e->dev.n_slots * sizeof(typename G::State);
        if (ds.alloc(bytes)) return BB_ERR_HIP;
        if constexpr (G::GID == BB_GAME_DRAGONCHESS)
            k_dc_get_roots<<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev, (DCState *)ds.p);
        else
            k_get_roots<G><<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev, (typename G::State *)ds.p);
        HIPCHK(hipGetLastError());
        HIPCHK(sync_all(e));
        HIPCHK(hipMemcpy(states_out, ds.p, bytes, hipMemcpyDefault));
        return BB_OK;
    });
}

extern "C" int bb_set_sims_per_move(bb_engine *e, int sims) {
    if (!e || sims <= 0) return fail(BB_ERR_ARG, "bad arguments");
    long need = (long)sims * e->cfg.max_plies + 2;
    if (e->cfg.node_capacity <= 0 && sims > e->cfg.sims_per_move && need > e->dev.node_cap)
        return fail(BB_ERR_CAPACITY, "node pool was sized for %d simulations per move", e->cfg.sims_per_move);
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    e->sims_now = sims;
    e->dev.sims_per_move = sims;
    for (int v = 0; v < e->n_views; v++) e->view[v].sims_per_move = sims;
    GAME_SWITCH(e->cfg.game, {
        k_add_sims<typename std::conditional<G::GID == BB_GAME_DRAGONCHESS, Connect4, G>::type><<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev, sims); // slots waiting for their next move
        HIPCHK(hipGetLastError());
        HIPCHK(sync_all(e));
        return BB_OK;
    });
}

extern "C" int bb_set_rng_stream(bb_engine *e, uint64_t seed, uint32_t first_game_id) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    e->cfg.seed = seed;
    e->cfg.first_game_id = first_game_id;
    e->dev.seed = seed;
    e->dev.first_game_id = first_game_id;
    for (int v = 0; v < 2; v++) {
        e->view[v].seed = seed;
        e->view[v].first_game_id = first_game_id;
    }
    e->net.seed = seed;
    return BB_OK;
}

// ---- self-play ----------------------------------------------------------------------------------------
extern "C" int bb_selfplay_begin(bb_engine *e, int n_games, double temp) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    if (n_games <= 0) return fail(BB_ERR_ARG, "Use a positive integer for number of games."); // Blackbird.py:235-236
    if (n_games > e->cfg.max_games)
        return fail(BB_ERR_CAPACITY, "n_games %d exceeds the engine's max_games %d", n_games, e->cfg.max_games);
    if (e->sims_now < 2 && temp != 0.0)
        return fail(BB_ERR_NAN, "probabilities contain NaN (a fresh root needs >= 2 simulations, MCTS.py:336-338)");
    int rc = check_eval(e);
    if (rc) return rc;
    HIPCHK(hipSetDevice(e->cfg.device));
    e->n_games_target = n_games;
    e->dev.n_games_target = n_games;
    e->dev.temp = temp;
    HIPCHK(sync_all(e));
    HIPCHK(hipMemsetAsync(e->dev.game_hdr, 0, (size_t)e->cfg.max_games * 16, e->stream));
    HIPCHK(hipMemsetAsync(e->dev.resume_cur, 0xFF, (size_t)e->dev.n_slots * 4, e->stream));
    HIPCHK(hipMemsetAsync(e->dev.post_count, 0, 8 * sizeof(int), e->stream));
    HIPCHK(sync_all(e));
    e->vround[0] = e->vround[1] = 0;
    GAME_SWITCH(e->cfg.game, {
        if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
            k_dc_selfplay_begin<<<nblk(e->dev.n_slots), 256, 0, e->stream>>>(e->dev, e->edges);
        } else {
            for (int v = 0; v < e->n_views; v++) {
                e->view[v].n_games_target = n_games;
                e->view[v].temp = temp;
                e->view[v].sims_per_move = e->sims_now;
                k_selfplay_begin<G><<<nblk(e->view[v].n_slots), 256, 0, e->vstream[v]>>>(e->view[v]);
            }
        }
        HIPCHK(hipGetLastError());
        return BB_OK;
    });
}

__global__ void k_set_i32(int *p, int v) { *p = v; }

template <class G>
static int selfplay_rounds_async(bb_engine *e, int rounds) {
    if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
        return fail(BB_ERR_ARG, "asynchronous self-play is for the dense-action games");
    } else {
        constexpr int PWMAX = NetPW<G>::v;
        if (e->mega && !e->general_net && e->net.R <= MEGA_RMAX && e->net.head_floats <= MEGA_HEAD_FLOATS) {
            // persistent launches of at most 64 steps' worth of visits each (3 s at 4096 Connect4 games x 800 visits): every
            // spin loop inside is bounded by BB_QUEUE_LIMIT_S of wall clock (30 s), so one launch must stay far below it
            // whatever the caller asks for.  Not shorter than necessary either: a launch ends when its SLOWEST workgroup has
            // given each of its games the requested visits, ~35 ms after the fastest one -- 4 % of a 16-step launch.
            TreeDev &d = e->dev;
            int nb = (d.n_slots + 15) / 16;
            int per_launch = e->tune.launch_steps * (e->sims_now > 0 ? e->sims_now : 1);
            if (per_launch > (1 << 30) / d.n_slots) per_launch = (1 << 30) / d.n_slots; // the launch's visit pool is an int
            const int all_rounds = rounds;
          for (int done_rounds = 0; done_rounds < all_rounds; done_rounds += per_launch) {
            rounds = all_rounds - done_rounds < per_launch ? all_rounds - done_rounds : per_launch;
            bool timed = e->time_every > 0 && e->ev_used + 2 <= e->ev_pool.size();
            if (timed) HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], e->stream));
            TreeDev dm = d;
            // tree levels per call: 10 / 12 / 16 / 20 / 24 -> 155.6 / 155.8 / 153.6 / 151.6 / 151.2 M sims/s (Connect4 @800, bf16-pipe network)
            if (!e->tune.level_budget_set) dm.level_budget = 12;
            // the launch's visits: 7/8 dealt to the workgroups (per slot), the rest in the launch-wide pool (mega2.hip.h)
            const int own = rounds - (rounds + 7) / 8;
            k_set_i32<<<1, 1, 0, e->stream>>>(d.visit_pool, d.n_slots * (rounds - own));
            const int lim = e->tune.queue_limit_s;
            const int netw = e->tune.queue_netw ? e->tune.queue_netw : 8; // network waves of the 12 (tuning)
            if (e->x3.w0) { // bf16-pipe network: 8 waves of 256 VGPRs -- Connect4 5 network + 3 tree waves, TicTacToe 4 + 4
                if constexpr (G::S <= 8) {
                    const int waves = e->tune.queue_waves;
                    const int nw = e->tune.queue_netw ? netw : (waves == 12 ? 8 : 5);
#define QX3(NW, WV) k_selfplay_queue<G, NW, true, WV><<<nb, WV * 64, 0, e->stream>>>(dm, e->net, e->x3, e->cfg.noise_on, lim, own)
                    if (waves == 12) { // 8 network + 4 tree waves of 168 VGPRs (default); BB_QUEUE_WAVES=8: 5 + 3 (6 + 2) waves of 256
                        if (nw == 6) QX3(6, 12);
                        else if (nw == 7) QX3(7, 12);
                        else if (nw == 9) QX3(9, 12);
                        else if (nw == 10) QX3(10, 12);
                        else QX3(8, 12);
                    } else {
                        if (nw == 4) QX3(4, 8);
                        else if (nw == 6) QX3(6, 8);
                        else QX3(5, 8);
                    }
#undef QX3
                } else {
                    k_selfplay_queue<G, 4, true, 8><<<nb, 512, 0, e->stream>>>(dm, e->net, e->x3, e->cfg.noise_on, lim, own);
                }
            } else if (netw == 7) k_selfplay_queue<G, 7><<<nb, 768, 0, e->stream>>>(dm, e->net, e->x3, e->cfg.noise_on, lim, own);
            else if (netw == 6) k_selfplay_queue<G, 6><<<nb, 768, 0, e->stream>>>(dm, e->net, e->x3, e->cfg.noise_on, lim, own);
            else k_selfplay_queue<G, 8><<<nb, 768, 0, e->stream>>>(dm, e->net, e->x3, e->cfg.noise_on, lim, own);
            HIPCHK(hipGetLastError());
            if (timed) {
                HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], e->stream));
                e->ev_used += 2;
            }
          }
            return BB_OK;
        }
        for (int r = 0; r < rounds; r++) {
            for (int v = 0; v < e->n_views; v++) {
                TreeDev &d = e->view[v];
                hipStream_t st = e->vstream[v];
                int tb = nblk((size_t)((d.n_slots + d.gpw - 1) / d.gpw) * 64);
                int nb = (d.n_slots + 4 * PWMAX - 1) / (4 * PWMAX);
                if (e->n_views == 2) nb = 256 > nb ? 256 : nb; // spread a half batch over every CU (pw <= 2)
                int round = e->vround[v]++;
                k_tree_async<G><<<tb, 256, 0, st>>>(d, round);
                HIPCHK(hipGetLastError());
                const typename G::State *ls = (const typename G::State *)d.leaf_state;
                if (d.evaluator == BB_EVAL_NET) {
                    bool timed = v == 0 && e->time_every > 0 && (e->eval_launches++ % (uint64_t)e->time_every) == 0 &&
                                 e->ev_used + 2 <= e->ev_pool.size();
                    if (timed) HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], st));
                    if (e->general_net) {
                        int rc = launch_gnet<G>(e, d.n_slots, d.post_count + (round & 3), d.post_slot, ls, nullptr, d.leaf_game_id,
                                                d.leaf_serial, e->cfg.noise_on, d.eval_value, nullptr, d.eval_policy, G::S, st,
                                                d.slot_offset);
                        if (rc) return rc;
                    } else if (e->x3.w0) {
                        if constexpr (G::C <= 4)
                            k_net_x3<G><<<(d.n_slots + 3) / 4, 256, 0, st>>>(e->net, e->x3, 0, d.post_count + (round & 3), d.post_slot, ls,
                                                                              nullptr, d.leaf_game_id, d.leaf_serial, e->cfg.noise_on,
                                                                              d.eval_value, nullptr, d.eval_policy, G::S);
                    } else
                    k_net_compact<G, PWMAX><<<nb, 256, 0, st>>>(e->net, d.post_count + (round & 3), d.post_slot, ls,
                                                                 d.leaf_game_id, d.leaf_serial, e->cfg.noise_on, d.eval_value,
                                                                 d.eval_policy, G::S);
                    HIPCHK(hipGetLastError());
                    if (timed) {
                        HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], st));
                        e->ev_used += 2;
                    }
                } else { // validation evaluator over every slot's mailbox of the view
                    k_hash_eval<G><<<nblk(d.n_slots), 256, 0, st>>>(d.n_slots, ls, d.leaf_game_id, d.salt, d.salt_per_game,
                                                                    d.first_game_id, d.eval_value, d.eval_policy, G::S);
                    HIPCHK(hipGetLastError());
                }
            }
        }
        return BB_OK;
    }
}

extern "C" int bb_selfplay_step(bb_engine *e, int plies) {
    if (!e || plies <= 0) return fail(BB_ERR_ARG, "bad arguments");
    if (e->n_games_target <= 0) return fail(BB_ERR_ARG, "bb_selfplay_begin has not been called");
    HIPCHK(hipSetDevice(e->cfg.device));
    GAME_SWITCH(e->cfg.game, {
        if (e->async_selfplay) return selfplay_rounds_async<G>(e, plies * e->sims_now);
        if constexpr (G::GID == BB_GAME_DRAGONCHESS) {
            if (e->dc_fused && e->has_weights && !e->general_net && e->net.head_floats <= DC_HEAD_FLOATS && e->net.R <= DC_RMAX) {
                // at most 64 plies' worth of simulations per launch (a launch is plies x sims x ~45 us long; nothing inside can
                // spin); the waves draw them from one pool (mega_dc.hip.h)
                int cap = 64;
                if ((long)cap * e->sims_now * e->dev.n_slots > (1l << 30)) cap = (int)((1l << 30) / ((long)e->sims_now * e->dev.n_slots));
                if (cap < 1) cap = 1;
                for (int done = 0; done < plies; done += cap) {
                    const int now = plies - done < cap ? plies - done : cap;
                    bool timed = e->time_every > 0 && e->ev_used + 2 <= e->ev_pool.size();
                    if (timed) HIPCHK(hipEventRecord(e->ev_pool[e->ev_used], e->stream));
                    const int per_slot = now * e->sims_now, own = per_slot - (per_slot + 7) / 8; // 7/8 dealt to the games, the rest pooled
                    k_set_i32<<<1, 1, 0, e->stream>>>(e->dev.visit_pool, e->dev.n_slots * (per_slot - own));
                    k_dc_selfplay_fused<<<nblk((size_t)e->dev.n_slots * 64), 256, 0, e->stream>>>(e->dev, e->edges, e->net, e->x3, e->cfg.noise_on, own);
                    HIPCHK(hipGetLastError());
                    if (timed) {
                        HIPCHK(hipEventRecord(e->ev_pool[e->ev_used + 1], e->stream));
                        e->ev_used += 2;
                    }
                }
                return BB_OK;
            }
        }
        for (int p = 0; p < plies; p++) {
            int rc = run_sims<G>(e, e->sims_now);
            if (rc) return rc;
            if constexpr (G::GID == BB_GAME_DRAGONCHESS)
                k_dc_selfplay_move<<<nblk((size_t)e->dev.n_slots * 64), 256, 0, e->stream>>>(e->dev, e->edges);
            else
                k_selfplay_move<G><<<nblk((size_t)e->dev.n_slots * G::S), 256, 0, e->stream>>>(e->dev);
            HIPCHK(hipGetLastError());
        }
        return BB_OK;
    });
}

static int sum_counters(bb_engine *e, bb_counters *out) {
    size_t n = (size_t)e->dev.n_slots * 8;
    std::vector<uint64_t> h(n);
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpy(h.data(), e->dev.ctr, n * 8, hipMemcpyDeviceToHost));
    uint64_t t[8] = {0};
    for (size_t i = 0; i < n; i++) t[i & 7] += h[i];
    std::vector<uint64_t> ev((size_t)e->dev.n_slots);
    HIPCHK(hipMemcpy(ev.data(), e->dev.evals, ev.size() * 8, hipMemcpyDeviceToHost));
    out->evals = 0;
    for (uint64_t v : ev) out->evals += v;
    out->sims = t[0];
    out->sum_depth = t[1];
    out->nodes = t[2];
    out->terminal_leaves = t[3];
    out->games_finished = t[4];
    out->plies = t[5];
    out->overflow = t[6];
    out->examples = t[7];
    return BB_OK;
}

extern "C" int bb_get_counters(bb_engine *e, bb_counters *out) {
    if (!e || !out) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    return sum_counters(e, out);
}

extern "C" int bb_reset_counters(bb_engine *e) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(hipMemsetAsync(e->dev.ctr, 0, (size_t)e->dev.n_slots * 64, e->stream));
    HIPCHK(hipMemsetAsync(e->dev.evals, 0, (size_t)e->dev.n_slots * 8, e->stream));
    return BB_OK;
}

extern "C" int bb_selfplay_done(bb_engine *e, int *done_out, int *games_finished_out) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    int ng = e->n_games_target;
    std::vector<int32_t> h((size_t)ng * 4);
    if (ng) HIPCHK(hipMemcpy(h.data(), e->dev.game_hdr, h.size() * 4, hipMemcpyDeviceToHost));
    int fin = 0;
    for (int g = 0; g < ng; g++) fin += h[(size_t)g * 4 + 3] != 0;
    if (done_out) *done_out = (ng > 0 && fin == ng) ? 1 : 0;
    if (games_finished_out) *games_finished_out = fin;
    return BB_OK;
}

extern "C" int bb_examples_fetch(bb_engine *e, int first_game, int n_games, void *records_out, int max_records,
                                 int32_t *game_offsets_out, int8_t *winner_out) {
    if (!e || first_game < 0 || n_games <= 0 || first_game + n_games > e->cfg.max_games || !records_out)
        return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    const size_t eb = (size_t)e->info.example_bytes, per = (size_t)(e->cfg.max_plies + 1) * eb;
    std::vector<int32_t> hdr((size_t)n_games * 4);
    HIPCHK(hipMemcpy(hdr.data(), e->dev.game_hdr + (size_t)first_game * 4, hdr.size() * 4, hipMemcpyDeviceToHost));
    std::vector<uint8_t> stage((size_t)n_games * per);
    HIPCHK(hipMemcpy(stage.data(), e->dev.examples + (size_t)first_game * per, stage.size(), hipMemcpyDeviceToHost));
    int total = 0;
    uint8_t *out = (uint8_t *)records_out;
    for (int g = 0; g < n_games; g++) {
        if (game_offsets_out) game_offsets_out[g] = total;
        int done = hdr[(size_t)g * 4 + 3], nex = hdr[(size_t)g * 4 + 0];
        if (winner_out) winner_out[g] = done ? (int8_t)hdr[(size_t)g * 4 + 1] : (int8_t)-2;
        if (!done) continue;
        if (total + nex > max_records) return fail(BB_ERR_CAPACITY, "records_out too small");
        memcpy(out + (size_t)total * eb, stage.data() + (size_t)g * per, (size_t)nex * eb);
        total += nex;
    }
    if (game_offsets_out) game_offsets_out[n_games] = total;
    return total;
}

extern "C" int bb_selfplay_headers(bb_engine *e, int first_game, int n_games, int32_t *hdr_out) {
    if (!e || first_game < 0 || n_games <= 0 || first_game + n_games > e->cfg.max_games || !hdr_out) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpy(hdr_out, e->dev.game_hdr + (size_t)first_game * 4, (size_t)n_games * 16, hipMemcpyDeviceToHost));
    return BB_OK;
}

// record r of the compacted output <- record (r - off[g]) of game ids[g]: one thread per 16 bytes
__global__ void __launch_bounds__(256) k_gather_examples(const uint8_t *store, size_t game_stride, int eb16, int n, const int32_t *ids,
                                                         const int32_t *off, uint4 *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)off[n] * eb16;
    if (i >= total) return;
    const int rec = (int)(i / eb16), part = (int)(i % eb16);
    int lo = 0, hi = n - 1; // the game this record belongs to: last g with off[g] <= rec
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= rec) lo = mid;
        else hi = mid - 1;
    }
    out[i] = ((const uint4 *)(store + (size_t)ids[lo] * game_stride))[(size_t)(rec - off[lo]) * eb16 + part];
}

extern "C" int bb_examples_fetch_games(bb_engine *e, int n, const int32_t *game_ids, void *records_out, int max_records,
                                       int32_t *game_offsets_out, int8_t *winner_out) {
    if (!e || n <= 0 || !game_ids || !records_out) return fail(BB_ERR_ARG, "bad arguments");
    HIPCHK(hipSetDevice(e->cfg.device));
    HIPCHK(sync_all(e));
    const size_t eb = (size_t)e->info.example_bytes, per = (size_t)(e->cfg.max_plies + 1) * eb;
    if (eb % 16) return fail(BB_ERR_ARG, "example records are copied in 16-byte units");
    std::vector<int32_t> hdr((size_t)e->cfg.max_games * 4);
    HIPCHK(hipMemcpy(hdr.data(), e->dev.game_hdr, hdr.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> off((size_t)n + 1, 0);
    for (int g = 0; g < n; g++) {
        const int id = game_ids[g];
        if (id < 0 || id >= e->cfg.max_games) return fail(BB_ERR_ARG, "game id %d out of range", id);
        const int done = hdr[(size_t)id * 4 + 3];
        if (winner_out) winner_out[g] = done ? (int8_t)hdr[(size_t)id * 4 + 1] : (int8_t)-2;
        off[g + 1] = off[g] + (done ? hdr[(size_t)id * 4 + 0] : 0);
    }
    if (game_offsets_out) memcpy(game_offsets_out, off.data(), off.size() * 4);
    const int total = off[n];
    if (total > max_records) return fail(BB_ERR_CAPACITY, "records_out too small");
    if (total == 0) return 0;
    DevBuf d_ids, d_off, d_out;
    if (d_ids.alloc((size_t)n * 4) || d_off.alloc(((size_t)n + 1) * 4) || d_out.alloc((size_t)total * eb)) return BB_ERR_HIP;
    HIPCHK(hipMemcpy(d_ids.p, game_ids, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(d_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice));
    k_gather_examples<<<nblk((size_t)total * (eb / 16)), 256, 0, e->stream>>>(e->dev.examples, per, (int)(eb / 16), n, (const int32_t *)d_ids.p,
                                                                            (const int32_t *)d_off.p, (uint4 *)d_out.p);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpy(records_out, d_out.p, (size_t)total * eb, hipMemcpyDeviceToHost));
    return total;
}

extern "C" int bb_examples_device(bb_engine *e, void **ptr_out, uint64_t *bytes_out, uint64_t *record_bytes_out,
                                  int32_t **game_hdr_out) {
    if (!e) return fail(BB_ERR_ARG, "null engine");
    if (ptr_out) *ptr_out = e->dev.examples;
    if (bytes_out) *bytes_out = (uint64_t)e->cfg.max_games * (uint64_t)(e->cfg.max_plies + 1) * (uint64_t)e->info.example_bytes;
    if (record_bytes_out) *record_bytes_out = (uint64_t)e->info.example_bytes;
    if (game_hdr_out) *game_hdr_out = e->dev.game_hdr;
    return BB_OK;
}

extern "C" int bb_timing_net(bb_engine *e, int iters, int noise, int ablate, double *ms_per_launch_out) {
    if (!e || iters <= 0) return fail(BB_ERR_ARG, "bad arguments");
    if (!e->has_weights) return fail(BB_ERR_WEIGHTS, "bb_load_weights has not been called");
#ifndef BB_DIAG
    if (ablate) return fail(BB_ERR_ARG, "the ablation switches exist in diagnostic builds only (-DBB_DIAG)");
#endif
    HIPCHK(hipSetDevice(e->cfg.device));
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    TreeDev &d = e->dev;
    int saved = e->net.dbg;
    e->net.dbg = ablate;
    int rc = BB_OK;
    GAME_SWITCH(e->cfg.game, {
        const typename G::State *ls = (const typename G::State *)d.leaf_state;
        for (int i = 0; i < 3 && !rc; i++)
            rc = launch_net<G>(e, d.n_slots, ls, nullptr, d.leaf_game_id, d.leaf_serial, noise, d.eval_value, nullptr,
                               d.eval_policy, G::S, e->stream);
        HIPCHK(hipEventRecord(a, e->stream));
        for (int i = 0; i < iters && !rc; i++)
            rc = launch_net<G>(e, d.n_slots, ls, nullptr, d.leaf_game_id, d.leaf_serial, noise, d.eval_value, nullptr,
                               d.eval_policy, G::S, e->stream);
        HIPCHK(hipEventRecord(b, e->stream));
        break;
    });
    e->net.dbg = saved;
    if (rc) return rc;
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (ms_per_launch_out) *ms_per_launch_out = (double)ms / iters;
    return BB_OK;
}


#ifdef BB_STAMPS
// diagnostic builds only: point the stamp buffer at host-coherent memory so that it can be read while a kernel runs
extern "C" int bb_debug_host_stamps(bb_engine *e, unsigned long long **host_out) {
    unsigned long long *p = nullptr;
    HIPCHK(hipHostMalloc((void **)&p, 16 * 64 * 8, hipHostMallocCoherent | hipHostMallocMapped));
    memset(p, 0, 16 * 64 * 8);
    e->dev.stamps = p;
    for (int v = 0; v < 2; v++) e->view[v].stamps = p;
    *host_out = p;
    return BB_OK;
}
extern "C" int bb_debug_net_stamps(bb_engine *e, unsigned long long *out8) {
    HIPCHK(sync_all(e));
    HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_net_stamps), 64));
    unsigned long long z[8] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_net_stamps), z, 64));
    return BB_OK;
}
extern "C" int bb_stream_done(bb_engine *e) { return hipStreamQuery(e->stream) == hipSuccess ? 1 : 0; }
// diagnostic builds only (tools/): in-kernel cycle stamps accumulated by the tree / persistent kernels
extern "C" int bb_debug_stamps(bb_engine *e, unsigned long long *out8) {
    HIPCHK(sync_all(e));
    unsigned long long all[16 * 64]; // 64 copies (the DragonChess kernels spread their flushes), summed here
    HIPCHK(hipMemcpy(all, e->dev.stamps, sizeof(all), hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) {
        out8[i] = 0;
        for (int c = 0; c < 64; c++) out8[i] += all[c * 16 + i];
    }
    HIPCHK(hipMemset(e->dev.stamps, 0, sizeof(all)));
    return BB_OK;
}
#endif
