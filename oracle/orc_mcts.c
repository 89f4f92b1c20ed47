/*
 * oracle/orc_mcts.c -- CPU ORACLE (test infrastructure, not product code).
 * Restatement of MCTS.py / DynamicMCTS.py / FixedMCTS.py and of the Model overrides and the
 * self-play loop in Blackbird.py.  Citations are relative to /root/reference/src/.
 *
 * Numeric types follow what the reference computes under numpy >= 2 (NEP 50) when the
 * evaluator returns numpy.float32 (Network.getEvaluation, Network.py:48-54):
 *   - SampleValue arithmetic, Node.Value accumulation and Node.WinRate() are float32,
 *   - ChildWinRates/ChildPlays/Priors and the PUCT expression are float64.
 * With the base-class rollout evaluator (MCTS.py:360-383) values are Python floats/ints,
 * so Value and WinRate are float64 there.
 */
#include "orc.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- Philox4x32-10 (Salmon et al. 2011); shared RNG spec with the HIP engine --------- */
void orc_philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define TAG_MOVE 0x4D4F5645u /* 'MOVE' */
#define TAG_NOISE 0x4E4F4953u /* 'NOIS' */
#define TAG_ROLL 0x524F4C4Cu /* 'ROLL' */

/* 53-bit uniform in [0,1), built like numpy's random_sample: (a>>5, b>>6) */
double orc_u53(uint64_t key, uint32_t game_id, uint32_t ply) {
    uint32_t x[4];
    orc_philox(key, game_id, ply, TAG_MOVE, 0, x);
    return ((double)(x[0] >> 5) * 67108864.0 + (double)(x[1] >> 6)) / 9007199254740992.0;
}

/* One Beta(alpha, 1-alpha) draw (NetworkFactory.py:176-180: Dirichlet([a,1-a]) first coord),
 * Johnk's method: X=U^(1/a), Y=V^(1/b); accept when X+Y<=1; return X/(X+Y). */
float orc_beta_noise(uint64_t key, uint32_t game_id, uint32_t node_serial, uint32_t action, float alpha) {
    float ia = 1.0f / alpha, ib = 1.0f / (1.0f - alpha);
    for (uint32_t trial = 0; trial < 64; trial++) { /* trial t: counter sub-index t/2, words 2(t&1), 2(t&1)+1 */
        uint32_t x[4];
        orc_philox(key, game_id, node_serial, TAG_NOISE, action * 64u + trial / 2, x);
        uint32_t h = trial & 1u;
        float u = ((float)(x[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float v = ((float)(x[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float X = powf(u, ia), Y = powf(v, ib);
        if (X + Y <= 1.0f && X + Y > 0.0f) return X / (X + Y);
    }
    return alpha;
}

/* numpy add.reduce over a contiguous float64 vector == pairwise sum
 * (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum; checked against numpy 2.2.6
 * in tests/test_oracle_games.py::test_np_sum_matches_numpy) */
double orc_np_sum(const double *a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return orc_np_sum(a, n2) + orc_np_sum(a + n2, n - n2);
    }
}

/* ---- deterministic synthetic evaluator ------------------------------------------------
 * value/policy are integer hashes of the AsInputArray bytes, exactly representable in
 * float32, so that the Python reference, this oracle and the HIP engine all see bit-identical
 * evaluator outputs (tree-search parity then has to be exact). */
static uint64_t splitmix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

void orc_hash_eval(int game, uint64_t salt, const orc_state *st, float *value, float *policy) {
    orc_dims d;
    orc_game_dims(game, &d);
    int8_t enc[8 * 8 * 17];
    orc_game_encode(game, st, enc);
    uint64_t h = 0xcbf29ce484222325ull ^ salt;
    int n = d.H * d.W * d.C;
    for (int i = 0; i < n; i++) {
        h ^= (uint8_t)enc[i];
        h *= 0x100000001b3ull;
    }
    uint64_t z = splitmix(h);
    *value = (float)(int32_t)(z >> 40) * (1.0f / 8388608.0f) - 1.0f;
    if (policy)
        for (int a = 0; a < d.A; a++) {
            uint64_t za = splitmix(z + (uint64_t)(a + 1) * 0x9e3779b97f4a7c15ull);
            policy[a] = (float)(int32_t)(1 + (za >> 44));
        }
}

/* ---- tree ---------------------------------------------------------------------------- */
typedef struct orc_node {
    orc_state st;
    int N;        /* Node.Plays */
    float Wf;     /* Node.Value when values are float32 */
    double Wd;    /* Node.Value when values are Python floats */
    struct orc_node *parent;
    int has_children; /* Children is not None */
    int prepared;     /* LegalActions / Priors known */
    int nlegal;       /* np.sum(LegalActions) */
    int *act;         /* legal action ids, ascending */
    double *prior;    /* Node.Priors restricted to legal ids (illegal ids hold exactly 0) */
    struct orc_node **ch;
    uint32_t serial; /* order of Node.__init__ calls (AddChildren creates every child at once, MCTS.py:136-138) */
    int32_t visit;   /* order in which searches first REACH nodes (root = 0); -1 = constructed but never reached.  This is
                      * the node's index in the HIP engine's pool (it materialises a child when a descent first selects it),
                      * and the key of the node's prior-noise stream in the RNG spec both sides share. */
} orc_node;

typedef struct arena_blk {
    struct arena_blk *next;
    size_t used, cap;
    char data[];
} arena_blk;

struct orc_search {
    orc_cfg cfg;
    orc_dims d;
    uint32_t game_id;
    orc_node *root;
    arena_blk *arena;
    orc_stats stats;
    uint32_t node_serial, sim_serial;
    int32_t visit_next;
    int value_f32;
    double *tmpA; /* dense scratch, A doubles */
    float *tmpP;  /* dense scratch, A floats */
};

static void *arena_alloc(orc_search *s, size_t n) {
    n = (n + 15) & ~(size_t)15;
    if (!s->arena || s->arena->used + n > s->arena->cap) {
        size_t cap = n > (1u << 20) ? n : (1u << 20);
        arena_blk *b = (arena_blk *)malloc(sizeof(arena_blk) + cap);
        b->next = s->arena;
        b->used = 0;
        b->cap = cap;
        s->arena = b;
    }
    void *p = s->arena->data + s->arena->used;
    s->arena->used += n;
    return p;
}

static void arena_free(orc_search *s) {
    while (s->arena) {
        arena_blk *n = s->arena->next;
        free(s->arena);
        s->arena = n;
    }
}

orc_search *orc_search_new(const orc_cfg *cfg, uint32_t game_id) {
    orc_search *s = (orc_search *)calloc(1, sizeof(*s));
    s->cfg = *cfg;
    orc_game_dims(cfg->game, &s->d);
    s->game_id = game_id;
    s->value_f32 = cfg->evaluator != ORC_EVAL_ROLLOUT;
    s->tmpA = (double *)malloc(sizeof(double) * (size_t)s->d.A);
    s->tmpP = (float *)malloc(sizeof(float) * (size_t)s->d.A);
    return s;
}

void orc_search_free(orc_search *s) {
    if (!s) return;
    arena_free(s);
    free(s->tmpA);
    free(s->tmpP);
    free(s);
}

void orc_drop_root(orc_search *s) { /* MCTS.DropRoot, MCTS.py:141-144 */
    s->root = NULL;
    arena_free(s); /* the reference leaves the old tree to the GC */
}

int orc_has_root(const orc_search *s) { return s->root != NULL; }
void orc_get_stats(const orc_search *s, orc_stats *out) { *out = s->stats; }

static orc_node *node_new(orc_search *s, const orc_state *st) { /* Node.__init__, MCTS.py:31-41 */
    orc_node *n = (orc_node *)arena_alloc(s, sizeof(orc_node));
    memset(n, 0, sizeof(*n));
    n->st = *st;
    n->serial = s->node_serial++;
    n->visit = -1;
    s->stats.nodes++;
    return n;
}

/* evaluator front: value in [-1,1] for the side to move + getPolicy()-shaped vector */
static void evaluate(orc_search *s, const orc_node *n, float *value, float *policy) {
    const orc_cfg *c = &s->cfg;
    s->stats.evals++;
    if (c->evaluator == ORC_EVAL_HASH) {
        orc_hash_eval(c->game, c->salt, &n->st, value, policy);
    } else if (c->evaluator == ORC_EVAL_CALLBACK) {
        c->cb(c->cb_ctx, &n->st, value, policy);
    } else if (c->evaluator == ORC_EVAL_CALLBACK_KEYED) { /* the caller's getPolicy draws the noise of (game, node) itself */
        c->cb2(c->cb_ctx, &n->st, s->game_id, (uint32_t)n->visit, value, policy);
    } else if (c->evaluator == ORC_EVAL_NET) {
        int8_t enc[8 * 8 * 17];
        orc_game_encode(c->game, &n->st, enc);
        float logits[4032];
        orc_net_forward(c->net, enc, 1, value, logits, policy);
        if (c->noise_on && policy) { /* NetworkFactory.py:176-182 */
            float tot = 0.f;
            for (int a = 0; a < s->d.A; a++) {
                float nz = orc_beta_noise(c->seed, s->game_id, (uint32_t)n->visit, (uint32_t)a, c->alpha);
                policy[a] = (1.0f - c->eps) * policy[a] + c->eps * nz;
                tot += policy[a];
            }
            for (int a = 0; a < s->d.A; a++) policy[a] /= tot;
        }
    }
}

/* Fill LegalActions and Priors of a node.
 * Model.GetPriors (Blackbird.py:372-389): policy = getPolicy(x) * LegalActions; policy /= np.sum(policy)
 * MCTS.GetPriors  (MCTS.py:346-358): ones -> Node.Priors = legal mask (unnormalised).
 * The reference evaluates a child's priors when its parent is expanded (MCTS.py:137); they are
 * first READ when the child itself has children, so evaluating them here, at the child's own
 * expansion, is observationally identical for a deterministic evaluator.  `policy` may carry
 * the evaluator output if the caller already has it. */
static void node_prepare(orc_search *s, orc_node *n, const float *policy) {
    if (n->prepared) return;
    int A = s->d.A;
    double *legal = s->tmpA;
    orc_game_legal(s->cfg.game, &n->st, legal);
    int nl = 0;
    for (int a = 0; a < A; a++) nl += legal[a] == 1.0;
    n->nlegal = nl;
    n->act = (int *)arena_alloc(s, sizeof(int) * (size_t)(nl ? nl : 1));
    n->prior = (double *)arena_alloc(s, sizeof(double) * (size_t)(nl ? nl : 1));
    int k = 0;
    for (int a = 0; a < A; a++)
        if (legal[a] == 1.0) n->act[k++] = a;
    if (s->cfg.evaluator == ORC_EVAL_ROLLOUT || s->cfg.priors_ones) {
        for (k = 0; k < nl; k++) n->prior[k] = 1.0;
    } else {
        float v;
        if (!policy) {
            evaluate(s, n, &v, s->tmpP);
            policy = s->tmpP;
        }
        for (int a = 0; a < A; a++) legal[a] = (double)policy[a] * legal[a]; /* f32 * f64 -> f64 */
        double tot = orc_np_sum(legal, A);
        for (k = 0; k < nl; k++) n->prior[k] = legal[n->act[k]] / tot;
    }
    n->prepared = 1;
}

/* MCTS.AddChildren, MCTS.py:122-139 */
static void add_children(orc_search *s, orc_node *n, const float *policy) {
    node_prepare(s, n, policy);
    n->ch = (orc_node **)arena_alloc(s, sizeof(orc_node *) * (size_t)(n->nlegal ? n->nlegal : 1));
    for (int k = 0; k < n->nlegal; k++) {
        orc_state c = n->st; /* _applyAction: Copy + ApplyAction, MCTS.py:227-236 */
        if (s->cfg.game != ORC_DC) c.prev = 0; /* Connect4/TicTacToe Copy() drops PreviousPlayer */
        orc_game_apply(s->cfg.game, &c, n->act[k]);
        n->ch[k] = node_new(s, &c);
        n->ch[k]->parent = n;
    }
    n->has_children = 1;
}

static double node_winrate(const orc_search *s, const orc_node *n) { /* Node.WinRate, MCTS.py:43-53 */
    if (n->N <= 0) return 0.0;
    if (s->value_f32) return (double)(n->Wf / (float)n->N); /* np.float32 / int -> np.float32 */
    return n->Wd / (double)n->N;
}

/* _selectAction, exploring branch, MCTS.py:327-334.  Returns child slot k (legal-list index). */
static int select_puct(const orc_search *s, const orc_node *n) {
    double allPlays = 0.0; /* sum(root.ChildPlays()) : integers, exact */
    for (int k = 0; k < n->nlegal; k++) allPlays += (double)n->ch[k]->N;
    double sq = sqrt(1.0 + allPlays);
    int best = -1;
    double bestU = 0.0; /* illegal ids score exactly 0.0; np.argmax keeps the first maximum */
    for (int k = 0; k < n->nlegal; k++) {
        double q = node_winrate(s, n->ch[k]);
        double u = q + ((s->cfg.c_puct * n->prior[k]) * sq) / (1.0 + (double)n->ch[k]->N);
        if (best < 0 || u > bestU) {
            /* a legal id whose score is exactly 0 would lose to an earlier illegal id in the
             * reference (which then trips its assert, MCTS.py:341); not reproduced. */
            best = k;
            bestU = u;
        }
    }
    return best;
}

/* sampling branch of _selectAction (MCTS.py:335-338) + np.random.choice(p=...) law:
 * cdf = cumsum(p); cdf /= cdf[-1]; index = searchsorted(cdf, u, side='right') */
int orc_sample_action(const double *child_plays, int A, double temp, double u) {
    double it = 1.0 / temp;
    double allPlays = 0.0;
    for (int a = 0; a < A; a++) allPlays += (it == 1.0) ? child_plays[a] : pow(child_plays[a], it);
    if (!(allPlays > 0.0)) return -3; /* ValueError: probabilities contain NaN */
    double cdf_last = 0.0;
    for (int a = 0; a < A; a++) {
        double p = ((it == 1.0) ? child_plays[a] : pow(child_plays[a], it)) / allPlays;
        cdf_last += p;
    }
    double run = 0.0;
    for (int a = 0; a < A; a++) {
        double p = ((it == 1.0) ? child_plays[a] : pow(child_plays[a], it)) / allPlays;
        run += p;
        if (run / cdf_last > u) return a;
    }
    return A - 1;
}

/* Model.SampleValue (Blackbird.py:350-370) / MCTS.SampleValue rollouts (MCTS.py:360-383).
 * Returns the value for `player` (= leaf.State.PreviousPlayer, 0 == None). */
static double sample_value(orc_search *s, orc_node *leaf, const float *net_value) {
    const orc_cfg *c = &s->cfg;
    int player = leaf->st.prev;
    if (c->evaluator == ORC_EVAL_ROLLOUT) {
        orc_state r = leaf->st;
        int winner = orc_game_winner(c->game, &r, -1);
        uint32_t step = 0;
        double *legal = s->tmpA;
        while (winner < 0) {
            orc_game_legal(c->game, &r, legal);
            int n = 0;
            for (int a = 0; a < s->d.A; a++) n += legal[a] == 1.0;
            uint32_t x[4];
            orc_philox(c->seed, s->game_id, s->sim_serial, TAG_ROLL, step++, x);
            int pick = (int)(((uint64_t)x[0] * (uint64_t)n) >> 32);
            int action = -1;
            for (int a = 0; a < s->d.A; a++)
                if (legal[a] == 1.0 && pick-- == 0) { action = a; break; }
            if (c->game != ORC_DC) r.prev = 0;
            orc_game_apply(c->game, &r, action);
            winner = orc_game_winner(c->game, &r, action);
        }
        return winner == 0 ? 0.5 : (double)(player == winner);
    }
    float value;
    if (net_value) value = *net_value;
    else evaluate(s, leaf, &value, NULL);
    value = (value + 1.0f) * 0.5f; /* float32 arithmetic under NEP 50 */
    if (leaf->st.player != player) value = 1.0f - value;
    return (double)value;
}

/* MCTS._backProp, MCTS.py:238-258 (recursion written as a loop; walks past the current root
 * through stale ancestors exactly like the reference, whose _moveRoot never clears Parent) */
static void back_prop(orc_node *leaf, double v, int player_for_value) {
    float vf = (float)v;
    for (orc_node *n = leaf; n; n = n->parent) {
        n->N += 1;
        if (!n->parent) break;
        if (n->parent->st.player == player_for_value) {
            n->Wf += vf;
            n->Wd += v;
        } else {
            n->Wf += 1.0f - vf;
            n->Wd += 1.0 - v;
        }
    }
}

/* One simulation: _findLeaf + SampleValue + _backProp (MCTS.py:298-303) */
static void run_sim(orc_search *s) {
    const orc_cfg *c = &s->cfg;
    orc_node *node = s->root;
    int last_action = -1;
    int depth = 0;
    float net_value = 0.f;
    int have_value = 0;
    if (c->kind == ORC_DYNAMIC) { /* DynamicMCTS._findLeaf, DynamicMCTS.py:14-34 */
        for (;;) {
            if (!node->has_children) {
                if (orc_game_winner(c->game, &node->st, last_action) >= 0) {
                    s->stats.terminal_leaves++;
                    break;
                }
                if (c->evaluator != ORC_EVAL_ROLLOUT) { /* one evaluator call yields value + priors */
                    evaluate(s, node, &net_value, s->tmpP);
                    have_value = 1;
                    add_children(s, node, s->tmpP);
                } else {
                    add_children(s, node, NULL);
                }
                break;
            }
            if (node->nlegal == 0) break;
            int k = select_puct(s, node);
            last_action = node->act[k];
            node = node->ch[k];
            if (node->visit < 0) node->visit = s->visit_next++;
            depth++;
        }
    } else { /* FixedMCTS._findLeaf, FixedMCTS.py:21-34 */
        for (int it = 0; it < c->max_depth; it++) {
            if (!node->has_children) {
                if (orc_game_winner(c->game, &node->st, last_action) >= 0) break;
                add_children(s, node, NULL);
            }
            if (node->nlegal == 0) break;
            int k = select_puct(s, node);
            last_action = node->act[k];
            node = node->ch[k];
            if (node->visit < 0) node->visit = s->visit_next++;
            depth++;
        }
    }
    double v = sample_value(s, node, have_value ? &net_value : NULL);
    back_prop(node, v, node->st.prev);
    s->stats.sims++;
    s->sim_serial++;
    s->stats.sum_depth += (uint64_t)depth;
    if (depth > s->stats.max_depth_seen) s->stats.max_depth_seen = depth;
}

int orc_select_puct(orc_search *s) {
    if (!s->root || !s->root->has_children) return -4;
    return s->root->act[select_puct(s, s->root)];
}

int orc_find_move(orc_search *s, const orc_state *st, double temp, int play_limit, double u,
                  uint32_t ply, int *action, orc_state *next, double *root_winrate,
                  double *child_prob, double *child_plays, double *child_winrates,
                  double *root_plays) { /* MCTS.FindMove, MCTS.py:146-199 */
    if (play_limit <= 0) return -4; /* ValueError: no stop rule (time limits are host-side) */
    int A = s->d.A;
    if (!s->root) { /* :184-186 */
        s->root = node_new(s, st);
        s->root->visit = 0;
        s->visit_next = 1;
    }
    if (!orc_game_equal(s->cfg.game, &s->root->st, st)) return -2; /* assert, :193 */
    int end_plays = s->root->N + play_limit; /* _runMCTS, :297-303 */
    while (s->root->N < end_plays) run_sim(s);

    orc_node *r = s->root;
    double *plays = s->tmpA;
    for (int a = 0; a < A; a++) plays[a] = 0.0;
    double all = 0.0;
    for (int k = 0; k < r->nlegal; k++) {
        plays[r->act[k]] = (double)r->ch[k]->N;
        all += (double)r->ch[k]->N;
    }
    int act;
    if (temp == 0.0) {
        act = r->act[select_puct(s, r)];
    } else {
        if (u < 0.0) u = orc_u53(s->cfg.seed, s->game_id, ply);
        act = orc_sample_action(plays, A, temp, u);
        if (act < 0) return act;
    }
    if (action) *action = act;
    if (next) {
        *next = *st;
        if (s->cfg.game != ORC_DC) next->prev = 0;
        orc_game_apply(s->cfg.game, next, act);
    }
    if (root_winrate) *root_winrate = node_winrate(s, r);
    if (root_plays) *root_plays = (double)r->N;
    for (int a = 0; a < A; a++) {
        if (child_plays) child_plays[a] = plays[a];
        if (child_prob) child_prob[a] = all > 0 ? plays[a] / all : 0.0; /* Node.ChildProbability, :55-68 */
        if (child_winrates) child_winrates[a] = 0.0;
    }
    if (child_winrates)
        for (int k = 0; k < r->nlegal; k++) child_winrates[r->act[k]] = node_winrate(s, r->ch[k]);
    return 0;
}

int orc_move_root(orc_search *s, const orc_state *st) { /* MCTS._moveRoot, MCTS.py:260-282 */
    if (!s->root) return 0;
    if (!s->root->has_children) {
        s->root = NULL;
        return 0;
    }
    for (int k = 0; k < s->root->nlegal; k++)
        if (orc_game_equal(s->cfg.game, &s->root->ch[k]->st, st)) {
            s->root = s->root->ch[k];
            if (s->root->visit < 0) s->root->visit = s->visit_next++; /* a move no simulation ever tried */
            return 1;
        }
    return 0;
}

int orc_selfplay_game(const orc_cfg *cfg, uint32_t game_id, double temp, int play_limit,
                      int max_plies, int8_t *boards, double *pi, int8_t *player, float *z,
                      int *actions, int *winner_out, orc_stats *stats) { /* Blackbird.py:238-268 */
    orc_search *s = orc_search_new(cfg, game_id);
    orc_dims d = s->d;
    int enc = d.H * d.W * d.C;
    orc_state st;
    orc_game_init(cfg->game, &st); /* state = model.Game() */
    int winner = -1;
    int n = 0;
    orc_drop_root(s);
    while (winner < 0 && n < max_plies) {
        orc_state next;
        int act;
        int rc = orc_find_move(s, &st, temp, play_limit, -1.0, (uint32_t)n, &act, &next, NULL,
                               pi + (size_t)n * d.A, NULL, NULL, NULL);
        if (rc < 0) {
            orc_search_free(s);
            return rc;
        }
        orc_game_encode(cfg->game, &st, boards + (size_t)n * enc); /* ExampleState(..., state.AsInputArray(), player) */
        player[n] = st.player;
        if (actions) actions[n] = act;
        st = next;
        orc_move_root(s, &st);
        winner = orc_game_winner(cfg->game, &st, -1); /* lastAction is always None, Blackbird.py:242,253 */
        n++;
    }
    /* terminal example with pi = zeros, Blackbird.py:256-258 */
    orc_game_encode(cfg->game, &st, boards + (size_t)n * enc);
    for (int a = 0; a < d.A; a++) pi[(size_t)n * d.A + a] = 0.0;
    player[n] = st.player;
    n++;
    for (int i = 0; i < n; i++) { /* Blackbird.py:260-264 */
        if (winner <= 0) z[i] = 0.f; /* draw (or ply cap: documented deviation) */
        else z[i] = (player[i] == winner) ? 1.f : -1.f;
    }
    if (winner_out) *winner_out = winner;
    if (stats) *stats = s->stats;
    orc_search_free(s);
    return n;
}
