/*
 * oracle/orc.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the BlackBird hot path
 *   Blackbird.GenerateTrainingSamples -> MCTS.FindMove -> Network.getEvaluation/getPolicy
 * written to follow the reference's Python line by line (arrays, loops and evaluation
 * order included) so that it can serve as the parity checker for the HIP engine.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (blackbird_amd/) never links or imports it.
 *
 * Pinning: every function here is checked against golden vectors produced by importing
 * the reference's own Python modules (tests/make_golden.py, fixtures in tests/golden/),
 * EXCEPT the network forward (Network.py/NetworkFactory.py need TensorFlow, which is not
 * installable here): that part is "parity unpinned" by the reference and is instead
 * cross-checked against independent PyTorch-CPU ops (tests/test_oracle_net.py).
 *
 * All reference citations are relative to /root/reference/src/.
 */
#ifndef ORC_H
#define ORC_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_C4 = 0, ORC_TTT = 1, ORC_DC = 2 };
enum { ORC_DYNAMIC = 0, ORC_FIXED = 1 };
/* evaluator kinds */
enum { ORC_EVAL_HASH = 0,    /* deterministic synthetic evaluator (parity tests)            */
       ORC_EVAL_NET = 1,     /* residual tower, Network.py:48-64 + NetworkFactory.py:22-183 */
       ORC_EVAL_ROLLOUT = 2, /* base-class MCTS: priors = ones, value = random rollouts     */
       ORC_EVAL_CALLBACK = 3, /* test hook: evaluator supplied by the caller                */
       ORC_EVAL_CALLBACK_KEYED = 4 /* the same, and the caller is told which (game, node) it evaluates so that its
                                      getPolicy can mix in that node's prior noise (NetworkFactory.py:176-182) */ };

#define ORC_MAX_CELLS 128

/* One game position, laid out like the reference objects:
 *  Connect4:   b = Board[6][7][2] int8 (Connect4.py:20), row 0 = bottom
 *  TicTacToe:  b = Board[3][3][2] int8 (TicTacToe.py:19)
 *  DragonChess: b = board[8][8] signed piece codes (DragonChess.py:39), row 0 = White's back rank
 * player/prev: 1|2, prev 0 == None. castle = {wK, wQ, bK, bQ} (DragonChess.py:56-59). */
typedef struct {
    int8_t b[ORC_MAX_CELLS];
    int8_t player;
    int8_t prev;
    int8_t castle[4];
    int8_t pad[2];
} orc_state;

typedef struct {
    int H, W, C, A; /* board height/width, input planes, LegalMoves */
} orc_dims;

/* ---- network weights (TF variable layout, NetworkFactory.py:37-183) -------------- */
typedef struct {
    int H, W, C, F, R, D, A;
    const float *conv0_k;  /* [3][3][C][F] HWIO  resTower/conv_block/conv/kernel */
    const float *conv0_b;  /* [F] */
    const float *conv0_bn; /* [4][F] gamma,beta,moving_mean,moving_variance */
    const float *blk_k;    /* [R][2][3][3][F][F] */
    const float *blk_b;    /* [R][2][F] */
    const float *blk_bn;   /* [R][2][4][F] */
    const float *v_conv_k; /* [F]      value/convolution/kernel [1,1,F,1] */
    const float *v_conv_b; /* [1] */
    const float *v_bn;     /* [4][1] */
    const float *v_d1_k;   /* [D]      value/dense_1/kernel [1,D] */
    const float *v_d1_b;   /* [D] */
    const float *v_d2_k;   /* [D]      value/dense_2/kernel [D,1] */
    const float *v_d2_b;   /* [1] */
    const float *p_conv_k; /* [F][2]   policy/convolution/kernel [1,1,F,2] */
    const float *p_conv_b; /* [2] */
    const float *p_bn;     /* [4][2] */
    const float *p_d_k;    /* [2][A]   policy/policy/kernel */
    const float *p_d_b;    /* [A] */
} orc_net;

/* value in [-1,1] from the side-to-move's perspective, policy[A] as getPolicy would return */
typedef void (*orc_eval_cb)(void *ctx, const orc_state *st, float *value, float *policy);

/* keyed form: node_serial = order in which the search first reached the node (root = 0), see orc_mcts.c */
typedef void (*orc_eval_cb2)(void *ctx, const orc_state *st, uint32_t game_id, uint32_t node_serial, float *value,
                             float *policy);

typedef struct {
    int game;
    int kind;       /* ORC_DYNAMIC / ORC_FIXED */
    int max_depth;  /* FixedMCTS.MaxDepth */
    int evaluator;
    double c_puct;  /* MCTS.ExplorationRate */
    uint64_t salt;  /* hash evaluator salt */
    uint64_t seed;  /* philox key */
    const orc_net *net;
    int noise_on;   /* NetworkFactory.py:176-180 */
    float alpha, eps;
    orc_eval_cb cb;
    void *cb_ctx;
    int priors_ones; /* MCTS.GetPriors default (MCTS.py:346-358): ones -> Priors = legal mask */
    orc_eval_cb2 cb2; /* ORC_EVAL_CALLBACK_KEYED */
} orc_cfg;

typedef struct {
    uint64_t sims, evals, sum_depth, nodes, terminal_leaves;
    int max_depth_seen;
} orc_stats;

typedef struct orc_search orc_search;

/* ---- games ----------------------------------------------------------------------- */
void orc_game_dims(int game, orc_dims *d);
void orc_game_init(int game, orc_state *st);
void orc_game_legal(int game, const orc_state *st, double *out /*A*/);
int  orc_game_apply(int game, orc_state *st, int action);             /* 0 ok, -1 ValueError */
int  orc_game_winner(int game, const orc_state *st, int prev_action); /* -1 None, else 0/1/2; prev_action <0 == None */
void orc_game_encode(int game, const orc_state *st, int8_t *out /*H*W*C*/);
int  orc_game_equal(int game, const orc_state *a, const orc_state *b);
int  orc_dc_is_legal(const orc_state *st, int r1, int c1, int r2, int c2);

/* ---- evaluators ------------------------------------------------------------------ */
void orc_hash_eval(int game, uint64_t salt, const orc_state *st, float *value, float *policy);
void orc_net_forward(const orc_net *w, const int8_t *boards /*[n][H][W][C]*/, int n,
                     float *value /*[n]*/, float *logits /*[n][A]*/, float *policy /*[n][A] softmax*/);
/* the same network with the heads in the reference's literal op order (dense per pixel, then reduce_sum): see orc_net.c */
void orc_net_forward_perpixel(const orc_net *w, const int8_t *boards, int n, float *value, float *logits, float *policy);
double orc_np_sum(const double *a, int n); /* numpy pairwise add.reduce */
void orc_philox(uint64_t key, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]);
double orc_u53(uint64_t key, uint32_t game_id, uint32_t ply);
float orc_beta_noise(uint64_t key, uint32_t game_id, uint32_t node_serial, uint32_t action, float alpha);

/* ---- tree search ----------------------------------------------------------------- */
orc_search *orc_search_new(const orc_cfg *cfg, uint32_t game_id);
void orc_search_free(orc_search *s);
void orc_drop_root(orc_search *s);
int  orc_has_root(const orc_search *s);
int  orc_move_root(orc_search *s, const orc_state *st);
void orc_get_stats(const orc_search *s, orc_stats *out);
/* MCTS.FindMove (MCTS.py:146-199). u in [0,1): the uniform np.random.choice would draw;
 * u < 0 -> drawn from philox(seed, game_id, ply).  Returns 0, or
 * -2 AssertionError (root != state), -3 ValueError (NaN probabilities), -4 bad args. */
int orc_find_move(orc_search *s, const orc_state *st, double temp, int play_limit, double u,
                  uint32_t ply, int *action, orc_state *next, double *root_winrate,
                  double *child_prob /*A*/, double *child_plays /*A*/, double *child_winrates /*A*/,
                  double *root_plays);
int orc_select_puct(orc_search *s); /* PUCT argmax at root (temp==0 branch) */
/* MCTS._selectAction(exploring=False) sampling law, MCTS.py:335-338 */
int orc_sample_action(const double *child_plays, int A, double temp, double u);

/* Blackbird.GenerateTrainingSamples, one game (Blackbird.py:238-268).
 * boards [(max_plies+1)][H*W*C], pi [(max_plies+1)][A], player/z [(max_plies+1)].
 * returns number of examples (plies+1), <0 on error. winner: 0 draw,1,2; -1 if capped. */
int orc_selfplay_game(const orc_cfg *cfg, uint32_t game_id, double temp, int play_limit,
                      int max_plies, int8_t *boards, double *pi, int8_t *player, float *z,
                      int *actions, int *winner, orc_stats *stats);

#ifdef __cplusplus
}
#endif
#endif
