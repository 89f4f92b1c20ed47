/*
 * include/blackbird_hip.h -- C ABI of libblackbird_hip.so (MI355X / gfx950).
 *
 * The reference (ZackAttack614/BlackBird) is pure Python and has no FFI of its own; its drop-in
 * boundary for the self-play hot path is the Python API
 *     Blackbird.GenerateTrainingSamples   (src/Blackbird.py:219-268)
 *     MCTS.FindMove / MoveRoot / DropRoot (src/MCTS.py:146-225)
 *     Network.getEvaluation / getPolicy   (src/Network.py:48-64)
 *     GameState.LegalActions / ApplyAction / Winner / AsInputArray
 *                                         (src/GameState.py:1-28, Connect4.py, TicTacToe.py, DragonChess.py)
 * The modules under blackbird_amd/ mirror those classes and binds the entry points below with ctypes
 * (INTEGRATION.md shows the binding).  Each entry point cites the reference code it replaces.
 *
 * Conventions: every function returns 0 on success or a negative bb_status; bb_last_error()
 * gives the message.  All buffers are caller-allocated; a pointer may be host or device memory
 * (copies use hipMemcpyDefault).  An engine is bound to one GPU and is not thread-safe (the
 * reference is single-threaded).  There is no CPU fallback: without a GPU every compute entry
 * point fails with BB_ERR_HIP.
 */
#ifndef BLACKBIRD_HIP_H
#define BLACKBIRD_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BB_OK = 0,
    BB_ERR_ARG = -1,      /* ValueError in the reference (bad nGames, no stop rule, ...) */
    BB_ERR_HIP = -2,      /* HIP runtime failure / no device */
    BB_ERR_STATE = -3,    /* AssertionError: tree root does not match the state (MCTS.py:193) */
    BB_ERR_NAN = -4,      /* ValueError: probabilities contain NaN (MCTS.py:336-338, 1 sim on a fresh root) */
    BB_ERR_CAPACITY = -5, /* node pool / example store exhausted */
    BB_ERR_WEIGHTS = -6   /* network evaluator requested before bb_load_weights */
} bb_status;

enum { BB_GAME_CONNECT4 = 0, BB_GAME_TICTACTOE = 1, BB_GAME_DRAGONCHESS = 2 };
enum { BB_MCTS_DYNAMIC = 0 /* DynamicMCTS.py:14-34 */, BB_MCTS_FIXED = 1 /* FixedMCTS.py:21-34 */ };
enum {
    BB_EVAL_HASH = 0,   /* deterministic synthetic evaluator (validation only; spec in DESIGN.md)  */
    BB_EVAL_NET = 1,    /* Model.SampleValue/GetPriors: residual tower (Blackbird.py:350-389)      */
    BB_EVAL_ROLLOUT = 2 /* MCTS.SampleValue/GetPriors: random rollouts, priors = ones (MCTS.py:346-383) */
};

/* ---- packed game states (the SoA/bit-plane form the engine keeps in HBM) ----------------
 * Connect4 / TicTacToe: 16 bytes, two little-endian u64 bit-planes.
 *   plane p (p=0,1) bit (row*STRIDE + col) = Board[row][col][p]   (Connect4.py:20, TicTacToe.py:19)
 *   STRIDE = 8 for Connect4 (6x7), 4 for TicTacToe (3x3); the spare column is always 0.
 *   plane 0 bits 56-57 = Player (1|2), bits 58-59 = PreviousPlayer (0 = None).
 * DragonChess: 80 bytes: int8 board[64] (row*8+col, signed codes K1 P2 N3 B4 R5 Q6,
 *   DragonChess.py:12-19), int8 player, int8 prev, int8 castle[4] {wK,wQ,bK,bQ}, 10 pad bytes. */
#define BB_GRID_STATE_BYTES 16
#define BB_DC_STATE_BYTES 80

typedef struct {
    int32_t H, W, C;       /* AsInputArray shape [1,H,W,C] */
    int32_t A;             /* LegalMoves */
    int32_t S;             /* child slots per node row (>= max legal moves of any position) */
    int32_t state_bytes;   /* packed state size */
    int32_t dense;         /* 1: slot i == action i; 0: compact child lists (DragonChess) */
    int32_t example_bytes; /* size of one bb_examples_fetch record */
} bb_game_info;

int bb_game_info_get(int game, bb_game_info *out);
const char *bb_last_error(void);
int bb_device_count(void);

/* ---- batched game kernels (stateless) ---------------------------------------------------
 * n boards per call, one packed state each; n == 0 is a no-op (BB_OK, nothing touched).      */
/* GameState.LegalActions (Connect4.py:30-36, TicTacToe.py:29-36, DragonChess.py:78-106):
 * legal_out[n][A] bytes 0/1. */
int bb_game_legal(int game, int n, const void *states, uint8_t *legal_out);
/* GameState.ApplyAction (Connect4.py:41-53, TicTacToe.py:41-48, DragonChess.py:127-159 + Move :172-214).
 * states updated in place; status_out[i] = 0, or -1 where the reference raises
 * ValueError('Tried to make an illegal move.') (state left unchanged). */
int bb_game_apply(int game, int n, void *states, const int32_t *actions, int32_t *status_out);
/* GameState.Winner(prevAction) (Connect4.py:62-83, TicTacToe.py:57-76, DragonChess.py:161-167).
 * prev_actions may be NULL (== None for every board); an entry < 0 is None.
 * winner_out[i] = -1 (None), 0 (draw), 1, 2. */
int bb_game_winner(int game, int n, const void *states, const int32_t *prev_actions, int8_t *winner_out);
/* GameState.AsInputArray (Connect4.py:55-60, TicTacToe.py:50-55, DragonChess.py:111-125):
 * planes_out[n][H][W][C] int8. */
int bb_game_encode(int game, int n, const void *states, int8_t *planes_out);
/* initial position (Connect4.py:19-22, TicTacToe.py:18-21, DragonChess.py:36-60) */
int bb_game_initial(int game, void *state_out);

/* ---- network weights: TF variable layout of NetworkFactory.py:37-183 ------ */
typedef struct {
    int32_t H, W, C, F, R, D, A;
    const float *conv0_k;  /* [3][3][C][F]  resTower/conv_block/conv/kernel (HWIO) */
    const float *conv0_b;  /* [F] */
    const float *conv0_bn; /* [4][F] gamma, beta, moving_mean, moving_variance */
    const float *blk_k;    /* [R][2][3][3][F][F]  resTower/block_i/conv_{1,2}/kernel */
    const float *blk_b;    /* [R][2][F] */
    const float *blk_bn;   /* [R][2][4][F] */
    const float *v_conv_k; /* [F]     value/convolution/kernel [1,1,F,1] */
    const float *v_conv_b; /* [1] */
    const float *v_bn;     /* [4][1] */
    const float *v_d1_k;   /* [D]     value/dense_1/kernel [1,D] */
    const float *v_d1_b;   /* [D] */
    const float *v_d2_k;   /* [D]     value/dense_2/kernel [D,1] */
    const float *v_d2_b;   /* [1] */
    const float *p_conv_k; /* [F][2]  policy/convolution/kernel [1,1,F,2] */
    const float *p_conv_b; /* [2] */
    const float *p_bn;     /* [4][2] */
    const float *p_d_k;    /* [2][A]  policy/policy/kernel */
    const float *p_d_b;    /* [A] */
} bb_net_weights;

/* ---- engine -------------------------------------------------------------------------------- */
typedef struct {
    int32_t game;          /* BB_GAME_* */
    int32_t n_slots;       /* concurrent games (trees) resident on this GPU */
    int32_t mcts_kind;     /* BB_MCTS_* */
    int32_t max_depth;     /* FixedMCTS.MaxDepth (FixedMCTS.py:8-18) */
    int32_t evaluator;     /* BB_EVAL_* */
    int32_t sims_per_move; /* MCTS.PlayLimit (MCTS.py:114-120) */
    int32_t max_plies;     /* cap on game length (Connect4 42, TicTacToe 9; DragonChess has none
                              in the reference -> documented deviation) */
    int32_t max_games;     /* self-play games this engine may store examples for */
    double c_puct;         /* MCTS.ExplorationRate */
    uint64_t seed;         /* Philox key */
    uint64_t hash_salt;    /* BB_EVAL_HASH salt */
    uint32_t first_game_id;/* global index of this engine's game 0 (rank offset; RNG stream id) */
    int32_t noise_on;      /* NetworkFactory.py:176-180 Beta(alpha,1-alpha) prior noise */
    float alpha, epsilon;
    int32_t device;        /* HIP device ordinal */
    int32_t salt_per_game; /* 1: hash salt += local game index (test fixtures) */
    int32_t node_capacity; /* nodes per slot; 0 = sims_per_move*max_plies + 2 */
    int32_t net_form;      /* BB_NET_FORM_*: the arithmetic of the conv tower (Network.getEvaluation/getPolicy, Network.py:48-64).
                              0 = AUTO: the fastest form that meets the 1e-5 bound (the split-operand form wherever it exists) */
    int32_t launch;        /* BB_LAUNCH_*: launch structure of bb_selfplay_step; 0 = AUTO (persistent kernels where the network
                              fits them).  The others exist for parity checks: every structure gives the same bits */
    int32_t general_net;   /* 1: run a 16-filter network through the launch-per-layer kernels of wider networks (parity checks) */
    int32_t track_ancestors; /* 1: keep, per slot, the chain of edges from the first root to the current one and back every
                              simulation up through it, as the reference's _backProp does (MCTS.py:238-258) -- what
                              MCTS.ResetRoot (:214-225) needs; dense-action games */
} bb_config;

/* bb_config.net_form */
#define BB_NET_FORM_AUTO 0
#define BB_NET_FORM_F32 1   /* float32 MFMA (v_mfma_f32_16x16x4_f32): bit-identical to a k-ordered fmaf chain */
#define BB_NET_FORM_SPLIT 2 /* float32 operands as three exact bf16 planes, six bf16 MFMA products per K slice, float32
                               accumulation: float32-grade (<= 1e-5 of the form above, tests/test_gpu_net.py), 2.4x the throughput */
/* bb_config.launch */
#define BB_LAUNCH_AUTO 0
#define BB_LAUNCH_LOCKSTEP 1 /* one tree + one evaluator launch per simulation, one move launch per ply */
#define BB_LAUNCH_ROUNDS 2   /* asynchronous rounds: k_tree_async + a compacted network launch (dense games) */

typedef struct bb_engine bb_engine;

typedef struct {
    uint64_t sims;            /* simulations completed */
    uint64_t sum_depth;       /* sum over simulations of the leaf depth (edges) */
    uint64_t nodes;           /* tree nodes created */
    uint64_t terminal_leaves; /* simulations that ended on a terminal leaf */
    uint64_t games_finished;
    uint64_t plies;           /* moves played in self-play */
    uint64_t overflow;        /* simulations cut short by pool/path limits (must be 0) */
    uint64_t examples;        /* examples stored */
    uint64_t evals;           /* leaves evaluated (< sims when known terminal values are reused) */
} bb_counters;

/* bb_create fails with BB_ERR_CAPACITY (before allocating anything) when the pools of cfg->n_slots games -- sized for the
 * worst case: sims_per_move * max_plies nodes per game -- do not fit the device's free memory. */
int bb_create(const bb_config *cfg, bb_engine **out);
/* The largest slot count <= cfg->n_slots whose pools fit the free memory of cfg->device, and the bytes one slot takes.
 * Games are independent and a slot plays game ids g, g+n_slots, ... one after another, so fewer slots change the
 * schedule, never the results.  The reference has no such limit (one Python object graph per game, Blackbird.py:238-251);
 * the batched GenerateTrainingSamples uses this to cap its concurrency.  BB_ERR_CAPACITY if not even one slot fits. */
int bb_fit_slots(const bb_config *cfg, int *n_slots_out, uint64_t *bytes_per_slot_out);
int bb_destroy(bb_engine *e);
/* Network.__init__/loadModel (Network.py:10-28, 100-112): install weights for BB_EVAL_NET */
int bb_load_weights(bb_engine *e, const bb_net_weights *w);
int bb_get_counters(bb_engine *e, bb_counters *out);
int bb_reset_counters(bb_engine *e);
int bb_synchronize(bb_engine *e);
/* MCTS.PlayLimit can be changed between moves (it is a plain attribute, MCTS.py:116-117). */
int bb_set_sims_per_move(bb_engine *e, int sims);
/* Measurement aid: bracket every `every_n`-th evaluator (network) launch with HIP
 * events on the engine's stream; bb_timing_read synchronises, returns mean/min launch duration in
 * milliseconds over the recorded launches and clears the record.  every_n = 0 turns it off. */
int bb_timing_enable(bb_engine *e, int every_n);
int bb_timing_read(bb_engine *e, double *mean_ms_out, double *min_ms_out, int *count_out);
/* Time `iters` back-to-back launches of the network kernel over the n_slots leaf mailbox (HIP events
 * on the engine stream).  ablate != 0 switches parts of the kernel off: only in a diagnostic build (-DBB_DIAG; kernel tuning,
 * results are wrong) -- the product library has no such switches and answers BB_ERR_ARG. */
int bb_timing_net(bb_engine *e, int iters, int noise, int ablate, double *ms_per_launch_out);
/* Which launch structure bb_selfplay_step uses: 0 lock-step (one tree + one evaluator launch per
 * simulation), 1 asynchronous rounds, 3 persistent per-CU kernel with a work queue between its tree and network
 * waves (default for networks that fit LDS), 5 DragonChess with the 16-filter network: one wave keeps its game for a
 * whole launch -- tree step, network and move in the same wave.  (2 and 4 were launch structures that measured slower
 * and were retired: tools/experimental/.) */
int bb_selfplay_mode(bb_engine *e);
/* Which arithmetic the loaded network's conv tower runs in (after bb_load_weights): 0 float32 MFMA, fused 16-filter
 * tower (bit-identical to the k-ordered fmaf chain); 1 float32 MFMA, one launch per conv layer (any multiple of 16
 * filters); 2 float32 results on the bf16 matrix pipe -- every operand split exactly into three bf16 values, six MFMA
 * products per K slice, float32 accumulation (every 16-filter network: Connect4, TicTacToe and DragonChess; 1e-5 of form 0, not
 * bit-identical; bb_config.net_form = BB_NET_FORM_F32 selects form 0 instead); 3 the launch-per-layer network with its tower
 * layers in the split-operand form of 2 (BB_NET_FORM_F32: form 1).  Negative: BB_ERR_*. */
int bb_net_form(bb_engine *e);

/* Network.getEvaluation + getPolicy for n positions (Network.py:48-64; graph NetworkFactory.py:22-183).
 * Exactly one of states (packed) / planes (int8 [n][H][W][C], what AsInputArray returns) is non-NULL.
 * value_out[n] tanh value for the side to move; logits_out[n][A] pre-softmax; policy_out[n][A]
 * softmax, mixed with Beta noise and re-normalised when noise != 0 (the value of `noise` selects the
 * random stream, so successive calls can draw fresh noise like successive sess.run calls do).
 * Outputs may be NULL. */
int bb_net_eval(bb_engine *e, int n, const void *states, const int8_t *planes, float *value_out,
                float *logits_out, float *policy_out, int noise);
/* The same forward pass with the prior-noise stream named explicitly: position i draws the Beta(alpha, 1-alpha) noise
 * of (global game id game_ids[i], node serial node_serials[i]) -- the key the self-play kernels use for the node they
 * expand -- so a checker can obtain exactly the priors the engine used inside a search
 * (Model.GetPriors with the graph's noise, Blackbird.py:372-389 + NetworkFactory.py:176-182). */
int bb_net_eval_keyed(bb_engine *e, int n, const void *states, const int8_t *planes, const uint32_t *game_ids,
                      const int32_t *node_serials, float *value_out, float *logits_out, float *policy_out);
/* the validation evaluator, same outputs (policy unnormalised, as getPolicy-shaped input to GetPriors) */
int bb_hash_eval(bb_engine *e, int n, const void *states, float *value_out, float *policy_out);

/* ---- tree search on resident slots (MCTS.FindMove split into its steps) ---------------------- */
/* DropRoot (MCTS.py:141-144) + prime slot i's root with states[i] (FindMove's `Root is None`
 * branch, MCTS.py:184-186).  game_ids may be NULL. */
int bb_set_roots(bb_engine *e, int n, const int32_t *slots, const void *states, const uint32_t *game_ids);
/* _runMCTS (MCTS.py:284-303): `sims` more simulations on every active slot, playLimit semantics
 * (added to Root.Plays). */
int bb_run_sims(bb_engine *e, int sims);
/* The same for the slots with mask[slot] != 0 only (mask[n_slots], host memory); the other slots' trees are left
 * untouched.  This is what a batched arena needs (Blackbird.py:177-216, TestModels: two searchers share the games and
 * each one only searches the positions where it is to move). */
int bb_run_sims_masked(bb_engine *e, int sims, const uint8_t *mask);
/* After bb_run_sims: Root statistics + the move _selectAction(exploring=False) picks (MCTS.py:335-338).
 * u[n_slots] uniforms in [0,1) for np.random.choice's law, or NULL to draw Philox(seed, game_id, ply).
 * Outputs per slot (S = bb_game_info.S): action (or BB_ERR_NAN), root_winrate = Root.WinRate(),
 * root_plays, child_action[S] (-1 pad), child_plays[S], child_value[S] (Node.Value of the child, f32).
 * Any output pointer may be NULL. */
int bb_sample_moves(bb_engine *e, double temp, const double *u, int32_t *action_out,
                    float *root_winrate_out, int32_t *root_plays_out, int32_t *child_action_out,
                    int32_t *child_plays_out, float *child_value_out);
/* MoveRoot (MCTS.py:201-212, 260-282) by action id; actions[i] < 0 leaves slot i alone. */
int bb_move_roots(bb_engine *e, const int32_t *actions);
int bb_get_root_states(bb_engine *e, void *states_out);
/* MCTS.ResetRoot (MCTS.py:214-225): every slot's root goes back to its top-most ancestor (the first position searched) with all
 * statistics intact, including the simulations run from the positions below it (bb_config.track_ancestors; BB_ERR_STATE otherwise). */
int bb_reset_roots(bb_engine *e);
/* One node of a slot's tree, for a host-side Node view (MCTS.py:7-98: Children, Plays, Value, State): node < 0 = the root, else a pool
 * index taken from an earlier view.  child_node_out / child_plays_out / child_value_out [S]: per child slot the child's pool index (-1:
 * no simulation has reached it yet -- the reference's eager AddChildren would hold a Node with zero statistics there when the move is
 * legal), its Plays and Value; state_out: the node's packed state; info_out[3]: flags (bit 0: expanded, bit 1: terminal), legal mask,
 * the node's own pool index.  Dense-action games. */
int bb_node_view(bb_engine *e, int slot, int node, int32_t *child_node_out, int32_t *child_plays_out, float *child_value_out,
                 void *state_out, int32_t *info_out);

/* Re-key the engine's random streams (Philox key `seed`; global id of local game 0).  The reference draws fresh numpy /
 * TensorFlow randomness on every GenerateTrainingSamples call (MCTS.py:336-338, NetworkFactory.py:176-180); a caller
 * that reuses an engine for another self-play run gives it a new stream here, otherwise the run repeats the last one. */
int bb_set_rng_stream(bb_engine *e, uint64_t seed, uint32_t first_game_id);

/* ---- batched self-play: Blackbird.GenerateTrainingSamples (Blackbird.py:219-268) ------------- */
/* Start `n_games` games (local ids 0..n_games-1; slot g plays ids g, g+n_slots, ...). */
int bb_selfplay_begin(bb_engine *e, int n_games, double temp);
/* Advance the self-play by `plies` moves' worth of search per active slot (a move: sims_per_move simulations, sample, record
 * the example, MoveRoot, Winner(); finished games hand their slot to the next game id).  Asynchronous.  In the lock-step and
 * rounds structures every slot makes exactly `plies` moves.  The persistent kernels hand out plies x sims_per_move tree visits
 * per slot instead: 7/8 of them go to the slots of each workgroup up front, the rest is one launch-wide pool the workgroups
 * draw from until it is dry -- so every slot advances, by about `plies` moves, a slot in a quicker workgroup by a few more
 * (a game's results do not depend on when its visits happen). */
int bb_selfplay_step(bb_engine *e, int plies);
/* 1 when every game started by bb_selfplay_begin has finished */
int bb_selfplay_done(bb_engine *e, int *done_out, int *games_finished_out);
/* Finished games' examples, in (game, ply) order.  record layout (little endian):
 *   u32 game_id, u16 ply, u8 player, i8 z, u32 total_visits, u32 n_children,
 *   packed state [state_bytes], u32 visits[S], (compact games only) u16 action[S]
 * The last record of a game is the terminal example (visits all zero, Blackbird.py:256-258).
 * Returns the number of records written (<= max_records), or a negative status. */
int bb_examples_fetch(bb_engine *e, int first_game, int n_games, void *records_out, int max_records,
                      int32_t *game_offsets_out /* n_games+1 */, int8_t *winner_out /* n_games */);
/* The [n_games][4] header words of games first_game ..: (n_examples, winner, plies, done).  What GenerateTrainingSamples needs
 * to hand finished games to Conn.PutGames (Blackbird.py:266-268) while the others still play. */
int bb_selfplay_headers(bb_engine *e, int first_game, int n_games, int32_t *hdr_out);
/* bb_examples_fetch for a LIST of games (any order): their records are compacted on the device and come back in one copy.
 * game_offsets_out[n + 1], winner_out[n] as in bb_examples_fetch; a game that has not finished contributes no records. */
int bb_examples_fetch_games(bb_engine *e, int n, const int32_t *game_ids, void *records_out, int max_records,
                            int32_t *game_offsets_out, int8_t *winner_out);
/* The example store where it lives, for a device-to-device exchange (the epoch-end RCCL all-gather of (s, pi, z),
 * SURVEY.md 8e): records_out = device pointer to [max_games][max_plies+1] records of record_bytes each (layout above),
 * game_hdr_out = device pointer to int32 [max_games][4] = {n_examples, winner, plies, done}; game g's records are the
 * first n_examples of its row once done != 0.  Valid until bb_destroy; synchronise (bb_synchronize) before reading. */
int bb_examples_device(bb_engine *e, void **records_out, uint64_t *bytes_out, uint64_t *record_bytes_out,
                       int32_t **game_hdr_out);

#ifdef __cplusplus
}
#endif
#endif
