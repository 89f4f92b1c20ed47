/*
 * oracle/orc_games.c -- CPU ORACLE (test infrastructure, not product code).
 * Literal restatement of the reference's three GameState classes.
 * Citations are relative to /root/reference/src/.
 */
#include "orc.h"
#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------------------
 * Connect4 / TicTacToe share one shape: Board[H][W][2] int8, Dirs, InARow.
 * Connect4.py:9-17, TicTacToe.py:9-16
 * ---------------------------------------------------------------------------------- */
typedef struct { int H, W, inarow; } grid_t;
static const grid_t GRID_C4 = {6, 7, 4};
static const grid_t GRID_TTT = {3, 3, 3};
static const int DIRS[4][2] = {{0, 1}, {1, 1}, {1, 0}, {1, -1}}; /* Connect4.py:14 */

#define BD(g, st, i, j, p) ((st)->b[((i) * (g)->W + (j)) * 2 + (p)])

static int cell_sum(const grid_t *g, const orc_state *st, int i, int j) {
    return BD(g, st, i, j, 0) + BD(g, st, i, j, 1); /* np.sum(Board[i, j, :]) */
}

/* _collapsed, Connect4.py:110-114: loop p over Players {0,1,2}; array[Board[:,:,p-1]==1] = p.
 * p=0 indexes plane -1 (== plane 1) and writes 0, so later p wins: plane1 -> 2, else plane0 -> 1. */
static void collapsed(const grid_t *g, const orc_state *st, int8_t *out) {
    for (int i = 0; i < g->H; i++)
        for (int j = 0; j < g->W; j++) {
            int8_t v = 0;
            if (BD(g, st, i, j, 0) == 1) v = 1;
            if (BD(g, st, i, j, 1) == 1) v = 2;
            out[i * g->W + j] = v;
        }
}

/* _checkVictory, Connect4.py:94-108 / TicTacToe.py:85-105.  Returns p (0,1,2) or -1 (None). */
static int check_victory(const grid_t *g, const int8_t *board, int i, int j) {
    int p = board[i * g->W + j];
    for (int d = 0; d < 4; d++) {
        int d0 = DIRS[d][0], d1 = DIRS[d][1];
        int inarow = 0;
        int r = 0;
        while (r * d0 + i < g->H && r * d1 + j < g->W && r * d1 + j >= 0 &&
               board[(r * d0 + i) * g->W + (r * d1 + j)] == p) {
            inarow++;
            r++;
        }
        r = -1;
        while (r * d0 + i >= 0 && r * d1 + j < g->W && r * d1 + j >= 0 &&
               board[(r * d0 + i) * g->W + (r * d1 + j)] == p) {
            inarow++;
            r--;
        }
        if (inarow >= g->inarow) return p;
    }
    return -1;
}

/* ---- Connect4 --------------------------------------------------------------------- */
static void c4_legal(const orc_state *st, double *out) { /* Connect4.py:30-36 */
    const grid_t *g = &GRID_C4;
    for (int j = 0; j < g->W; j++) out[j] = (cell_sum(g, st, g->H - 1, j) == 0) ? 1.0 : 0.0;
}

static int c4_apply(orc_state *st, int action) { /* Connect4.py:41-53 */
    const grid_t *g = &GRID_C4;
    if (action < 0 || action >= g->W) return -1;
    if (cell_sum(g, st, g->H - 1, action) != 0) return -1; /* ValueError */
    int top = -1;
    for (int i = g->H - 1; i >= 0; i--)
        if (cell_sum(g, st, i, action) != 0) { top = i; break; }
    BD(g, st, top + 1, action, st->player - 1) = 1;
    st->prev = st->player;
    st->player = (st->player == 2) ? 1 : 2;
    return 0;
}

static int c4_is_over(const orc_state *st) { /* Connect4.py:88-92 */
    const grid_t *g = &GRID_C4;
    for (int j = 0; j < g->W; j++)
        if (cell_sum(g, st, g->H - 1, j) == 0) return 0;
    return 1;
}

static int c4_winner(const orc_state *st, int prev) { /* Connect4.py:62-83 */
    const grid_t *g = &GRID_C4;
    int8_t board[42];
    collapsed(g, st, board);
    if (prev >= 0) {
        int i = 0;
        for (i = g->H - 1; i >= 0; i--) /* for i in reversed(range(H)): if sum != 0: break */
            if (cell_sum(g, st, i, prev) != 0) break;
        if (i < 0) i = 0; /* loop ran out: Python leaves i == 0 */
        int win = check_victory(g, board, i, prev);
        if (win >= 0) return win; /* "is not None": 0.0 counts (empty-column quirk) */
    } else {
        for (int i = 0; i < g->H; i++)
            for (int j = 0; j < g->W; j++) {
                if (board[i * g->W + j] == 0) continue;
                int win = check_victory(g, board, i, j);
                if (win >= 0) return win;
            }
    }
    if (c4_is_over(st)) return 0;
    return -1;
}

/* AsInputArray, Connect4.py:55-60 / TicTacToe.py:50-55: int8[1][H][W][3] */
static void grid_encode(const grid_t *g, const orc_state *st, int8_t *out) {
    int8_t pl = (st->player == 1) ? 1 : -1;
    for (int i = 0; i < g->H; i++)
        for (int j = 0; j < g->W; j++) {
            int8_t *o = out + (i * g->W + j) * 3;
            o[0] = BD(g, st, i, j, 0);
            o[1] = BD(g, st, i, j, 1);
            o[2] = pl;
        }
}

/* ---- TicTacToe -------------------------------------------------------------------- */
static void ttt_legal(const orc_state *st, double *out) { /* TicTacToe.py:29-36 */
    const grid_t *g = &GRID_TTT;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = (cell_sum(g, st, i, j) == 0) ? 1.0 : 0.0;
}

static int ttt_apply(orc_state *st, int action) { /* TicTacToe.py:41-48 */
    const grid_t *g = &GRID_TTT;
    if (action < 0 || action >= 9) return -1;
    int i = action / 3, j = action % 3;
    if (cell_sum(g, st, i, j) != 0) return -1;
    BD(g, st, i, j, st->player - 1) = 1;
    st->prev = st->player;
    st->player = (st->player == 2) ? 1 : 2;
    return 0;
}

static int ttt_winner(const orc_state *st, int prev) { /* TicTacToe.py:57-76 */
    const grid_t *g = &GRID_TTT;
    int8_t board[9];
    collapsed(g, st, board);
    if (prev >= 0) {
        int win = check_victory(g, board, prev / 3, prev % 3);
        if (win >= 0) return win;
    } else {
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                if (board[i * 3 + j] == 0) continue;
                int win = check_victory(g, board, i, j);
                if (win >= 0) return win;
            }
    }
    int cnt = 0; /* _isOver: np.sum(board > 0) == Size*Size, TicTacToe.py:82-83 */
    for (int k = 0; k < 9; k++) cnt += board[k] > 0;
    if (cnt == 9) return 0;
    return -1;
}

/* ------------------------------------------------------------------------------------
 * DragonChess, DragonChess.py:10-371.  board[r][c] signed codes K1 P2 N3 B4 R5 Q6.
 * ---------------------------------------------------------------------------------- */
#define DCB(st, r, c) ((st)->b[(r) * 8 + (c)])
static int iabs(int x) { return x < 0 ? -x : x; }
static int isign(int x) { return (x > 0) - (x < 0); }

static void dc_init(orc_state *st) { /* DragonChess.py:36-60, fen 'rnbqkbnr/pppppppp/8/8/8/8/3PPP2/4K3 w kq' */
    memset(st, 0, sizeof(*st));
    static const int8_t back[8] = {-5, -3, -4, -6, -1, -4, -3, -5};
    for (int c = 0; c < 8; c++) {
        DCB(st, 7, c) = back[c];
        DCB(st, 6, c) = -2;
    }
    DCB(st, 1, 3) = 2;
    DCB(st, 1, 4) = 2;
    DCB(st, 1, 5) = 2;
    DCB(st, 0, 4) = 1;
    st->player = 1;
    st->prev = 0;
    st->castle[0] = 0; /* 'K' in 'kq' */
    st->castle[1] = 0;
    st->castle[2] = 1;
    st->castle[3] = 1;
}

/* _sanity_check, DragonChess.py:242-259 */
static int dc_sanity(const orc_state *st, int lr, int lc, int nr, int nc) {
    if (!(-1 < nr && nr < 8) || !(-1 < nc && nc < 8)) return 0;
    if (nr == lr && nc == lc) return 0;
    if (DCB(st, lr, lc) > 0) {
        if (DCB(st, nr, nc) > 0) return 0;
    } else {
        if (DCB(st, nr, nc) < 0) return 0;
    }
    if (DCB(st, lr, lc) < 0 && st->player == 1) return 0;
    else if (DCB(st, lr, lc) > 0 && st->player == 2) return 0;
    else if (DCB(st, lr, lc) == 0) return 0;
    return 1;
}

/* _is_legal_move_pawn, DragonChess.py:282-319.  Promotion branches return Python 0 == falsy. */
static int dc_pawn(const orc_state *st, int lr, int lc, int nr, int nc) {
    if (DCB(st, lr, lc) > 0) {
        if (lr == 6 && lc == nc && nr == 7 && DCB(st, 7, nc) == 0) return 0;
        else if (lr == 6 && iabs(lc - nc) == 1 && nr == 7 && DCB(st, 7, nc) < 0) return 0;
        else if (lr == 1) {
            if (lc == nc) {
                if (nr == 3 && DCB(st, 3, nc) == 0 && DCB(st, 2, nc) == 0) return 1;
                else if (nr == 2 && DCB(st, 2, nc) == 0) return 1;
                return 0;
            }
        } else if (lr > 1 && lr < 6) {
            if (lc == nc && DCB(st, nr, nc) == 0 && nr == lr + 1) return 1;
        }
        if (iabs(lc - nc) == 1)
            if (nr == lr + 1)
                if (DCB(st, nr, nc) < 0) return 1;
    } else {
        if (lr == 1 && lc == nc && nr == 0 && DCB(st, 0, nc) == 0) return 0;
        else if (lr == 1 && iabs(lc - nc) == 1 && nr == 0 && DCB(st, 0, nc) > 0) return 0;
        else if (lr == 6) {
            if (lc == nc) {
                if (nr == 4 && DCB(st, 4, nc) == 0 && DCB(st, 5, nc) == 0) return 1;
                else if (nr == 5 && DCB(st, 5, nc) == 0) return 1;
                return 0;
            }
        } else if (lr > 1 && lr < 6) {
            if (lc == nc && DCB(st, nr, nc) == 0 && nr == lr - 1) return 1;
        }
        if (iabs(lc - nc) == 1)
            if (nr == lr - 1)
                if (DCB(st, nr, nc) > 0) return 1;
    }
    return 0;
}

static int dc_rook(const orc_state *st, int lr, int lc, int nr, int nc) { /* DragonChess.py:321-339 */
    int direction = (nr == lr || nc == lc);
    if (!direction) return 0;
    int no_obstacle = 1;
    if (nr == lr) {
        int lo = nc < lc ? nc : lc, hi = nc < lc ? lc : nc;
        for (int c = lo + 1; c < hi; c++)
            if (DCB(st, nr, c) != 0) { no_obstacle = 0; break; }
    } else {
        int lo = nr < lr ? nr : lr, hi = nr < lr ? lr : nr;
        for (int r = lo + 1; r < hi; r++)
            if (DCB(st, r, nc) != 0) { no_obstacle = 0; break; }
    }
    return direction && no_obstacle;
}

static int dc_bishop(const orc_state *st, int lr, int lc, int nr, int nc) { /* DragonChess.py:341-347 */
    if (iabs(lr - nr) != iabs(lc - nc)) return 0;
    for (int d = 1; d < iabs(lr - nr); d++)
        if (DCB(st, lr + d * isign(nr - lr), lc + d * isign(nc - lc)) != 0) return 0;
    return 1;
}

static int dc_queen(const orc_state *st, int lr, int lc, int nr, int nc) { /* :349-351 */
    return dc_bishop(st, lr, lc, nr, nc) || dc_rook(st, lr, lc, nr, nc);
}

static int dc_king(const orc_state *st, int lr, int lc, int nr, int nc) { /* :353-364 */
    (void)st;
    if (iabs(lr - nr) <= 1 && iabs(lc - nc) <= 1) return 1;
    /* the four castling branches return Python 0 (falsy) -> castling is never legal */
    return 0;
}

static int dc_knight(int lr, int lc, int nr, int nc) { /* :366-371 */
    if (iabs(lr - nr) == 1 && iabs(lc - nc) == 2) return 1;
    else if (iabs(lr - nr) == 2 && iabs(lc - nc) == 1) return 1;
    return 0;
}

/* _is_legal_move (promote=None, castle=None), DragonChess.py:261-280 */
int orc_dc_is_legal(const orc_state *st, int lr, int lc, int nr, int nc) {
    if (!dc_sanity(st, lr, lc, nr, nc)) return 0;
    int pt = iabs(DCB(st, lr, lc));
    switch (pt) {
    case 1: return dc_king(st, lr, lc, nr, nc);
    case 2: return dc_pawn(st, lr, lc, nr, nc);
    case 3: return dc_knight(lr, lc, nr, nc);
    case 4: return dc_bishop(st, lr, lc, nr, nc);
    case 5: return dc_rook(st, lr, lc, nr, nc);
    case 6: return dc_queen(st, lr, lc, nr, nc);
    }
    return 0;
}

static void dc_legal(const orc_state *st, double *out) { /* DragonChess.py:78-106 */
    int idx = 0; /* move_to_int enumeration order, DragonChess.py:26-34 */
    for (int s1 = 0; s1 < 64; s1++)
        for (int s2 = 0; s2 < 64; s2++) {
            if (s2 == s1) continue;
            int c1 = s1 % 8, r1 = s1 / 8, c2 = s2 % 8, r2 = s2 / 8;
            out[idx++] = orc_dc_is_legal(st, r1, c1, r2, c2) ? 1.0 : 0.0;
        }
}

static int dc_apply(orc_state *st, int action) { /* ApplyAction :127-159 + Move :172-214 (action < 4032) */
    if (action < 0 || action >= 4032) return -1;
    int s1 = action / 63, rem = action % 63;
    int s2 = rem + (rem >= s1); /* inverse of idx = s1*63 + s2 - (s2 > s1) */
    int lc = s1 % 8, lr = s1 / 8, nc = s2 % 8, nr = s2 / 8;
    if (!orc_dc_is_legal(st, lr, lc, nr, nc)) return -1; /* ValueError('Tried to make an illegal move.') */
    DCB(st, nr, nc) = DCB(st, lr, lc);
    DCB(st, lr, lc) = 0;
    if (st->prev == 1 && st->player == 1) { /* :192-197 */
        st->player = 2;
        st->prev = 1;
    } else {
        st->prev = st->player;
        st->player = 1;
    }
    if (DCB(st, 0, 4) != 1) { /* :199-212 */
        st->castle[0] = 0;
        st->castle[1] = 0;
    } else if (DCB(st, 0, 7) != 5) {
        st->castle[0] = 0;
    }
    if (DCB(st, 0, 0) != 5) st->castle[1] = 0;
    if (DCB(st, 7, 4) != -1) {
        st->castle[2] = 0;
        st->castle[3] = 0;
    } else if (DCB(st, 7, 7) != -5) {
        st->castle[2] = 0;
    }
    if (DCB(st, 7, 0) != -5) st->castle[3] = 0;
    return 0;
}

static int dc_winner(const orc_state *st) { /* DragonChess.py:161-167 */
    int has_bk = 0, has_wk = 0;
    for (int k = 0; k < 64; k++) {
        has_bk |= st->b[k] == -1;
        has_wk |= st->b[k] == 1;
    }
    if (!has_bk) return 1;
    else if (!has_wk) return 2;
    return -1;
}

static void dc_encode(const orc_state *st, int8_t *out) { /* DragonChess.py:111-125 */
    /* piece_map, DragonChess.py:11 */
    memset(out, 0, 8 * 8 * 17);
    for (int r = 0; r < 8; r++)
        for (int c = 0; c < 8; c++) {
            int v = DCB(st, r, c);
            int8_t *o = out + (r * 8 + c) * 17;
            if (v != 0) {
                int plane = -1;
                switch (v) {
                case 1: plane = 10; break;
                case -1: plane = 11; break;
                case 2: plane = 0; break;
                case -2: plane = 1; break;
                case 3: plane = 4; break;
                case -3: plane = 5; break;
                case 4: plane = 6; break;
                case -4: plane = 7; break;
                case 5: plane = 2; break;
                case -5: plane = 3; break;
                case 6: plane = 8; break;
                case -6: plane = 9; break;
                }
                if (plane >= 0) o[plane] = 1;
            }
            o[12] = st->castle[0];
            o[13] = st->castle[1];
            o[14] = st->castle[2];
            o[15] = st->castle[3];
            o[16] = (st->player == 1 && st->prev == 1) ? 1 : 0;
        }
}

/* ---- dispatch --------------------------------------------------------------------- */
void orc_game_dims(int game, orc_dims *d) {
    switch (game) {
    case ORC_C4: d->H = 6; d->W = 7; d->C = 3; d->A = 7; break;
    case ORC_TTT: d->H = 3; d->W = 3; d->C = 3; d->A = 9; break;
    default: d->H = 8; d->W = 8; d->C = 17; d->A = 4032; break;
    }
}

void orc_game_init(int game, orc_state *st) {
    if (game == ORC_DC) { dc_init(st); return; }
    memset(st, 0, sizeof(*st));
    st->player = 1;
    st->prev = 0;
}

void orc_game_legal(int game, const orc_state *st, double *out) {
    if (game == ORC_C4) c4_legal(st, out);
    else if (game == ORC_TTT) ttt_legal(st, out);
    else dc_legal(st, out);
}

int orc_game_apply(int game, orc_state *st, int action) {
    if (game == ORC_C4) return c4_apply(st, action);
    if (game == ORC_TTT) return ttt_apply(st, action);
    return dc_apply(st, action);
}

int orc_game_winner(int game, const orc_state *st, int prev) {
    if (game == ORC_C4) return c4_winner(st, prev);
    if (game == ORC_TTT) return ttt_winner(st, prev);
    return dc_winner(st);
}

void orc_game_encode(int game, const orc_state *st, int8_t *out) {
    if (game == ORC_C4) grid_encode(&GRID_C4, st, out);
    else if (game == ORC_TTT) grid_encode(&GRID_TTT, st, out);
    else dc_encode(st, out);
}

/* __eq__: Connect4.py:129-132, TicTacToe.py:131-134 (Player + Board);
 * DragonChess.py:226-237 (Player + 4 castle flags + board; PreviousPlayer ignored) */
int orc_game_equal(int game, const orc_state *a, const orc_state *b) {
    if (a->player != b->player) return 0;
    int n = game == ORC_C4 ? 84 : game == ORC_TTT ? 18 : 64;
    if (game == ORC_DC && memcmp(a->castle, b->castle, 4) != 0) return 0;
    return memcmp(a->b, b->b, (size_t)n) == 0;
}
