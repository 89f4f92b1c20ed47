// games.hip.h -- device-side game rules on packed bit-plane states (gfx950).
//
// Connect4 / TicTacToe boards are two u64 bit-planes (layout in include/blackbird_hip.h).
// Every function reproduces the reference bit-exactly, including its behaviour on positions
// that cannot be reached in play (the parity fixtures probe those):
//   LegalActions  Connect4.py:30-36   TicTacToe.py:29-36
//   ApplyAction   Connect4.py:41-53   TicTacToe.py:41-48
//   Winner        Connect4.py:62-108  TicTacToe.py:57-105
//   AsInputArray  Connect4.py:55-60   TicTacToe.py:50-55
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// A pointer the caller KNOWS to be LDS (or global), re-derived through an explicit address-space cast: the compiler's
// address-space inference then turns the accesses behind it into ds_* (global_*) instructions.  Through a generic pointer
// every access is a flat_* instruction -- it takes the LDS and the memory path, counts on both vmcnt and lgkmcnt, and its
// LDS round trip is several times a ds_read's.  Device functions that are not inlined into the kernel that owns the
// __shared__ array (the out-of-line phase functions of mega_dc.hip.h) only see generic pointers unless told.
template <class T>
__device__ __forceinline__ T *as_lds(T *p) {
#if __HIP_DEVICE_COMPILE__
    __builtin_assume(__builtin_amdgcn_is_shared((const void *)p));
#endif
    return p;
}
template <class T>
__device__ __forceinline__ T *as_global(T *p) {
#if __HIP_DEVICE_COMPILE__
    __builtin_assume(!__builtin_amdgcn_is_shared((const void *)p) && !__builtin_amdgcn_is_private((const void *)p));
#endif
    return p;
}

struct GridState {
    uint64_t p1, p2;
};
static_assert(sizeof(GridState) == 16, "packed grid state is 16 bytes");

#define BB_HD __host__ __device__ __forceinline__

BB_HD int gs_player(const GridState &s) { return (int)((s.p1 >> 56) & 3); }
BB_HD int gs_prev(const GridState &s) { return (int)((s.p1 >> 58) & 3); }
BB_HD uint64_t gs_cells(uint64_t p) { return p & 0x00FFFFFFFFFFFFFFull; }
BB_HD GridState gs_make(uint64_t b1, uint64_t b2, int player, int prev) {
    GridState s;
    s.p1 = gs_cells(b1) | ((uint64_t)(player & 3) << 56) | ((uint64_t)(prev & 3) << 58);
    s.p2 = gs_cells(b2);
    return s;
}

BB_HD int bb_clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}
BB_HD int bb_ctz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffsll((unsigned long long)x) - 1;
#else
    return __builtin_ctzll(x);
#endif
}
BB_HD int bb_popc64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popcll(x);
#else
    return __builtin_popcountll(x);
#endif
}

// H x W board, K in a row, bit index = row*STR + col with one spare (always empty) column.
template <int H_, int W_, int K_, int STR_, int A_, int S_, int GID_>
struct GridGame {
    static constexpr int H = H_, W = W_, K = K_, STR = STR_;
    static constexpr int C = 3;      // input planes
    static constexpr int A = A_;     // LegalMoves
    static constexpr int S = S_;     // child slots per node row (lanes per game)
    static constexpr int GID = GID_; // BB_GAME_*
    static constexpr int MAXPATH = H_ * W_ + 2;
    static constexpr bool DROP = (GID_ == 0); // Connect4: pieces fall
    static constexpr bool CELL_BF16 = false;  // (DragonChess: encode_cell_bf16)
    using State = GridState;

    BB_HD static uint64_t board_mask() {
        uint64_t m = 0;
        for (int r = 0; r < H; r++) m |= ((1ull << W) - 1) << (r * STR);
        return m;
    }
    BB_HD static State initial() { return gs_make(0, 0, 1, 0); }

    // cells that belong to a run of >= K equal cells in b (the set _checkVictory accepts)
    BB_HD static uint64_t win_cells(uint64_t b) {
        const int sh[4] = {1, STR + 1, STR, STR - 1}; // Dirs (0,1),(1,1),(1,0),(1,-1)
        uint64_t out = 0;
        for (int d = 0; d < 4; d++) {
            int s = sh[d];
            uint64_t start;
            if (K == 4) {
                uint64_t m = b & (b >> s);
                start = m & (m >> (2 * s));
            } else {
                start = b & (b >> s) & (b >> (2 * s));
            }
            uint64_t cells = start;
            for (int k = 1; k < K; k++) cells |= start << (k * s);
            out |= cells;
        }
        return out;
    }

    BB_HD static uint32_t legal_mask(const State &s) {
        uint64_t occ = gs_cells(s.p1) | gs_cells(s.p2);
        if (DROP) return (uint32_t)((~occ >> ((H - 1) * STR)) & ((1u << W) - 1));
        uint32_t m = 0;
        for (int r = 0; r < H; r++) m |= (uint32_t)((~occ >> (r * STR)) & ((1u << W) - 1)) << (r * W);
        return m;
    }

    // returns false where the reference raises ValueError; s is left unchanged then
    BB_HD static bool apply(State &s, int a) {
        if (a < 0 || a >= A) return false;
        uint64_t b1 = gs_cells(s.p1), b2 = gs_cells(s.p2), occ = b1 | b2;
        int player = gs_player(s);
        int bit;
        if (DROP) {
            if ((occ >> ((H - 1) * STR + a)) & 1) return false;
            uint64_t col = 0;
            for (int r = 0; r < H; r++) col |= 1ull << (r * STR);
            uint64_t cb = (occ >> a) & col; // stones in that column
            int top = cb ? (63 - bb_clz64(cb)) / STR : -1;
            bit = (top + 1) * STR + a;
        } else {
            bit = (a / W) * STR + (a % W);
            if ((occ >> bit) & 1) return false;
        }
        if (player == 1) b1 |= 1ull << bit;
        else b2 |= 1ull << bit;
        s = gs_make(b1, b2, player == 2 ? 1 : 2, player);
        return true;
    }

    BB_HD static bool is_over(uint64_t occ) {
        if (DROP) { // _isOver: no empty cell in the top row (Connect4.py:88-92)
            uint64_t top = ((1ull << W) - 1) << ((H - 1) * STR);
            return (occ & top) == top;
        }
        return bb_popc64(occ) == H * W; // TicTacToe.py:82-83
    }

    // Winner(prevAction); prev < 0 == None.  -1 None, 0 draw, 1, 2.
    BB_HD static int winner(const State &s, int prev) {
        // _collapsed: plane 1 wins where both planes are set
        uint64_t b2 = gs_cells(s.p2), b1 = gs_cells(s.p1) & ~b2, occ = b1 | b2;
        if (prev >= 0) {
            int cell;
            if (DROP) {
                uint64_t col = 0;
                for (int r = 0; r < H; r++) col |= 1ull << (r * STR);
                uint64_t cb = (occ >> prev) & col;
                int top = cb ? (63 - bb_clz64(cb)) / STR : 0; // empty column: Python's loop leaves i == 0
                cell = top * STR + prev;
            } else {
                cell = (prev / W) * STR + (prev % W);
            }
            uint64_t bit = 1ull << cell;
            int p = (b1 & bit) ? 1 : (b2 & bit) ? 2 : 0;
            uint64_t own = p == 1 ? b1 : p == 2 ? b2 : (~occ & board_mask()); // p == 0: a run of empties counts
            if (win_cells(own) & bit) return p;
        } else {
            uint64_t w1 = win_cells(b1), w2 = win_cells(b2), any = w1 | w2;
            if (any) return (w1 >> bb_ctz64(any)) & 1 ? 1 : 2; // first hit of the row-major scan
        }
        if (is_over(occ)) return 0;
        return -1;
    }

    // AsInputArray -> int8[H][W][3]
    BB_HD static void encode_cell(const State &s, int r, int c, int8_t out[3]) {
        int bit = r * STR + c;
        out[0] = (int8_t)((s.p1 >> bit) & 1);
        out[1] = (int8_t)((s.p2 >> bit) & 1);
        out[2] = gs_player(s) == 1 ? 1 : -1;
    }
};

using Connect4 = GridGame<6, 7, 4, 8, 7, 8, 0>;
using TicTacToe = GridGame<3, 3, 3, 4, 9, 16, 1>;

// ------------------------------------------------------------------------------------------------
// DragonChess (DragonChess.py:10-371): White = K + 3 pawns and moves twice per turn, Black = full army,
// win by capturing the king.  Pseudo-legal rules exactly as the reference codes them: promotions
// and castling return Python 0 (falsy) => pawns on their 7th rank cannot advance and castling is
// never legal; no en passant; no check rule.  Packed state = 80 B (include/blackbird_hip.h).
// ------------------------------------------------------------------------------------------------
struct alignas(16) DCState {
    int8_t b[64]; // row*8+col, signed codes K1 P2 N3 B4 R5 Q6, row 0 = White's back rank
    int8_t player, prev;
    int8_t castle[4]; // wK wQ bK bQ
    int8_t pad[10];
};
static_assert(sizeof(DCState) == 80, "packed DragonChess state is 80 bytes");

struct DragonChess {
    static constexpr int H = 8, W = 8, C = 17, A = 4032, S = 144, GID = 2;
    static constexpr int MAXPATH = 128;
    static constexpr bool CELL_BF16 = true;
    using State = DCState;

    BB_HD static State initial() { // fen 'rnbqkbnr/pppppppp/8/8/8/8/3PPP2/4K3 w kq' (DragonChess.py:36-60)
        State s;
        for (int i = 0; i < 64; i++) s.b[i] = 0;
        const int8_t back[8] = {-5, -3, -4, -6, -1, -4, -3, -5};
        for (int c = 0; c < 8; c++) {
            s.b[56 + c] = back[c];
            s.b[48 + c] = -2;
        }
        s.b[8 + 3] = 2;
        s.b[8 + 4] = 2;
        s.b[8 + 5] = 2;
        s.b[4] = 1;
        s.player = 1;
        s.prev = 0;
        s.castle[0] = 0;
        s.castle[1] = 0;
        s.castle[2] = 1;
        s.castle[3] = 1;
        for (int i = 0; i < 10; i++) s.pad[i] = 0;
        return s;
    }
    BB_HD static int iabs(int x) { return x < 0 ? -x : x; }
    BB_HD static int isgn(int x) { return (x > 0) - (x < 0); }

    // _is_legal_move(loc, new) with promote=None, castle=None (DragonChess.py:242-371)
    BB_HD static bool is_legal(const int8_t *b, int player, int lr, int lc, int nr, int nc) {
        int piece = b[lr * 8 + lc], target = b[nr * 8 + nc];
        // _sanity_check :242-259
        if (nr == lr && nc == lc) return false;
        if (piece > 0) {
            if (target > 0) return false;
        } else {
            if (target < 0) return false;
        }
        if (piece < 0 && player == 1) return false;
        if (piece > 0 && player == 2) return false;
        if (piece == 0) return false;
        int adr = iabs(lr - nr), adc = iabs(lc - nc);
        switch (iabs(piece)) {
        case 1: return adr <= 1 && adc <= 1; // castling branches return 0 (:356-363)
        case 2: // pawn :282-319
            if (piece > 0) {
                if (lr == 6 && lc == nc && nr == 7 && b[56 + nc] == 0) return false;          // promotion -> 0
                else if (lr == 6 && adc == 1 && nr == 7 && b[56 + nc] < 0) return false;      // capture-promotion -> 0
                else if (lr == 1) {
                    if (lc == nc) {
                        if (nr == 3 && b[24 + nc] == 0 && b[16 + nc] == 0) return true;
                        else if (nr == 2 && b[16 + nc] == 0) return true;
                        return false;
                    }
                } else if (lr > 1 && lr < 6) {
                    if (lc == nc && target == 0 && nr == lr + 1) return true;
                }
                if (adc == 1 && nr == lr + 1 && target < 0) return true;
            } else {
                if (lr == 1 && lc == nc && nr == 0 && b[nc] == 0) return false;
                else if (lr == 1 && adc == 1 && nr == 0 && b[nc] > 0) return false;
                else if (lr == 6) {
                    if (lc == nc) {
                        if (nr == 4 && b[32 + nc] == 0 && b[40 + nc] == 0) return true;
                        else if (nr == 5 && b[40 + nc] == 0) return true;
                        return false;
                    }
                } else if (lr > 1 && lr < 6) {
                    if (lc == nc && target == 0 && nr == lr - 1) return true;
                }
                if (adc == 1 && nr == lr - 1 && target > 0) return true;
            }
            return false;
        case 3: return (adr == 1 && adc == 2) || (adr == 2 && adc == 1);
        case 4: return bishop(b, lr, lc, nr, nc);
        case 5: return rook(b, lr, lc, nr, nc);
        case 6: return bishop(b, lr, lc, nr, nc) || rook(b, lr, lc, nr, nc);
        }
        return false;
    }
    BB_HD static bool rook(const int8_t *b, int lr, int lc, int nr, int nc) { // :321-339
        if (!(nr == lr || nc == lc)) return false;
        if (nr == lr) {
            int lo = nc < lc ? nc : lc, hi = nc < lc ? lc : nc;
            for (int c = lo + 1; c < hi; c++)
                if (b[nr * 8 + c] != 0) return false;
        } else {
            int lo = nr < lr ? nr : lr, hi = nr < lr ? lr : nr;
            for (int r = lo + 1; r < hi; r++)
                if (b[r * 8 + nc] != 0) return false;
        }
        return true;
    }
    BB_HD static bool bishop(const int8_t *b, int lr, int lc, int nr, int nc) { // :341-347
        if (iabs(lr - nr) != iabs(lc - nc)) return false;
        int sr = isgn(nr - lr), sc = isgn(nc - lc);
        for (int d = 1; d < iabs(lr - nr); d++)
            if (b[(lr + d * sr) * 8 + (lc + d * sc)] != 0) return false;
        return true;
    }

    // legal targets of one from-square as a 64-bit mask (bit = to-square)
    // b may point at a copy of the board in LDS: the 63 legality tests index the board dynamically, which on a
    // register-resident State means scratch-memory round trips
    BB_HD static uint64_t targets_from(const int8_t *b, int player, int sq1) {
        uint64_t m = 0;
        int piece = b[sq1];
        if (piece == 0 || (piece < 0 && player == 1) || (piece > 0 && player == 2)) return 0;
        for (int sq2 = 0; sq2 < 64; sq2++)
            if (sq2 != sq1 && is_legal(b, player, sq1 >> 3, sq1 & 7, sq2 >> 3, sq2 & 7)) m |= 1ull << sq2;
        return m;
    }
    BB_HD static uint64_t targets(const State &s, int sq1) { return targets_from(s.b, s.player, sq1); }
    // The same legality rules (DragonChess.py:225-363 via is_legal above) as bit operations for one from-square, given
    // the board's occupancy: `white` / `black` = squares holding a positive / negative piece.  A wave whose lane i holds
    // square i gets the three masks from two ballots.  Sliders walk their rays until the first occupied square (own:
    // stop before it, enemy: capture it); king / knight by offsets; pawns follow the reference's literal rules (no
    // promotion, no en passant, double step from the start row only over two empty squares, a pawn that would promote
    // has no move, captures one row ahead from any row that has one).  Checked against is_legal on the reference's
    // golden boards, which include unreachable positions (tests/test_gpu_games.py).
    BB_HD static uint64_t targets_bits(int piece, int player, int sq1, uint64_t white, uint64_t black) {
        if (piece == 0 || (piece < 0 && player == 1) || (piece > 0 && player == 2)) return 0;
        const uint64_t occ = white | black, own = piece > 0 ? white : black, enemy = piece > 0 ? black : white;
        const int r = sq1 >> 3, c = sq1 & 7;
        const int kind = piece < 0 ? -piece : piece;
        uint64_t m = 0;
        if (kind == 1 || kind == 3) {
            const int8_t kdr[8] = {-1, -1, -1, 0, 0, 1, 1, 1}, kdc[8] = {-1, 0, 1, -1, 1, -1, 0, 1};
            const int8_t ndr[8] = {-2, -2, -1, -1, 1, 1, 2, 2}, ndc[8] = {-1, 1, -2, 2, -2, 2, -1, 1};
            for (int k = 0; k < 8; k++) {
                int nr = r + (kind == 1 ? kdr[k] : ndr[k]), nc = c + (kind == 1 ? kdc[k] : ndc[k]);
                if (nr >= 0 && nr < 8 && nc >= 0 && nc < 8) m |= 1ull << (nr * 8 + nc);
            }
            return m & ~own;
        }
        if (kind == 2) {
            const int dir = piece > 0 ? 1 : -1;          // white pawns walk up the rows, black pawns down
            const int start = piece > 0 ? 1 : 6, last = piece > 0 ? 6 : 1; // `last`: the next row would be a promotion -> no move
            if (r == last) return 0;
            const int nr = r + dir;
            if (nr < 0 || nr > 7) return 0;
            if (r == start) {
                const uint64_t one = 1ull << (nr * 8 + c), two = 1ull << ((nr + dir) * 8 + c);
                if (!(occ & one)) {
                    m |= one;
                    if (!(occ & two)) m |= two;
                }
            } else if (r > 1 && r < 6) {
                const uint64_t one = 1ull << (nr * 8 + c);
                if (!(occ & one)) m |= one;
            }
            if (c > 0) m |= (1ull << (nr * 8 + c - 1)) & enemy;
            if (c < 7) m |= (1ull << (nr * 8 + c + 1)) & enemy;
            return m;
        }
        const bool diag = kind == 4 || kind == 6, ortho = kind == 5 || kind == 6;
        const int8_t sdr[8] = {-1, -1, 1, 1, -1, 1, 0, 0}, sdc[8] = {-1, 1, -1, 1, 0, 0, -1, 1};
        for (int k = 0; k < 8; k++) {
            if (k < 4 ? !diag : !ortho) continue;
            int nr = r + sdr[k], nc = c + sdc[k];
            while (nr >= 0 && nr < 8 && nc >= 0 && nc < 8) {
                const uint64_t bit = 1ull << (nr * 8 + nc);
                if (occ & bit) {
                    m |= bit & enemy;
                    break;
                }
                m |= bit;
                nr += sdr[k];
                nc += sdc[k];
            }
        }
        return m;
    }

    BB_HD static int action_id(int sq1, int sq2) { return sq1 * 63 + sq2 - (sq2 > sq1); } // DragonChess.py:26-34
    BB_HD static void action_squares(int a, int &sq1, int &sq2) {
        sq1 = a / 63;
        int rem = a % 63;
        sq2 = rem + (rem >= sq1);
    }

    // ApplyAction (:127-159) + Move (:172-214) for action < 4032; false == ValueError
    BB_HD static bool apply(State &s, int a) {
        if (a < 0 || a >= A) return false;
        int sq1, sq2;
        action_squares(a, sq1, sq2);
        if (!is_legal(s.b, s.player, sq1 >> 3, sq1 & 7, sq2 >> 3, sq2 & 7)) return false;
        s.b[sq2] = s.b[sq1];
        s.b[sq1] = 0;
        if (s.prev == 1 && s.player == 1) { // :192-197: W, W, B, W, W, B ...
            s.player = 2;
            s.prev = 1;
        } else {
            s.prev = s.player;
            s.player = 1;
        }
        if (s.b[4] != 1) { // :199-212
            s.castle[0] = 0;
            s.castle[1] = 0;
        } else if (s.b[7] != 5) {
            s.castle[0] = 0;
        }
        if (s.b[0] != 5) s.castle[1] = 0;
        if (s.b[60] != -1) {
            s.castle[2] = 0;
            s.castle[3] = 0;
        } else if (s.b[63] != -5) {
            s.castle[2] = 0;
        }
        if (s.b[56] != -5) s.castle[3] = 0;
        return true;
    }

    BB_HD static int winner(const State &s, int /*prev*/) { // :161-167
        bool bk = false, wk = false;
        for (int i = 0; i < 64; i++) {
            bk |= s.b[i] == -1;
            wk |= s.b[i] == 1;
        }
        if (!bk) return 1;
        if (!wk) return 2;
        return -1;
    }

    // AsInputArray cell -> 17 planes (:111-125, piece_map :11)
    BB_HD static void encode_cell(const State &s, int r, int c, int8_t out[17]) {
        for (int k = 0; k < 17; k++) out[k] = 0;
        int v = s.b[r * 8 + c];
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
        if (plane >= 0) out[plane] = 1;
        out[12] = s.castle[0];
        out[13] = s.castle[1];
        out[14] = s.castle[2];
        out[15] = s.castle[3];
        out[16] = (s.player == 1 && s.prev == 1) ? 1 : 0;
    }
    // The same cell as the 32 bf16 values the split-operand first conv reads (17 planes + zero padding), two per word,
    // without the 12-way switch (a divergent wave walks every case): the piece plane from a nibble table indexed by code + 6
    // (15 = empty square), bf16 1.0 = 0x3F80.  tests/test_gpu_net.py compares the network on encode_cell's planes.
    __device__ __forceinline__ static void encode_cell_bf16(const State &s, int r, int c, uint32_t out[16]) {
        const int v = s.b[r * 8 + c];
        const int plane = (int)((0x82640AFB15739ull >> (4 * (v + 6))) & 15);
        const uint32_t one = 0x3F80u << (16 * (plane & 1));
        for (int j = 0; j < 6; j++) out[j] = (plane >> 1) == j ? one : 0u;
        auto bf = [](int x) { return __builtin_bit_cast(uint32_t, (float)x) >> 16; };
        out[6] = bf(s.castle[0]) | (bf(s.castle[1]) << 16);
        out[7] = bf(s.castle[2]) | (bf(s.castle[3]) << 16);
        out[8] = (s.player == 1 && s.prev == 1) ? 0x3F80u : 0u;
        for (int j = 9; j < 16; j++) out[j] = 0u;
    }
};
