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
