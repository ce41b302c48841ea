/* blokus_oracle.c -- scalar CPU restatement of the Blokus hot path.  TEST INFRASTRUCTURE
 * (see crl_oracle.h).  Plain nested loops in the reference's own enumeration order:
 *   piece (inventory order) -> anchor (row-major) -> orientation -> shift id
 * following colosseumrl/envs/blokus/board.py:170-193, computation.py:122-246,
 * BlokusEnvironment.py:357-451 and ai.py:44-54.  It shares no code and no formulation with the
 * HIP kernels (those fit whole shapes against row bitboards); agreement between the two is the test.
 * Parity: pinned by tests/golden/blokus_*.npz (generated from the reference run in the build
 * container) and the KATs of SURVEY.md 8c (21 pieces / 89 cells, 712 combos, 116 opening actions).
 */
#include "crl_oracle.h"
#include <stdlib.h>
#include <string.h>

#define BN 20
#define NP 21

/* the 21 pieces as (dx, dy) cell offsets, (0,0) first; order = inventory order = action order
 * (board.py:24-44; values ai.py:12-22 equal the cell counts) */
static const int8_t PIECES[NP][5][2] = {
    {{0, 0}},                                              /* monomino1 */
    {{0, 0}, {1, 0}},                                      /* domino1 */
    {{0, 0}, {1, 0}, {1, 1}},                              /* trominoe1 */
    {{0, 0}, {1, 0}, {2, 0}},                              /* trominoe2 */
    {{0, 0}, {1, 0}, {0, 1}, {1, 1}},                      /* tetrominoes1 */
    {{0, 0}, {1, -1}, {1, 0}, {2, 0}},                     /* tetrominoes2 */
    {{0, 0}, {1, 0}, {2, 0}, {3, 0}},                      /* tetrominoes3 */
    {{0, 0}, {1, 0}, {2, 0}, {2, -1}},                     /* tetrominoes4 */
    {{0, 0}, {1, 0}, {1, -1}, {2, -1}},                    /* tetrominoes5 */
    {{0, 0}, {0, -1}, {1, 0}, {2, 0}, {3, 0}},             /* pentominoe1 */
    {{0, 0}, {0, -1}, {0, 1}, {1, 0}, {2, 0}},             /* pentominoe2 */
    {{0, 0}, {0, -1}, {0, -2}, {1, -2}, {2, -2}},          /* pentominoe3 */
    {{0, 0}, {1, 0}, {1, -1}, {2, -1}, {3, -1}},           /* pentominoe4 */
    {{0, 0}, {0, 1}, {1, 0}, {2, 0}, {2, -1}},             /* pentominoe5 */
    {{0, 0}, {1, 0}, {2, 0}, {3, 0}, {4, 0}},              /* pentominoe6 */
    {{0, 0}, {1, 0}, {2, 0}, {1, -1}, {2, -1}},            /* pentominoe7 */
    {{0, 0}, {0, 1}, {1, 0}, {1, -1}, {2, -1}},            /* pentominoe8 */
    {{0, 0}, {1, 0}, {0, 1}, {0, 2}, {1, 2}},              /* pentominoe9 */
    {{0, 0}, {1, 0}, {1, -1}, {1, 1}, {2, -1}},            /* pentominoe10 */
    {{0, 0}, {-1, 0}, {1, 0}, {0, -1}, {0, 1}},            /* pentominoe11 */
    {{0, 0}, {1, 0}, {1, -1}, {2, 0}, {3, 0}},             /* pentominoe12 */
};
static const int8_t PIECE_CELLS[NP] = {1, 2, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5};

int orc_blokus_piece_cells(int piece) { return (piece >= 0 && piece < NP) ? PIECE_CELLS[piece] : -1; }

/* computation.py:54-86 rotate_piece about the origin, ORIENTATIONS order of board.py:47:
 * 0 north 1 northeast 2 east 3 southeast 4 south 5 southwest 6 west 7 northwest */
static void orient_offset(int o, int dx, int dy, int *ox, int *oy)
{
    switch (o) {
    case 0: *ox = dy;  *oy = -dx; break;   /* rotate 270 deg                 :67-68 */
    case 1: *ox = dx;  *oy = -dy; break;   /* 0 deg, flip y                  :82-84 */
    case 2: *ox = dx;  *oy = dy;  break;   /* default                        :85-86 */
    case 3: *ox = dy;  *oy = dx;  break;   /* 90 deg, flip x                 :74-76 */
    case 4: *ox = -dy; *oy = dx;  break;   /* 90 deg                         :72-73 */
    case 5: *ox = -dx; *oy = dy;  break;   /* 180 deg, flip y                :79-81 */
    case 6: *ox = -dx; *oy = -dy; break;   /* 180 deg                        :77-78 */
    default: *ox = -dy; *oy = -dx; break;  /* 270 deg, flip x (northwest)    :69-71 */
    }
}

/* cells of (piece, orient, shift) relative to the anchor: rotate every offset, then re-origin on
 * the shift-th rotated cell (computation.py:184-246) */
void orc_blokus_placement(int piece, int orient, int shift, int8_t cells[5][2])
{
    int n = PIECE_CELLS[piece], rx[5], ry[5];
    for (int j = 0; j < n; ++j) orient_offset(orient, PIECES[piece][j][0], PIECES[piece][j][1], &rx[j], &ry[j]);
    for (int j = 0; j < n; ++j) { cells[j][0] = (int8_t)(rx[j] - rx[shift]); cells[j][1] = (int8_t)(ry[j] - ry[shift]); }
}

typedef struct {
    int8_t board[BN][BN];   /* [y][x], 0 empty else colour 1..4 (board.py:85) */
    uint32_t inv[4];
    int32_t score[4];
    int32_t round, to_move;
} blk_env;

/* computation.py:89-119 */
static int valid_adjacents(const blk_env *e, int y, int x, int color)
{
    if (y != 0 && e->board[y - 1][x] == color) return 0;
    if (x != 0 && e->board[y][x - 1] == color) return 0;
    if (y != BN - 1 && e->board[y + 1][x] == color) return 0;
    if (x != BN - 1 && e->board[y][x + 1] == color) return 0;
    return 1;
}

/* computation.py:122-142 */
static int valid_cell(const blk_env *e, int x, int y, int color)
{
    if (x < 0 || x >= BN || y < 0 || y >= BN) return 0;
    return e->board[y][x] == 0 && valid_adjacents(e, y, x, color);
}

/* board.py:127-154 */
static int valid_corner(const blk_env *e, int color, int row, int col)
{
    if (!valid_adjacents(e, row, col, color)) return 0;
    if (row != 0 && col != BN - 1 && e->board[row - 1][col + 1] == color) return 1;
    if (row != 0 && col != 0 && e->board[row - 1][col - 1] == color) return 1;
    if (row != BN - 1 && col != BN - 1 && e->board[row + 1][col + 1] == color) return 1;
    if (row != BN - 1 && col != 0 && e->board[row + 1][col - 1] == color) return 1;
    return 0;
}

/* board.py:170-193 get_all_valid_moves, flattened in BlokusEnvironment.valid_actions order (:453-500).
 * ids (may be NULL) receives encoded actions ((piece*400 + y*20 + x)*8 + orient)*5 + shift; returns
 * the number of actions; stops early once `limit` have been found (limit <= 0: no limit). */
/* Work counter for bench.py: the number of (piece, anchor, orientation, shift) placements the REFERENCE tests for the
 * calls restated here -- get_all_valid_moves has no early exit (board.py:184-189), and ai.check_moves (ai.py:31-42,
 * called per player by next_state :424 until one of them has a move) runs that full enumeration too.  So every
 * enumerate_moves() call counts n_anchor * sum over held pieces of 8 * cells, whatever `limit` lets this code skip. */
static uint64_t g_placement_tests = 0;
uint64_t orc_blokus_placement_tests(int reset)
{
    uint64_t v;
#ifdef _OPENMP
#pragma omp atomic read
#endif
    v = g_placement_tests;
    if (reset) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
        g_placement_tests = 0;
    }
    return v;
}

static int enumerate_moves(const blk_env *e, int player, int round, uint32_t inv, int32_t *ids, int cap, int limit)
{
    static const int CORNER_X[4] = {0, BN - 1, 0, BN - 1}, CORNER_Y[4] = {0, 0, BN - 1, BN - 1};   /* board.py:50 */
    int color = player + 1, n_anchor = 0, ax[BN * BN], ay[BN * BN], count = 0;
    if (round == 0) {                                     /* board.py:177-179 */
        ax[0] = CORNER_X[player]; ay[0] = CORNER_Y[player]; n_anchor = 1;
    } else {                                              /* board.py:114-125 row-major scan */
        for (int row = 0; row < BN; ++row)
            for (int col = 0; col < BN; ++col)
                if (e->board[row][col] == 0 && valid_corner(e, color, row, col)) { ax[n_anchor] = col; ay[n_anchor] = row; n_anchor++; }
    }
    {
        uint64_t combos = 0;
        for (int piece = 0; piece < NP; ++piece) if ((inv >> piece) & 1u) combos += 8u * (uint64_t)PIECE_CELLS[piece];
        combos *= (uint64_t)n_anchor;
#ifdef _OPENMP
#pragma omp atomic
#endif
        g_placement_tests += combos;
    }
    for (int piece = 0; piece < NP; ++piece) {            /* board.py:184 inventory order */
        if (!((inv >> piece) & 1u)) continue;
        for (int a = 0; a < n_anchor; ++a)                /* :186 */
            for (int o = 0; o < 8; ++o)                   /* :187 */
                for (int k = 0; k < PIECE_CELLS[piece]; ++k) {   /* computation.py:163 */
                    int8_t cells[5][2];
                    orc_blokus_placement(piece, o, k, cells);
                    int ok = 1;
                    for (int j = 0; j < PIECE_CELLS[piece]; ++j)   /* computation.py:165-175 (no early exit there either) */
                        if (!valid_cell(e, ax[a] + cells[j][0], ay[a] + cells[j][1], color)) ok = 0;
                    if (ok) {
                        if (ids && count < cap) ids[count] = ((piece * 400 + ay[a] * 20 + ax[a]) * 8 + o) * 5 + k;
                        count++;
                        if (limit > 0 && count >= limit) return count;
                    }
                }
    }
    return count;
}

static void load_env(blk_env *e, const uint32_t *occ, const uint32_t *inv, const int32_t *score, int32_t round, int32_t to_move)
{
    memset(e->board, 0, sizeof(e->board));
    for (int c = 0; c < 4; ++c)
        for (int y = 0; y < BN; ++y)
            for (int x = 0; x < BN; ++x)
                if ((occ[c * BN + y] >> x) & 1u) e->board[y][x] = (int8_t)(c + 1);
    for (int c = 0; c < 4; ++c) { e->inv[c] = inv[c]; e->score[c] = score[c]; }
    e->round = round; e->to_move = to_move;
}

static void store_env(const blk_env *e, uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move)
{
    memset(occ, 0, sizeof(uint32_t) * 4 * BN);
    for (int y = 0; y < BN; ++y)
        for (int x = 0; x < BN; ++x)
            if (e->board[y][x] > 0) occ[(e->board[y][x] - 1) * BN + y] |= 1u << x;
    for (int c = 0; c < 4; ++c) { inv[c] = e->inv[c]; score[c] = e->score[c]; }
    *round = e->round; *to_move = e->to_move;
}

static void reset_env(blk_env *e)
{
    memset(e, 0, sizeof(*e));
    for (int c = 0; c < 4; ++c) e->inv[c] = (1u << NP) - 1u;      /* ai.py:29 all 21 pieces */
}

/* An action id -> (piece, index x, index y, orientation, shift).  Ids below ORC_BLOKUS_ACTION_IDS are the dense ids of the
 * action strings valid_actions can emit (index on the board); ids from ORC_BLOKUS_EXT_BASE on name an index anywhere in
 * [-20, 20) x [-20, 20) -- next_state takes whatever string_to_action parsed (BlokusEnvironment.py:83-106, :417-419).
 * Returns 0 when the id names no action at all. */
#define ORC_BLOKUS_ACTION_IDS 336000
#define ORC_BLOKUS_EXT_BASE   336000
#define ORC_BLOKUS_EXT_IDS    (21 * 1600 * 40)
static int decode_action(int action, int *piece, int *x, int *y, int *o, int *shift)
{
    if (action < 0 || action >= ORC_BLOKUS_EXT_BASE + ORC_BLOKUS_EXT_IDS) return 0;
    if (action < ORC_BLOKUS_ACTION_IDS) {
        int cell = (action / 40) % 400;
        *shift = action % 5; *o = (action / 5) % 8; *piece = action / 16000; *x = cell % BN; *y = cell / BN;
    } else {
        int a = action - ORC_BLOKUS_EXT_BASE, cell = (a / 40) % 1600;
        *shift = a % 5; *o = (a / 5) % 8; *piece = a / 64000; *x = cell % 40 - 20; *y = cell / 40 - 20;
    }
    return 1;
}

/* BlokusEnvironment.next_state (:357-451); action < 0 is ''.  No legality test (:417-420): Board.update_board writes
 * board_contents[y][x] = colour cell by cell under numpy's index rules (board.py:87-103: -20..-1 wrap, anything else outside
 * 0..19 raises IndexError; a shift id that names no cell of the piece raises IndexError in shift_offsets, computation.py:218),
 * other colours' cells are overwritten; then AI.update_player (ai.py:44-54) raises ValueError for a piece not held.
 * next_state works on copies (:408-409), so a raise leaves the state as it was.  Here: *status = 0 fine, -1 IndexError,
 * -2 ValueError, -3 the id names no action; on a non-zero status the state is untouched and reward / terminal / winners are 0. */
static void step_env(blk_env *e, int action, int *reward, int *terminal, int *winners, int *status)
{
    int pl = e->to_move, color = pl + 1;
    blk_env old = *e;                                              /* :424 checks the PRE-move board */
    *reward = 0; *terminal = 0; *winners = 0; *status = 0;
    if (action >= 0) {
        int shift, o, piece, ax, ay;
        if (!decode_action(action, &piece, &ax, &ay, &o, &shift)) { *status = -3; return; }
        if (shift >= PIECE_CELLS[piece]) { *status = -1; return; }  /* computation.py:218 offsets[offset_id] */
        int8_t cells[5][2];
        orc_blokus_placement(piece, o, shift, cells);
        for (int j = 0; j < PIECE_CELLS[piece]; ++j) {             /* board.py:87-103, no legality check */
            int x = ax + cells[j][0], y = ay + cells[j][1];
            if (x < -BN || x >= BN || y < -BN || y >= BN) { *e = old; *status = -1; return; }   /* numpy IndexError */
            e->board[y < 0 ? y + BN : y][x < 0 ? x + BN : x] = (int8_t)color;
        }
        if (!((e->inv[pl] >> piece) & 1u)) { *e = old; *status = -2; return; }   /* ai.py:47 list.remove */
        e->inv[pl] &= ~(1u << piece);                              /* ai.py:47 */
        if (e->inv[pl] == 0) e->score[pl] += (piece == 0) ? 20 : 15;   /* ai.py:49-52 */
        e->score[pl] += PIECE_CELLS[piece];                        /* ai.py:54 */
    }
    int any = 0;
    for (int q = 0; q < 4 && !any; ++q)                            /* :424 old board, old round, NEW inventories */
        any = enumerate_moves(&old, q, old.round, e->inv[q], NULL, 0, 1) > 0;
    if (!any) {
        *terminal = 1;
        int max_score = 0;                                         /* :426-437 */
        for (int q = 0; q < 4; ++q) if (e->score[q] > max_score) max_score = e->score[q];
        for (int q = 0; q < 4; ++q) if (e->score[q] == max_score) *winners |= 1 << q;
        int rank = 0;                                              /* :439-440 index in the stable ascending sort */
        for (int q = 0; q < 4; ++q)
            if (e->score[q] < e->score[pl] || (e->score[q] == e->score[pl] && q < pl)) rank++;
        *reward = rank;
    }
    if (pl == 3) e->round += 1;                                    /* :446-447 */
    e->to_move = (pl + 1) % 4;
}

/* ---- batched entry points (state layout of include/colosseum_hip.h) ---- */
void orc_blokus_reset(int64_t B, uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move)
{
    for (int64_t b = 0; b < B; ++b) {
        blk_env e;
        reset_env(&e);
        store_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round + b, to_move + b);
    }
}

/* valid_actions of `player[b]` (NULL: the player to move): count[b] and, when ids != NULL, up to cap ids per env */
void orc_blokus_valid(int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score, const int32_t *round,
                      const int32_t *to_move, const int8_t *player, int32_t *count, int32_t *ids, int cap, int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t b = 0; b < B; ++b) {
        blk_env e;
        load_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round[b], to_move[b]);
        int pl = player ? player[b] : e.to_move;
        count[b] = enumerate_moves(&e, pl, e.round, e.inv[pl], ids ? ids + b * cap : NULL, cap, 0);
    }
}

void orc_blokus_step(int64_t B, uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                     const int32_t *action, int8_t *reward, uint8_t *terminal, uint8_t *winners, int n_threads)
{
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t b = 0; b < B; ++b) {
        blk_env e;
        load_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round[b], to_move[b]);
        int r, t, w, st;
        step_env(&e, action[b], &r, &t, &w, &st);
        store_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round + b, to_move + b);
        reward[b] = (int8_t)(st ? st : r); terminal[b] = (uint8_t)t; winners[b] = (uint8_t)w;   /* the reward slot carries the code */
    }
}

/* Board.check_valid_corner (board.py:127-154) for every cell and colour of reference-layout boards int8 [K][20][20]:
 * grid uint8 [K][4][20][20].  The method does not test the cell itself (gather_empty_corner_indexes does, :121). */
void orc_blokus_valid_corner_grid(int64_t K, const int8_t *board, uint8_t *grid)
{
    for (int64_t k = 0; k < K; ++k) {
        blk_env e;
        memset(&e, 0, sizeof(e));
        memcpy(e.board, board + k * BN * BN, BN * BN);
        for (int c = 1; c <= 4; ++c)
            for (int row = 0; row < BN; ++row)
                for (int col = 0; col < BN; ++col)
                    grid[((k * 4 + c - 1) * BN + row) * BN + col] = (uint8_t)valid_corner(&e, c, row, col);
    }
}

void orc_blokus_board(int64_t B, const uint32_t *occ, int8_t *board)
{
    for (int64_t b = 0; b < B; ++b)
        for (int y = 0; y < BN; ++y)
            for (int x = 0; x < BN; ++x) {
                int v = 0;
                for (int c = 0; c < 4; ++c) if ((occ[(b * 4 + c) * BN + y] >> x) & 1u) v = c + 1;
                board[(b * BN + y) * BN + x] = (int8_t)v;
            }
}

/* BlokusEnvironment.state_to_observation (:721-768): board = -1 empty else (owner - observer) % 4, rotated with
 * np.rot90(k=-observer); pieces[r][i] = inventory bit i of player (r + observer) % 4; score rolled by -observer */
void orc_blokus_observe(int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score, const int8_t *player,
                        int8_t *obs_board, uint8_t *obs_pieces, int32_t *obs_score)
{
    for (int64_t b = 0; b < B; ++b) {
        int pl = player[b] & 3, rel[BN][BN];
        for (int y = 0; y < BN; ++y)
            for (int x = 0; x < BN; ++x) {
                int v = -1;
                for (int c = 0; c < 4; ++c) if ((occ[(b * 4 + c) * BN + y] >> x) & 1u) v = ((c - pl) % 4 + 4) % 4;
                rel[y][x] = v;
            }
        for (int i = 0; i < BN; ++i)
            for (int j = 0; j < BN; ++j) {
                int v;
                switch (pl) {                                    /* np.rot90(m, k=-pl) */
                case 0: v = rel[i][j]; break;
                case 1: v = rel[BN - 1 - j][i]; break;           /* one clockwise quarter turn */
                case 2: v = rel[BN - 1 - i][BN - 1 - j]; break;
                default: v = rel[j][BN - 1 - i]; break;          /* three clockwise = one counter-clockwise */
                }
                obs_board[(b * BN + i) * BN + j] = (int8_t)v;
            }
        for (int r = 0; r < 4; ++r) {
            int src = (r + pl) % 4;
            for (int i = 0; i < NP; ++i) obs_pieces[(b * 4 + r) * NP + i] = (uint8_t)((inv[b * 4 + src] >> i) & 1u);
            obs_score[b * 4 + r] = score[b * 4 + src];
        }
    }
}

static inline uint32_t mulhi32b(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* random agent: the mover plays the r-th action of valid_actions(), r = mulhi32(w, n), '' if n == 0;
 * w = philox(ctr={g, c>>2, 0, TAG_BLOKUS}, key=seed)[c&3], c = tcount.  auto-reset on terminal. */
void orc_blokus_rollout(int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                        uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                        orc_blokus_stats st, int n_threads)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t b = 0; b < B; ++b) {
        blk_env e;
        load_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round[b], to_move[b]);
        uint32_t tc = st.tcount[b], ts = st.tstep[b];
        uint32_t g = (uint32_t)(first_env_id + (uint64_t)b);
        int32_t *ids = (int32_t *)malloc(sizeof(int32_t) * 8192);
        for (int t = 0; t < T; ++t) {
            int n = enumerate_moves(&e, e.to_move, e.round, e.inv[e.to_move], ids, 8192, 0);
            uint32_t ctr[4] = { g, tc >> 2, 0u, ORC_TAG_BLOKUS }, w[4];
            orc_philox4x32(ctr, key, w);
            int action = n > 0 ? ids[mulhi32b(w[tc & 3u], (uint32_t)n)] : -1;
            tc += 1;
            int r, term, wm, status;
            step_env(&e, action, &r, &term, &wm, &status);                     /* a listed action: status is 0 */
            ts += 1;
            if (term) {
                st.n_episodes[b] += 1;
                st.len_sum[b] += ts;
                for (int q = 0; q < 4; ++q) {
                    if (wm >> q & 1) st.win_count[q * B + b] += 1;
                    st.score_sum[q * B + b] += e.score[q];
                }
                reset_env(&e);
                ts = 0;
            }
        }
        free(ids);
        store_env(&e, occ + b * 4 * BN, inv + b * 4, score + b * 4, round + b, to_move + b);
        st.tcount[b] = tc; st.tstep[b] = ts;
    }
}
