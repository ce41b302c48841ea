/* crl_oracle.c -- scalar CPU restatement of the Tron and TicTacToe hot path.
 * TEST INFRASTRUCTURE (see crl_oracle.h).  One environment at a time, plain C,
 * written from the reference's semantics; every function cites the reference
 * lines it restates.  Parity: pinned by tests/golden/ (generated from the
 * reference run in the build container) and the KATs of SURVEY.md section 8c.
 */
#include "crl_oracle.h"
#include <stdlib.h>
#include <string.h>

/* ======================= Philox-4x32-10 ==================================== */
/* Published algorithm: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as
 * easy as 1, 2, 3" (SC'11), Random123 v1.x constants. KATs in tests/test_oracle_rng.py. */
#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* ======================= Tron ============================================== */

/* Python slice [lo:hi:step] of a length-n sequence, step > 0, with Python's clipping. */
static int py_slice(const int *src, int n, long lo, long hi, int step, int *dst)
{
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    int m = 0;
    for (long i = lo; i < hi; i += step) dst[m++] = src[i];
    return m;
}

/* TronGridEnvironment.py:183-226.  The strided "right"/"left" slices and the
 * 4*side != len(ring) mismatch are reproduced as they are. */
int orc_tron_start_positions(int N, int P, int ring_offset, const int *offs,
                             int16_t *heads, int8_t *dirs)
{
    if (N < 2 || P < 1 || P > ORC_TRON_MAX_P) return -1;
    int size = N / 2, odd = N % 2;                 /* :187-188 */
    int r1 = size - ring_offset - 1, r2 = size - ring_offset;  /* :190 */
    int side = 2 * (r1 + 1);                       /* :191 */
    if (side <= 0) return -2;
    /* coordinates c[k] = -size + center + k with center = 0.5 (even N) or 0 (odd N) (:189,194);
     * compare doubled values so everything stays integer */
    int *idx = (int *)malloc(sizeof(int) * (size_t)N * N);
    int n_idx = 0;
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            int cy2 = 2 * (-size + y) + (odd ? 0 : 1), cx2 = 2 * (-size + x) + (odd ? 0 : 1);
            int ay = cy2 < 0 ? -cy2 : cy2, ax = cx2 < 0 ? -cx2 : cx2;
            int m1 = (ax <= 2 * r1) && (ay <= 2 * r1);   /* :195 */
            int m2 = (ax <= 2 * r2) && (ay <= 2 * r2);   /* :196 */
            if (m1 != m2) idx[n_idx++] = y * N + x;      /* :197-200 row-major np.where */
        }
    int cap = n_idx + 4 * side + 8;
    int *top = (int *)malloc(sizeof(int) * cap), *right = (int *)malloc(sizeof(int) * cap);
    int *bottom = (int *)malloc(sizeof(int) * cap), *left = (int *)malloc(sizeof(int) * cap);
    int *order = (int *)malloc(sizeof(int) * 4 * cap);
    int nt = py_slice(idx, n_idx, 0, side, 1, top);                    /* :203 */
    int nr = py_slice(idx, n_idx, side, 3L * side, 2, right);          /* :204 */
    int nb = py_slice(idx, n_idx, 3L * side, n_idx, 1, bottom);        /* :205 */
    int nl = py_slice(idx, n_idx, side + 1, 3L * side + 1, 2, left);   /* :206 */
    int L = 0;
    for (int i = 0; i < nt; ++i) order[L++] = top[i];                  /* :213 concatenate */
    for (int i = 0; i < nr; ++i) order[L++] = right[i];
    for (int i = nb - 1; i >= 0; --i) order[L++] = bottom[i];
    for (int i = nl - 1; i >= 0; --i) order[L++] = left[i];
    int LD = 4 * side;                                                 /* :209 */
    int rc = 0;
    /* np.array_split: first (len % P) sections hold len/P+1 items, the rest len/P (:213-214) */
    int so = 0, sd = 0;
    for (int p = 0; p < P; ++p) {
        int ls = L / P + (p < L % P ? 1 : 0);
        int ld = LD / P + (p < LD % P ? 1 : 0);
        if (ls <= 0 || ld <= 0) { rc = -3; break; }
        int is = ls / 2 + offs[p];                                     /* :218 clamp */
        if (is < 0) is = 0; if (is > ls - 1) is = ls - 1;
        int id = ld / 2 + offs[p];
        if (id < 0) id = 0; if (id > ld - 1) id = ld - 1;
        heads[p] = (int16_t)order[so + is];
        dirs[p] = (int8_t)((((sd + id) / side) + 2) % 4);              /* :209-210 */
        so += ls; sd += ld;
    }
    free(idx); free(top); free(right); free(bottom); free(left); free(order);
    return rc;
}

/* TronGridEnvironment.py:256-261 */
void orc_tron_reset(int N, int P, int64_t B, const int16_t *start_heads, const int8_t *start_dirs,
                    int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths)
{
    int NN = N * N;
    for (int64_t b = 0; b < B; ++b) {
        int8_t *bd = board + b * NN;
        memset(bd, 0, (size_t)NN);
        for (int p = 0; p < P; ++p) {
            heads[p * B + b] = start_heads[p];
            dirs[p * B + b] = start_dirs[p];
            deaths[p * B + b] = 0;
            bd[start_heads[p]] = (int8_t)(p + 1);
        }
    }
}

/* one env; h/d/k are gathered copies of heads/dirs/deaths */
static void tron_step_env(int N, int P, int8_t *bd, int *h, int *d, int *k, const int *act,
                          int *rew, int *term, int *winmask)
{
    /* CyTronGrid.pyx:15-62 */
    for (int i = 0; i < P; ++i) {
        if (k[i] > 0) continue;                        /* :16 */
        int x = h[i] % N, y = h[i] / N;                /* :21-22 */
        int dir = (d[i] + act[i] + 4) % 4;             /* :31 */
        if (dir == 0) y -= 1;                          /* :34-41 */
        else if (dir == 1) x += 1;
        else if (dir == 2) y += 1;
        else x -= 1;
        d[i] = dir;                                    /* :44 written even if the move dies */
        if (x < 0 || x >= N || y < 0 || y >= N) {
            k[i] = i + 1;                              /* :47-48 */
        } else if (bd[y * N + x] > 0) {
            int enemy = bd[y * N + x];                 /* :51-53 */
            k[i] = enemy;
            if (h[enemy - 1] == N * y + x)             /* :56-57 head-on: owner dies too */
                k[enemy - 1] = i + 1;
        } else {
            bd[y * N + x] = (int8_t)(i + 1);           /* :60-62 */
            h[i] = N * y + x;
        }
    }
    /* TronGridEnvironment.py:309-321 */
    int alive = 0, mask = 0;
    for (int i = 0; i < P; ++i) if (k[i] == 0) { alive++; mask |= 1 << i; }
    *term = alive <= 1;
    *winmask = *term ? mask : 0;
    for (int i = 0; i < P; ++i) {
        rew[i] = k[i] > 0 ? -1 : 1;
        if (*term && k[i] == 0) rew[i] += 9;
    }
}

void orc_tron_step(int N, int P, int64_t B,
                   int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                   const int8_t *actions, int8_t *rewards, uint8_t *terminal, uint8_t *winners)
{
    int NN = N * N;
    for (int64_t b = 0; b < B; ++b) {
        int h[ORC_TRON_MAX_P], d[ORC_TRON_MAX_P], k[ORC_TRON_MAX_P], a[ORC_TRON_MAX_P], r[ORC_TRON_MAX_P];
        for (int p = 0; p < P; ++p) {
            h[p] = heads[p * B + b]; d[p] = dirs[p * B + b]; k[p] = deaths[p * B + b]; a[p] = actions[p * B + b];
        }
        int term, wm;
        tron_step_env(N, P, board + b * NN, h, d, k, a, r, &term, &wm);
        for (int p = 0; p < P; ++p) {
            heads[p * B + b] = (int16_t)h[p]; dirs[p * B + b] = (int8_t)d[p];
            deaths[p * B + b] = (int8_t)k[p]; rewards[p * B + b] = (int8_t)r[p];
        }
        terminal[b] = (uint8_t)term;
        winners[b] = (uint8_t)wm;
    }
}

void orc_tron_rollout(int N, int P, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                      const int16_t *start_heads, const int8_t *start_dirs,
                      int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                      orc_tron_stats st, int n_threads)
{
    int NN = N * N;
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t b = 0; b < B; ++b) {
        int8_t *bd = board + b * NN;
        int h[ORC_TRON_MAX_P], d[ORC_TRON_MAX_P], k[ORC_TRON_MAX_P], a[ORC_TRON_MAX_P], r[ORC_TRON_MAX_P];
        for (int p = 0; p < P; ++p) { h[p] = heads[p * B + b]; d[p] = dirs[p * B + b]; k[p] = deaths[p * B + b]; }
        uint32_t tc = st.tcount[b], ts = st.tstep[b];
        uint32_t g = (uint32_t)(first_env_id + (uint64_t)b);
        static const uint32_t POW3[8] = { 1u, 3u, 9u, 27u, 81u, 243u, 729u, 2187u };
        uint32_t wq[2][4];                              /* the Philox block of the current 8 steps, players 0-3 / 4-7 */
        for (int t = 0; t < T; ++t) {
            for (int q = 0; q < (P + 3) / 4; ++q) {
                uint32_t *w = wq[q];
                if (t == 0 || (tc & 7u) == 0u) {            /* one block serves 8 consecutive steps (the contract in the header): */
                    uint32_t ctr[4] = { g, tc >> 3, (uint32_t)q, ORC_TAG_TRON };   /* recomputed only when tc >> 3 moves on */
                    orc_philox4x32(ctr, key, w);
                }
                uint32_t j = tc & 7u, word = w[j >> 1];
                for (int i = 0; i < 4 && q * 4 + i < P; ++i) {
                    uint32_t v = word * POW3[(j & 1u) * 4u + (uint32_t)i];
                    uint32_t a3 = mulhi32(v, 3u);
                    a[q * 4 + i] = a3 == 0 ? 0 : (a3 == 1 ? 1 : -1);
                }
            }
            tc += 1;
            int term, wm;
            tron_step_env(N, P, bd, h, d, k, a, r, &term, &wm);
            ts += 1;
            for (int p = 0; p < P; ++p) st.ret_sum[p * B + b] += r[p];
            if (term) {
                st.n_episodes[b] += 1;
                st.len_sum[b] += ts;
                st.last_len[b] = (uint16_t)ts;
                st.last_winners[b] = (uint8_t)wm;
                for (int p = 0; p < P; ++p) if (wm >> p & 1) st.win_count[p * B + b] += 1;
                memset(bd, 0, (size_t)NN);
                for (int p = 0; p < P; ++p) {
                    h[p] = start_heads[p]; d[p] = start_dirs[p]; k[p] = 0;
                    bd[start_heads[p]] = (int8_t)(p + 1);
                }
                ts = 0;
            }
        }
        for (int p = 0; p < P; ++p) {
            heads[p * B + b] = (int16_t)h[p]; dirs[p * B + b] = (int8_t)d[p]; deaths[p * B + b] = (int8_t)k[p];
        }
        st.tcount[b] = tc; st.tstep[b] = ts;
    }
}

/* TronGridEnvironment.py:385-405 + CyTronGrid.pyx:65-71 (fully observable branch).
 * relabel v>0 -> ((v-(pl+1)+P)%P)+1 with C's remainder (CyTronGrid.pyx:1 cdivision=True: for an observer id beyond P low
 * trail ids come out <= 0); roll heads/dirs/deaths with numpy's modulo so index 0 is the observer (any integer id);
 * killer ids stored inside deaths stay absolute. */
void orc_tron_observe(int N, int P, int64_t B, const int8_t *board, const int16_t *heads,
                      const int8_t *dirs, const int8_t *deaths, const int8_t *player,
                      int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths)
{
    int NN = N * N;
    for (int64_t b = 0; b < B; ++b) {
        int pl = player[b];
        for (int c = 0; c < NN; ++c) {
            int v = board[b * NN + c];
            obs_board[b * NN + c] = (int8_t)(v > 0 ? ((v - (pl + 1) + P) % P) + 1 : v);
        }
        for (int i = 0; i < P; ++i) {
            int src = (((i + pl) % P) + P) % P;                   /* numpy's modulo (:393): negative observer ids wrap */
            obs_heads[i * B + b] = heads[src * B + b];
            obs_dirs[i * B + b] = dirs[src * B + b];
            obs_deaths[i * B + b] = deaths[src * B + b];
        }
    }
}

/* TronGridEnvironment.compute_ranking (:483-508).  score = cells carrying the player's id; for every i
 * with deaths[deaths[i]-1] == i+1 (an alive i reads deaths[-1], i.e. the LAST player's entry) the score becomes
 * min(score[i], score[deaths[i]-1]) -- and for an alive i that "killer" index is -1, a key the Counter no
 * longer holds, so the min is taken with 0; ranks are competition ranks by descending score. */
void orc_tron_ranking(int N, int P, int64_t B, const int8_t *board, const int8_t *deaths, int8_t *rank)
{
    int NN = N * N;
    for (int64_t b = 0; b < B; ++b) {
        int score[ORC_TRON_MAX_P] = {0};
        for (int c = 0; c < NN; ++c) {
            int v = board[b * NN + c];
            if (v >= 1 && v <= P) score[v - 1]++;
        }
        for (int i = 0; i < P; ++i) {                             /* :492-495, ascending like np.where */
            int di = deaths[i * B + b];
            int via = di > 0 ? di - 1 : P - 1;                    /* python index -1 = last player */
            if (deaths[via * B + b] == i + 1) {
                int other = di > 0 ? score[di - 1] : 0;           /* scores[-1] is a missing Counter key -> 0 */
                if (other < score[i]) score[i] = other;
            }
        }
        for (int i = 0; i < P; ++i) {                             /* :497-506 */
            int higher = 0;
            for (int q = 0; q < P; ++q) higher += score[q] > score[i];
            rank[i * B + b] = (int8_t)higher;
        }
    }
}

/* ======================= TicTacToe ========================================= */

/* WINNING_SHAPES (2p:12-17; 4p:19-38) slid over the board in 'valid' mode: every window of K
 * collinear cells along an axis, a face diagonal or a space diagonal.  Board cell (i,j,k)
 * has flat index (i*D1+j)*D2+k; 2-D boards use D0=1... note the 2-D envs index (row, col),
 * i.e. D1=rows, D2=cols.  13 directions = half of the 26 neighbours. */
int orc_ttt_lines(int D0, int D1, int D2, int K, uint32_t *lines)
{
    int n = 0;
    if (D0 * D1 * D2 > 32) return -1;
    for (int di = 0; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj)
            for (int dk = -1; dk <= 1; ++dk) {
                /* canonical half-space: first non-zero component positive */
                if (di == 0 && dj < 0) continue;
                if (di == 0 && dj == 0 && dk <= 0) continue;
                for (int i = 0; i < D0; ++i)
                    for (int j = 0; j < D1; ++j)
                        for (int k = 0; k < D2; ++k) {
                            int ei = i + di * (K - 1), ej = j + dj * (K - 1), ek = k + dk * (K - 1);
                            if (ei < 0 || ei >= D0 || ej < 0 || ej >= D1 || ek < 0 || ek >= D2) continue;
                            uint32_t m = 0;
                            for (int s = 0; s < K; ++s)
                                m |= 1u << (((i + di * s) * D1 + (j + dj * s)) * D2 + (k + dk * s));
                            if (n >= ORC_TTT_MAX_LINES) return -2;
                            lines[n++] = m;
                        }
            }
    return n;
}

static void ttt_step_env(int n_cells, int P, int n_lines, const uint32_t *lines,
                         uint32_t *occ /*[P]*/, int *winner, int *to_move, int action,
                         int *reward, int *terminal, int *winners)
{
    int pl = *to_move;
    uint32_t all = 0;
    for (int p = 0; p < P; ++p) all |= occ[p];
    uint32_t full = n_cells == 32 ? 0xffffffffu : ((1u << n_cells) - 1u);
    *reward = 0; *terminal = 0; *winners = -1;
    /* 2p:293  valid iff action non-empty, target cell empty, and no sticky winner */
    if (action >= 0 && action < n_cells && !((all >> action) & 1u) && *winner < 0) {
        occ[pl] |= 1u << action;                         /* :295 */
        all |= 1u << action;
        for (int l = 0; l < n_lines; ++l)                /* :296-300 any complete window of the mover */
            if ((occ[pl] & lines[l]) == lines[l]) { *winner = pl; break; }
    }
    if (*winner >= 0) {                                  /* :302-308 */
        *reward = (*winner == pl) ? 1 : -1;
        *winners = *winner;
        *terminal = 1;
    }
    if (all == full) *terminal = 1;                      /* :310-311 no empty cell left */
    *to_move = (pl + 1) % P;                             /* :313 always advances */
}

void orc_ttt_step(int n_cells, int P, int n_lines, const uint32_t *lines, int64_t B,
                  uint32_t *occ, int8_t *winner, int8_t *to_move,
                  const int8_t *action, int8_t *reward, uint8_t *terminal, int8_t *winners)
{
    for (int64_t b = 0; b < B; ++b) {
        uint32_t o[ORC_TTT_MAX_P];
        for (int p = 0; p < P; ++p) o[p] = occ[p * B + b];
        int w = winner[b], tm = to_move[b], r, t, ws;
        ttt_step_env(n_cells, P, n_lines, lines, o, &w, &tm, action[b], &r, &t, &ws);
        for (int p = 0; p < P; ++p) occ[p * B + b] = o[p];
        winner[b] = (int8_t)w; to_move[b] = (int8_t)tm;
        reward[b] = (int8_t)r; terminal[b] = (uint8_t)t; winners[b] = (int8_t)ws;
    }
}

/* index of the r-th (0-based) set bit of m, scanning from bit 0 (row-major np.where order, 2p:345) */
static int nth_set_bit(uint32_t m, int r)
{
    for (int c = 0; c < 32; ++c)
        if ((m >> c) & 1u) { if (r == 0) return c; --r; }
    return -1;
}

void orc_ttt_rollout(int n_cells, int P, int n_lines, const uint32_t *lines, int64_t B,
                     uint64_t seed, uint64_t first_env_id, int T,
                     uint32_t *occ, int8_t *winner, int8_t *to_move, orc_ttt_stats st, int n_threads)
{
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    uint32_t full = n_cells == 32 ? 0xffffffffu : ((1u << n_cells) - 1u);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t b = 0; b < B; ++b) {
        uint32_t o[ORC_TTT_MAX_P];
        for (int p = 0; p < P; ++p) o[p] = occ[p * B + b];
        int w = winner[b], tm = to_move[b];
        uint32_t tc = st.tcount[b], ts = st.tstep[b];
        uint32_t g = (uint32_t)(first_env_id + (uint64_t)b);
        for (int t = 0; t < T; ++t) {
            uint32_t all = 0;
            for (int p = 0; p < P; ++p) all |= o[p];
            uint32_t empty = full & ~all;
            int n_empty = __builtin_popcount(empty);
            /* RNG contract (round 3): one Philox call serves EIGHT plies, a 32-bit word two.  The ply at an even step counter
               reads the word w itself; the ply at an odd one reads lo32(w * (n_empty + 1)) -- what the even ply's extraction
               hi32(w * n) left over, when that ply was the one before in the same game (n = n_empty + 1 then): the two
               choices are the two leading digits of w in the mixed radix (n, n - 1).  The definition does not look back,
               so a ply can be computed from (seed, game, step counter, board) alone. */
            uint32_t ctr[4] = { g, tc >> 3, 0u, ORC_TAG_TTT }, rnd[4];
            orc_philox4x32(ctr, key, rnd);
            uint32_t word = rnd[(tc >> 1) & 3u];
            if (tc & 1u) word *= (uint32_t)(n_empty + 1);
            int action = n_empty ? nth_set_bit(empty, (int)mulhi32(word, (uint32_t)n_empty)) : -1;
            tc += 1;
            int r, term, ws;
            ttt_step_env(n_cells, P, n_lines, lines, o, &w, &tm, action, &r, &term, &ws);
            ts += 1;
            if (term) {
                st.n_episodes[b] += 1;
                st.len_sum[b] += ts;
                if (ws >= 0) st.win_count[ws * B + b] += 1; else st.draw_count[b] += 1;
                for (int p = 0; p < P; ++p) o[p] = 0;
                w = -1; tm = 0; ts = 0;
            }
        }
        for (int p = 0; p < P; ++p) occ[p * B + b] = o[p];
        winner[b] = (int8_t)w; to_move[b] = (int8_t)tm;
        st.tcount[b] = tc; st.tstep[b] = ts;
    }
}
