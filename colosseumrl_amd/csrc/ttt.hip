// ttt.hip -- batched n-player TicTacToe stepper for gfx950 (MI355X).  Hand-written HIP, wave64.
//
// Restates, for B independent games at once:
//   colosseumrl/envs/tictactoe/tictactoe_2p_env.py:139-169  new_state
//   colosseumrl/envs/tictactoe/tictactoe_2p_env.py:240-315  next_state   (3p/4p files: same logic,
//   colosseumrl/envs/tictactoe/tictactoe_2p_env.py:317-348  valid_actions  other board shape / P)
//   WINNING_SHAPES: 2p:12-17, 3p:12-17, 4p:19-38
//
// Layout: a board has <= 32 cells, so each player's marks are ONE uint32 per game: occ[P][B].
// One lane per game; a wave reads 64 consecutive words per player (fully coalesced 256-B rows).
// Win test: the reference correlates the mover's mask with each line pattern in 'valid' mode,
// i.e. "some K-window along one of the 13 (3-D) / 4 (2-D) directions is fully the mover's".
// With the mask in a register that is K-1 shift-ANDs per direction against a per-direction
// start mask (windows that stay on the board).  Direction strides and start masks are
// wave-uniform kernel arguments (SGPRs) -- no table traffic at all.
#include "crl_common.hpp"
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace {


// ND = 4 (boards with at most 4 line directions: everything 1-D / 2-D) or 13 (3-D).  Strides and start masks are plain
// SGPR operands and no direction needs its own branch (unused slots have start == 0).  A K-window along stride s is
// built by doubling: t = m & (m >> s) is "two in a row", t & (t >> 2s) four, ... -- for the pinned K = 3 / 4 / 5 that is
// two or three shift-and pairs per direction (K is wave-uniform: one scalar branch) where the plain K-1 rounds of the
// generic loop take K-1 pairs plus the copies that seed them.
// KC: K when the caller has branched on it already (3, 4, 5), else 0 = ask dd.K here.
template <int ND, int KC = 0>
__device__ __forceinline__ bool ttt_has_line(const ttt_dirs &dd, const uint32_t m)
{
    uint32_t hit = 0;
    const int K = KC ? KC : dd.K;
    if (K == 3) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int s = dd.stride[d];
            hit |= m & (m >> s) & (m >> (2 * s)) & dd.start[d];
        }
    } else if (K == 4) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int s = dd.stride[d];
            const uint32_t two = m & (m >> s);
            hit |= two & (two >> (2 * s)) & dd.start[d];
        }
    } else if (K == 5) {
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const int s = dd.stride[d];
            const uint32_t two = m & (m >> s);
            hit |= two & (two >> (2 * s)) & (m >> (4 * s)) & dd.start[d];
        }
    } else {
        uint32_t t[ND], run[ND];
#pragma unroll
        for (int d = 0; d < ND; ++d) t[d] = run[d] = m;
        for (int s = 1; s < K; ++s) {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                t[d] >>= dd.stride[d];              // cell c + s * stride of direction d, seen from c
                run[d] &= t[d];
            }
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) hit |= run[d] & dd.start[d];
    }
    return hit != 0;
}

template <int P, int ND>
__device__ __forceinline__ void ttt_step_core(const ttt_dirs &dd, uint32_t (&o)[P], int &winner, int &to_move,
                                              const int action, int &reward, int &term, int &winners)
{
    const int pl = to_move;
    uint32_t all = 0, mine = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) { all |= o[p]; mine = (p == pl) ? o[p] : mine; }
    // tictactoe_2p_env.py:293: non-empty action, target cell empty, no sticky winner
    const bool in_range = (action >= 0) & (action < dd.n_cells);
    const uint32_t bit = in_range ? (1u << (action & 31)) : 0u;
    const bool ok = in_range && !(all & bit) && winner < 0;
    if (ok) {
        mine |= bit;                                   // :295
        all |= bit;
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] = (p == pl) ? mine : o[p];
        if (ttt_has_line<ND>(dd, mine)) winner = pl;   // :296-300
    }
    reward = 0; term = 0; winners = -1;
    if (winner >= 0) {                                 // :302-308
        reward = (winner == pl) ? 1 : -1;
        winners = winner;
        term = 1;
    }
    if (all == dd.full) term = 1;                      // :310-311
    to_move = (pl + 1 == P) ? 0 : pl + 1;              // :313
}

// index of the r-th (0-based) set bit of m (row-major np.where order, tictactoe_2p_env.py:345): binary search over
// bit fields.  Per level: the count below the midpoint (bit-field extract + popcount), the rank moves on by an UNSIGNED
// min (r - c wraps to a huge value exactly when r < c), the position by a compare.
__device__ __forceinline__ int nth_set_bit(const uint32_t m, int r)
{
    // (Round 3 tried the decision of a level as a sign mask -- t = rr - c, neg = t >> 31 (arithmetic), pos |= w & ~neg --
    //  which keeps the chain off VCC and so saves the wait state (s_nop) each level's select needs behind the VALU write
    //  of VCC; but the mask costs a VALU instruction and lengthens the dependent chain: 1.107 -> 1.180 ms per 2048 plies.)
    uint32_t rr = (uint32_t)r, pos = 0, c;
    c = (uint32_t)__popc(m & 0xffffu);                              pos = (rr >= c) ? 16u : 0u;          rr = min(rr, rr - c);
    c = (uint32_t)__popc(__builtin_amdgcn_ubfe(m, pos, 8u));        pos = (rr >= c) ? pos + 8u : pos;    rr = min(rr, rr - c);
    c = (uint32_t)__popc(__builtin_amdgcn_ubfe(m, pos, 4u));        pos = (rr >= c) ? pos + 4u : pos;    rr = min(rr, rr - c);
    c = (uint32_t)__popc(__builtin_amdgcn_ubfe(m, pos, 2u));        pos = (rr >= c) ? pos + 2u : pos;    rr = min(rr, rr - c);
    c = __builtin_amdgcn_ubfe(m, pos, 1u);                          pos = (rr >= c) ? pos + 1u : pos;
    return (int)pos;
}

// The same search with its last three levels out of a table: [byte][rank] = position of the rank-th set bit of the byte,
// 2 KB of LDS filled by the 256 threads of the workgroup (ttt_fill_rank_table; the caller puts a __syncthreads behind it).
// Two levels by arithmetic find the byte, one LDS read the bit in it: 15 vector instructions + 1 LDS read where the
// five-level search takes 26 (in the rollout's ply, where this chain is a third of the instructions).
__device__ __forceinline__ void ttt_fill_rank_table(uint8_t (&tab)[256 * 8])
{
    const uint32_t t = threadIdx.x & 255u;
    uint32_t lo = 0, hi = 0, n = 0;
#pragma unroll
    for (uint32_t bit = 0; bit < 8; ++bit) {
        const bool set = (t >> bit) & 1u;
        const uint32_t v = set ? bit << ((n & 3u) * 8u) : 0u;
        lo |= (n < 4u) ? v : 0u;
        hi |= (n < 4u) ? 0u : v;
        n += set ? 1u : 0u;
    }
    *reinterpret_cast<uint2 *>(&tab[t * 8]) = make_uint2(lo, hi);
}

// SMALL: the board has at most 16 cells (3x3, 3x5): the first level falls away.
template <bool SMALL = false>
__device__ __forceinline__ uint32_t nth_set_bit_tab(const uint8_t (&tab)[256 * 8], const uint32_t m, const uint32_t r)
{
    uint32_t rr = r, pos = 0, c;
    if (!SMALL) { c = (uint32_t)__popc(m & 0xffffu);                pos = (rr >= c) ? 16u : 0u;          rr = min(rr, rr - c); }
    c = (uint32_t)__popc(__builtin_amdgcn_ubfe(m, pos, 8u));        pos = (rr >= c) ? pos + 8u : pos;    rr = min(rr, rr - c);
    CRL_BOUNDS_LT(rr, 8u, 201);                                  // the rank left for the byte table is a rank INSIDE that byte
    return pos + tab[__builtin_amdgcn_ubfe(m, pos, 8u) * 8u + rr];
}

// the random agent's 32-bit draw for the ply at step counter c of game g with n_empty empty cells (the RNG contract of
// ttt_rollout_kernel, stated there): the ply picks empty cell number hi32(draw * n_empty)
__device__ __forceinline__ uint32_t ttt_agent_word(const uint32_t g, const uint32_t c, const int n_empty, const uint32_t seed_lo,
                                                   const uint32_t seed_hi)
{
    const philox_out rnd = philox4x32_10<true>(g, c >> 3, 0u, CRL_TAG_TTT, seed_lo, seed_hi);
    const uint32_t sel = (c >> 1) & 3u;
    const uint32_t w = sel == 0 ? rnd.w[0] : sel == 1 ? rnd.w[1] : sel == 2 ? rnd.w[2] : rnd.w[3];
    return (c & 1u) ? w * (uint32_t)(n_empty + 1) : w;
}

template <int P, int ND>
__global__ void __launch_bounds__(256)
ttt_step_kernel(const ttt_dirs dd, const int64_t B, uint32_t *__restrict__ occ, int8_t *__restrict__ winner,
                int8_t *__restrict__ to_move, const int8_t *__restrict__ action, int8_t *__restrict__ reward,
                uint8_t *__restrict__ terminal, int8_t *__restrict__ winners, const uint32_t flags)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t o[P];
#pragma unroll
    for (int p = 0; p < P; ++p) o[p] = occ[p * B + b];
    int w = winner[b], tm = to_move[b], r, t, ws;
    ttt_step_core<P, ND>(dd, o, w, tm, action[b], r, t, ws);
    reward[b] = (int8_t)r;
    terminal[b] = (uint8_t)t;
    winners[b] = (int8_t)ws;
    if (t && (flags & CRL_STEP_AUTO_RESET)) {
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] = 0;
        w = -1; tm = 0;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) occ[p * B + b] = o[p];
    winner[b] = (int8_t)w;
    to_move[b] = (int8_t)tm;
}

// WT: the board has at most 16 cells and win_tab is its table of winning masks (crl_ttt_create); ND = 4 then
template <int P, int ND, bool WT = false>
__global__ void __launch_bounds__(256)
ttt_rollout_kernel(const ttt_dirs dd, const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi,
                   const uint64_t first_env_id, const int T, uint32_t *__restrict__ occ,
                   int8_t *__restrict__ winner, int8_t *__restrict__ to_move, const crl_ttt_stats st,
                   const uint32_t *__restrict__ win_tab)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ uint8_t rank_tab[256 * 8];
    __shared__ uint32_t win_bits[WT ? 2048 : 1];             // boards of <= 16 cells: bit m = the mask m holds a K-line
    ttt_fill_rank_table(rank_tab);
    if (WT) {
        const uint4 *src = reinterpret_cast<const uint4 *>(win_tab);
        uint4 *dst = reinterpret_cast<uint4 *>(win_bits);
        dst[threadIdx.x] = src[threadIdx.x];
        dst[threadIdx.x + 256] = src[threadIdx.x + 256];
    }
    __syncthreads();
    if (b >= B) return;
    uint32_t o[P], wins[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { o[p] = occ[p * B + b]; wins[p] = 0; }
    int w = winner[b], tm = to_move[b];
    uint32_t tc = st.tcount[b], ts = st.tstep[b], n_ep = 0, draws = 0, len_sum = 0;
    const uint32_t g = (uint32_t)(first_env_id + (uint64_t)b);
    // RNG contract (round 3, identical in the oracle): one Philox call serves EIGHT plies, a 32-bit word two -- block
    // tc >> 3, word (tc >> 1) & 3.  The ply at an even step counter reads the word w itself, the ply at an odd one
    // lo32(w * (n_empty + 1)): what the even ply's extraction hi32(w * n) left over when that ply was the one before in the
    // same game (n = n_empty + 1 then; the two choices are the leading digits of w in the mixed radix (n, n - 1), bias
    // < n^2 / 2^32).  The definition does not look back, so any ply follows from (seed, game, step counter, board) alone.
    // Round 2 drew one word per ply: the ten Philox rounds were 36 % of a ply's vector instructions.
    philox_out rnd = philox4x32_10<true>(g, tc >> 3, 0u, CRL_TAG_TTT, seed_lo, seed_hi);
    // one ply with the Philox word `pw` of its step counter (odd: that counter is odd)
    auto ply = [&](const uint32_t pw, const uint32_t odd) {
        uint32_t all = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) all |= o[p];
        const uint32_t empty = dd.full & ~all;
        const int n_empty = __popc(empty);
        const uint32_t word = odd ? pw * (uint32_t)(n_empty + 1) : pw;
        const int action = n_empty ? nth_set_bit(empty, (int)__umulhi(word, (uint32_t)n_empty)) : -1;
        int r, term, ws;
        ttt_step_core<P, ND>(dd, o, w, tm, action, r, term, ws);
        ts += 1;
        if (term) {
            n_ep += 1;
            len_sum += ts;
            draws += (ws < 0);
#pragma unroll
            for (int p = 0; p < P; ++p) { wins[p] += (ws == p); o[p] = 0; }
            w = -1; tm = 0; ts = 0;
        }
    };
    // The same ply for a game that is RUNNING when the ply starts (no sticky winner, board not full) -- which every game is
    // once it has been through a ply of this kernel, because a finished game restarts at once: the random agent's move
    // is legal by construction, so next_state's checks (in range, cell empty, no winner yet: tictactoe_2p_env.py:293) and
    // the empty-board-is-full corner fall away, and the winner is known the moment the line test says so.
    // Round 3: the players' masks ROTATE through the registers r[] -- r[0] is always the mover's, because the players move
    // in strict rotation (:313) and a restart empties every mask, so that the rotation's phase means nothing across it.
    // The mover's mask then costs no select chain (round 2: P compares, 2 P selects and an or3 per ply) and, the plies
    // being unrolled, the rotation itself is register renaming.  Episode lengths are not summed per episode either: the
    // lengths of the episodes a lane finished add up to the plies it played minus the length of the unfinished one, so
    // a lane only remembers where its current episode began (ep0, a ply index of this launch).  And the outcomes are
    // counted in ONE register of byte fields -- field p the wins of player p, field P the draws (P = 8: the draws are the
    // episodes nobody won) -- which a finished episode bumps with a select, a shift and an add (per-player counters: a
    // compare, a select and an add EACH, in the block some lane of the wave runs on practically every ply); the fields are
    // flushed into the 32-bit counters before any can reach 256, i.e. at least every 248 plies.
    uint32_t all_run = 0;                                       // every player's marks (kept across running plies)
    uint32_t r[P];
    int ep0 = 0;                                                // ply index (of this launch) at which the current episode began
    constexpr bool DRAW_FIELD = P < 8;
    using acc_t = std::conditional_t<(P <= 3), uint32_t, uint64_t>;
    acc_t acc = 0;
    int tm8 = 0;                                                // 8 * (player to move) while the running plies are at it
    auto flush = [&]() {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint32_t f = (uint32_t)(acc >> (8 * p)) & 0xffu;
            wins[p] += f;
            n_ep += f;
        }
        if (DRAW_FIELD) {
            const uint32_t f = (uint32_t)(acc >> (8 * (P & 7))) & 0xffu;
            draws += f;
            n_ep += f;
        }
        acc = 0;
    };
    auto ply_running = [&](auto k_tag, const uint32_t w, const uint32_t odd, const int t_after) {   // k_tag: K as a compile-time constant (0: read it from dd)
        constexpr int KC = decltype(k_tag)::value & 7;            // k_tag: K (0: read it from dd) | 8 for boards of <= 16 cells
        constexpr bool SMALL = (decltype(k_tag)::value & 8) != 0;  //        | 16: ... whose win test is a table lookup
        constexpr bool TABLE = WT && (decltype(k_tag)::value & 16) != 0;
        const uint32_t empty = dd.full & ~all_run;              // (not 0: the game is running)
        const uint32_t n_empty = (uint32_t)__popc(empty);
        const uint32_t word = odd ? w * (n_empty + 1u) : w;     // (odd is a constant in the unrolled trips)
        const uint32_t bit = 1u << nth_set_bit_tab<SMALL>(rank_tab, empty, __umulhi(word, n_empty));
        const uint32_t mine = r[0] | bit;                                              // :295
        bool won;                                                                      // :296-300
        if constexpr (TABLE) {
            CRL_BOUNDS_LT(mine >> 5, 2048u, 202);               // masks of boards with at most 16 cells
            won = ((win_bits[mine >> 5] >> (mine & 31u)) & 1u) != 0u;
        }
        else won = ttt_has_line<ND, KC>(dd, mine);
        all_run |= bit;
        const bool term = won | (all_run == dd.full);                                  // :302-311
#pragma unroll
        for (int p = 0; p + 1 < P; ++p) r[p] = r[p + 1];
        r[P - 1] = mine;
        const int pl8 = tm8;
        if constexpr (P <= 4) {                                                        // :313: the next mover out of a constant of
            constexpr uint32_t kNext = P == 1 ? 0u : P == 2 ? 0x0008u : P == 3 ? 0x001008u : 0x00181008u;   // bytes (one v_bfe)
            tm8 = (int)__builtin_amdgcn_ubfe(kNext, (uint32_t)pl8, 8u);
        } else {
            tm8 = (pl8 + 8 == 8 * P) ? 0 : pl8 + 8;
        }
        // (The end of an episode as selects instead of a branch -- some lane of the wave ends one on practically every ply
        //  -- is a wash: +0.7 % at 5x5, -2.3 % at 3x3x3; the selects cost 6 more vector instructions per ply.)
        if (term) {
            if (DRAW_FIELD) {
                acc += (acc_t)1 << (won ? pl8 : 8 * P);
            } else {
                acc += (acc_t)(won ? 1u : 0u) << pl8;
                draws += won ? 0u : 1u;
                n_ep += won ? 0u : 1u;
            }
#pragma unroll
            for (int p = 0; p < P; ++p) r[p] = 0;
            tm8 = 0; all_run = 0; ep0 = t_after;
        }
    };
    auto next_word = [&](uint32_t &odd) -> uint32_t {           // the word of step counter tc, then on to tc + 1
        const uint32_t sel = (tc >> 1) & 3u;
        odd = tc & 1u;
        uint32_t word = rnd.w[0];
        word = (sel == 1) ? rnd.w[1] : word;
        word = (sel == 2) ? rnd.w[2] : word;
        word = (sel == 3) ? rnd.w[3] : word;
        tc += 1;
        if ((tc & 7u) == 0) rnd = philox4x32_10<true>(g, tc >> 3, 0u, CRL_TAG_TTT, seed_lo, seed_hi);
        return word;
    };
    int t = 0;
    {   // a state that came in finished but not restarted (stepped without auto-reset, or hand-made): one general ply
        uint32_t all = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) all |= o[p];
        if (T > 0 && __builtin_amdgcn_ballot_w64(w >= 0 || all == dd.full || (unsigned)tm >= (unsigned)P) != 0ull) {
            uint32_t odd;
            const uint32_t w0 = next_word(odd);
            ply(w0, odd);
            t = 1;
        }
    }
    // into the rotating order: r[i] = the marks of player (tm + i) mod P
    const int tm_in = (unsigned)tm < (unsigned)P ? tm : 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        int q = tm_in + i;
        q = q >= P ? q - P : q;
        uint32_t v = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) { v = (q == p) ? o[p] : v; }
        r[i] = v;
        all_run |= v;
    }
    tm8 = 8 * tm_in;
    ep0 = t - (int)ts;                                          // the current episode is ts plies old
    const int ep0_in = ep0;
    // When every game of the wave stands at a step counter that is a multiple of eight (launches of 8 k steps keep it so)
    // the plies run in trips of eight -- one Philox call -- with word and parity picked at compile time: no per-ply select
    // chain, no refill test.
    // (K is wave-uniform: one branch here instead of one per ply)
    auto run_plies = [&](auto k_tag) {
        if (__builtin_amdgcn_ballot_w64((tc & 7u) != 0u) == 0ull) {
            for (int trips = 0; t + 8 <= T; t += 8) {
                if (++trips == 32) { flush(); trips = 1; }      // <= 31 trips = 248 plies between flushes
                ply_running(k_tag, rnd.w[0], 0u, t + 1);
                ply_running(k_tag, rnd.w[0], 1u, t + 2);
                ply_running(k_tag, rnd.w[1], 0u, t + 3);
                ply_running(k_tag, rnd.w[1], 1u, t + 4);
                ply_running(k_tag, rnd.w[2], 0u, t + 5);
                ply_running(k_tag, rnd.w[2], 1u, t + 6);
                ply_running(k_tag, rnd.w[3], 0u, t + 7);
                ply_running(k_tag, rnd.w[3], 1u, t + 8);
                tc += 8;
                rnd = philox4x32_10<true>(g, tc >> 3, 0u, CRL_TAG_TTT, seed_lo, seed_hi);
            }
        }
        flush();
        for (; t < T; ++t) {
            uint32_t odd;
            const uint32_t w1 = next_word(odd);
            ply_running(k_tag, w1, odd, t + 1);
            if ((t & 127) == 127) flush();
        }
        flush();
    };
    if constexpr (WT) {
        run_plies(std::integral_constant<int, 8 | 16>{});        // 3x3, 3x5 (the reference's 2p / 3p boards), 4x4, ...
    } else {
        if (dd.K == 3 && dd.n_cells <= 16) run_plies(std::integral_constant<int, 3 | 8>{});
        else if (dd.K == 3) run_plies(std::integral_constant<int, 3>{});
        else if (dd.K == 4) run_plies(std::integral_constant<int, 4>{});
        else run_plies(std::integral_constant<int, 0>{});
    }
    // back out of the rotating order, and the bookkeeping the running plies left implicit
    {
        tm = tm8 >> 3;
        const int tm_out = tm;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int i = p - tm_out;
            i = i < 0 ? i + P : i;
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < P; ++k) { v = (i == k) ? r[k] : v; }
            o[p] = v;
        }
        ts = (uint32_t)(T - ep0);                               // plies of the unfinished episode
        len_sum += (uint32_t)(ep0 - ep0_in);                    // = the lengths of the episodes finished by running plies
    }
    int32_t *row = st.results ? st.results + b * (3 + P) : nullptr;    // packed result row for the gather
#pragma unroll
    for (int p = 0; p < P; ++p) {
        occ[p * B + b] = o[p];
        const uint32_t wc = st.win_count[p * B + b] + wins[p];
        st.win_count[p * B + b] = wc;
        if (row) row[3 + p] = (int32_t)wc;
    }
    winner[b] = (int8_t)w;
    to_move[b] = (int8_t)tm;
    st.tcount[b] = tc;
    st.tstep[b] = ts;
    const uint32_t ne = st.n_episodes[b] + n_ep, dr = st.draw_count[b] + draws, ls = st.len_sum[b] + len_sum;
    st.n_episodes[b] = ne;
    st.draw_count[b] = dr;
    st.len_sum[b] = ls;
    if (row) { row[0] = (int32_t)ne; row[1] = (int32_t)ls; row[2] = (int32_t)dr; }
}

// int8 [B][cells] boards of the 256 games of a workgroup from their occupancy masks in LDS: -1 empty, else the owner's
// id -- relative to s_rel[game] when that is >= 0 (_relative_player_id, tictactoe_2p_env.py:26-27: python's non-negative
// modulo by rel_mod).  One dword (four cells) per thread and trip, written coalesced; the bytes of a workgroup start at
// a multiple of 256 * cells, so dword stores are aligned whenever the buffer is.
template <int P>
__device__ __forceinline__ void ttt_write_boards(const uint32_t (&s_occ)[P][256], const int (&s_rel)[256], const int cells,
                                                 const uint32_t inv_cells, const int rel_mod, const int64_t g0, const int64_t B,
                                                 int8_t *__restrict__ obs)
{
    const int n_game = (int)((B - g0) < 256 ? (B - g0) : 256);
    const int total = n_game * cells;
    int8_t *out = obs + g0 * cells;
    for (int d = threadIdx.x; d * 4 < total; d += 256) {
        int e = cells == 1 ? d * 4 : (int)__umulhi((uint32_t)(d * 4), inv_cells);
        int c = d * 4 - e * cells;
        uint32_t word = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ee = e < 256 ? e : 255;       // bytes past the last game of the block are never stored
            int v = -1;
#pragma unroll
            for (int p = 0; p < P; ++p) v = ((s_occ[p][ee] >> c) & 1u) ? p : v;
            const int rel = s_rel[ee];
            if (v >= 0 && rel >= 0) {
                const int rr = (v - rel) % rel_mod;
                v = rr < 0 ? rr + rel_mod : rr;
            }
            word |= (uint32_t)(v & 0xff) << (8 * k);
            if (++c == cells) { c = 0; ++e; }
        }
        if (d * 4 + 4 <= total) {
            *reinterpret_cast<uint32_t *>(out + d * 4) = word;
        } else {
            for (int k = 0; d * 4 + k < total; ++k) out[d * 4 + k] = (int8_t)(word >> (8 * k));
        }
    }
}

// ---- fused per-step call: [sample ->] next_state (auto-reset) -> valid_actions mask + state_to_observation of the
// player to move next.  What a learner / vector env runs every ply (tictactoe_2p_env.py:240-315, :317-348, :382-407);
// as separate launches (crl_ttt_sample, crl_ttt_step, crl_ttt_valid, crl_ttt_board) the launch boundaries cost more than
// the work.  One lane per game plays the move; the post-move masks go through LDS so that the int8 [B][cells] observation
// is produced a dword per thread and written coalesced (a lane writing its own game's 9..27 bytes would not be).
template <int P, int ND>
__global__ void __launch_bounds__(256)
ttt_step_observe_kernel(const ttt_dirs dd, const uint32_t inv_cells, const int64_t B, const uint32_t seed_lo,
                        const uint32_t seed_hi, const uint64_t first_env_id, uint32_t *__restrict__ occ,
                        int8_t *__restrict__ winner, int8_t *__restrict__ to_move, const int8_t *__restrict__ action,
                        uint32_t *__restrict__ tcount, int8_t *__restrict__ reward, uint8_t *__restrict__ terminal,
                        int8_t *__restrict__ winners, int8_t *__restrict__ obs, uint32_t *__restrict__ valid,
                        const int rel_mod, const uint32_t flags)
{
    __shared__ uint32_t s_occ[P][256];
    __shared__ int s_mover[256];
    const int64_t g0 = (int64_t)blockIdx.x * 256;
    const int64_t b = g0 + threadIdx.x;
    uint32_t o[P];
    int tm = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) o[p] = 0u;
    if (b < B) {
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] = occ[p * B + b];
        int w = winner[b];
        tm = to_move[b];
        int act;
        if (action) {
            act = action[b];
        } else {                                    // the rollout's random agent at this game's step counter
            uint32_t all = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) all |= o[p];
            const uint32_t empty = dd.full & ~all, c = tcount[b];
            const int n_empty = __popc(empty);
            act = n_empty ? nth_set_bit(empty, (int)__umulhi(ttt_agent_word((uint32_t)(first_env_id + (uint64_t)b), c, n_empty, seed_lo, seed_hi), (uint32_t)n_empty)) : -1;
            tcount[b] = c + 1u;
        }
        int r, t, ws;
        ttt_step_core<P, ND>(dd, o, w, tm, act, r, t, ws);
        reward[b] = (int8_t)r;
        terminal[b] = (uint8_t)t;
        winners[b] = (int8_t)ws;
        if (t && (flags & CRL_STEP_AUTO_RESET)) {
#pragma unroll
            for (int p = 0; p < P; ++p) o[p] = 0;
            w = -1; tm = 0;
        }
        uint32_t all = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) { occ[p * B + b] = o[p]; all |= o[p]; }
        winner[b] = (int8_t)w;
        to_move[b] = (int8_t)tm;
        valid[b] = dd.full & ~all;                  // tictactoe_2p_env.py:317-348 for the player to move next
    }
#pragma unroll
    for (int p = 0; p < P; ++p) s_occ[p][threadIdx.x] = o[p];
    s_mover[threadIdx.x] = tm;
    __syncthreads();
    ttt_write_boards<P>(s_occ, s_mover, dd.n_cells, inv_cells, rel_mod, g0, B, obs);
}

// the rollout's random agent for one step
__global__ void __launch_bounds__(256)
ttt_sample_kernel(const int P, const uint32_t full, const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi,
                  const uint64_t first_env_id, const uint32_t *__restrict__ occ, uint32_t *__restrict__ tcount,
                  const int advance, int8_t *__restrict__ action)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t all = 0;
    for (int p = 0; p < P; ++p) all |= occ[p * B + b];
    const uint32_t empty = full & ~all, c = tcount[b];
    const int n_empty = __popc(empty);
    const uint32_t word = ttt_agent_word((uint32_t)(first_env_id + (uint64_t)b), c, n_empty, seed_lo, seed_hi);
    action[b] = (int8_t)(n_empty ? nth_set_bit(empty, (int)__umulhi(word, (uint32_t)n_empty)) : -1);
    if (advance) tcount[b] = c + 1u;
}

__global__ void __launch_bounds__(256)
ttt_reset_kernel(const int P, const int64_t B, const uint8_t *__restrict__ mask, uint32_t *__restrict__ occ,
                 int8_t *__restrict__ winner, int8_t *__restrict__ to_move)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mask && !mask[b]) return;
    for (int p = 0; p < P; ++p) occ[p * B + b] = 0;
    winner[b] = -1;
    to_move[b] = 0;
}

__global__ void __launch_bounds__(256)
ttt_valid_kernel(const int P, const uint32_t full, const int64_t B, const uint32_t *__restrict__ occ, uint32_t *__restrict__ valid)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t all = 0;
    for (int p = 0; p < P; ++p) all |= occ[p * B + b];
    valid[b] = full & ~all;
}

// board export / state_to_observation for 256 games per workgroup: masks into LDS (coalesced), then ttt_write_boards
template <int P>
__global__ void __launch_bounds__(256)
ttt_board_kernel(const int n_cells, const uint32_t inv_cells, const int64_t B, const uint32_t *__restrict__ occ,
                 const int8_t *__restrict__ player, const int rel_mod, int8_t *__restrict__ board)
{
    __shared__ uint32_t s_occ[P][256];
    __shared__ int s_rel[256];
    const int64_t g0 = (int64_t)blockIdx.x * 256;
    const int64_t b = g0 + threadIdx.x;
#pragma unroll
    for (int p = 0; p < P; ++p) s_occ[p][threadIdx.x] = b < B ? occ[p * B + b] : 0u;
    s_rel[threadIdx.x] = (player && b < B) ? (int)player[b] : -1;
    __syncthreads();
    ttt_write_boards<P>(s_occ, s_rel, n_cells, inv_cells, rel_mod, g0, B, board);
}

// ---- the same calls on states in the REFERENCE's own layout (tictactoe_2p_env.py:165-169): board int8 [B][cells], -1 =
// empty, else the owner's id.  For callers that hold reference states -- the single-state drop-in classes above all
// (one launch per next_state on host-mapped memory, crl_host_alloc) -- and never see the occupancy masks.  One lane per
// game: it gathers its cells into the per-player masks, plays the very ttt_step_core of crl_ttt_step, and writes back
// what changed.
template <int P>
__device__ __forceinline__ void ttt_masks_of_board(const int8_t *__restrict__ cells, const int n_cells, uint32_t (&o)[P])
{
#pragma unroll
    for (int p = 0; p < P; ++p) o[p] = 0u;
    for (int c = 0; c < n_cells; ++c) {
        const int v = cells[c];
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] |= (v == p) ? (1u << c) : 0u;
    }
}

// observation of one game: ids relative to `rel` (python's non-negative modulo by rel_mod), -1 stays -1
template <int P>
__device__ __forceinline__ void ttt_write_relative(const uint32_t (&o)[P], const int n_cells, const int rel, const int rel_mod,
                                                   int8_t *__restrict__ out)
{
    for (int c = 0; c < n_cells; ++c) {
        int v = -1;
#pragma unroll
        for (int p = 0; p < P; ++p) v = ((o[p] >> c) & 1u) ? p : v;
        if (v >= 0 && rel >= 0) {
            const int rr = (v - rel) % rel_mod;
            v = rr < 0 ? rr + rel_mod : rr;
        }
        out[c] = (int8_t)v;
    }
}

// one reference-layout state BY VALUE (kernel arguments): what the single-state call passes instead of letting the kernel
// fetch board, winner, mover and action over PCIe first (crl_ttt_step_board_host)
struct TttVec {
    uint32_t occ[CRL_TTT_MAX_P];                               // the board as the masks ttt_masks_of_board would build
    int32_t winner, to_move, action;
};

template <int P, int ND>
__global__ void __launch_bounds__(256)
ttt_step_board_kernel(const ttt_dirs dd, const int64_t B, int8_t *__restrict__ board, int8_t *__restrict__ winner,
                      int8_t *__restrict__ to_move, const int8_t *__restrict__ action, int8_t *__restrict__ reward,
                      uint8_t *__restrict__ terminal, int8_t *__restrict__ winners, uint32_t *__restrict__ valid,
                      int8_t *__restrict__ obs, const int rel_mod, const uint32_t flags,
                      const TttVec vec, const bool by_value, uint32_t *flag, const uint32_t seq)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int8_t *cells = board + b * dd.n_cells;
    uint32_t o[P];
    if (by_value) {
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] = vec.occ[p];
    } else {
        ttt_masks_of_board<P>(cells, dd.n_cells, o);
    }
    uint32_t before = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) before |= o[p];
    int w = by_value ? vec.winner : (int)winner[b], tm = by_value ? vec.to_move : (int)to_move[b], r, t, ws;
    const int pl = tm, act = by_value ? vec.action : (int)action[b];
    ttt_step_core<P, ND>(dd, o, w, tm, act, r, t, ws);
    reward[b] = (int8_t)r;
    terminal[b] = (uint8_t)t;
    winners[b] = (int8_t)ws;
    uint32_t all = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) all |= o[p];
    if (t && (flags & CRL_STEP_AUTO_RESET)) {
#pragma unroll
        for (int p = 0; p < P; ++p) o[p] = 0;
        all = 0; w = -1; tm = 0;
        for (int c = 0; c < dd.n_cells; ++c) cells[c] = (int8_t)-1;
    } else if (all != before) {
        cells[act] = (int8_t)pl;                   // the one cell the move filled (tictactoe_2p_env.py:295)
    }
    winner[b] = (int8_t)w;
    to_move[b] = (int8_t)tm;
    if (valid) valid[b] = dd.full & ~all;          // valid_actions of the player to move next (:317-348)
    if (obs) ttt_write_relative<P>(o, dd.n_cells, tm, rel_mod, obs + b * dd.n_cells);
    // (single-state call, B = 1: this thread wrote everything, so its release store publishes completion)
    if (flag && b == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// valid_actions (empties mask) and / or state_to_observation (ids relative to player[b]) of reference-layout boards
template <int P>
__global__ void __launch_bounds__(256)
ttt_observe_board_kernel(const int n_cells, const uint32_t full, const int64_t B, const int8_t *__restrict__ board,
                         const int8_t *__restrict__ player, const int rel_mod, int8_t *__restrict__ obs,
                         uint32_t *__restrict__ valid)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint32_t o[P];
    ttt_masks_of_board<P>(board + b * n_cells, n_cells, o);
    if (valid) {
        uint32_t all = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) all |= o[p];
        valid[b] = full & ~all;
    }
    if (obs) ttt_write_relative<P>(o, n_cells, player ? (int)player[b] : -1, rel_mod, obs + b * n_cells);
}

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// host: enumerate directions / lines exactly like the oracle does, independently written
int build_dirs(const crl_ttt_cfg &c, ttt_dirs &dd, uint32_t *lines, int &n_lines)
{
    memset(&dd, 0, sizeof(dd));
    dd.K = c.K; dd.P = c.P; dd.n_cells = c.n_cells; dd.full = c.full;
    n_lines = 0;
    for (int di = 0; di <= 1; ++di)
        for (int dj = -1; dj <= 1; ++dj)
            for (int dk = -1; dk <= 1; ++dk) {
                if (di == 0 && dj < 0) continue;
                if (di == 0 && dj == 0 && dk <= 0) continue;
                const int stride = (di * c.D1 + dj) * c.D2 + dk;
                uint32_t start = 0;
                for (int i = 0; i < c.D0; ++i)
                    for (int j = 0; j < c.D1; ++j)
                        for (int k = 0; k < c.D2; ++k) {
                            const int ei = i + di * (c.K - 1), ej = j + dj * (c.K - 1), ek = k + dk * (c.K - 1);
                            if (ei < 0 || ei >= c.D0 || ej < 0 || ej >= c.D1 || ek < 0 || ek >= c.D2) continue;
                            const int cell = (i * c.D1 + j) * c.D2 + k;
                            start |= 1u << cell;
                            uint32_t m = 0;
                            for (int s = 0; s < c.K; ++s) m |= 1u << (cell + s * stride);
                            if (n_lines >= CRL_TTT_MAX_LINES) return -1;
                            lines[n_lines++] = m;
                        }
                if (start) {
                    if (stride <= 0 || dd.n_dirs >= 13) return -2;
                    dd.stride[dd.n_dirs] = stride;
                    dd.start[dd.n_dirs] = start;
                    dd.n_dirs++;
                }
            }
    return 0;
}

inline const ttt_dirs &dirs_of(const crl_ctx *ctx) { return ctx->ttt_dd; }     // built once, in crl_ttt_create

} // namespace

#define TTT_DISPATCH_P(P_, CALL)           \
    switch (P_) {                          \
        case 1: { constexpr int PP = 1; CALL; } break; \
        case 2: { constexpr int PP = 2; CALL; } break; \
        case 3: { constexpr int PP = 3; CALL; } break; \
        case 4: { constexpr int PP = 4; CALL; } break; \
        case 5: { constexpr int PP = 5; CALL; } break; \
        case 6: { constexpr int PP = 6; CALL; } break; \
        case 7: { constexpr int PP = 7; CALL; } break; \
        case 8: { constexpr int PP = 8; CALL; } break; \
        default: crl_set_error("ttt: P=%d out of range 1..8", P_); return CRL_EINVAL; \
    }

#define TTT_CTX_CHECK(fn)                                                                  \
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_TTT, fn ": ctx is not a tictactoe context"); \
    CRL_REQUIRE(B > 0 && B <= ((int64_t)1 << 31), fn ": B=%lld out of range", (long long)B)

int crl_ttt_bounds(unsigned int *out4)
{
    CRL_BOUNDS_READBACK(out4);
    return CRL_OK;
}

extern "C" {

int crl_ttt_create(int D0, int D1, int D2, int K, int P, crl_ctx **out)
{
    CRL_REQUIRE(out != nullptr, "crl_ttt_create: out is NULL");
    CRL_REQUIRE(D0 >= 1 && D1 >= 1 && D2 >= 1 && (int64_t)D0 * D1 * D2 <= 32, "crl_ttt_create: board %dx%dx%d must have 1..32 cells", D0, D1, D2);
    CRL_REQUIRE(K >= 2 && K <= 32, "crl_ttt_create: K=%d out of range", K);
    CRL_REQUIRE(P >= 1 && P <= CRL_TTT_MAX_P, "crl_ttt_create: P=%d out of range 1..%d", P, CRL_TTT_MAX_P);
    crl_ctx *c = new crl_ctx();
    memset(c, 0, sizeof(*c));
    c->game = CRL_GAME_TTT;
    c->ttt.D0 = D0; c->ttt.D1 = D1; c->ttt.D2 = D2; c->ttt.K = K; c->ttt.P = P;
    c->ttt.n_cells = D0 * D1 * D2;
    c->ttt.full = c->ttt.n_cells == 32 ? 0xffffffffu : ((1u << c->ttt.n_cells) - 1u);
    ttt_dirs dd;
    int n_lines = 0;
    if (build_dirs(c->ttt, dd, c->ttt_lines_host, n_lines) != 0) {
        delete c;
        crl_set_error("crl_ttt_create: too many win lines for %dx%dx%d K=%d", D0, D1, D2, K);
        return CRL_EUNSUPPORTED;
    }
    c->ttt.n_lines = n_lines;
    c->ttt_dd = dd;
    // Boards of at most 16 cells (the reference's 3x3 and 3x5 among them): "does this mask hold a K-line" for all 65,536
    // masks as a bit table on the device, which crl_ttt_rollout stages into LDS -- one LDS read and four vector
    // instructions per ply instead of the four-direction shift-and test.  Optional: without it (no device at create
    // time, or a rollout on another device) the rollout computes the test as everywhere else.
    if (c->ttt.n_cells <= 16 && getenv("CRL_TTT_NO_WIN_TABLE") == nullptr) {     // (the switch: tests of the table-less path)
        std::vector<uint32_t> tab(2048, 0u);
        for (uint32_t m = 0; m < 65536u; ++m)
            for (int i = 0; i < n_lines; ++i)
                if ((m & c->ttt_lines_host[i]) == c->ttt_lines_host[i]) { tab[m >> 5] |= 1u << (m & 31u); break; }
        int dev = -1;
        void *d = nullptr;
        if (hipGetDevice(&dev) == hipSuccess && hipMalloc(&d, tab.size() * sizeof(uint32_t)) == hipSuccess) {
            if (hipMemcpy(d, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess) {
                c->ttt_win_dev = (uint32_t *)d;
                c->ttt_win_device = dev;
            } else {
                (void)hipFree(d);
            }
        }
        (void)hipGetLastError();                           // (a failure here is not an error of the call)
    }
    *out = c;
    return CRL_OK;
}

int crl_ttt_lines(const crl_ctx *ctx, uint32_t *lines, int cap)
{
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_TTT, "crl_ttt_lines: ctx is not a tictactoe context");
    if (lines)
        for (int i = 0; i < ctx->ttt.n_lines && i < cap; ++i) lines[i] = ctx->ttt_lines_host[i];
    return ctx->ttt.n_lines;
}

int crl_ttt_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask, uint32_t *occ, int8_t *winner, int8_t *to_move, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_reset");
    CRL_REQUIRE(occ && winner && to_move, "crl_ttt_reset: NULL state pointer");
    hipLaunchKernelGGL(ttt_reset_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, ctx->ttt.P, B, mask, occ, winner, to_move);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_step(const crl_ctx *ctx, int64_t B, uint32_t *occ, int8_t *winner, int8_t *to_move,
                 const int8_t *action, int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t flags, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_step");
    CRL_REQUIRE(occ && winner && to_move, "crl_ttt_step: NULL state pointer");
    CRL_REQUIRE(action && reward && terminal && winners, "crl_ttt_step: NULL action/output pointer");
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "crl_ttt_step: unknown flags 0x%x", flags);
    const ttt_dirs dd = dirs_of(ctx);
    TTT_DISPATCH_P(ctx->ttt.P, {
        if (dd.n_dirs <= 4)
            hipLaunchKernelGGL((ttt_step_kernel<PP, 4>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               occ, winner, to_move, action, reward, terminal, winners, flags);
        else
            hipLaunchKernelGGL((ttt_step_kernel<PP, 13>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               occ, winner, to_move, action, reward, terminal, winners, flags);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, uint32_t *valid, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_valid");
    CRL_REQUIRE(occ && valid, "crl_ttt_valid: NULL pointer");
    hipLaunchKernelGGL(ttt_valid_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, ctx->ttt.P, ctx->ttt.full, B, occ, valid);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_board(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const int8_t *player, int rel_mod, int8_t *board, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_board");
    CRL_REQUIRE(occ && board, "crl_ttt_board: NULL pointer");
    CRL_REQUIRE(player == nullptr || rel_mod >= 1, "crl_ttt_board: rel_mod must be >= 1 when player is given");
    CRL_REQUIRE((((uintptr_t)board) & 3) == 0, "crl_ttt_board: board must be 4-byte aligned");
    const int cells = ctx->ttt.n_cells;
    const uint32_t inv_cells = cells == 1 ? 0u : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)cells) + 1u;
    TTT_DISPATCH_P(ctx->ttt.P, {
        hipLaunchKernelGGL((ttt_board_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream,
                           cells, inv_cells, B, occ, player, rel_mod < 1 ? 1 : rel_mod, board);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, const uint32_t *occ,
                   uint32_t *tcount, int advance, int8_t *action, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_sample");
    CRL_REQUIRE(occ && tcount && action, "crl_ttt_sample: NULL pointer");
    const ttt_dirs dd = dirs_of(ctx);
    hipLaunchKernelGGL(ttt_sample_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, ctx->ttt.P, dd.full, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, occ, tcount, advance, action);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                         uint32_t *occ, int8_t *winner, int8_t *to_move, const int8_t *action, uint32_t *tcount,
                         int8_t *reward, uint8_t *terminal, int8_t *winners, int8_t *obs_board, uint32_t *valid,
                         int rel_mod, uint32_t flags, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_step_observe");
    CRL_REQUIRE(occ && winner && to_move, "crl_ttt_step_observe: NULL state pointer");
    CRL_REQUIRE(action || tcount, "crl_ttt_step_observe: action and tcount are both NULL (nothing to play)");
    CRL_REQUIRE(reward && terminal && winners && obs_board && valid, "crl_ttt_step_observe: NULL output pointer");
    CRL_REQUIRE(rel_mod >= 1, "crl_ttt_step_observe: rel_mod must be >= 1");
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "crl_ttt_step_observe: unknown flags 0x%x", flags);
    CRL_REQUIRE((((uintptr_t)obs_board) & 3) == 0, "crl_ttt_step_observe: obs_board must be 4-byte aligned");
    const ttt_dirs dd = dirs_of(ctx);
    const uint32_t inv_cells = dd.n_cells == 1 ? 0u : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)dd.n_cells) + 1u;
    TTT_DISPATCH_P(ctx->ttt.P, {
        if (dd.n_dirs <= 4)
            hipLaunchKernelGGL((ttt_step_observe_kernel<PP, 4>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd,
                               inv_cells, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, occ, winner, to_move, action,
                               tcount, reward, terminal, winners, obs_board, valid, rel_mod, flags);
        else
            hipLaunchKernelGGL((ttt_step_observe_kernel<PP, 13>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd,
                               inv_cells, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, occ, winner, to_move, action,
                               tcount, reward, terminal, winners, obs_board, valid, rel_mod, flags);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

static int ttt_step_board_launch(const char *fn, const crl_ctx *ctx, int64_t B, int8_t *board, int8_t *winner, int8_t *to_move,
                                 const int8_t *action, int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t *valid,
                                 int8_t *obs_board, int rel_mod, uint32_t flags, void *stream, const bool by_value, uint32_t *flag,
                                 const uint32_t seq)
{
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_TTT, "%s: ctx is not a tictactoe context", fn);
    CRL_REQUIRE(B > 0 && B <= ((int64_t)1 << 31), "%s: B=%lld out of range", fn, (long long)B);
    CRL_REQUIRE(board && winner && to_move, "%s: NULL state pointer", fn);
    CRL_REQUIRE(action && reward && terminal && winners, "%s: NULL action/output pointer", fn);
    CRL_REQUIRE(obs_board == nullptr || rel_mod >= 1, "%s: rel_mod must be >= 1 when obs_board is given", fn);
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "%s: unknown flags 0x%x", fn, flags);
    const ttt_dirs dd = dirs_of(ctx);
    const int rm = rel_mod < 1 ? 1 : rel_mod;
    TttVec vec = {};
    if (by_value) {                                             // host-visible state (crl_host_alloc): read here, passed as arguments
        for (int c = 0; c < dd.n_cells; ++c) {
            const int v = board[c];
            if (v >= 0 && v < ctx->ttt.P) vec.occ[v] |= 1u << c;
        }
        vec.winner = winner[0]; vec.to_move = to_move[0]; vec.action = action[0];
    }
    TTT_DISPATCH_P(ctx->ttt.P, {
        if (dd.n_dirs <= 4)
            hipLaunchKernelGGL((ttt_step_board_kernel<PP, 4>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               board, winner, to_move, action, reward, terminal, winners, valid, obs_board, rm, flags, vec, by_value, flag, seq);
        else
            hipLaunchKernelGGL((ttt_step_board_kernel<PP, 13>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               board, winner, to_move, action, reward, terminal, winners, valid, obs_board, rm, flags, vec, by_value, flag, seq);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_step_board(const crl_ctx *ctx, int64_t B, int8_t *board, int8_t *winner, int8_t *to_move, const int8_t *action,
                       int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t *valid, int8_t *obs_board, int rel_mod,
                       uint32_t flags, void *stream)
{
    return ttt_step_board_launch("crl_ttt_step_board", ctx, B, board, winner, to_move, action, reward, terminal, winners, valid,
                                 obs_board, rel_mod, flags, stream, false, nullptr, 0u);
}

int crl_ttt_step_board_host(const crl_ctx *ctx, int8_t *board, int8_t *winner, int8_t *to_move, const int8_t *action,
                            int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t *valid, int8_t *obs_board, int rel_mod,
                            uint32_t flags, void *stream, uint32_t *flag, uint32_t seq, double timeout_s)
{
    CRL_REQUIRE(flag != nullptr, "crl_ttt_step_board_host: flag is NULL");
    const int rc = ttt_step_board_launch("crl_ttt_step_board_host", ctx, 1, board, winner, to_move, action, reward, terminal, winners,
                                         valid, obs_board, rel_mod, flags, stream, true, flag, seq);
    if (rc != CRL_OK) return rc;
    return crl_spin_mapped(stream, flag, seq, timeout_s, "crl_ttt_step_board_host");
}

int crl_ttt_observe_board(const crl_ctx *ctx, int64_t B, const int8_t *board, const int8_t *player, int rel_mod,
                          int8_t *obs_board, uint32_t *valid, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_observe_board");
    CRL_REQUIRE(board != nullptr, "crl_ttt_observe_board: board is NULL");
    CRL_REQUIRE(obs_board || valid, "crl_ttt_observe_board: nothing to compute (obs_board and valid are NULL)");
    CRL_REQUIRE(player == nullptr || rel_mod >= 1, "crl_ttt_observe_board: rel_mod must be >= 1 when player is given");
    TTT_DISPATCH_P(ctx->ttt.P, {
        hipLaunchKernelGGL((ttt_observe_board_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream,
                           ctx->ttt.n_cells, ctx->ttt.full, B, board, player, rel_mod < 1 ? 1 : rel_mod, obs_board, valid);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_ttt_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                    uint32_t *occ, int8_t *winner, int8_t *to_move, crl_ttt_stats st, void *stream)
{
    TTT_CTX_CHECK("crl_ttt_rollout");
    CRL_REQUIRE(occ && winner && to_move, "crl_ttt_rollout: NULL state pointer");
    CRL_REQUIRE(st.tcount && st.tstep && st.n_episodes && st.win_count && st.draw_count && st.len_sum, "crl_ttt_rollout: NULL stats pointer");
    CRL_REQUIRE(T >= 0 && T <= (1 << 24), "crl_ttt_rollout: T=%d out of range", T);
    if (T == 0) return CRL_OK;
    const ttt_dirs dd = dirs_of(ctx);
    const uint32_t *win_tab = nullptr;                      // the <= 16-cell win table, when it lives on this device
    if (ctx->ttt_win_dev) {
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && dev == ctx->ttt_win_device) win_tab = ctx->ttt_win_dev;
    }
    TTT_DISPATCH_P(ctx->ttt.P, {
        if (dd.n_dirs <= 4 && dd.n_cells <= 16 && win_tab != nullptr)
            hipLaunchKernelGGL((ttt_rollout_kernel<PP, 4, true>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, occ, winner, to_move, st, win_tab);
        else if (dd.n_dirs <= 4)
            hipLaunchKernelGGL((ttt_rollout_kernel<PP, 4>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, occ, winner, to_move, st, win_tab);
        else
            hipLaunchKernelGGL((ttt_rollout_kernel<PP, 13>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, dd, B,
                               (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, occ, winner, to_move, st, win_tab);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

} // extern "C"
