// tron.hip -- batched Tron stepper for gfx950 (MI355X).  Hand-written HIP, wave64.
//
// Restates, for B independent games at once:
//   colosseumrl/envs/tron/CyTronGrid.pyx:3-62      next_state_inplace
//   colosseumrl/envs/tron/TronGridEnvironment.py:228-263  new_state
//   colosseumrl/envs/tron/TronGridEnvironment.py:309-323  rewards / terminal / winners
//   colosseumrl/envs/tron/CyTronGrid.pyx:65-71 + TronGridEnvironment.py:385-405  observation
//
// Mapping: one lane per game (the <=P-player loop of a game is inherently serial: the result of
// player i depends on what players j<i did in the same step); per-player state is
// [P][B] struct-of-arrays so a wave's 64 lanes load 64 consecutive elements; the board is int8
// [B][N*N].  A step issues its P byte probes together (the targets depend only on the pre-step
// heads) and patches same-step interactions in registers, so it costs one round of loads and one
// round of stores instead of P dependent round trips.
//
// Seven interchangeable rollout kernels (T fused steps, random agent, auto-reset; crl_tron_rollout picks one):
//   * lane per PLAYER, four lanes per game (at most 4 players; the defaults where they apply): the quad shares what is
//     per game by DPP (alive count, random stream, reset), the reference's sequential order is resolved only on the
//     ~1 % of wave-steps where players interact.  64 games per workgroup, 4 waves per SIMD:
//       - tron_rollout_quad_kernel, LDS byte slabs, boards up to 20x20;
//       - tron_rollout_pair_kernel, the same with TWO lanes per game (one or two players);
//       - tron_rollout_qbits_kernel, LDS bitboards + replay kernel, boards up to 40x40 (default for 21..40);
//       - tron_rollout_gquad_kernel, boards in global memory, no tags, no copies: no fixed cost per launch (launches too
//         short to earn the LDS kernels' copies back; boards above 40x40).
//   * lane per GAME (more than 4 players; or pinned):
//       - LDS byte slabs (boards up to 20x20: 256 games per workgroup; up to 40x40: 64): every wave copies its boards
//         into LDS once, plays all T steps there (wall-bordered slabs, heads as LDS addresses, episode-tagged cells
//         with a rolling one-row rewrite instead of board clears), and writes the boards back once;
//       - LDS bitboards + replay (above 20x20, T >= 256): occupancy only while playing, deaths as lane masks; the
//         unfinished episode is replayed on byte slabs at the end to recover owners.
//   * global-memory, lane per game (more than 4 players or long launches on boards above 44x44): boards stay in HBM /
//     Infinity Cache, same tagged cells, tags stripped in place at the end.
// plus crl_tron_ranking (compute_ranking, TronGridEnvironment.py:483-508), one wave per game.
#include "crl_common.hpp"
#include <hip/hip_ext.h>
// Observation boards are written once and not read again by the kernel that writes them; whether their 16-byte stores should
// be NONTEMPORAL depends on where they land (round-5 A/B on one box, tools/debug/stream_ab.py -> profiles/r5_stream_ab.txt;
// Tron 20x20 P4, GPU us per call, plain / nontemporal):
//   games                    65,536        131,072       262,144        524,288        1,048,576
//   crl_tron_step_observe    23.1 / 25.9   51.0 / 47.4   120.8 / 88.9   244.4 / 168.4  480-558 / 415-472
//   crl_tron_observe_all     21.7 / 20.6   43.7 / 38.9    98.1 / 79.0   191.9 / 149.8  385-391 / 397-448
// observe_all: nontemporal always.  step_observe: nontemporal once the call's boards in + out (131 MB at 65,536 games,
// where plain stores are absorbed by the 256 MiB Infinity Cache) exceed 192 MiB.  CRL_NT_STREAM=0 / 1 (A/B builds) forces
// one form everywhere.
typedef uint32_t crl_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void crl_stream_store16(void *ptr, const uint4 v, const bool nt)
{
    if (nt) __builtin_nontemporal_store(crl_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<crl_u32x4 *>(ptr));
    else *reinterpret_cast<uint4 *>(ptr) = v;
}
constexpr int64_t kStreamNtBytes = (int64_t)192 << 20;
inline bool crl_stream_nt(const int64_t bytes_per_call, const bool small_default)
{
#ifdef CRL_NT_STREAM
    return CRL_NT_STREAM != 0;
#else
    return bytes_per_call > kStreamNtBytes ? true : small_default;
#endif
}
#ifndef CRL_SO_HOIST
#define CRL_SO_HOIST 1                 // tron_step_observe_kernel: the stepping lanes' player vectors are loaded ahead of the board staging (A/B switch)
#endif
#ifndef CRL_STEP_DEFAULT_STAGED
#define CRL_STEP_DEFAULT_STAGED 0      // crl_tron_step's default kernel: 0 = byte probes, 1 = boards staged through LDS (A/B: profiles/r5_step_ab.json)
#endif
#include <type_traits>
#include <algorithm>

namespace {

struct TronGeom {
    int N, NN;
    uint32_t inv_n;    // floor(2^32 / N) + 1 : y = umulhi(h, inv_n) is exact for h < N*N <= 2^15
};

// ---- per-game step, everything in registers -------------------------------------------------
template <int P>
struct TronRegs {
    int h[P];   // head (flat)
    int x[P];   // head column
    int y[P];   // head row
    int d[P];   // dirs
    int k[P];   // deaths
};

template <int P>
__device__ __forceinline__ void tron_split_heads(const TronGeom &g, TronRegs<P> &s)
{
#pragma unroll
    for (int i = 0; i < P; ++i) {
        s.y[i] = (int)__umulhi((uint32_t)s.h[i], g.inv_n);
        s.x[i] = s.h[i] - s.y[i] * g.N;
    }
}

// Board views: how a step reads "who owns this cell" (0 = empty) and writes a trail cell.
// (the cell-indexed views below carry the board's cell count in the bounds-assert build: every probe / trail index of the
//  step, step_observe and int64 in-place kernels is checked against it -- codes 13x; a probe of a dead or out-of-board
//  player reads cell 0 by construction)
#ifdef CRL_BOUNDS
#define CRL_CELLS_MEMBER int cells;
#define CRL_CELLS_INIT(nn) , (nn)
#define CRL_CELLS_CHECK(c, code) CRL_BOUNDS_LT((c), cells, (code))
#else
#define CRL_CELLS_MEMBER
#define CRL_CELLS_INIT(nn)
#define CRL_CELLS_CHECK(c, code) do { } while (0)
#endif
struct PlainBoard {                 // canonical int8 cells, global memory
    int8_t *p;
    CRL_CELLS_MEMBER
    __device__ __forceinline__ int raw(const int c) const { CRL_CELLS_CHECK(c, 130); return p[c]; }
    __device__ __forceinline__ int owner(const int r) const { return r; }
    __device__ __forceinline__ void put(const int c, const int who) const { CRL_CELLS_CHECK(c, 131); p[c] = (int8_t)who; }
};
// LDS cells carry an episode tag: byte = tag << OB | owner.  A cell counts as occupied only when its
// tag equals the game's current tag, so "new_state" is tag+1 instead of clearing N*N bytes; a real clear
// happens once per 2^(8-OB) episodes.  Canonical HBM boards are tag 0, converted back on copy-out.
template <int OB>
struct TaggedBoard {
    uint8_t *p;
    uint32_t tagbits;               // tag << OB
    __device__ __forceinline__ int raw(const int c) const { return p[c]; }
    __device__ __forceinline__ int owner(const int r) const
    {
        const uint32_t x = (uint32_t)r ^ tagbits;       // same tag: the tag bits cancel and x IS the owner
        return x < (1u << OB) ? (int)x : 0;
    }
    __device__ __forceinline__ void put(const int c, const int who) const { p[c] = (uint8_t)(tagbits | (uint32_t)who); }
};

// phase 1 of a step: every player's target cell and the raw probe of it (CyTronGrid.pyx:21-41).
// All P probes are issued back to back; probes of dead / out-of-board players read cell 0 and are ignored.
template <int P>
struct TronProbe {
    int tgt[P], raw[P], ndir[P], nx[P], ny[P];
    bool oob[P];
};

// EXACT (the int64 entries with the Cython signature only): the direction as CyTronGrid.pyx:32 computes it on C longs,
// (directions[i] + action + 4) % 4 with C's remainder (cdivision=True) -- for the values TronGridEnvironment ever passes
// (directions 0..3, actions -1 / 0 / +1) that is the `& 3` of the batched entries; for anything that sums below -4 the
// remainder is NEGATIVE, none of :35-42's four branches fires, the player "moves" onto the cell it stands on (and dies by
// its own id there, :51-57) and the negative direction is stored (:44).  Reproduced bit for bit (tests/golden/tron_wild64.npz).
template <int P, typename BOARD, bool EXACT = false>
__device__ __forceinline__ void tron_probe(const TronGeom &g, const BOARD &bd, const TronRegs<P> &s,
                                           const int (&act)[P], TronProbe<P> &pr)
{
    const int N = g.N;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int exact = EXACT ? (s.d[i] + act[i] + 4) % 4 : 0;
        const bool stay = EXACT && exact < 0;
        const int dir = EXACT ? exact : (s.d[i] + act[i]) & 3;  // act in {0, 1, 3}: forward, right, left (= -1 mod 4)
        // unit step of direction dir from two 4-entry byte tables (one v_bfe_i32 each): dx = 0,+1,0,-1  dy = -1,0,+1,0
        const int sh8 = (dir & 3) << 3;
        pr.nx[i] = s.x[i] + (stay ? 0 : __builtin_amdgcn_sbfe((int)0xff000100u, sh8, 8));
        pr.ny[i] = s.y[i] + (stay ? 0 : __builtin_amdgcn_sbfe((int)0x000100ffu, sh8, 8));
        pr.ndir[i] = dir;
        pr.oob[i] = ((unsigned)pr.nx[i] >= (unsigned)N) | ((unsigned)pr.ny[i] >= (unsigned)N);
        // in-board coordinates are < 2^15: the 24-bit multiply-add is a full-rate VALU op (a 32-bit one is not)
        pr.tgt[i] = pr.oob[i] ? 0 : (int)(__umul24((unsigned)pr.ny[i], (unsigned)N) + (unsigned)pr.nx[i]);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) pr.raw[i] = bd.raw(pr.tgt[i]);
}

// the same for a wall-bordered slab: target = head + one of four byte steps, no bounds test (the probe answers it)
template <int P, typename BOARD>
__device__ __forceinline__ void tron_probe_padded(const uint32_t delta4, const BOARD &bd, const TronRegs<P> &s,
                                                  const int (&act)[P], TronProbe<P> &pr)
{
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int dir = (s.d[i] + act[i]) & 3;
        pr.ndir[i] = dir;
        pr.tgt[i] = s.h[i] + __builtin_amdgcn_sbfe((int)delta4, dir << 3, 8);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) pr.raw[i] = bd.raw(pr.tgt[i]);
}

// reward / terminal / winners tail of a step (TronGridEnvironment.py:309-321)
template <int P>
__device__ __forceinline__ void tron_outcome(const TronRegs<P> &s, int (&rew)[P], int &term, int &wmask)
{
    int alive = 0;
    wmask = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        alive += (s.k[i] == 0);
        wmask |= (s.k[i] == 0) << i;
    }
    term = alive <= 1;
    wmask = term ? wmask : 0;
#pragma unroll
    for (int i = 0; i < P; ++i) rew[i] = (s.k[i] > 0) ? -1 : (term ? 10 : 1);
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16 bytes of global memory at a dword boundary

// phases 2+3: the reference's sequential resolution on registers (CyTronGrid.pyx:15-62), trail writes, and the
// reward / terminal / winners tail (TronGridEnvironment.py:309-321).  Straight-line code: every decision is a
// select, so a wave never diverges inside a step.
template <int P, typename BOARD>
__device__ __forceinline__ void tron_resolve(const BOARD &bd, const bool valid, TronRegs<P> &s, const TronProbe<P> &pr,
                                             int (&rew)[P], int &term, int &wmask)
{
    bool moved[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool run = s.k[i] == 0;                   // :16 (may have been killed head-on by j < i)
        int v = bd.owner(pr.raw[i]);
#pragma unroll
        for (int j = 0; j < i; ++j)                     // a lower id that moved into the same cell this step
            v = (moved[j] & (pr.tgt[j] == pr.tgt[i])) ? j + 1 : v;
        const bool wall = run & pr.oob[i];              // :47-48
        const bool crash = run & !pr.oob[i] & (v > 0);  // :51-57
        moved[i] = run & !pr.oob[i] & (v <= 0);         // :60-62
        s.d[i] = run ? pr.ndir[i] : s.d[i];             // :44 direction is committed even if the move dies
        s.k[i] = wall ? i + 1 : (crash ? v : s.k[i]);
        // :56-57 the owner's head is this very cell -> the owner dies too (a head is never its owner's own target,
        // so q == i cannot hit).  One head select + one compare instead of P compares against every head.
        int hv = -1;
#pragma unroll
        for (int q = 0; q < P; ++q) hv = (q != i && v == q + 1) ? s.h[q] : hv;
        const bool hit = crash & (hv == pr.tgt[i]);
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (q != i) s.k[q] = (hit & (v == q + 1)) ? i + 1 : s.k[q];
        s.h[i] = moved[i] ? pr.tgt[i] : s.h[i];
        s.x[i] = moved[i] ? pr.nx[i] : s.x[i];
        s.y[i] = moved[i] ? pr.ny[i] : s.y[i];
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (valid & moved[i]) bd.put(pr.tgt[i], i + 1);
    tron_outcome<P>(s, rew, term, wmask);
}

// The LDS rollout kernel's resolve.  Its board is a slab with a wall border (cells == kWallCell), heads are LDS
// addresses, and it relies on the invariant every state produced by new_state / next_state has: the cell under a
// player's head holds that player's id (nobody ever overwrites an occupied cell).  Hence
//   * "the cell I move into is q's head" already says the cell is occupied, by q: one compare per (i, q) answers
//     both the :56-57 head-on test and, for q < i, the "q moved there earlier in this very step" patch;
//   * a wall kills exactly like the player's own trail would (deaths[i] = i + 1, nobody else involved), so a wall
//     cell is treated as owned by the mover and :47-48 folds into :51-57;
//   * a player that does not move stores to the slab's junk byte, which keeps the trail store unconditional;
//     `stamp[i]` is the byte to store (tag | i + 1), refreshed by the caller when the tag changes.
constexpr int kWallCell = 0xff;

template <int P, typename BOARD>
__device__ __forceinline__ void tron_resolve_lds(const BOARD &bd, TronRegs<P> &s, const TronProbe<P> &pr,
                                                 const uint32_t (&stamp)[P], const int junk, const bool active = true)
{
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const bool run = active & (s.k[i] == 0);        // :16 (may have been killed head-on by j < i)
        bool on_head[P];
#pragma unroll
        for (int q = 0; q < P; ++q) on_head[q] = (q != i) && (pr.tgt[i] == s.h[q]);   // h[q] already moved for q < i
        int v = bd.owner(pr.raw[i]);
        v = (pr.raw[i] == kWallCell) ? i + 1 : v;
#pragma unroll
        for (int j = 0; j < i; ++j) v = on_head[j] ? j + 1 : v;
        const bool crash = run & (v > 0);               // :47-48, :51-57
        const bool moved = run & !(v > 0);              // :60-62
        s.d[i] = run ? pr.ndir[i] : s.d[i];             // :44 direction is committed even if the move dies
        s.k[i] = crash ? v : s.k[i];
#pragma unroll
        for (int q = 0; q < P; ++q)
            if (q != i) s.k[q] = (run & on_head[q]) ? i + 1 : s.k[q];                 // :56-57
        s.h[i] = moved ? pr.tgt[i] : s.h[i];
#ifdef CRL_BOUNDS
        CRL_BOUNDS_IN(moved ? pr.tgt[i] : junk, bd.lo, bd.hi, 123);
#endif
        *(lds_u8 *)(uintptr_t)(uint32_t)(moved ? pr.tgt[i] : junk) = (uint8_t)stamp[i];
    }
}

template <int P, typename BOARD, bool EXACT = false>
__device__ __forceinline__ void tron_step_core(const TronGeom &g, const BOARD &bd, const bool valid,
                                               TronRegs<P> &s, const int (&act)[P],
                                               int (&rew)[P], int &term, int &wmask)
{
    TronProbe<P> pr;
    tron_probe<P, BOARD, EXACT>(g, bd, s, act, pr);
    tron_resolve<P>(bd, valid, s, pr, rew, term, wmask);
}

// uniform random actions for step c of global env g (contract: include/colosseum_hip.h, crl_tron_rollout).
// One Philox call = 4 words = 8 steps x 4 players: each word is a base-3 fraction giving 8 digits.
// Scalars (no arrays) so the words stay in VGPRs; "current word" is w0, rotated every second step.
struct TronRng4 {
    uint32_t w0, w1, w2, w3;
    __device__ __forceinline__ void refill(const uint32_t g, const uint32_t block, const uint32_t q,
                                           const uint32_t k0, const uint32_t k1)
    {
        const philox_out r = philox4x32_10(g, block, q, CRL_TAG_TRON, k0, k1);
        w0 = r.w[0]; w1 = r.w[1]; w2 = r.w[2]; w3 = r.w[3];
    }
    __device__ __forceinline__ void rotate() { w0 = w1; w1 = w2; w2 = w3; }
    // position the stream at step c (kernel entry): drop the words of the (c & 7) >> 1 finished step pairs
    __device__ __forceinline__ void seek(const uint32_t c)
    {
        const uint32_t pairs = (c & 7u) >> 1;
        if (pairs >= 1) rotate();
        if (pairs >= 2) rotate();
        if (pairs >= 3) rotate();
    }
};

template <int P>
struct TronRng {
    TronRng4 lo, hi;   // hi serves players 4..7 (unused when P <= 4)
    __device__ __forceinline__ void start(const uint32_t g, const uint32_t c, const uint32_t k0, const uint32_t k1)
    {
        lo.refill(g, c >> 3, 0u, k0, k1);
        lo.seek(c);
        if (P > 4) { hi.refill(g, c >> 3, 1u, k0, k1); hi.seek(c); }
    }
    // actions of step c, then advance to step c+1
    __device__ __forceinline__ void next(const uint32_t g, const uint32_t c, const uint32_t k0, const uint32_t k1, int (&act)[P])
    {
        const bool odd = c & 1u;
        uint32_t v = lo.w0 * (odd ? 81u : 1u);          // 3^4: odd steps use digits 4..7 of the word
        uint32_t u = (P > 4) ? hi.w0 * (odd ? 81u : 1u) : 0u;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            uint32_t &x = (i < 4) ? v : u;
            const uint32_t a3 = __umulhi(x, 3u);        // next base-3 digit of the fraction x / 2^32
            x *= 3u;
            act[i] = (int)(a3 + (a3 >> 1));             // 0, 1, 3 == forward, right, left (-1 mod 4)
        }
        advance(g, c, k0, k1);
    }
    // (Tried in round 2, tools/README.md: select-based rotation and an out-of-line `unlikely` refill, to save
    // the lone wave its taken branches -- +0.9 % on the byte kernel, -1 % on the bitboard kernel: placement noise.)
    __device__ __forceinline__ void advance(const uint32_t g, const uint32_t c, const uint32_t k0, const uint32_t k1)
    {
        if (c & 1u) {
            if ((c & 7u) == 7u) {
                lo.refill(g, (c + 1u) >> 3, 0u, k0, k1);
                if (P > 4) hi.refill(g, (c + 1u) >> 3, 1u, k0, k1);
            } else {
                lo.rotate();
                if (P > 4) hi.rotate();
            }
        }
    }
    // the same draw through a table: the four base-3 digits a step takes from a word are the base-3 digits of
    // t = umulhi(v, 81) (floor(81 v / 2^32) = 27 a0 + 9 a1 + 3 a2 + a3), and lut[t] holds the four action codes
    // (0, 1, 3) as 2-bit fields -- one multiply, one LDS byte and one bit-field extract per player.
    __device__ __forceinline__ void next_lut(const uint32_t g, const uint32_t c, const uint32_t k0, const uint32_t k1,
                                             const uint8_t *lut, int (&act)[P])
    {
        const bool odd = c & 1u;
        const uint32_t code_lo = lut[__umulhi(lo.w0 * (odd ? 81u : 1u), 81u)];
        const uint32_t code_hi = (P > 4) ? lut[__umulhi(hi.w0 * (odd ? 81u : 1u), 81u)] : 0u;
#pragma unroll
        for (int i = 0; i < P; ++i) act[i] = (int)((((i < 4) ? code_lo : code_hi) >> (2 * (i & 3))) & 3u);
        advance(g, c, k0, k1);
    }
};

constexpr int kTronLutSize = 81;
// fills the action table of next_lut (one entry per thread; the caller synchronises the workgroup afterwards)
__device__ __forceinline__ void tron_fill_action_lut(uint8_t *lut)
{
    for (uint32_t t = threadIdx.x; t < (uint32_t)kTronLutSize; t += blockDim.x) {
        const uint32_t a[4] = {t / 27u, (t / 9u) % 3u, (t / 3u) % 3u, t % 3u};
        uint32_t code = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) code |= (a[i] + (a[i] >> 1)) << (2 * i);
        lut[t] = (uint8_t)code;
    }
}

// 16 bytes of a freshly reset board starting at byte offset `off` (heads stamped)
template <int P>
__device__ __forceinline__ uint4 tron_fresh_chunk16(const crl_tron_cfg &cfg, const int off)
{
    uint32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int r = (int)cfg.start_heads[p] - off;
        if (r >= 0 && r < 16) {
            const uint32_t byte = (uint32_t)(p + 1) << ((r & 3) * 8);
            const int w = r >> 2;
            v0 |= (w == 0) ? byte : 0u;
            v1 |= (w == 1) ? byte : 0u;
            v2 |= (w == 2) ? byte : 0u;
            v3 |= (w == 3) ? byte : 0u;
        }
    }
    return make_uint4(v0, v1, v2, v3);
}

template <int P>
__device__ __forceinline__ void tron_regs_to_start(const crl_tron_cfg &cfg, const TronGeom &g, TronRegs<P> &s)
{
#pragma unroll
    for (int p = 0; p < P; ++p) { s.h[p] = cfg.start_heads[p]; s.d[p] = cfg.start_dirs[p]; s.k[p] = 0; }
    tron_split_heads<P>(g, s);
}

// ---- kernels ---------------------------------------------------------------------------------

// board part of new_state: one thread per 16-byte chunk (or per byte when N*N % 16 != 0)
template <int P, bool WIDE>
__global__ void __launch_bounds__(256)
tron_reset_board_kernel(const crl_tron_cfg cfg, const int64_t B, const uint8_t *__restrict__ mask,
                        int8_t *__restrict__ board)
{
    const int NN = cfg.N * cfg.N;
    const int per_env = WIDE ? NN / 16 : NN;
    const int64_t total = B * (int64_t)per_env;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per_env;
        const int c = (int)(i - b * per_env);
        if (mask && !mask[b]) continue;
        if (WIDE) {
            *reinterpret_cast<uint4 *>(board + b * NN + c * 16) = tron_fresh_chunk16<P>(cfg, c * 16);
        } else {
            int8_t v = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) v = (cfg.start_heads[p] == c) ? (int8_t)(p + 1) : v;
            board[b * NN + c] = v;
        }
    }
}

// board part of new_state for boards that are NOT whole 16-byte chunks (19 x 19, the reference's default): the boards of the
// batch are one byte stream, written 16 bytes at a time; a chunk that straddles two games takes the fresh cells of both
// (tron_fresh_flat16, below), and under a mask its two halves are stored apart when only one of the games is reset.
// The stream's last odd bytes (B * N * N % 16) go one by one.  (Round 5: 36 -> 10 us at 65,536 games of 19 x 19.)
template <int P>
__device__ __forceinline__ uint4 tron_fresh_flat16(const crl_tron_cfg &cfg, const int r, const int NN);

template <int P>
__global__ void __launch_bounds__(256)
tron_reset_board_flat_kernel(const crl_tron_cfg cfg, const int64_t B, const uint8_t *__restrict__ mask, int8_t *__restrict__ board,
                             const uint64_t inv_nn64)
{
    const int NN = cfg.N * cfg.N;
    const int64_t total = B * (int64_t)NN, chunks = total >> 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < chunks; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t byte0 = i << 4;
        const int64_t e0 = (int64_t)__umul64hi((uint64_t)byte0, inv_nn64);      // byte0 / NN (exact: byte0 * NN < 2^64)
        const int r = (int)(byte0 - e0 * NN);
        const int first = NN - r;                               // bytes of the chunk that belong to game e0 (>= 16: all)
        const bool m0 = !mask || mask[e0] != 0;
        const bool m1 = first >= 16 ? m0 : (!mask || mask[e0 + 1] != 0);
        if (!(m0 | m1)) continue;
        const uint4 fr = tron_fresh_flat16<P>(cfg, r, NN);
        if (m0 & m1) {
            *reinterpret_cast<uint4 *>(board + byte0) = fr;
        } else {                                                // one of the two games only: its bytes one by one
            const uint32_t w[4] = {fr.x, fr.y, fr.z, fr.w};
            for (int j = m0 ? 0 : first; j < (m0 ? first : 16); ++j) board[byte0 + j] = (int8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
        }
    }
    for (int64_t i = (chunks << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = (int64_t)__umul64hi((uint64_t)i, inv_nn64);
        if (mask && !mask[e]) continue;
        const int c = (int)(i - e * NN);
        int8_t v = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) v = (cfg.start_heads[p] == c) ? (int8_t)(p + 1) : v;
        board[i] = v;
    }
}

template <int P>
__global__ void __launch_bounds__(256)
tron_reset_players_kernel(const crl_tron_cfg cfg, const int64_t B, const uint8_t *__restrict__ mask,
                          int16_t *__restrict__ heads, int8_t *__restrict__ dirs, int8_t *__restrict__ deaths)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (mask && !mask[b]) return;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        heads[p * B + b] = cfg.start_heads[p];
        dirs[p * B + b] = cfg.start_dirs[p];
        deaths[p * B + b] = 0;
    }
}

template <int P>
__global__ void __launch_bounds__(256)
tron_step_kernel(const crl_tron_cfg cfg, const TronGeom g, const int64_t B,
                 int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                 int8_t *__restrict__ deaths, const int8_t *__restrict__ actions,
                 int8_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                 const uint32_t flags)
{
    const int NN = g.NN;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    const int64_t bb = valid ? b : 0;
    TronRegs<P> s;
    int act[P], rew[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s.h[p] = valid ? heads[p * B + bb] : 0;
        s.d[p] = valid ? dirs[p * B + bb] : 0;
        s.k[p] = valid ? deaths[p * B + bb] : 1;
        act[p] = valid ? actions[p * B + bb] : 0;
    }
    tron_split_heads<P>(g, s);
    int term, wm;
    const PlainBoard bd{board + bb * NN CRL_CELLS_INIT(NN)};
    tron_step_core<P>(g, bd, valid, s, act, rew, term, wm);
    if (valid) {
#pragma unroll
        for (int p = 0; p < P; ++p) rewards[p * B + b] = (int8_t)rew[p];
        terminal[b] = (uint8_t)term;
        winners[b] = (uint8_t)wm;
    }
    if (valid && term && (flags & CRL_STEP_AUTO_RESET)) {
        // new_state for the games that just ended: every such lane clears its OWN board (all ending games of the
        // wave in parallel, 16 bytes per store) and stamps the heads afterwards (same lane, same addresses: ordered)
        int8_t *own = board + b * NN;
        if ((NN & 15) == 0)
            for (int off = 0; off < NN; off += 16) *reinterpret_cast<uint4 *>(own + off) = make_uint4(0, 0, 0, 0);
        else
            for (int off = 0; off < NN; ++off) own[off] = 0;
        tron_regs_to_start<P>(cfg, g, s);
#pragma unroll
        for (int p = 0; p < P; ++p) own[s.h[p]] = (int8_t)(p + 1);
    }
    if (valid) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            heads[p * B + b] = (int16_t)s.h[p];
            dirs[p * B + b] = (int8_t)s.d[p];
            deaths[p * B + b] = (int8_t)s.k[p];
        }
    }
}

// u16 entries per game of the 16-bit gather row (crl_tron_stats.packed): n_episodes, len_sum, last_winners, tstep,
// ret_sum[P], rounded up to a whole number of dwords
template <int P> constexpr int kTronPackedRow = (4 + P + 1) & ~1;

// per-lane rollout bookkeeping shared by the rollout kernels.  The running totals a launch adds to are read at kernel
// entry (with every other global load of the prologue), not in the epilogue: a short launch otherwise ends on a chain
// of exposed load -> add -> store round trips.
template <int P>
struct TronAcc {
    int ret[P];
    uint32_t wins[P];
    uint32_t tc, ts, n_ep, len_sum;
    int last_w, last_len;
    int old_ret[P];
    uint32_t old_wins[P], old_n_ep, old_len_sum, old_last_w;
    __device__ __forceinline__ void load(const crl_tron_stats &st, const bool valid, const int64_t b, const int64_t B)
    {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            ret[p] = 0; wins[p] = 0;
            old_ret[p] = valid ? st.ret_sum[p * B + b] : 0;
            old_wins[p] = valid ? st.win_count[p * B + b] : 0u;
        }
        tc = valid ? st.tcount[b] : 0;
        ts = valid ? st.tstep[b] : 0;
        old_n_ep = valid ? st.n_episodes[b] : 0u;
        old_len_sum = valid ? st.len_sum[b] : 0u;
        old_last_w = valid ? st.last_winners[b] : 0u;
        n_ep = 0; len_sum = 0; last_w = -1; last_len = 0;
    }
    __device__ __forceinline__ void finish_episode(const int wm)
    {
        n_ep += 1;
        len_sum += ts;
        last_len = (int)ts;
        last_w = wm;
#pragma unroll
        for (int p = 0; p < P; ++p) wins[p] += (wm >> p) & 1;
        ts = 0;
    }
    // adds this launch's share to the running per-game totals and, when the caller asked for it, also writes the
    // packed result row [n_episodes, len_sum, last_winners, win_count[P], ret_sum[P]] the end-of-rollout gather ships
    __device__ __forceinline__ void store(const crl_tron_stats &st, const int64_t B, const int64_t b) const
    {
        int32_t *row = st.results ? st.results + b * (3 + 2 * P) : nullptr;
        uint16_t *pk = st.packed ? st.packed + b * kTronPackedRow<P> : nullptr;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int r = old_ret[p] + ret[p];
            const uint32_t w = old_wins[p] + wins[p];
            st.ret_sum[p * B + b] = r;
            st.win_count[p * B + b] = w;
            if (row) { row[3 + p] = (int32_t)w; row[3 + P + p] = r; }
            if (pk) pk[4 + p] = (uint16_t)r;
        }
        st.tcount[b] = tc;
        st.tstep[b] = ts;
        const uint32_t ne = old_n_ep + n_ep, ls = old_len_sum + len_sum;
        st.n_episodes[b] = ne;
        st.len_sum[b] = ls;
        if (last_w >= 0) {
            st.last_winners[b] = (uint8_t)last_w;
            st.last_len[b] = (uint16_t)last_len;
        }
        if (row) {
            row[0] = (int32_t)ne;
            row[1] = (int32_t)ls;
            row[2] = last_w >= 0 ? last_w : (int32_t)old_last_w;
        }
        if (pk) {
            pk[0] = (uint16_t)ne;
            pk[1] = (uint16_t)ls;
            pk[2] = (uint16_t)(last_w >= 0 ? (uint32_t)last_w : old_last_w);
            pk[3] = (uint16_t)ts;
        }
    }
};

// T fused steps, boards in global memory (any board size; L2 / Infinity-Cache resident at the benchmark sizes).
// Same loop as the LDS kernel below: cells written during the launch carry an episode tag, so a reset is tag+1
// plus P head stamps (no N*N-byte clear), the next step's actions are drawn while the probes are in flight,
// and one wave-cooperative pass at the end strips the tags from the wave's 64 boards (16-byte loads / stores).
template <int P>
__global__ void __launch_bounds__(256)
tron_rollout_kernel(const crl_tron_cfg cfg, const TronGeom g, const int64_t B, const uint32_t seed_lo,
                    const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                    int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                    int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    const int NN = g.NN;
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    const int64_t bb = valid ? b : 0;
    const int64_t env0 = b - lane;
    const int n_env = (int)((B - env0) < CRL_WAVE ? (B - env0 > 0 ? B - env0 : 0) : CRL_WAVE);
    TronRegs<P> s, fresh;
    int act[P], rew[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        s.h[p] = valid ? heads[p * B + bb] : 0;
        s.d[p] = valid ? dirs[p * B + bb] : 0;
        s.k[p] = valid ? deaths[p * B + bb] : 1;
    }
    tron_split_heads<P>(g, s);
    tron_regs_to_start<P>(cfg, g, fresh);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        asm volatile("" : "+v"(fresh.h[p]), "+v"(fresh.x[p]), "+v"(fresh.y[p]), "+v"(fresh.d[p]));
    }
    TronAcc<P> acc;
    acc.load(st, valid, bb, B);
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);
    TronRng<P> rng;
    rng.start(gid, acc.tc, seed_lo, seed_hi);
    constexpr int OB = (P <= 7) ? 3 : 4;
    constexpr uint32_t kTags = 1u << (8 - OB);
    TaggedBoard<OB> bd{reinterpret_cast<uint8_t *>(board + bb * NN), 0u};
    __shared__ uint8_t act_lut[84];
    tron_fill_action_lut(act_lut);
    __syncthreads();
    rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);
    for (int t = 0; t < T; ++t) {
        TronProbe<P> pr;
        tron_probe<P>(g, bd, s, act, pr);                       // P byte loads in flight ...
        acc.tc += 1;
        rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);   // ... while the NEXT step's actions are drawn
        int term, wm;
        tron_resolve<P>(bd, valid, s, pr, rew, term, wm);
        acc.ts += 1;
#pragma unroll
        for (int p = 0; p < P; ++p) acc.ret[p] += rew[p];
        if (valid && term) {
            uint32_t tag = (bd.tagbits >> OB) + 1u;
            if (tag == kTags) {                                 // tag space exhausted: one real clear of this board
                tag = 0;
                if ((NN & 15) == 0)
                    for (int off = 0; off < NN; off += 16) *reinterpret_cast<uint4 *>(bd.p + off) = make_uint4(0, 0, 0, 0);
                else
                    for (int off = 0; off < NN; ++off) bd.p[off] = 0;
            }
            bd.tagbits = tag << OB;
#pragma unroll
            for (int p = 0; p < P; ++p) bd.put(fresh.h[p], p + 1);
            acc.finish_episode(wm);
            s = fresh;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- strip the tags in place: cells of older episodes become 0, current ones keep their owner
    constexpr uint32_t OM = 0x01010101u * ((1u << OB) - 1u);
    constexpr uint32_t TM = 0x01010101u * (kTags - 1u);
    uint8_t *gslab = reinterpret_cast<uint8_t *>(board + env0 * NN);
    if ((NN & 15) == 0) {
        const int bytes = n_env * NN;
        for (int base = 0; base < bytes; base += CRL_WAVE * 16) {       // uniform trip count: the shuffle below reads
            const int off = base + lane * 16;                            // lanes that have no piece of their own left
            const int e = (off < bytes ? off : 0) / NN;
            const uint32_t trep = (uint32_t)__shfl((int)(bd.tagbits >> OB), e, CRL_WAVE) * 0x01010101u;
            if (off >= bytes) continue;
            const uint4 raw = *reinterpret_cast<const uint4 *>(gslab + off);
            uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t diff = ((w[q] >> OB) & TM) ^ trep;
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                w[q] = w[q] & OM & ~(stale * 0xffu);
            }
            *reinterpret_cast<uint4 *>(gslab + off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int e = 0; e < n_env; ++e) {
            const uint32_t tb = (uint32_t)__shfl((int)bd.tagbits, e, CRL_WAVE);
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const uint32_t raw = gslab[(int64_t)e * NN + c];
                gslab[(int64_t)e * NN + c] = (uint8_t)((((raw ^ tb) >> OB) == 0) ? (raw & ((1u << OB) - 1u)) : 0u);
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            heads[p * B + b] = (int16_t)s.h[p];
            dirs[p * B + b] = (int8_t)s.d[p];
            deaths[p * B + b] = (int8_t)s.k[p];
        }
        acc.store(st, B, b);
    }
}

// T fused steps, boards resident in LDS.  256 threads = 4 independent waves; every lane owns one board slab.
//
// Slab layout (TronPad): (N+2) rows of RS bytes (24 for boards up to 20x20, 44 up to 40x40).  Row 0 and row N+1 are wall, and
// so are the bytes x >= N of every row -- which makes the byte before a row's first cell wall as well.  Cell (x, y)
// sits at (y+1)*RS + x; heads are kept as LDS ADDRESSES (slab base + cell), so a probe is head + step[dir] and one
// ds_read_u8, leaving the board is "the probe read a wall", and no x / y / bounds arithmetic exists in the loop.
// After the rows comes one junk byte that players who do not move write to, which keeps the trail store
// unconditional.  The slab is a whole, ODD number of dwords: the 32 lanes of a half-wave that probe the same cell
// of their boards (every game right after a reset!) hit 32 different LDS banks instead of 8.
//
// Cells carry an episode tag (TaggedBoard).  Random agents finish an episode every ~10 steps, so a reset must be
// cheap and must never clear a board in one go: it bumps the tag, stamps the heads and rewrites `sweep_rows` rows
// (6 dword stores each, cells = 0, walls = 0xff) round-robin, so every row is rewritten at least once per
// (usable tags - 2) episodes and a stale cell never survives until its tag value comes round again.
// Returns are not accumulated per step: with A_p = number of steps after which player p was alive,
// ret_p = 2 A_p - T + 9 wins_p (alive +1, dead -1, alive at a terminal step +10 = the win count).
// No barrier inside the loop: a lane only ever touches its own slab.
constexpr int kRowBytesSmall = 24, kLdsMaxNSmall = 20;   // 256 games per workgroup (4 waves, one per SIMD)
constexpr int kRowBytesLarge = 44, kLdsMaxNLarge = 40;   // 64 games per workgroup (LDS holds one wave's boards)

struct TronPad {
    int junk;        // slab offset of the junk byte = (N + 2) * RS
    int stride;      // bytes per slab
    int sweep_rows;  // rows rewritten per reset = ceil(N / (usable tags - 2))
    uint32_t inv_nn; // floor(2^32 / (N*N)) + 1: exact quotients for byte offsets inside a wave's 64 boards
    uint32_t inv_nq; // floor(2^32 / (N/4)) + 1 (boards with N % 4 == 0 only)
};

// board view over absolute LDS addresses (no base add in front of every access)
template <int OB>
struct LdsBoard {
    uint32_t tagbits;               // tag << OB
#ifdef CRL_BOUNDS                   /* the bounds-assert build: [lo, hi) = the slab (junk byte included) every access must hit */
    int lo = 0, hi = 0x7fffffff;
    __device__ __forceinline__ void within(const int lo_, const int hi_) { lo = lo_; hi = hi_; }
    __device__ __forceinline__ int raw(const int a) const { CRL_BOUNDS_IN(a, lo, hi, 121); return *(const lds_u8 *)(uintptr_t)(uint32_t)a; }
#else
    __device__ __forceinline__ void within(const int, const int) {}
    __device__ __forceinline__ int raw(const int a) const { return *(const lds_u8 *)(uintptr_t)(uint32_t)a; }
#endif
    __device__ __forceinline__ int owner(const int r) const
    {
        const uint32_t x = (uint32_t)r ^ tagbits;
        return x < (1u << OB) ? (int)x : 0;
    }
    __device__ __forceinline__ void put(const int a, const int who) const
    {
#ifdef CRL_BOUNDS
        CRL_BOUNDS_IN(a, lo, hi, 122);
#endif
        *(lds_u8 *)(uintptr_t)(uint32_t)a = (uint8_t)(tagbits | (uint32_t)who);
    }
};

// Slab byte offsets (relative to the slab start) of the four dwords of a 16-byte piece of a board whose first cell
// dword is cq: cell dword d = row d / nq, dword d % nq of that row (nq = N / 4); slab rows are RS bytes apart and
// row 0 is wall.  One exact division for the piece; its other dwords follow by carry (a piece crosses at most one
// row boundary when a row has at least 4 dwords), which is what keeps the copy loops short.
template <int RS>
__device__ __forceinline__ void tron_piece_offsets(const int cq, const int nq, const uint32_t inv_nq, int (&o)[4])
{
    const int y0 = (int)__umulhi((uint32_t)cq, inv_nq);
    const int r0 = cq - y0 * nq;
    const int base = (y0 + 1) * RS + 4 * r0;
    if (nq >= 4) {                                              // wave-uniform
        const int wrap = RS - 4 * nq;                           // extra bytes on entering the next row
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = base + 4 * q + ((r0 + q >= nq) ? wrap : 0);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = (int)__umulhi((uint32_t)(cq + q), inv_nq);
            o[q] = (y + 1) * RS + 4 * (cq + q - y * nq);
        }
    }
}

template <int P, int RS>
__global__ void __launch_bounds__(256)
tron_rollout_lds_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const int64_t B,
                        const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                        int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                        int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    tron_fill_action_lut(act_lut);
    constexpr int kRowDwords = RS / 4;
    constexpr uint32_t step4 = (uint32_t)((-RS) & 0xff) | (1u << 8) | ((uint32_t)RS << 16) | (0xffu << 24);
    const int N = g.N, NN = g.NN;
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    const int64_t bb = valid ? b : 0;
    const int64_t env0 = b - lane;                              // first game of this wave
    const int n_env = (int)((B - env0) < CRL_WAVE ? (B - env0 > 0 ? B - env0 : 0) : CRL_WAVE);
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;   // LDS address of the dynamic part
    const int mine = lds0 + (int)threadIdx.x * pad.stride;      // this lane's slab
    const bool wide = (N & 3) == 0 && N >= 8;                  // rows are whole dwords in HBM too (4x4: inv_nq would overflow)
    constexpr int OB = (P <= 7) ? 3 : 4;                        // owner bits; 8 - OB tag bits
    constexpr uint32_t kTags = (1u << (8 - OB)) - 1u;           // the all-ones tag is never used: 0xff stays "wall"
    // a fresh row: cells 0, walls 0xff (kept in registers for the rolling rewrite)
    uint32_t rowpat[kRowDwords];
#pragma unroll
    for (int j = 0; j < kRowDwords; ++j) {
        uint32_t w = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) w |= (4 * j + k < N) ? 0u : (0xffu << (8 * k));
        rowpat[j] = w;
        asm volatile("" : "+v"(rowpat[j]));
    }
    // ---- copy in: walls everywhere, then the cells.  Canonical HBM cells are tag 0.
    // The wave streams its 64 boards as one contiguous run, 16 bytes per lane and load (coalesced), and scatters the
    // dwords into the owners' slabs (a 16-byte piece never straddles two boards, its dwords may change row).  A lone
    // wave per SIMD hides HBM latency only through its own loads in flight, so the first kCopyBatch loads (all 25 of a
    // 20x20 launch) are issued BEFORE the wall fill and scattered after it; short launches are dominated by this copy.
    constexpr int kCopyBatch = 26;
    const int8_t *gslab_in = board + (n_env > 0 ? env0 : 0) * NN;   // (a wave wholly beyond the batch still issues its loads: from game 0)
    const int slab0_in = mine - lane * pad.stride;
    const int bytes_in = n_env * NN;
    uint4 cin[kCopyBatch];
    if (wide) {
#pragma unroll
        for (int k = 0; k < kCopyBatch; ++k) {                  // unconditional loads (a piece beyond the wave's boards
            const int off = lane * 16 + k * (CRL_WAVE * 16);    // re-reads piece 0): no exec branches between them, so the
            cin[k] = *reinterpret_cast<const uint4 *>(gslab_in + (off < bytes_in ? off : 0));   // scatter below can wait for
        }                                                       // them one by one (vmcnt(N)) instead of for all (vmcnt(0))
    }
    // the per-player state and the step counters are requested now as well, so that every global load of the prologue
    // is in flight together (their latency used to be exposed one after the other behind the board copy)
    int h_in[P], d_in[P], k_in[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        h_in[p] = valid ? heads[p * B + bb] : 0;
        d_in[p] = valid ? dirs[p * B + bb] : 0;
        k_in[p] = valid ? deaths[p * B + bb] : 1;
    }
    TronAcc<P> acc;
    acc.load(st, valid, bb, B);
    for (int off = 0; off < pad.stride; off += 4) *(lds_u32 *)(uintptr_t)(uint32_t)(mine + off) = 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the wall fill above vs other lanes' cell writes
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wide) {
        auto scatter = [&](const uint4 &v, const int off) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            const int sb = slab0_in + e * pad.stride;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]) = w[q];
        };
#pragma unroll
        for (int k = 0; k < kCopyBatch; ++k) {
            const int off = lane * 16 + k * (CRL_WAVE * 16);
            if (off < bytes_in) scatter(cin[k], off);
        }
#pragma unroll 5
        for (int off = lane * 16 + kCopyBatch * (CRL_WAVE * 16); off < bytes_in; off += CRL_WAVE * 16)   // boards above 20x20
            scatter(*reinterpret_cast<const uint4 *>(gslab_in + off), off);
    } else {
        for (int e = 0; e < n_env; ++e)
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                *(lds_u8 *)(uintptr_t)(uint32_t)(slab0_in + e * pad.stride + (y + 1) * RS + (c - y * N)) = (uint8_t)gslab_in[(int64_t)e * NN + c];
            }
    }
    // heads as LDS addresses
    TronRegs<P> s, fresh;                                       // fresh = the start layout, kept in VGPRs for resets
    int act[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int h = min(max(h_in[p], 0), NN - 1);             // a hand-made state must not turn into a wild LDS address
        const int y = (int)__umulhi((uint32_t)h, g.inv_n);
        s.h[p] = mine + (y + 1) * RS + (h - y * N);
        s.d[p] = d_in[p] & 3;
        s.k[p] = k_in[p];
        const int fh = cfg.start_heads[p];
        const int fy = (int)__umulhi((uint32_t)fh, g.inv_n);
        fresh.h[p] = mine + (fy + 1) * RS + (fh - fy * N);
        fresh.d[p] = cfg.start_dirs[p];
        fresh.k[p] = 0;
        asm volatile("" : "+v"(fresh.h[p]), "+v"(fresh.d[p]));
    }
    const int junk = mine + pad.junk;
    int sweep = mine + RS;                                      // next row of the rolling rewrite
    const int sweep_end = mine + (N + 1) * RS;
    const uint32_t ts_at_entry = acc.ts;
    uint32_t alive_steps[P];
    int last_k[P];                                              // deaths at the latest terminal step
#pragma unroll
    for (int p = 0; p < P; ++p) { alive_steps[p] = 0; last_k[p] = 1; }
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);
    TronRng<P> rng;
    rng.start(gid, acc.tc, seed_lo, seed_hi);
    LdsBoard<OB> bd{0u};
    bd.within(mine, mine + pad.stride);
    uint32_t stamp[P];                                          // the byte a trail cell of player p gets: tag | p + 1
#pragma unroll
    for (int p = 0; p < P; ++p) stamp[p] = (uint32_t)(p + 1);
    __syncthreads();                                            // action table; (non-wide) slabs written by other lanes
    rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);
    for (int t = 0; t < T; ++t) {
        TronProbe<P> pr;
        tron_probe_padded<P>(step4, bd, s, act, pr);            // P ds_read_u8 in flight ...
        acc.tc += 1;
        rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);   // ... while the NEXT step's actions are drawn
        tron_resolve_lds<P>(bd, s, pr, stamp, junk);
        int alive = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) alive += (s.k[p] == 0);
        const bool term = alive <= 1;                           // TronGridEnvironment.py:309-321
        acc.ts += 1;
#pragma unroll
        for (int p = 0; p < P; ++p) alive_steps[p] += (s.k[p] == 0);
        if (valid && term) {
            // new_state for this game only: bump the tag (every stale cell becomes empty), rewrite the next row(s)
            // of the rolling clear, stamp the heads
            uint32_t tag = (bd.tagbits >> OB) + 1u;
            tag = (tag == kTags) ? 0u : tag;
            bd.tagbits = tag << OB;
#pragma unroll
            for (int p = 0; p < P; ++p) stamp[p] = bd.tagbits | (uint32_t)(p + 1);
            for (int r = 0; r < pad.sweep_rows; ++r) {
#pragma unroll
                for (int j = 0; j < kRowDwords; ++j) *(lds_u32 *)(uintptr_t)(uint32_t)(sweep + 4 * j) = rowpat[j];
                sweep += RS;
                sweep = (sweep == sweep_end) ? mine + RS : sweep;
            }
#pragma unroll
            for (int p = 0; p < P; ++p) *(lds_u8 *)(uintptr_t)(uint32_t)fresh.h[p] = (uint8_t)stamp[p];
            acc.n_ep += 1;
            acc.last_len = (int)acc.ts;
            acc.ts = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                acc.wins[p] += (s.k[p] == 0);                   // the winners are whoever is alive at the terminal step
                last_k[p] = s.k[p];
                s.h[p] = fresh.h[p]; s.d[p] = fresh.d[p]; s.k[p] = 0;
            }
        }
    }
    acc.len_sum = ts_at_entry + (uint32_t)T - acc.ts;          // steps of this launch that belong to finished episodes
    if (acc.n_ep > 0) {
        acc.last_w = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) acc.last_w |= (last_k[p] == 0) << p;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) acc.ret[p] = 2 * (int)alive_steps[p] - T + 9 * (int)acc.wins[p];
    // ---- copy out: LDS -> HBM, dropping the tags (cells of older episodes become 0)
    constexpr uint32_t OM = 0x01010101u * ((1u << OB) - 1u);    // owner bits of 4 cells
    constexpr uint32_t TM = 0x01010101u * ((1u << (8 - OB)) - 1u);   // tag bits of 4 cells, shifted down
    *(lds_u8 *)(uintptr_t)(uint32_t)junk = (uint8_t)(bd.tagbits >> OB);   // the junk byte hands this board's tag to its copier
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // slabs are read by other lanes of the wave below
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wide) {
        int8_t *gslab = board + env0 * NN;
        const int slab0 = mine - lane * pad.stride;
        const int bytes = n_env * NN;
#pragma unroll 5
        for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            const int sb = slab0 + e * pad.stride;
            const uint32_t trep = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk) * 0x01010101u;
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c4 = *(const lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]);
                const uint32_t diff = ((c4 >> OB) & TM) ^ trep;                  // per byte: 0 iff the tag matches
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u; // per byte: 1 iff diff != 0 (diff <= 0x1f)
                w[q] = c4 & OM & ~(stale * 0xffu);
            }
            *reinterpret_cast<uint4 *>(gslab + off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        int8_t *gslab = board + env0 * NN;
        const int slab0 = mine - lane * pad.stride;
        for (int e = 0; e < n_env; ++e) {
            const uint32_t tb = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + pad.junk) << OB;
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                const uint32_t raw = *(const lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + (y + 1) * RS + (c - y * N));
                gslab[(int64_t)e * NN + c] = (int8_t)((((raw ^ tb) >> OB) == 0) ? (raw & ((1u << OB) - 1u)) : 0u);
            }
        }
    }
    if (valid) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int rel = s.h[p] - mine;
            const int row = rel / RS;                                                // = y + 1
            heads[p * B + b] = (int16_t)((row - 1) * N + (rel - row * RS));
            dirs[p * B + b] = (int8_t)s.d[p];
            deaths[p * B + b] = (int8_t)s.k[p];
        }
        acc.store(st, B, b);
    }
}

// ---- quad rollout: one lane per PLAYER, four lanes per game (P <= 4, boards up to 20x20) ---------------------------
// 65,536 games are 1,024 waves when a lane is a game: one wave per SIMD, and a lone wave issues one instruction per ~5.9
// cycles where an occupied SIMD issues one per ~2.2-2.9 (profiles/r2_valu_issue_calibration.json).  More waves need
// more lanes per game, and that pays only if the per-lane work shrinks with it.  It does, because players almost never
// interact: a player's target is another player's head, or two players target the same cell, in 0.08 % of game-steps
// (5 % of 64-game wave-steps, 1.3 % of 16-game wave-steps).  So a lane plays ONE player -- probe, decode, die or
// move, trail write -- and the quad only shares what is per game: the terminal test (two DPP adds), the reset, the
// random word.  The sequential semantics of the reference (CyTronGrid.pyx:15-62) are needed only when a wave detects an
// interaction (six DPP xors + min3 per lane); that wave-step then gathers the four players of every quad with DPP
// broadcasts and runs the same tron_resolve_lds as the lane-per-game kernel, in all four lanes redundantly.
// Slabs, tags, rolling row rewrite, RNG and copy-in / copy-out are those of tron_rollout_lds_kernel; a workgroup holds
// 64 games (34 KB of LDS at 20x20), four workgroups fit a CU: 16 waves = 4 per SIMD.  Results are bit-identical.
template <int CTRL>
__device__ __forceinline__ int tron_quad(const int v)          // v of the lane quad_perm CTRL selects inside the quad
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}

// The game's random stream, shared out over a quad of lanes (one lane per player; both lane-per-player kernels).  The
// contract (include/colosseum_hip.h) gives every step c a code byte -- four 2-bit actions -- from word (c & 7) >> 1 of
// Philox block c >> 3.  Computed per lane that is a Philox call every 8 steps in each of the four lanes.  Instead lane q
// computes block 4 (c >> 5) + q once per 32 steps, turns its four words into the block's 8 code bytes (two registers:
// steps 0-3, 4-7), transposes each register as a 4x4 matrix of 2-bit fields (steps x players -> players x steps: two delta
// swaps), so that byte p' holds player p''s four actions, and the quad exchanges them with DPP broadcasts: lane p ends up
// with its player's 32 actions of group `group` = c >> 5 in two registers (steps 0-15 / 16-31 of the group).
// Every lane of the quad must call this together (the broadcasts read all four lanes).
__device__ __forceinline__ void tron_quad_actions(const uint32_t gid, const uint32_t group, const int p, const uint32_t seed_lo,
                                                  const uint32_t seed_hi, const uint8_t *act_lut, uint32_t &a_lo, uint32_t &a_hi)
{
    auto transpose2 = [](uint32_t x) -> uint32_t {              // 4x4 transpose of the 2-bit fields of x (bit 8r + 2c)
        uint32_t t = (x ^ (x >> 6)) & 0x00CC00CCu;
        x ^= t ^ (t << 6);
        t = (x ^ (x >> 12)) & 0x0000F0F0u;
        x ^= t ^ (t << 12);
        return x;
    };
    // (the seed through an opaque copy: the ten rounds' keys are then derived here, every 32 steps, instead of living in
    //  20 scalar registers across the caller's step loop)
    uint32_t k0 = seed_lo, k1 = seed_hi;
    asm volatile("" : "+s"(k0), "+s"(k1));
    const philox_out r = philox4x32_10<true>(gid, 4u * group + (uint32_t)p, 0u, CRL_TAG_TRON, k0, k1);
    uint32_t code[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t frac, first;                                   // w * 81 = first : frac (one v_mad_u64_u32)
        crl_mul_wide<true>(81u, r.w[i], frac, first);
        code[2 * i] = act_lut[first];
        code[2 * i + 1] = act_lut[__umulhi(frac, 81u)];
    }
    const uint32_t lo = transpose2(code[0] | code[1] << 8 | code[2] << 16 | code[3] << 24);   // byte p' = player p', steps 0-3
    const uint32_t hi = transpose2(code[4] | code[5] << 8 | code[6] << 16 | code[7] << 24);   //                 steps 4-7
    const uint32_t r01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);   // {lo.b0, hi.b0, lo.b1, hi.b1}: players 0, 1 (16 bits each)
    const uint32_t r23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);   // {lo.b2, hi.b2, lo.b3, hi.b3}: players 2, 3
    // (every broadcast runs in all four lanes, THEN the lane picks: a DPP read of a lane that sits out a divergent branch
    //  returns 0)
    const int sh = (p & 1) * 16;
    const int b01[4] = {tron_quad<0x00>((int)r01), tron_quad<0x55>((int)r01), tron_quad<0xAA>((int)r01), tron_quad<0xFF>((int)r01)};
    const int b23[4] = {tron_quad<0x00>((int)r23), tron_quad<0x55>((int)r23), tron_quad<0xAA>((int)r23), tron_quad<0xFF>((int)r23)};
    uint32_t s16[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) s16[q] = ((uint32_t)(p < 2 ? b01[q] : b23[q]) >> sh) & 0xffffu;
    a_lo = s16[0] | s16[1] << 16;
    a_hi = s16[2] | s16[3] << 16;
}

// ---- diagnostic build only (-DCRL_QUAD_STAMPS): when each wave of the lane-per-player byte kernel passes its phases
// (100 MHz wall clock: entry, boards in LDS, steps done, kernel end).  Leaves the kernel through g_quad_stamps alone; the
// shipped build compiles none of it (tools/debug/quad_phases.py).
#ifdef CRL_QUAD_STAMPS
__device__ unsigned long long g_quad_stamps[4096 * 4];
#define QUAD_STAMP(i)                                                                                          \
    do {                                                                                                       \
        const unsigned wid_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                             \
        if ((threadIdx.x & 63) == 0 && wid_ < 4096u) g_quad_stamps[wid_ * 4u + (i)] = wall_clock64();           \
    } while (0)
#else
#define QUAD_STAMP(i) do { } while (0)
#endif

#ifndef CRL_QUAD_WG
#define CRL_QUAD_WG 256          /* threads per workgroup (diagnostic builds: 512 / 1024, tools/README.md) */
#endif
template <int RS>
__global__ void __launch_bounds__(CRL_QUAD_WG, 1024 / CRL_QUAD_WG)
tron_rollout_quad_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const int64_t B,
                         const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                         int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                         int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    constexpr int kGames = CRL_QUAD_WG / 4, kWaveGames = 16;
    constexpr int kRowDwords = RS / 4;
    constexpr uint32_t step4 = (uint32_t)((-RS) & 0xff) | (1u << 8) | ((uint32_t)RS << 16) | (0xffu << 24);
    constexpr int OB = 3;                                       // P <= 4: owners 1..4, 5 tag bits
    constexpr uint32_t kTags = (1u << (8 - OB)) - 1u;           // the all-ones tag is never used: 0xff stays "wall"
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    QUAD_STAMP(0);
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN, P = cfg.P;
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int p = lane & 3;                                     // my player
    const int slot = threadIdx.x >> 2;                          // my game's slab in this workgroup
    const int64_t b = (int64_t)blockIdx.x * kGames + slot;
    const bool gvalid = b < B;
    const bool pvalid = gvalid && p < P;
    const int64_t bb = gvalid ? b : 0;
    const int64_t env0 = (int64_t)blockIdx.x * kGames + wave * kWaveGames;   // first game of this wave
    const int n_env = (int)((B - env0) < kWaveGames ? (B - env0 > 0 ? B - env0 : 0) : kWaveGames);
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;
    const int mine = lds0 + slot * pad.stride;                  // my game's slab
    const int slab0 = lds0 + wave * kWaveGames * pad.stride;    // first slab of this wave
    const bool wide = (N & 3) == 0 && N >= 8;
    // ---- prologue: every global load in flight together (the wave's boards, my player's state, the running totals)
    constexpr int kCopyBatch = 7;                               // 16 boards of <= 400 bytes = <= 6.25 KiB per wave
    const int8_t *gslab_in = board + (n_env > 0 ? env0 : 0) * NN;   // (a wave wholly beyond the batch still issues its loads: from game 0)
    const int bytes_in = n_env * NN;
    // 20x20 boards (the headline shape) move by ROWS: a row is 20 bytes = five dwords in HBM and six in the slab (its four
    // wall bytes included), so a lane takes rows lane, lane + 64, ... of the wave's 320: a 16-byte + a 4-byte load each
    // (consecutive lanes read consecutive rows: whole cache lines), three two-dword LDS stores, one division for the
    // game -- no per-dword slab offsets, no wall fill to overwrite.  Other sizes move by 16-byte pieces as before.
    constexpr int kRowsPerLane = (kWaveGames * 20) / CRL_WAVE;  // 5
    const bool rows20 = N == 20;
    const int rows_in = n_env * 20;
    uint4 cin[kCopyBatch];
    u32x4_a4 rin4[kRowsPerLane];
    uint32_t rin1[kRowsPerLane];
    // Boards whose width is not a multiple of four (19 x 19, the reference's default; round 5) move by rows as well: row R of the
    // wave's 16 * N starts at byte R * N of its run -- any alignment --, so a lane loads the six dwords around it and funnels
    // them to the row's start (v_alignbyte); the copy back stores a row as leading bytes, aligned dwords, trailing bytes.  Not
    // for the wave that holds the batch's last bytes when those are not a whole dword (it keeps the byte loop).
    const bool rowsN = !rows20 && !wide && N >= 4 && N <= 20 && (((uintptr_t)board & 3) == 0) &&
                       !(env0 + kWaveGames >= B && ((B * (int64_t)NN) & 3) != 0);
    const int rows_inN = n_env * N;
    uint32_t rwN[kRowsPerLane][6];
    if (rowsN) {
#pragma unroll
        for (int k = 0; k < kRowsPerLane; ++k) {
            const int R = lane + k * CRL_WAVE;
            const uint32_t boff = (uint32_t)((R < rows_inN ? R : 0) * N);
            const int8_t *src = gslab_in + (boff & ~3u);
#pragma unroll
            for (int i = 0; i < 6; ++i) rwN[k][i] = *reinterpret_cast<const uint32_t *>(src + 4 * i);
        }
    } else if (rows20) {
#pragma unroll
        for (int k = 0; k < kRowsPerLane; ++k) {
            const int R = lane + k * CRL_WAVE;
            const int8_t *src = gslab_in + (R < rows_in ? R * 20 : 0);
            rin4[k] = *reinterpret_cast<const u32x4_a4 *>(src);
            rin1[k] = *reinterpret_cast<const uint32_t *>(src + 16);
        }
    } else if (wide) {
#pragma unroll
        for (int k = 0; k < kCopyBatch; ++k) {
            const int off = lane * 16 + k * (CRL_WAVE * 16);
            cin[k] = *reinterpret_cast<const uint4 *>(gslab_in + (off < bytes_in ? off : 0));
        }
    }
    // (unconditional, from a clamped index: under `pvalid ? load : 0` every load sits in its own branch with its own wait)
    const int64_t pb = (int64_t)(p < P ? p : 0) * B + bb;
    const int h_in = heads[pb];
    const int d_in = dirs[pb];
    int k = deaths[pb];
    const int old_ret = st.ret_sum[pb];
    const uint32_t old_wins = st.win_count[pb];
    uint32_t tc = st.tcount[bb], ts = st.tstep[bb];
    const uint32_t old_n_ep = st.n_episodes[bb], old_len_sum = st.len_sum[bb];
    const uint32_t old_last_w = st.last_winners[bb];
    k = pvalid ? k : 1;                                         // a seat without a player counts as dead for good
    if (rows20 || rowsN) {
        // the two wall rows and the junk dword of my game (three dwords per lane), then my rows of the wave's boards
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int dw = 3 * p + i;
            *(lds_u32 *)(uintptr_t)(uint32_t)(mine + (dw < 6 ? 4 * dw : (N + 1) * RS + 4 * (dw - 6))) = 0xffffffffu;
        }
        if (p == 0) *(lds_u32 *)(uintptr_t)(uint32_t)(mine + pad.junk) = 0xffffffffu;
    }
    if (rows20) {
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {
            const int R = lane + k2 * CRL_WAVE;
            const bool have = R < rows_in;                      // rows of games beyond the batch: empty
            const int e = (int)__umulhi((uint32_t)R, g.inv_n);  // R = 20 e + r
            const int a = slab0 + e * pad.stride + (R - 20 * e + 1) * RS;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a) = have ? rin4[k2].x : 0u;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a + 4) = have ? rin4[k2].y : 0u;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a + 8) = have ? rin4[k2].z : 0u;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a + 12) = have ? rin4[k2].w : 0u;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a + 16) = have ? rin1[k2] : 0u;
            *(lds_u32 *)(uintptr_t)(uint32_t)(a + 20) = 0xffffffffu;
        }
    } else if (!rowsN) {
        // walls everywhere (each lane a quarter of its game's slab), then the cells
        for (int off = 4 * p; off < pad.stride; off += 16) *(lds_u32 *)(uintptr_t)(uint32_t)(mine + off) = 0xffffffffu;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (rows20) {
    } else if (rowsN) {
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {
            const int R = lane + k2 * CRL_WAVE;
            if (R < kWaveGames * N) {
                const bool have = R < rows_inN;                 // rows of games beyond the batch: empty
                const uint32_t m8 = ((uint32_t)((have ? R : 0) * N) & 3u);
                const int e = (int)__umulhi((uint32_t)R, g.inv_n);      // R = N e + y
                const int a = slab0 + e * pad.stride + (R - N * e + 1) * RS;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    uint32_t d = __builtin_amdgcn_alignbyte(rwN[k2][q + 1], rwN[k2][q], m8);
                    d = have ? d : 0u;
                    const int keep = N - 4 * q;                 // cells of this dword that are on the board; the rest is wall
                    d = keep >= 4 ? d : (keep <= 0 ? 0xffffffffu : (d | (0xffffffffu << (8 * keep))));
                    *(lds_u32 *)(uintptr_t)(uint32_t)(a + 4 * q) = d;
                }
                *(lds_u32 *)(uintptr_t)(uint32_t)(a + 20) = 0xffffffffu;
            }
        }
    } else if (wide) {
#pragma unroll
        for (int k2 = 0; k2 < kCopyBatch; ++k2) {
            const int off = lane * 16 + k2 * (CRL_WAVE * 16);
            if (off < bytes_in) {
                const uint32_t w[4] = {cin[k2].x, cin[k2].y, cin[k2].z, cin[k2].w};
                const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
                int o[4];
                tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
                const int sb = slab0 + e * pad.stride;
#pragma unroll
                for (int q = 0; q < 4; ++q) *(lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]) = w[q];
            }
        }
    } else {
        for (int e = 0; e < n_env; ++e)
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                *(lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + (y + 1) * RS + (c - y * N)) = (uint8_t)gslab_in[(int64_t)e * NN + c];
            }
    }
    // ---- my player: head as an LDS address, the start layout for resets
    const int junk = mine + pad.junk + p;                       // four junk bytes per slab: one per lane of the quad
    int fh = cfg.start_heads[0], fd = cfg.start_dirs[0];
    fh = (p == 1) ? cfg.start_heads[1] : fh; fd = (p == 1) ? cfg.start_dirs[1] : fd;
    fh = (p == 2) ? cfg.start_heads[2] : fh; fd = (p == 2) ? cfg.start_dirs[2] : fd;
    fh = (p == 3) ? cfg.start_heads[3] : fh; fd = (p == 3) ? cfg.start_dirs[3] : fd;
    const int fy = (int)__umulhi((uint32_t)(fh < 0 ? 0 : fh), g.inv_n);
    const int fresh_h = (p < P) ? mine + (fy + 1) * RS + (fh - fy * N) : junk;
    const int fresh_d8 = fd << 3;
    const int fresh_a = (p < P) ? 1 : 0;
    int h, d8 = (d_in & 3) << 3;                                // directions live pre-scaled (the bit offset into step4),
    {                                                           // bits above 4:3 are garbage
        const int hc = min(max(h_in, 0), NN - 1);
        const int y = (int)__umulhi((uint32_t)hc, g.inv_n);
        h = pvalid ? mine + (y + 1) * RS + (hc - y * N) : junk;   // a seat without a player never matches a target
    }
    // my share of a fresh row: dwords p and p + 2 of its six (dwords 2 and 3 are stored twice, with the same value)
    static_assert(kRowDwords == 6, "the row rewrite of the quad kernel shares out six dwords");
    uint32_t rp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = p + 2 * i;
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) w |= (4 * j + c < N) ? 0u : (0xffu << (8 * c));
        rp[i] = w;
    }
    int sweep = mine + RS + 4 * p;                              // my first dword of the row the next reset rewrites
    const int sweep_end = mine + (N + 1) * RS + 4 * p, sweep_first = mine + RS + 4 * p;
    uint32_t tagbits = 0, stamp = (uint32_t)(p + 1);
    uint32_t alive_steps = 0, wn = 0;                           // wn: episodes << 16 | my wins
    uint32_t marks = 0;                                         // the latest reset in bits 15:0, the one before in 31:16:
                                                                // (launch steps done at the reset) << 1 | I was alive
    if (!gvalid) k = 1;
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);
    __syncthreads();                                            // action table; (non-wide) slabs written by other lanes
    QUAD_STAMP(1);
    uint32_t a_lo = 0, a_hi = 0;                                // my player's actions of steps 0-15 / 16-31 of the group
    auto refill = [&](const uint32_t group) { tron_quad_actions(gid, group, p, seed_lo, seed_hi, act_lut, a_lo, a_hi); };
    refill(tc >> 5);
    uint32_t acts = ((tc & 16u) ? a_hi : a_lo) >> ((tc & 15u) * 2u);   // bits 1:0 = this step's action code (0, 1, 3)
    // `acts` runs dry after dry2 / 2 steps of this launch; neg2 = 2 (launch steps done) - dry2 counts up to zero
    uint32_t dry2 = 32u - 2u * (tc & 15u);
    int neg2 = -(int)dry2;
    const uint32_t tc_in = tc;
    // my player is alive (k only says why it is not): carried as a 0 / 1 vector register -- one compare per step makes
    // the lane mask; carried AS a lane mask it costs ~8 scalar instructions per step to merge around the reset branch
    int a = (k == 0) ? 1 : 0;
    // Cell decode.  A cell is tag << 3 | owner; walls are 0xff and the tag 31 is never used, so with x = cell ^ tag << 3
    // and w = cell ^ 0xf8 the cell is occupied for THIS episode iff 1 <= min(x, w) <= 7: x is the owner under the
    // current tag (>= 8 under a stale one), w is 7 for a wall (>= 8 for every other cell).  That minimum also is what
    // deaths[] records, except that a wall reads 7 where the reference stores the mover's own id (CyTronGrid.pyx:47-48):
    // translated once, at the end.
    // The action countdown: per lane in general (games may enter with different step counters); when the whole wave
    // shares one (the usual case: the games of a batch are rolled out together) it is a scalar -- add, compare and branch
    // on the scalar unit instead of two vector instructions and an exec-masked branch per step.
    const bool uni_wave = __builtin_amdgcn_ballot_w64(tc != (uint32_t)__builtin_amdgcn_readfirstlane((int)tc)) == 0ull;
    int sneg2 = __builtin_amdgcn_readfirstlane(neg2);
    uint32_t sdry2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dry2);
#ifdef CRL_DIAG_STALE_PROBE
    uint32_t stale_raw = 0;
#endif
    auto one_step = [&](auto uni_tag) {
        constexpr bool UNI = decltype(uni_tag)::value;
        const bool run = a != 0;
        const int dir8 = (int)((acts << 3) + (uint32_t)d8);     // bits 4:3: (d + action) & 3 (a bit-field offset reads 5 bits)
        const int tgt = h + __builtin_amdgcn_sbfe((int)step4, dir8, 8);
        const int tq = run ? tgt : junk;                        // a dead player probes its own junk byte
        // (the probe as inline asm: the compiler would mask the byte it loads with 0xff once more; its own waitcnts stay
        //  correct, LDS operations retire in order and an extra one in flight only makes them conservative)
        uint32_t raw;
        CRL_BOUNDS_IN(tq, mine, mine + pad.stride, 101);       // the probe stays inside my game's slab (junk byte included)
        asm volatile("ds_read_u8 %0, %1" : "=v"(raw) : "v"(tq) : "memory");
        // does anything in this wave need the reference's sequential order?  my target against the other players'
        // heads (head-on, CyTronGrid.pyx:51-57) and targets (two players entering one cell; every pair is seen from
        // one of its two lanes by the rotations by one and two)
        const int x1 = tq ^ tron_quad<0x39>(h), x2 = tq ^ tron_quad<0x4E>(h), x3 = tq ^ tron_quad<0x93>(h);
        const int y1 = tq ^ tron_quad<0x39>(tq), y2 = tq ^ tron_quad<0x4E>(tq);
        uint32_t near = min(min(min(min((uint32_t)x1, (uint32_t)x2), (uint32_t)x3), (uint32_t)y1), (uint32_t)y2);
#ifdef CRL_DIAG_STALE_PROBE      /* diagnostic builds only (WRONG results): the step decides on the PREVIOUS step's probe word, so that the
                                   probe's latency is hidden completely -- the upper bound of what a software-pipelined probe could buy */
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(stale_raw), "+v"(near) : : "memory");
        { const uint32_t t_ = raw; raw = stale_raw; stale_raw = t_; }
#else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw), "+v"(near) : : "memory");   // (near: the test runs while the probe is out)
#endif
        // The common path, for every lane and without a branch: each player on its own (correct unless players interact).
        const uint32_t m = min(raw ^ tagbits, raw ^ 0xf8u);
        const bool dead = run & ((m - 1u) < 7u);                // :47-57
        const bool moved = run ^ dead;                          // :60-62
        const int h_was = h;
        k = dead ? (int)m : k;
        d8 = run ? dir8 : d8;                                   // :44 the direction is committed even if the move dies
        h = moved ? tgt : h;
        // the trail: a head cell already holds its player's stamp (new_state / every earlier move put it there), so a
        // player that stays where it is restamps its own head and the store needs no condition
        CRL_BOUNDS_IN(h, mine, mine + pad.stride, 102);        // ... and so does the trail store
        *(lds_u8 *)(uintptr_t)(uint32_t)h = (uint8_t)stamp;
        bool alive_now = moved;
#if defined(CRL_DIAG_NO_SLOW)      /* diagnostic builds only (WRONG results): what the interaction path costs the common one */
        if (false && near == 0u) {
#elif defined(CRL_DIAG_DETECT_ONLY) /* ... and with the detection kept but nothing behind it */
        if (__builtin_amdgcn_ballot_w64(near == 0u) != 0ull) asm volatile("s_nop 0");
        if (false) {
#else
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(near == 0u) != 0ull, 0)) {
#endif
            // Rare (1.3 % of wave-steps): some quad of the wave needs the reference's order.  A fix-up AFTER the common
            // path rather than an alternative to it -- as an if / else the join cost every step a dozen scalar
            // instructions merging lane masks and a copy of the selects (13 % of the kernel's time).  The pre-step state
            // is rebuilt from what is still in registers: a player that ran stood at tgt - step and had direction
            // dir - action (acts is shifted only at the end of the step); one that did not run is unchanged.
            // Cells the common path stamped go back to "empty" first (0 reads as empty under every tag; they were empty:
            // the player moved there), then the wave gathers every quad's four players and resolves them in order,
            // redundantly in the quad's four lanes: that rewrites the stamp of every player that really moved.
            if (moved) *(lds_u8 *)(uintptr_t)(uint32_t)tgt = (uint8_t)0;
            TronRegs<4> s;
            TronProbe<4> pr;
            uint32_t stamp4[4];
            const int h0 = h_was;
            const int d0 = (run ? (dir8 - (int)(acts << 3)) >> 3 : d8 >> 3) & 3, dir = (dir8 >> 3) & 3, kk = run ? 0 : k;
            s.h[0] = tron_quad<0x00>(h0); s.h[1] = tron_quad<0x55>(h0); s.h[2] = tron_quad<0xAA>(h0); s.h[3] = tron_quad<0xFF>(h0);
            s.d[0] = tron_quad<0x00>(d0); s.d[1] = tron_quad<0x55>(d0); s.d[2] = tron_quad<0xAA>(d0); s.d[3] = tron_quad<0xFF>(d0);
            s.k[0] = tron_quad<0x00>(kk); s.k[1] = tron_quad<0x55>(kk); s.k[2] = tron_quad<0xAA>(kk); s.k[3] = tron_quad<0xFF>(kk);
            pr.tgt[0] = tron_quad<0x00>(tgt); pr.tgt[1] = tron_quad<0x55>(tgt); pr.tgt[2] = tron_quad<0xAA>(tgt); pr.tgt[3] = tron_quad<0xFF>(tgt);
            pr.raw[0] = tron_quad<0x00>((int)raw); pr.raw[1] = tron_quad<0x55>((int)raw); pr.raw[2] = tron_quad<0xAA>((int)raw); pr.raw[3] = tron_quad<0xFF>((int)raw);
            pr.ndir[0] = tron_quad<0x00>(dir); pr.ndir[1] = tron_quad<0x55>(dir); pr.ndir[2] = tron_quad<0xAA>(dir); pr.ndir[3] = tron_quad<0xFF>(dir);
#pragma unroll
            for (int q = 0; q < 4; ++q) stamp4[q] = tagbits | (uint32_t)(q + 1);
            LdsBoard<OB> bd{tagbits};
            bd.within(mine, mine + pad.stride);
            tron_resolve_lds<4>(bd, s, pr, stamp4, junk);       // trail writes of all four players from every lane: identical
            int hS = s.h[0], dS = s.d[0], kS = s.k[0];
            hS = (p == 1) ? s.h[1] : hS; dS = (p == 1) ? s.d[1] : dS; kS = (p == 1) ? s.k[1] : kS;
            hS = (p == 2) ? s.h[2] : hS; dS = (p == 2) ? s.d[2] : dS; kS = (p == 2) ? s.k[2] : kS;
            hS = (p == 3) ? s.h[3] : hS; dS = (p == 3) ? s.d[3] : dS; kS = (p == 3) ? s.k[3] : kS;
            h = (p < P) ? hS : junk;
            d8 = dS << 3;
            k = kS;
            alive_now = kS == 0;
        }
        // TronGridEnvironment.py:309-321 for the game: alive players over the quad
        a = alive_now ? 1 : 0;
        int alive = a + tron_quad<0xB1>(a);
        alive += tron_quad<0x4E>(alive);
        alive_steps += (uint32_t)a;
        if constexpr (UNI) sneg2 += 2; else neg2 += 2;          // the action countdown (its refill: end of the step)
        if (alive <= 1) {                                       // (a game beyond the batch is "over" at every step)
            // new_state: bump the tag, rewrite the next row of the rolling clear (boards up to 20x20 with 5 tag bits:
            // crl_tron_rollout checks sweep_rows == 1), stamp the heads
            tagbits += 1u << OB;
            tagbits = (tagbits == (kTags << OB)) ? 0u : tagbits;
            stamp = tagbits | (uint32_t)(p + 1);
            *(lds_u32 *)(uintptr_t)(uint32_t)sweep = rp[0];
            *(lds_u32 *)(uintptr_t)(uint32_t)(sweep + 8) = rp[1];
            sweep += RS;
            sweep = (sweep == sweep_end) ? sweep_first : sweep;
            *(lds_u8 *)(uintptr_t)(uint32_t)fresh_h = (uint8_t)stamp;    // seats without a player: their junk byte
            // episodes + 1, my wins + a (the winners are whoever is alive at the terminal step), and the reset's step count
            // pushed into `marks` (launch steps done = (dry2 + neg2) / 2: no use of the scalar t, which would turn it into a
            // vector register).  As inline asm: the compiler folds `0x10000 + a` into a select of two constants plus an
            // add, and the shift-add pair into three instructions.
            asm("v_add3_u32 %0, %0, %1, %2" : "+v"(wn) : "v"(a), "s"(0x10000u));
            if constexpr (UNI) {
                const uint32_t sdone2 = sdry2 + (uint32_t)sneg2;
                asm("v_lshl_add_u32 %0, %0, 16, %1\n\tv_add_u32 %0, %0, %2" : "+v"(marks) : "s"(sdone2), "v"(a));
            } else {
                asm("v_lshl_add_u32 %0, %0, 16, %1\n\tv_add3_u32 %0, %0, %2, %3" : "+v"(marks) : "v"(dry2), "v"(neg2), "v"(a));
            }
            h = fresh_h; d8 = fresh_d8;
            a = fresh_a;
        }
        acts >>= 2;
        if constexpr (UNI) {
            if (sneg2 == 0) {                                   // the whole wave at once
                const uint32_t c = tc_in + (sdry2 >> 1);        // the step the new actions are for
                if ((c & 16u) == 0u) refill(c >> 5);
                acts = (c & 16u) ? a_hi : a_lo;
                sdry2 += 32u;
                sneg2 = -32;
            }
        } else {
            if (neg2 == 0) {                                    // a quad shares its step counter: whole quads take this branch
                const uint32_t c = tc_in + (dry2 >> 1);
                if ((c & 16u) == 0u) refill(c >> 5);
                acts = (c & 16u) ? a_hi : a_lo;
                dry2 += 32u;
                neg2 = -32;
            }
        }
    };
    if (uni_wave) {
        int t = 0;
        for (; t + 4 <= T; t += 4) {                            // (four steps per trip: a quarter of the loop control)
            one_step(std::true_type{});
            one_step(std::true_type{});
            one_step(std::true_type{});
            one_step(std::true_type{});
        }
        for (; t < T; ++t) one_step(std::true_type{});
    } else {
        for (int t = 0; t < T; ++t) one_step(std::false_type{});
    }
    QUAD_STAMP(2);
    const uint32_t n_ep = wn >> 16, wins = wn & 0xffffu;
    const int done_last = (int)((marks & 0xffffu) >> 1), done_prev = (int)(marks >> 17), last_alive = (int)(marks & 1u);
    tc = tc_in + (uint32_t)T;
    const uint32_t ts_at_entry = ts;
    ts = n_ep ? (uint32_t)(T - done_last) : ts_at_entry + (uint32_t)T;
    const int last_len = (n_ep > 1u) ? done_last - done_prev : (int)ts_at_entry + done_last;
    k = a ? 0 : (k == 7 ? p + 1 : k);
    const int d = (d8 >> 3) & 3;
    // ---- epilogue: the junk dword hands this board's tag to its copier, boards LDS -> HBM without tags
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p == 0) *(lds_u8 *)(uintptr_t)(uint32_t)(mine + pad.junk) = (uint8_t)(tagbits >> OB);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr uint32_t OM = 0x01010101u * ((1u << OB) - 1u);
    constexpr uint32_t TM = 0x01010101u * ((1u << (8 - OB)) - 1u);
    int8_t *gslab = board + env0 * NN;
    if (rowsN) {
        // a row of N bytes back to byte R * N of the wave's run, whatever its alignment: the bytes up to the next dword
        // boundary one by one, then aligned dwords (the row's dwords funnelled by that many bytes), then the rest one by one --
        // every byte of the run is written by exactly one lane, no store touches a neighbour's bytes
        const bool flat_out = n_env == kWaveGames && (((uintptr_t)board & 15) == 0);
        uint32_t c[kRowsPerLane][5], trep[kRowsPerLane];
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {
            const int R = lane + k2 * CRL_WAVE;
            const int Rc = R < kWaveGames * N ? R : 0;
            const int e = (int)__umulhi((uint32_t)Rc, g.inv_n);
            const int sb = slab0 + e * pad.stride;
            const int a = sb + (Rc - N * e + 1) * RS;
            trep[k2] = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk);
#pragma unroll
            for (int q = 0; q < 5; ++q) c[k2][q] = *(const lds_u32 *)(uintptr_t)(uint32_t)(a + 4 * q);
        }
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {
            const int R = lane + k2 * CRL_WAVE;
            const uint32_t tr = trep[k2] * 0x01010101u;
            uint32_t w[6];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const uint32_t diff = ((c[k2][q] >> OB) & TM) ^ tr;
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                w[q] = c[k2][q] & OM & ~(stale * 0xffu);
            }
            w[5] = 0u;
            const int pre = (int)((4u - ((uint32_t)(R * N) & 3u)) & 3u);        // bytes up to the next dword boundary (N >= 4 > pre)
            const int nb = (N - pre) >> 2, suf = (N - pre) & 3;
            uint32_t sft[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) sft[q] = __builtin_amdgcn_alignbyte(w[q + 1], w[q], (uint32_t)pre);
            uint32_t tail = sft[0];
#pragma unroll
            for (int q = 0; q < 5; ++q) tail = (q == nb) ? sft[q] : tail;
            if (flat_out) {
                // (a full wave with aligned boards) the rows are packed into ONE flat image of the wave's 16 boards first, in
                // LDS, over the slabs -- every lane has read its rows by now --, where byte stores cost nothing; the image then
                // leaves in aligned 16-byte chunks.  Stored straight to HBM the rows' odd leading and trailing bytes made the
                // copy back 12 us of a 20-step launch at 19 x 19 (against 1.9 us at 20 x 20): partial-line writes.
                if (R < kWaveGames * N) {
                    const int fa = slab0 + R * N;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < pre) *(lds_u8 *)(uintptr_t)(uint32_t)(fa + j) = (uint8_t)((w[0] >> (8 * j)) & 0xffu);
#pragma unroll
                    for (int q = 0; q < 5; ++q)
                        if (q < nb) *(lds_u32 *)(uintptr_t)(uint32_t)(fa + pre + 4 * q) = sft[q];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if (j < suf) *(lds_u8 *)(uintptr_t)(uint32_t)(fa + pre + 4 * nb + j) = (uint8_t)((tail >> (8 * j)) & 0xffu);
                }
            } else if (R < rows_inN) {
                int8_t *dst = gslab + R * N;
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < pre) dst[j] = (int8_t)((w[0] >> (8 * j)) & 0xffu);
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    if (q < nb) *reinterpret_cast<uint32_t *>(dst + pre + 4 * q) = sft[q];
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    if (j < suf) dst[pre + 4 * nb + j] = (int8_t)((tail >> (8 * j)) & 0xffu);
            }
        }
        if (flat_out) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 7
            for (int i = lane; i < NN; i += CRL_WAVE)           // 16 boards = N * N chunks of 16 bytes
                *reinterpret_cast<crl_u32x4 *>(gslab + 16 * i) = *(const __attribute__((address_space(3))) crl_u32x4 *)(uintptr_t)(uint32_t)(slab0 + 16 * i);
        }
    } else if (rows20) {
        uint32_t c[kRowsPerLane][5], trep[kRowsPerLane];
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {             // all LDS reads in flight, then the arithmetic
            const int R = lane + k2 * CRL_WAVE;
            const int e = (int)__umulhi((uint32_t)R, g.inv_n);
            const int sb = slab0 + e * pad.stride;
            const int a = sb + (R - 20 * e + 1) * RS;
            trep[k2] = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk);
#pragma unroll
            for (int q = 0; q < 5; ++q) c[k2][q] = *(const lds_u32 *)(uintptr_t)(uint32_t)(a + 4 * q);
        }
#pragma unroll
        for (int k2 = 0; k2 < kRowsPerLane; ++k2) {
            const int R = lane + k2 * CRL_WAVE;
            const uint32_t tr = trep[k2] * 0x01010101u;
            uint32_t w[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const uint32_t diff = ((c[k2][q] >> OB) & TM) ^ tr;
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                w[q] = c[k2][q] & OM & ~(stale * 0xffu);
            }
            if (R < rows_in) {
                *reinterpret_cast<u32x4_a4 *>(gslab + R * 20) = (u32x4_a4){w[0], w[1], w[2], w[3]};
                *reinterpret_cast<uint32_t *>(gslab + R * 20 + 16) = w[4];
            }
        }
    } else if (wide) {
        const int bytes = n_env * NN;
#pragma unroll 7
        for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            const int sb = slab0 + e * pad.stride;
            const uint32_t trep = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk) * 0x01010101u;
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c4 = *(const lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]);
                const uint32_t diff = ((c4 >> OB) & TM) ^ trep;
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                w[q] = c4 & OM & ~(stale * 0xffu);
            }
            *reinterpret_cast<uint4 *>(gslab + off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int e = 0; e < n_env; ++e) {
            const uint32_t tb = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + pad.junk) << OB;
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                const uint32_t rawc = *(const lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + (y + 1) * RS + (c - y * N));
                gslab[(int64_t)e * NN + c] = (int8_t)((((rawc ^ tb) >> OB) == 0) ? (rawc & ((1u << OB) - 1u)) : 0u);
            }
        }
    }
    // ---- per-player state and statistics (my columns), per-game statistics (lane 0 of the quad)
    int lw = (last_alive & 1) << p;                             // winners of the latest terminal step, over the quad
    lw |= tron_quad<0xB1>(lw);
    lw |= tron_quad<0x4E>(lw);
    const int ret = 2 * (int)alive_steps - T + 9 * (int)wins;   // alive +1, dead -1, alive at a terminal step +10
    int32_t *row = st.results ? st.results + b * (3 + 2 * P) : nullptr;
    uint16_t *pk = st.packed ? st.packed + b * ((4 + P + 1) & ~1) : nullptr;        // kTronPackedRow, P a run-time value here
    if (pvalid) {
        const int rel = h - mine;
        const int rowi = rel / RS;                               // = y + 1
        heads[p * B + b] = (int16_t)((rowi - 1) * N + (rel - rowi * RS));
        dirs[p * B + b] = (int8_t)d;
        deaths[p * B + b] = (int8_t)k;
        const int rs = old_ret + ret;
        const uint32_t wc = old_wins + wins;
        st.ret_sum[p * B + b] = rs;
        st.win_count[p * B + b] = wc;
        if (row) { row[3 + p] = (int32_t)wc; row[3 + P + p] = rs; }
        if (pk) pk[4 + p] = (uint16_t)rs;
    }
    if (gvalid && p == 0) {
        const uint32_t ne = old_n_ep + n_ep, ls = old_len_sum + (ts_at_entry + (uint32_t)T - ts);
        st.tcount[b] = tc;
        st.tstep[b] = ts;
        st.n_episodes[b] = ne;
        st.len_sum[b] = ls;
        if (n_ep > 0) {
            st.last_winners[b] = (uint8_t)lw;
            st.last_len[b] = (uint16_t)last_len;
        }
        if (row) { row[0] = (int32_t)ne; row[1] = (int32_t)ls; row[2] = n_ep > 0 ? lw : (int32_t)old_last_w; }
        if (pk) { pk[0] = (uint16_t)ne; pk[1] = (uint16_t)ls; pk[2] = (uint16_t)(n_ep > 0 ? (uint32_t)lw : old_last_w); pk[3] = (uint16_t)ts; }
    }
#ifdef CRL_QUAD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the stores of this wave have left
#endif
    QUAD_STAMP(3);
}

// ---- two lanes per game: the byte-slab kernel for ONE or TWO players (round 5) ---------------------------------------------
// tron_rollout_quad_kernel gives every game four lanes; a two-player game leaves half of them idle (its launches cost what a
// four-player game's cost).  Here a wave holds 32 games, two lanes each: same slabs, tags, rolling row rewrite and copy loops, the
// DPP traffic of a quad reduced to one swap with the neighbour (target against the other head / the other target, alive count,
// winners), the fix-up on a DPP-gathered pair.  64 games per workgroup of 128 threads (34 KB of LDS at 20x20): four per CU,
// two waves per SIMD -- a step is then bound by its own dependent chain rather than by issue, and carries twice the games.
// The random stream (include/colosseum_hip.h): lane q of the pair computes Philox block 2 (c >> 4) + q once per 16 steps, turns
// it into the block's 8 code bytes, gathers player 0's and player 1's two bits of each (16 bits apiece), and the pair swaps
// halves: every lane ends up with its own player's 16 actions in one register.
__device__ __forceinline__ uint32_t tron_pair_actions(const uint32_t gid, const uint32_t group16, const int p, const uint32_t seed_lo,
                                                      const uint32_t seed_hi, const uint8_t *act_lut)
{
    uint32_t k0 = seed_lo, k1 = seed_hi;
    asm volatile("" : "+s"(k0), "+s"(k1));
    const philox_out r = philox4x32_10<true>(gid, 2u * group16 + (uint32_t)p, 0u, CRL_TAG_TRON, k0, k1);
    uint32_t code[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t frac, first;
        crl_mul_wide<true>(81u, r.w[i], frac, first);
        code[2 * i] = act_lut[first];
        code[2 * i + 1] = act_lut[__umulhi(frac, 81u)];
    }
    const uint32_t w0 = code[0] | code[1] << 8 | code[2] << 16 | code[3] << 24;    // steps 0-3 of my block: a code byte each
    const uint32_t w1 = code[4] | code[5] << 8 | code[6] << 16 | code[7] << 24;    // steps 4-7
    auto gather = [](const uint32_t w, const int q) -> uint32_t {      // player q's two bits of four code bytes -> one byte
        const uint32_t v = (w >> (2 * q)) & 0x03030303u;
        const uint32_t t = v | (v >> 6);
        return (t | (t >> 12)) & 0xffu;
    };
    const uint32_t mine0 = gather(w0, 0) | gather(w1, 0) << 8;          // player 0's eight actions of my block
    const uint32_t mine1 = gather(w0, 1) | gather(w1, 1) << 8;          // player 1's
    // (both swaps run in both lanes, THEN the lane picks: see tron_quad_actions)
    const uint32_t other0 = (uint32_t)tron_quad<0xB1>((int)mine0), other1 = (uint32_t)tron_quad<0xB1>((int)mine1);
    // lane 0 holds the even block (steps 0-7 of the group), lane 1 the odd one (steps 8-15)
    return p == 0 ? (mine0 | other0 << 16) : (other1 | mine1 << 16);
}

template <int RS>
__global__ void __launch_bounds__(128, 6)
tron_rollout_pair_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const int64_t B,
                         const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                         int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                         int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    constexpr int kGames = 64, kWaveGames = 32;
    constexpr int kRowDwords = RS / 4;
    constexpr uint32_t step4 = (uint32_t)((-RS) & 0xff) | (1u << 8) | ((uint32_t)RS << 16) | (0xffu << 24);
    constexpr int OB = 3;
    constexpr uint32_t kTags = (1u << (8 - OB)) - 1u;           // the all-ones tag is never used: 0xff stays "wall"
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN, P = cfg.P;                    // P <= 2
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int p = lane & 1;                                     // my player
    const int slot = threadIdx.x >> 1;                          // my game's slab in this workgroup
    const int64_t b = (int64_t)blockIdx.x * kGames + slot;
    const bool gvalid = b < B;
    const bool pvalid = gvalid && p < P;
    const int64_t bb = gvalid ? b : 0;
    const int64_t env0 = (int64_t)blockIdx.x * kGames + wave * kWaveGames;   // first game of this wave
    const int n_env = (int)((B - env0) < kWaveGames ? (B - env0 > 0 ? B - env0 : 0) : kWaveGames);
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;
    const int mine = lds0 + slot * pad.stride;
    const int slab0 = lds0 + wave * kWaveGames * pad.stride;
    const bool wide = (N & 3) == 0 && N >= 8;
    const int8_t *gslab_in = board + (n_env > 0 ? env0 : 0) * NN;
    const int bytes_in = n_env * NN;
    const int64_t pb = (int64_t)(p < P ? p : 0) * B + bb;
    const int h_in = heads[pb];
    const int d_in = dirs[pb];
    int k = deaths[pb];
    const int old_ret = st.ret_sum[pb];
    const uint32_t old_wins = st.win_count[pb];
    uint32_t tc = st.tcount[bb], ts = st.tstep[bb];
    const uint32_t old_n_ep = st.n_episodes[bb], old_len_sum = st.len_sum[bb];
    const uint32_t old_last_w = st.last_winners[bb];
    k = pvalid ? k : 1;
    // walls everywhere (each lane half of its game's slab), then the cells
    for (int off = 4 * p; off < pad.stride; off += 8) *(lds_u32 *)(uintptr_t)(uint32_t)(mine + off) = 0xffffffffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wide) {
#pragma unroll 4
        for (int off = lane * 16; off < bytes_in; off += CRL_WAVE * 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(gslab_in + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            const int sb = slab0 + e * pad.stride;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]) = w[q];
        }
    } else {
        for (int i = lane; i < bytes_in; i += CRL_WAVE) {
            const int e = (int)__umulhi((uint32_t)i, pad.inv_nn);
            const int c = i - e * NN;
            const int y = (int)__umulhi((uint32_t)c, g.inv_n);
            *(lds_u8 *)(uintptr_t)(uint32_t)(slab0 + e * pad.stride + (y + 1) * RS + (c - y * N)) = (uint8_t)gslab_in[i];
        }
    }
    // ---- my player: head as an LDS address, the start layout for resets
    const int junk = mine + pad.junk + p;
    int fh = cfg.start_heads[0], fd = cfg.start_dirs[0];
    fh = (p == 1) ? cfg.start_heads[1] : fh; fd = (p == 1) ? cfg.start_dirs[1] : fd;
    const int fy = (int)__umulhi((uint32_t)(fh < 0 ? 0 : fh), g.inv_n);
    const int fresh_h = (p < P) ? mine + (fy + 1) * RS + (fh - fy * N) : junk;
    const int fresh_d8 = fd << 3;
    const int fresh_a = (p < P) ? 1 : 0;
    int h, d8 = (d_in & 3) << 3;
    {
        const int hc = min(max(h_in, 0), NN - 1);
        const int y = (int)__umulhi((uint32_t)hc, g.inv_n);
        h = pvalid ? mine + (y + 1) * RS + (hc - y * N) : junk;
    }
    // my share of a fresh row: three of its six dwords
    static_assert(kRowDwords == 6, "the row rewrite of the pair kernel shares out six dwords");
    uint32_t rp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = 3 * p + i;
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) w |= (4 * j + c < N) ? 0u : (0xffu << (8 * c));
        rp[i] = w;
    }
    int sweep = mine + RS + 12 * p;
    const int sweep_end = mine + (N + 1) * RS + 12 * p, sweep_first = mine + RS + 12 * p;
    uint32_t tagbits = 0, stamp = (uint32_t)(p + 1);
    uint32_t alive_steps = 0, wn = 0, marks = 0;
    if (!gvalid) k = 1;
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);
    __syncthreads();                                            // action table; slabs written by other lanes
    uint32_t acts = tron_pair_actions(gid, tc >> 4, p, seed_lo, seed_hi, act_lut) >> ((tc & 15u) * 2u);
    uint32_t dry2 = 32u - 2u * (tc & 15u);
    int neg2 = -(int)dry2;
    const uint32_t tc_in = tc;
    int a = (k == 0) ? 1 : 0;
    for (int t = 0; t < T; ++t) {
        const bool run = a != 0;
        const int dir8 = (int)((acts << 3) + (uint32_t)d8);
        const int tgt = h + __builtin_amdgcn_sbfe((int)step4, dir8, 8);
        const int tq = run ? tgt : junk;
        uint32_t raw;
        CRL_BOUNDS_IN(tq, mine, mine + pad.stride, 143);
        asm volatile("ds_read_u8 %0, %1" : "=v"(raw) : "v"(tq) : "memory");
        const int x1 = tq ^ tron_quad<0xB1>(h), y1 = tq ^ tron_quad<0xB1>(tq);
        uint32_t near = min((uint32_t)x1, (uint32_t)y1);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw), "+v"(near) : : "memory");
        const uint32_t m = min(raw ^ tagbits, raw ^ 0xf8u);
        const bool dead = run & ((m - 1u) < 7u);
        const bool moved = run ^ dead;
        const int h_was = h;
        k = dead ? (int)m : k;
        d8 = run ? dir8 : d8;
        h = moved ? tgt : h;
        CRL_BOUNDS_IN(h, mine, mine + pad.stride, 144);
        *(lds_u8 *)(uintptr_t)(uint32_t)h = (uint8_t)stamp;
        bool alive_now = moved;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(near == 0u) != 0ull, 0)) {
            if (moved) *(lds_u8 *)(uintptr_t)(uint32_t)tgt = (uint8_t)0;
            TronRegs<2> s;
            TronProbe<2> pr;
            uint32_t stamp2[2];
            const int h0 = h_was;
            const int d0 = (run ? (dir8 - (int)(acts << 3)) >> 3 : d8 >> 3) & 3, dir = (dir8 >> 3) & 3, kk = run ? 0 : k;
            s.h[0] = tron_quad<0xA0>(h0); s.h[1] = tron_quad<0xF5>(h0);
            s.d[0] = tron_quad<0xA0>(d0); s.d[1] = tron_quad<0xF5>(d0);
            s.k[0] = tron_quad<0xA0>(kk); s.k[1] = tron_quad<0xF5>(kk);
            pr.tgt[0] = tron_quad<0xA0>(tgt); pr.tgt[1] = tron_quad<0xF5>(tgt);
            pr.raw[0] = tron_quad<0xA0>((int)raw); pr.raw[1] = tron_quad<0xF5>((int)raw);
            pr.ndir[0] = tron_quad<0xA0>(dir); pr.ndir[1] = tron_quad<0xF5>(dir);
            stamp2[0] = tagbits | 1u; stamp2[1] = tagbits | 2u;
            LdsBoard<OB> bd{tagbits};
            bd.within(mine, mine + pad.stride);
            tron_resolve_lds<2>(bd, s, pr, stamp2, junk);
            const int hS = p ? s.h[1] : s.h[0], dS = p ? s.d[1] : s.d[0], kS = p ? s.k[1] : s.k[0];
            h = (p < P) ? hS : junk;
            d8 = dS << 3;
            k = kS;
            alive_now = kS == 0;
        }
        a = alive_now ? 1 : 0;
        const int alive = a + tron_quad<0xB1>(a);
        alive_steps += (uint32_t)a;
        neg2 += 2;
        if (alive <= 1) {                                       // (a game beyond the batch is "over" at every step)
            tagbits += 1u << OB;
            tagbits = (tagbits == (kTags << OB)) ? 0u : tagbits;
            stamp = tagbits | (uint32_t)(p + 1);
            *(lds_u32 *)(uintptr_t)(uint32_t)sweep = rp[0];
            *(lds_u32 *)(uintptr_t)(uint32_t)(sweep + 4) = rp[1];
            *(lds_u32 *)(uintptr_t)(uint32_t)(sweep + 8) = rp[2];
            sweep += RS;
            sweep = (sweep == sweep_end) ? sweep_first : sweep;
            *(lds_u8 *)(uintptr_t)(uint32_t)fresh_h = (uint8_t)stamp;
            asm("v_add3_u32 %0, %0, %1, %2" : "+v"(wn) : "v"(a), "s"(0x10000u));
            asm("v_lshl_add_u32 %0, %0, 16, %1\n\tv_add3_u32 %0, %0, %2, %3" : "+v"(marks) : "v"(dry2), "v"(neg2), "v"(a));
            h = fresh_h; d8 = fresh_d8;
            a = fresh_a;
        }
        acts >>= 2;
        if (neg2 == 0) {                                        // a pair shares its step counter: whole pairs take this branch
            const uint32_t c = tc_in + (dry2 >> 1);             // the step the new actions are for (a multiple of 16)
            acts = tron_pair_actions(gid, c >> 4, p, seed_lo, seed_hi, act_lut);
            dry2 += 32u;
            neg2 = -32;
        }
    }
    const uint32_t n_ep = wn >> 16, wins = wn & 0xffffu;
    const int done_last = (int)((marks & 0xffffu) >> 1), done_prev = (int)(marks >> 17), last_alive = (int)(marks & 1u);
    tc = tc_in + (uint32_t)T;
    const uint32_t ts_at_entry = ts;
    ts = n_ep ? (uint32_t)(T - done_last) : ts_at_entry + (uint32_t)T;
    const int last_len = (n_ep > 1u) ? done_last - done_prev : (int)ts_at_entry + done_last;
    k = a ? 0 : (k == 7 ? p + 1 : k);
    const int d = (d8 >> 3) & 3;
    // ---- epilogue: the junk dword hands this board's tag to its copier, boards LDS -> HBM without tags
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (p == 0) *(lds_u8 *)(uintptr_t)(uint32_t)(mine + pad.junk) = (uint8_t)(tagbits >> OB);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr uint32_t OM = 0x01010101u * ((1u << OB) - 1u);
    constexpr uint32_t TM = 0x01010101u * ((1u << (8 - OB)) - 1u);
    int8_t *gslab = board + env0 * NN;
    if (wide) {
        const int bytes = n_env * NN;
#pragma unroll 4
        for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            const int sb = slab0 + e * pad.stride;
            const uint32_t trep = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk) * 0x01010101u;
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t c4 = *(const lds_u32 *)(uintptr_t)(uint32_t)(sb + o[q]);
                const uint32_t diff = ((c4 >> OB) & TM) ^ trep;
                const uint32_t stale = ((diff + 0x7f7f7f7fu) >> 7) & 0x01010101u;
                w[q] = c4 & OM & ~(stale * 0xffu);
            }
            *reinterpret_cast<uint4 *>(gslab + off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int i = lane; i < n_env * NN; i += CRL_WAVE) {
            const int e = (int)__umulhi((uint32_t)i, pad.inv_nn);
            const int c = i - e * NN;
            const int y = (int)__umulhi((uint32_t)c, g.inv_n);
            const int sb = slab0 + e * pad.stride;
            const uint32_t tb = (uint32_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(sb + pad.junk) << OB;
            const uint32_t rawc = *(const lds_u8 *)(uintptr_t)(uint32_t)(sb + (y + 1) * RS + (c - y * N));
            gslab[i] = (int8_t)((((rawc ^ tb) >> OB) == 0) ? (rawc & ((1u << OB) - 1u)) : 0u);
        }
    }
    // ---- per-player state and statistics (my columns), per-game statistics (lane 0 of the pair)
    int lw = (last_alive & 1) << p;
    lw |= tron_quad<0xB1>(lw);
    const int ret = 2 * (int)alive_steps - T + 9 * (int)wins;
    int32_t *row = st.results ? st.results + b * (3 + 2 * P) : nullptr;
    uint16_t *pk = st.packed ? st.packed + b * ((4 + P + 1) & ~1) : nullptr;
    if (pvalid) {
        const int rel = h - mine;
        const int rowi = rel / RS;
        heads[p * B + b] = (int16_t)((rowi - 1) * N + (rel - rowi * RS));
        dirs[p * B + b] = (int8_t)d;
        deaths[p * B + b] = (int8_t)k;
        const int rs = old_ret + ret;
        const uint32_t wc = old_wins + wins;
        st.ret_sum[p * B + b] = rs;
        st.win_count[p * B + b] = wc;
        if (row) { row[3 + p] = (int32_t)wc; row[3 + P + p] = rs; }
        if (pk) pk[4 + p] = (uint16_t)rs;
    }
    if (gvalid && p == 0) {
        const uint32_t ne = old_n_ep + n_ep, ls = old_len_sum + (ts_at_entry + (uint32_t)T - ts);
        st.tcount[b] = tc;
        st.tstep[b] = ts;
        st.n_episodes[b] = ne;
        st.len_sum[b] = ls;
        if (n_ep > 0) {
            st.last_winners[b] = (uint8_t)lw;
            st.last_len[b] = (uint16_t)last_len;
        }
        if (row) { row[0] = (int32_t)ne; row[1] = (int32_t)ls; row[2] = n_ep > 0 ? lw : (int32_t)old_last_w; }
        if (pk) { pk[0] = (uint16_t)ne; pk[1] = (uint16_t)ls; pk[2] = (uint16_t)(n_ep > 0 ? (uint32_t)lw : old_last_w); pk[3] = (uint16_t)ts; }
    }
}

// ---- one lane per player on boards in GLOBAL memory (any board size, P <= 4) --------------------------------------
// What tron_rollout_quad_kernel is to the byte-slab kernel, this is to tron_rollout_kernel: a lane plays ONE player -- one
// byte probe, die or move, one byte store per step --, the quad shares what is per game (alive count by two DPP adds, the
// random stream of tron_quad_actions, the reset), and the reference's sequential order (CyTronGrid.pyx:15-62) is resolved,
// redundantly in the quad's four lanes on DPP-gathered copies, only in the wave-steps where a target meets another player's
// head or target.  65,536 games are 4,096 waves (4 per SIMD) instead of 1,024: a step is bound by the latency of its probe,
// and four times the waves hide four times as much of it.  No tags and no pass over the boards at either end of the launch:
// a game that ends is rewritten with the start layout by the WHOLE wave (16-byte stores between the board's unaligned ends),
// so a launch costs what its steps cost -- which is what short launches on boards that do not fit the byte slabs want
// (crl_tron_rollout: boards above 20x20 up to ~96 steps, boards above 40x40 always; profiles/r5_shape_sweep.txt).
// Cross-lane traffic through memory (a cell one lane stamps is probed by another lane of the wave in a later step; a reset's
// stores come from all lanes) relies on what the lane-per-game kernel relies on: one wave's vector memory operations reach
// its CU's cache in program order.
template <int P>
__global__ void __launch_bounds__(256)
tron_rollout_gquad_kernel(const crl_tron_cfg cfg, const TronGeom g, const int64_t B,
                          const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                          int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                          int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    static_assert(P <= 4, "one lane per player, four lanes per game");
    constexpr int kGames = 64, kWaveGames = 16;
    __shared__ uint8_t act_lut[84];
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN;
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int p = lane & 3;                                     // my player
    const int slot = threadIdx.x >> 2;
    const int64_t b = (int64_t)blockIdx.x * kGames + slot;
    const bool gvalid = b < B;
    const bool seat = p < P;                                    // this lane has a player at all
    const bool pvalid = gvalid && seat;
    const int64_t bb = gvalid ? b : 0;
    const int64_t env0 = (int64_t)blockIdx.x * kGames + wave * kWaveGames;   // first game of this wave
    uint8_t *gb = reinterpret_cast<uint8_t *>(board) + bb * NN;  // my game's board
    // (unconditional loads from a clamped index, as in the byte-slab kernel)
    const int64_t pb = (int64_t)(seat ? p : 0) * B + bb;
    const int h_in = heads[pb];
    const int d_in = dirs[pb];
    int k = deaths[pb];
    const int old_ret = st.ret_sum[pb];
    const uint32_t old_wins = st.win_count[pb];
    uint32_t tc = st.tcount[bb], ts = st.tstep[bb];
    const uint32_t old_n_ep = st.n_episodes[bb], old_len_sum = st.len_sum[bb];
    const uint32_t old_last_w = st.last_winners[bb];
    k = pvalid ? k : 1;                                         // a seat without a player counts as dead for good
    int fh = cfg.start_heads[0], fd = cfg.start_dirs[0];
    fh = (p == 1) ? cfg.start_heads[1] : fh; fd = (p == 1) ? cfg.start_dirs[1] : fd;
    fh = (p == 2) ? cfg.start_heads[2] : fh; fd = (p == 2) ? cfg.start_dirs[2] : fd;
    fh = (p == 3) ? cfg.start_heads[3] : fh; fd = (p == 3) ? cfg.start_dirs[3] : fd;
    fh = seat ? fh : 0;
    const int fy = (int)__umulhi((uint32_t)fh, g.inv_n), fx = fh - fy * N;
    const int fresh_a = pvalid ? 1 : 0;
    // a head that is not a cell of the board never equals a target: seats without a player carry one of their own
    const int no_head = -6 - p, no_tgt = -2 - p;
    int h = seat ? min(max(h_in, 0), NN - 1) : no_head;
    int hy = seat ? (int)__umulhi((uint32_t)h, g.inv_n) : 0, hx = seat ? h - hy * N : 0;
    int d = d_in & 3;
    uint32_t alive_steps = 0, wn = 0, marks = 0;                // as in tron_rollout_quad_kernel
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);
    __syncthreads();                                            // action table
    uint32_t a_lo = 0, a_hi = 0;
    auto refill = [&](const uint32_t group) { tron_quad_actions(gid, group, p, seed_lo, seed_hi, act_lut, a_lo, a_hi); };
    refill(tc >> 5);
    uint32_t acts = ((tc & 16u) ? a_hi : a_lo) >> ((tc & 15u) * 2u);   // bits 1:0 = this step's action code (0, 1, 3)
    uint32_t dry2 = 32u - 2u * (tc & 15u);                      // `acts` runs dry after dry2 / 2 steps of this launch
    int neg2 = -(int)dry2;
    const uint32_t tc_in = tc;
    int a = (k == 0) ? 1 : 0;
    for (int t = 0; t < T; ++t) {
        const bool run = a != 0;
        const int dir = (d + (int)(acts & 3u)) & 3;
        const int sh8 = dir << 3;
        const int nx = hx + __builtin_amdgcn_sbfe((int)0xff000100u, sh8, 8);
        const int ny = hy + __builtin_amdgcn_sbfe((int)0x000100ffu, sh8, 8);
        const bool oob = ((unsigned)nx >= (unsigned)N) | ((unsigned)ny >= (unsigned)N);
        const int tgt = (int)(__umul24((unsigned)ny, (unsigned)N) + (unsigned)nx);
        const bool look = run & !oob;
        CRL_BOUNDS_LT(look ? tgt : 0, NN, 138);                 // the probe stays on my game's board
        const int raw = gb[look ? tgt : 0];                     // (a player that does not look reads cell 0 and ignores it)
        // does anything in this wave need the reference's sequential order?  (tron_rollout_quad_kernel, on cell indices)
        const int tq = look ? tgt : no_tgt;
        const int x1 = tq ^ tron_quad<0x39>(h), x2 = tq ^ tron_quad<0x4E>(h), x3 = tq ^ tron_quad<0x93>(h);
        const int y1 = tq ^ tron_quad<0x39>(tq), y2 = tq ^ tron_quad<0x4E>(tq);
        const uint32_t near = min(min(min(min((uint32_t)x1, (uint32_t)x2), (uint32_t)x3), (uint32_t)y1), (uint32_t)y2);
        // the common path: each player on its own
        const int h0 = h, hx0 = hx, hy0 = hy, d0 = d, k0 = k;     // (k == 0 exactly while my player is alive)
        bool dead = run & (oob | (raw != 0));                   // :47-57
        bool moved = run ^ dead;                                // :60-62
        k = dead ? (oob ? p + 1 : raw) : k;
        d = run ? dir : d;                                      // :44
        h = moved ? tgt : h; hx = moved ? nx : hx; hy = moved ? ny : hy;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(near == 0u) != 0ull, 0)) {
            // rare: the quad's four players in the reference's order from the pre-step state, redundantly in its four lanes
            TronRegs<4> s;
            TronProbe<4> pr;
            const int oobi = oob ? 1 : 0;
            s.h[0] = tron_quad<0x00>(h0); s.h[1] = tron_quad<0x55>(h0); s.h[2] = tron_quad<0xAA>(h0); s.h[3] = tron_quad<0xFF>(h0);
            s.x[0] = tron_quad<0x00>(hx0); s.x[1] = tron_quad<0x55>(hx0); s.x[2] = tron_quad<0xAA>(hx0); s.x[3] = tron_quad<0xFF>(hx0);
            s.y[0] = tron_quad<0x00>(hy0); s.y[1] = tron_quad<0x55>(hy0); s.y[2] = tron_quad<0xAA>(hy0); s.y[3] = tron_quad<0xFF>(hy0);
            s.d[0] = tron_quad<0x00>(d0); s.d[1] = tron_quad<0x55>(d0); s.d[2] = tron_quad<0xAA>(d0); s.d[3] = tron_quad<0xFF>(d0);
            s.k[0] = tron_quad<0x00>(k0); s.k[1] = tron_quad<0x55>(k0); s.k[2] = tron_quad<0xAA>(k0); s.k[3] = tron_quad<0xFF>(k0);
            pr.tgt[0] = tron_quad<0x00>(tgt); pr.tgt[1] = tron_quad<0x55>(tgt); pr.tgt[2] = tron_quad<0xAA>(tgt); pr.tgt[3] = tron_quad<0xFF>(tgt);
            pr.raw[0] = tron_quad<0x00>(raw); pr.raw[1] = tron_quad<0x55>(raw); pr.raw[2] = tron_quad<0xAA>(raw); pr.raw[3] = tron_quad<0xFF>(raw);
            pr.ndir[0] = tron_quad<0x00>(dir); pr.ndir[1] = tron_quad<0x55>(dir); pr.ndir[2] = tron_quad<0xAA>(dir); pr.ndir[3] = tron_quad<0xFF>(dir);
            pr.nx[0] = tron_quad<0x00>(nx); pr.nx[1] = tron_quad<0x55>(nx); pr.nx[2] = tron_quad<0xAA>(nx); pr.nx[3] = tron_quad<0xFF>(nx);
            pr.ny[0] = tron_quad<0x00>(ny); pr.ny[1] = tron_quad<0x55>(ny); pr.ny[2] = tron_quad<0xAA>(ny); pr.ny[3] = tron_quad<0xFF>(ny);
            const int o0 = tron_quad<0x00>(oobi), o1 = tron_quad<0x55>(oobi), o2 = tron_quad<0xAA>(oobi), o3 = tron_quad<0xFF>(oobi);
            pr.oob[0] = o0 != 0; pr.oob[1] = o1 != 0; pr.oob[2] = o2 != 0; pr.oob[3] = o3 != 0;
            int rew4[4], term4, wm4;
            const PlainBoard nowhere{nullptr CRL_CELLS_INIT(NN)};
            tron_resolve<4>(nowhere, false, s, pr, rew4, term4, wm4);   // valid = false: no stores here (below, with everybody's)
            int hS = s.h[0], xS = s.x[0], yS = s.y[0], dS = s.d[0], kS = s.k[0];
            hS = (p == 1) ? s.h[1] : hS; xS = (p == 1) ? s.x[1] : xS; yS = (p == 1) ? s.y[1] : yS; dS = (p == 1) ? s.d[1] : dS; kS = (p == 1) ? s.k[1] : kS;
            hS = (p == 2) ? s.h[2] : hS; xS = (p == 2) ? s.x[2] : xS; yS = (p == 2) ? s.y[2] : yS; dS = (p == 2) ? s.d[2] : dS; kS = (p == 2) ? s.k[2] : kS;
            hS = (p == 3) ? s.h[3] : hS; xS = (p == 3) ? s.x[3] : xS; yS = (p == 3) ? s.y[3] : yS; dS = (p == 3) ? s.d[3] : dS; kS = (p == 3) ? s.k[3] : kS;
            moved = hS != h0;                                   // (a head is never its owner's own target)
            h = hS; hx = xS; hy = yS; d = dS;
            k = kS;                                             // (also a dead player's entry, overwritten by a head-on killer: :56-57)
        }
        // TronGridEnvironment.py:309-321 for the game: alive players over the quad
        a = (run && k == 0) ? 1 : 0;
        int alive = a + tron_quad<0xB1>(a);
        alive += tron_quad<0x4E>(alive);
        alive_steps += (uint32_t)a;
        neg2 += 2;
        const bool over = alive <= 1;                           // (a game beyond the batch is "over" at every step)
        // the trail -- unless the game ends with this step: its board is about to be rewritten as a whole
        if (moved && !over && gvalid) {
            CRL_BOUNDS_LT(h, NN, 139);                          // ... and so does the trail store
            gb[h] = (uint8_t)(p + 1);
        }
        if (over) {
            asm("v_add3_u32 %0, %0, %1, %2" : "+v"(wn) : "v"(a), "s"(0x10000u));
            asm("v_lshl_add_u32 %0, %0, 16, %1\n\tv_add3_u32 %0, %0, %2, %3" : "+v"(marks) : "v"(dry2), "v"(neg2), "v"(a));
            h = seat ? fh : no_head; hx = fx; hy = fy; d = fd;
            k = pvalid ? 0 : 1;
            a = fresh_a;
        }
        // new_state for the games of this wave that ended: the whole wave writes each one's start board
        uint64_t ending = __builtin_amdgcn_ballot_w64(over && gvalid && p == 0);
        while (ending) {
            const int l = (int)__builtin_ctzll(ending);
            ending &= ending - 1;
            uint8_t *eb = reinterpret_cast<uint8_t *>(board) + (env0 + (l >> 2)) * NN;
            const int lead = min((int)((16u - (uint32_t)((uintptr_t)eb & 15u)) & 15u), NN);
            const int chunks = (NN - lead) >> 4, tail0 = lead + (chunks << 4);
            for (int c = lane; c < chunks; c += CRL_WAVE)
                *reinterpret_cast<uint4 *>(eb + lead + (c << 4)) = tron_fresh_chunk16<P>(cfg, lead + (c << 4));
            const int odd = lane < lead ? lane : (lane - lead < NN - tail0 ? tail0 + lane - lead : -1);   // (lead + tail < 31 bytes)
            if (odd >= 0) {
                int v = 0;
#pragma unroll
                for (int q = 0; q < P; ++q) v = (cfg.start_heads[q] == odd) ? q + 1 : v;
                eb[odd] = (uint8_t)v;
            }
        }
        acts >>= 2;
        if (neg2 == 0) {                                        // a quad shares its step counter: whole quads take this branch
            const uint32_t c = tc_in + (dry2 >> 1);             // the step the new actions are for
            if ((c & 16u) == 0u) refill(c >> 5);
            acts = (c & 16u) ? a_hi : a_lo;
            dry2 += 32u;
            neg2 = -32;
        }
    }
    // ---- per-player state and statistics (my columns), per-game statistics (lane 0 of the quad)
    const uint32_t n_ep = wn >> 16, wins = wn & 0xffffu;
    const int done_last = (int)((marks & 0xffffu) >> 1), done_prev = (int)(marks >> 17), last_alive = (int)(marks & 1u);
    tc = tc_in + (uint32_t)T;
    const uint32_t ts_at_entry = ts;
    ts = n_ep ? (uint32_t)(T - done_last) : ts_at_entry + (uint32_t)T;
    const int last_len = (n_ep > 1u) ? done_last - done_prev : (int)ts_at_entry + done_last;
    int lw = (last_alive & 1) << p;
    lw |= tron_quad<0xB1>(lw);
    lw |= tron_quad<0x4E>(lw);
    const int ret = 2 * (int)alive_steps - T + 9 * (int)wins;   // alive +1, dead -1, alive at a terminal step +10
    int32_t *row = st.results ? st.results + b * (3 + 2 * P) : nullptr;
    uint16_t *pk = st.packed ? st.packed + b * kTronPackedRow<P> : nullptr;
    if (pvalid) {
        heads[p * B + b] = (int16_t)h;
        dirs[p * B + b] = (int8_t)d;
        deaths[p * B + b] = (int8_t)k;
        const int rs = old_ret + ret;
        const uint32_t wc = old_wins + wins;
        st.ret_sum[p * B + b] = rs;
        st.win_count[p * B + b] = wc;
        if (row) { row[3 + p] = (int32_t)wc; row[3 + P + p] = rs; }
        if (pk) pk[4 + p] = (uint16_t)rs;
    }
    if (gvalid && p == 0) {
        const uint32_t ne = old_n_ep + n_ep, ls = old_len_sum + (ts_at_entry + (uint32_t)T - ts);
        st.tcount[b] = tc;
        st.tstep[b] = ts;
        st.n_episodes[b] = ne;
        st.len_sum[b] = ls;
        if (n_ep > 0) {
            st.last_winners[b] = (uint8_t)lw;
            st.last_len[b] = (uint16_t)last_len;
        }
        if (row) { row[0] = (int32_t)ne; row[1] = (int32_t)ls; row[2] = n_ep > 0 ? lw : (int32_t)old_last_w; }
        if (pk) { pk[0] = (uint16_t)ne; pk[1] = (uint16_t)ls; pk[2] = (uint16_t)(n_ep > 0 ? (uint32_t)lw : old_last_w); pk[3] = (uint16_t)ts; }
    }
}

// ---- replay of unfinished episodes on byte slabs (epilogue of both bitboard kernels) --------------------------------
// The bitboard kernels do not know who owns a cell; the state they hand back is rebuilt by replaying each game's unfinished
// episode with the byte-slab stepper (same tron_resolve_lds, same random stream), from the start layout or -- for a game
// that was not reset during the launch -- from the state the launch came in with.  The pieces below work on `n_slabs` byte
// slabs laid out from LDS address lds0 (pad.stride apart), which hold the games [gbase, gbase + n_games).

// (1) fresh boards in every slab: walls, empty cells.  Called by `nthreads` threads (tid = 0 .. nthreads - 1), a multiple of
// n_slabs; the caller synchronises the workgroup afterwards.
template <int RS>
__device__ __forceinline__ void tron_replay_fresh_slabs(const int lds0, const TronPad &pad, const int N, const int n_slabs,
                                                        const int tid, const int nthreads)
{
    // a thread takes one slab and every `parts`-th dword COLUMN of it: the value of a column is the same in all N board rows
    // (cells 0, walls 0xff behind column N - 1), so it is computed once and stored down the rows -- two instructions per
    // dword (round 5: a division and the row / column tests per dword made this 3 us of a one-wave workgroup's replay)
    constexpr int kRowDwords = RS / 4;
    const int parts = nthreads / n_slabs, slab = tid % n_slabs;
    const int base = lds0 + slab * pad.stride, sd = pad.stride >> 2;
    for (int j = tid / n_slabs; j < kRowDwords; j += parts) {
        const int left = N - 4 * j;                             // cells from this dword to the row's end
        const uint32_t v = (left >= 4) ? 0u : (left <= 0 ? 0xffffffffu : 0xffffffffu << (8 * left));
        int a = base + 4 * j;
        *(lds_u32 *)(uintptr_t)(uint32_t)a = 0xffffffffu;       // row 0: wall
        for (int row = 1; row <= N; ++row) {
            a += RS;
            *(lds_u32 *)(uintptr_t)(uint32_t)a = v;
        }
        *(lds_u32 *)(uintptr_t)(uint32_t)(a + RS) = 0xffffffffu;    // row N + 1: wall
    }
    // what lies behind row N + 1 (the junk dword, padding)
    for (int d = (N + 2) * kRowDwords + tid / n_slabs; d < sd; d += parts)
        *(lds_u32 *)(uintptr_t)(uint32_t)(base + 4 * d) = 0xffffffffu;
}

// the P start heads of new_state stamped into the slab at `base`
template <int P, int RS>
__device__ __forceinline__ void tron_replay_stamp_heads(const crl_tron_cfg &cfg, const TronGeom &g, const int base)
{
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const int sh = cfg.start_heads[q];
        const int sy = (int)__umulhi((uint32_t)sh, g.inv_n);
        *(lds_u8 *)(uintptr_t)(uint32_t)(base + (sy + 1) * RS + (sh - sy * g.N)) = (uint8_t)(q + 1);
    }
}

// (2a) one wave copies the n_games incoming boards at `gslab` into the slabs (for the games that resume from them)
template <int RS>
__device__ __forceinline__ void tron_replay_copy_in(const int8_t *__restrict__ gslab, const int n_games, const int lds0,
                                                    const TronPad &pad, const TronGeom &g, const int lane)
{
    const int N = g.N, NN = g.NN;
    if ((N & 3) == 0 && N >= 8) {
        const int bytes = n_games * NN;
        for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {
            const uint4 v = *reinterpret_cast<const uint4 *>(gslab + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            const int cq = (off - e * NN) >> 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y = (int)__umulhi((uint32_t)(cq + q), pad.inv_nq);
                *(lds_u32 *)(uintptr_t)(uint32_t)(lds0 + e * pad.stride + (y + 1) * RS + 4 * (cq + q - y * (N >> 2))) = w[q];
            }
        }
    } else {
        // (rows that are not whole dwords: cell by cell; aligned 16-byte chunks dealt out cell by cell were tried in round 5 and
        //  change nothing measurable -- this copy runs only in waves that hold a game the launch did not reset)
        for (int e = 0; e < n_games; ++e)
            for (int c = lane; c < NN; c += CRL_WAVE) {
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                *(lds_u8 *)(uintptr_t)(uint32_t)(lds0 + e * pad.stride + (y + 1) * RS + (c - y * N)) = (uint8_t)gslab[(int64_t)e * NN + c];
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a lane's own slab back to the start layout (after tron_replay_copy_in overwrote it with an incoming board)
template <int P, int RS>
__device__ __forceinline__ void tron_replay_refresh_slab(const crl_tron_cfg &cfg, const TronGeom &g, const int bmine)
{
    constexpr int kRowDwords = RS / 4;
    for (int y = 0; y < g.N; ++y)
#pragma unroll
        for (int j = 0; j < kRowDwords; ++j) {
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) w |= (4 * j + k < g.N) ? 0u : (0xffu << (8 * k));
            *(lds_u32 *)(uintptr_t)(uint32_t)(bmine + (y + 1) * RS + 4 * j) = w;
        }
    tron_replay_stamp_heads<P, RS>(cfg, g, bmine);
}

// (2b) this lane replays game `bg` (global id gid) on the slab at bmine and writes its heads / dirs / deaths.  Every lane
// of the wave runs `replay_len` iterations and takes part in the LAST steps_r of them (the games of a wave normally
// share their step counter, so counting backwards from the end, tc_end, keeps the Philox refills wave-uniform).  A lane
// without a game (lvalid false) idles along: its stores go to the slab's junk byte.
template <int P, int RS>
__device__ __forceinline__ void tron_replay_lane(const crl_tron_cfg &cfg, const TronGeom &g, const TronPad &pad, const int bmine,
                                                 const uint8_t *act_lut, const bool lvalid, const bool from_start,
                                                 const int steps_r, const int replay_len, const uint32_t tc_end,
                                                 const uint32_t gid, const uint32_t seed_lo, const uint32_t seed_hi,
                                                 const int64_t bg, const int64_t B, int16_t *__restrict__ heads,
                                                 int8_t *__restrict__ dirs, int8_t *__restrict__ deaths)
{
    constexpr uint32_t step4 = (uint32_t)((-RS) & 0xff) | (1u << 8) | ((uint32_t)RS << 16) | (0xffu << 24);
    const int N = g.N, NN = g.NN;
    const int first_r = replay_len - steps_r;                   // this lane joins at iteration first_r
    uint32_t c_r = tc_end - (uint32_t)replay_len;
    const int64_t bbg = lvalid ? bg : 0;
    TronRegs<P> s;
    uint32_t stamp[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
        int h = cfg.start_heads[q];
        s.d[q] = cfg.start_dirs[q];
        s.k[q] = lvalid ? 0 : 1;
        if (!from_start) {
            h = lvalid ? min(max((int)heads[q * B + bbg], 0), NN - 1) : 0;
            s.d[q] = lvalid ? dirs[q * B + bbg] & 3 : 0;
            s.k[q] = lvalid ? deaths[q * B + bbg] : 1;
        }
        const int y = (int)__umulhi((uint32_t)h, g.inv_n);
        s.h[q] = bmine + (y + 1) * RS + (h - y * N);
        stamp[q] = (uint32_t)(q + 1);
    }
    const int junk = bmine + pad.junk;
    LdsBoard<(P <= 7) ? 3 : 4> bd{0u};                          // single episode: tag 0, cells hold the plain owner
    bd.within(bmine, bmine + pad.stride);
    TronRng<P> rr;
    int act[P];
    rr.start(gid, c_r, seed_lo, seed_hi);
    rr.next_lut(gid, c_r, seed_lo, seed_hi, act_lut, act);
    for (int t = 0; t < replay_len; ++t) {
        TronProbe<P> pr;
        tron_probe_padded<P>(step4, bd, s, act, pr);
        c_r += 1;
        rr.next_lut(gid, c_r, seed_lo, seed_hi, act_lut, act);
        tron_resolve_lds<P>(bd, s, pr, stamp, junk, lvalid && t >= first_r);
    }
    if (lvalid) {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int rel = s.h[q] - bmine;
            const int row = rel / RS;
            heads[q * B + bg] = (int16_t)((row - 1) * N + (rel - row * RS));
            dirs[q * B + bg] = (int8_t)s.d[q];
            deaths[q * B + bg] = (int8_t)s.k[q];
        }
    }
}

// (3) the slabs' boards to HBM (cells are plain owners; walls are never copied), by `nthreads` threads
template <int RS>
__device__ __forceinline__ void tron_replay_copy_out(int8_t *__restrict__ gslab, const int n_games, const int lds0,
                                                     const TronPad &pad, const TronGeom &g, const int tid, const int nthreads)
{
    const int N = g.N, NN = g.NN;
    if ((N & 3) == 0 && N >= 8) {
        const int bytes = n_games * NN;
        for (int off = tid * 16; off < bytes; off += nthreads * 16) {
            const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
            int o[4];
            tron_piece_offsets<RS>((off - e * NN) >> 2, N >> 2, pad.inv_nq, o);
            uint32_t w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = *(const lds_u32 *)(uintptr_t)(uint32_t)(lds0 + e * pad.stride + o[q]);
            *reinterpret_cast<uint4 *>(gslab + off) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        // rows that are not whole dwords: a cell per thread and store.  (Gathering 16 cells per aligned 16-byte store was tried
        // in round 5 and is SLOWER here -- 39x39 at T = 64: 163 against 145 us, 30x30: 111 against 100: sixteen LDS byte reads with
        // their row / game carries per chunk cost a wave more than the byte stores they save.)
        const int cells = n_games * NN;
        const int full = 0;
        for (int i = full + tid; i < cells; i += nthreads) {
            const int e = (int)__umulhi((uint32_t)i, pad.inv_nn);
            const int c = i - e * NN;
            const int y = (int)__umulhi((uint32_t)c, g.inv_n);
            gslab[i] = (int8_t)*(const lds_u8 *)(uintptr_t)(uint32_t)(lds0 + e * pad.stride + (y + 1) * RS + (c - y * N));
        }
    }
}

// ---- bitboard rollout ---------------------------------------------------------------------------------------
// For long launches the rollout does not need to know WHO owns a cell while it plays: the owner only decides the
// value stored in deaths[], and deaths[] / the board bytes of all but the LAST episode of a launch are never
// output (statistics only need "dead or alive").  So the T steps are played on an occupancy bitboard -- one row
// word per board row, walls as set bits (x >= N, row 0, row N+1), heads as bit addresses, deaths as lane masks --
// and a reset rewrites the whole board from a register-resident pattern (no tags, no rolling clear).  The state
// the launch hands back is rebuilt afterwards by REPLAYING the unfinished episode with the byte-slab stepper of
// tron_rollout_lds_kernel: actions are a pure function of (seed, env id, step counter), so a game that was reset
// during the launch is replayed from the start layout for its `tstep` steps, and the rare game whose episode
// spans the whole launch is replayed from the state it came in with.  Random agents live ~10 steps, so a replay
// is a few dozen steps per wave against thousands of fused steps; crl_tron_rollout picks this kernel for T >= 256
// on boards above 20x20 (where bitboards fit four times the games of byte slabs into a CU's LDS).
// A 32-bit row holds boards up to 30x30, but the replay slabs limit it to the byte kernel's sizes (20 / 40).
struct TronBits {
    int stride;      // bytes per game: two bit slabs of kMaxW pattern words + junk, a multiple of 16 (16-byte rewrite stores)
    uint32_t inv_s;  // floor(2^32 / (N + 1)) + 1
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4 lds_u128;

// Bitboard layout: ONE linear bit string per game, bit (y+1)*S + x for cell (x, y) with S = N + 1.  Row 0 and row
// N+1 are all ones, and so is column N of every row -- which also is "column -1" of the next one.  (N+2)(N+1) bits:
// 216 bytes at 40x40, 58 at 20x20 -- what a reset has to rewrite.  A head is the LDS bit address 8*slab + bit; a
// step adds +-1 or +-S, the word holding a position is (pos >> 3) & ~3 and the bit pos & 31.
template <int P, bool LARGE>
__global__ void __launch_bounds__(256)
tron_rollout_bits_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const TronBits bits, const int64_t B,
                         const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                         int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                         int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    constexpr int kMaxN = LARGE ? kLdsMaxNLarge : kLdsMaxNSmall;
    constexpr int kMaxW = (((kMaxN + 2) * (kMaxN + 1) + 31) / 32 + 3) & ~3;     // pattern words, a multiple of 4
    constexpr int RS = LARGE ? kRowBytesLarge : kRowBytesSmall;                 // byte slabs of the replay
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    __shared__ uint32_t wall_words[kMaxW];
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN, S = N + 1;
    const uint32_t bstep4 = (uint32_t)((-S) & 0xff) | (1u << 8) | ((uint32_t)S << 16) | (0xffu << 24);
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = b < B;
    const int64_t bb = valid ? b : 0;
    const int64_t env0 = b - lane;                              // first game of this wave
    const int n_env = (int)((B - env0) < CRL_WAVE ? (B - env0 > 0 ? B - env0 : 0) : CRL_WAVE);
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;
    const int mine = lds0 + (int)threadIdx.x * bits.stride;     // this lane's bit slab
    const bool wide = (N & 3) == 0 && N >= 8;
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);

    // the empty board (walls only), one word per thread, shared through LDS ...
    if (threadIdx.x < kMaxW) {
        uint32_t w = 0;
        for (int k = 0; k < 32; ++k) {
            const uint32_t bit = threadIdx.x * 32 + k;
            const uint32_t row = __umulhi(bit, bits.inv_s), col = bit - row * S;
            w |= (uint32_t)((row == 0) | (row > (uint32_t)N) | (col == (uint32_t)N)) << k;
        }
        wall_words[threadIdx.x] = w;
    }
    __syncthreads();
    // ... and the start layout (walls + heads) in registers, for resets
    // Two bit slabs per game: the one being played on (`cur`) and a spare that always holds -- or is on its way to
    // holding -- the start layout.  A reset is a swap of the two; the slab just retired is rewritten with the start
    // layout by ALL lanes together, one quarter per step (kPerGroup 16-byte stores), so it is fresh again four steps
    // later, before the shortest episode can end.  (Rewriting a whole slab at every reset, as the first version did,
    // costs kChunks exec-masked 16-byte stores per wave-step -- some lane of a wave resets on practically every step --
    // each at the price of a full-width store: the LDS pipe was 41 % busy with them.)
    // The two slabs of a game lie a multiple of 128 bytes apart: a lane's 16-byte stores then fall on the same four LDS
    // banks whichever slab is its spare, and the eight lanes a ds_write_b128 services together stay on distinct banks
    // (per-game stride = 4 dwords mod 32).
    constexpr int kChunks = kMaxW / 4, kPerGroup = (kChunks + 3) / 4, kSlab = (kMaxW * 4 + 127) & ~127;
    uint32_t fresh[kMaxW];
#pragma unroll
    for (int j = 0; j < kMaxW; ++j) {
        uint32_t w = wall_words[j];
        *(lds_u32 *)(uintptr_t)(uint32_t)(mine + 4 * j) = w;    // this lane's first slab starts out empty
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int fh = cfg.start_heads[p];
            const int fy = (int)__umulhi((uint32_t)fh, g.inv_n);
            const int fb = (fy + 1) * S + (fh - fy * N);
            w |= ((fb >> 5) == j) ? 1u << (fb & 31) : 0u;
        }
        fresh[j] = w;
        asm volatile("" : "+v"(fresh[j]));                      // VGPRs, not SGPRs: the reset stores them as they are
        *(lds_u32 *)(uintptr_t)(uint32_t)(mine + kSlab + 4 * j) = w;   // the spare slab: start layout
    }
    const int junk_row = mine + 2 * kSlab;                      // a word nobody reads
    int cur = mine;                                             // slab in play; the spare is cur ^ flip
    const int flip = mine ^ (mine + kSlab);
    // ---- copy in: the wave ORs the occupied cells of its 64 boards into the slabs
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const int8_t *gslab = board + env0 * NN;
        const int slab0 = mine - lane * bits.stride;
        if (wide) {
            const int bytes = n_env * NN;
#pragma unroll 5
            for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {      // unrolled: several loads in flight
                const uint4 v = *reinterpret_cast<const uint4 *>(gslab + off);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
                const int cq = (off - e * NN) >> 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // one bit per non-zero byte of the dword (bit 7 of (b | (b & 0x7f) + 0x7f) <=> b != 0), gathered to a nibble
                    const uint32_t nz = (((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[q]) & 0x80808080u;
                    const uint64_t nib = ((nz >> 7) * 0x10204080u) >> 28;
                    const int y = (int)__umulhi((uint32_t)(cq + q), pad.inv_nq);
                    const int bit = (y + 1) * S + 4 * (cq + q - y * (N >> 2));   // 4 cells of one row: contiguous bits,
                    const uint64_t two = nib << (bit & 31);                       // possibly across a word boundary
                    unsigned int *wp = (unsigned int *)(lds + (slab0 - lds0) + e * bits.stride + ((bit >> 5) << 2));
                    atomicOr(wp, (unsigned int)two);
                    atomicOr(wp + 1, (unsigned int)(two >> 32));
                }
            }
        } else {
            for (int e = 0; e < n_env; ++e)
                for (int c = lane; c < NN; c += CRL_WAVE) {
                    const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                    const int bit = (y + 1) * S + (c - y * N);
                    if (gslab[(int64_t)e * NN + c] != 0)
                        atomicOr((unsigned int *)(lds + (slab0 - lds0) + e * bits.stride + ((bit >> 5) << 2)), 1u << (bit & 31));
                }
        }
    }
    // heads as BIT addresses: 8 * slab address + bit index; a step is +-1 or +-S
    // deaths are LANE MASKS (one 64-bit scalar per player, bit = lane): every boolean of the resolve is then a
    // scalar and/or on the side of the vector pipe; compares feed them through ballots, selects read them back
    int pos[P], dir_[P], fresh_pos[P], fresh_dir[P];
    uint64_t dead[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int h = valid ? heads[p * B + bb] : 0;
        h = min(max(h, 0), NN - 1);                             // a hand-made state must not turn into a wild LDS address
        const int y = (int)__umulhi((uint32_t)h, g.inv_n);
        pos[p] = 8 * mine + (y + 1) * S + (h - y * N);
        dir_[p] = valid ? dirs[p * B + bb] & 3 : 0;
        dead[p] = __builtin_amdgcn_ballot_w64(valid ? (deaths[p * B + bb] != 0) : true);
        const int fh = cfg.start_heads[p];
        const int fy = (int)__umulhi((uint32_t)fh, g.inv_n);
        fresh_pos[p] = (fy + 1) * S + (fh - fy * N);             // bit index inside a slab
        fresh_dir[p] = cfg.start_dirs[p];
        asm volatile("" : "+v"(fresh_pos[p]), "+v"(fresh_dir[p]));
    }
    TronAcc<P> acc;
    acc.load(st, valid, bb, B);
    const uint32_t ts_at_entry = acc.ts;
    uint32_t alive_steps[P];
    int last_alive = 0;                                         // who was alive at this lane's latest terminal step
#pragma unroll
    for (int p = 0; p < P; ++p) alive_steps[p] = 0;
    const uint64_t valid_m = __builtin_amdgcn_ballot_w64(valid);
    TronRng<P> rng;
    rng.start(gid, acc.tc, seed_lo, seed_hi);
    int act[P];
    __syncthreads();                                            // action table
    rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);
    auto one_step = [&](auto grp_tag) {
        constexpr int GRP = decltype(grp_tag)::value;
        // positions are linear bit addresses, so the 32-bit word holding a cell is (pos >> 3) & ~3 whatever the row width
        int np[P], nd[P], ra[P];
        uint32_t roww[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {                           // CyTronGrid.pyx:21-41, all P probes in flight
            nd[i] = (dir_[i] + act[i]) & 3;
            np[i] = pos[i] + __builtin_amdgcn_sbfe((int)bstep4, nd[i] << 3, 8);
            ra[i] = (np[i] >> 3) & ~3;
        }
#pragma unroll
        for (int i = 0; i < P; ++i) roww[i] = *(const lds_u32 *)(uintptr_t)(uint32_t)ra[i];
        // every lane rewrites quarter GRP of its spare slab with the start layout.  GRP is a compile-time constant (the
        // step loop is unrolled four times below): as a run-time `switch (t & 3)` the compiler merged the four cases
        // into one set of stores fed by ~16 register copies per step, which cost more than the burst resets had.
        auto rewrite = [&](const int grp, const int base) {
#pragma unroll
            for (int c = 0; c < kPerGroup; ++c) {
                const int j = 4 * (grp * kPerGroup + c);
                if (j < kMaxW)
                    *(lds_u128 *)(uintptr_t)(uint32_t)(base + 4 * j) = (u32x4){fresh[j], fresh[j + 1], fresh[j + 2], fresh[j + 3]};
            }
        };
        // Issued right behind the probes: the LDS serves a wave's requests in order, so here the stores run while the
        // wave waits for its probe words anyway; at the end of the step they would sit in front of the NEXT step's probes.
        rewrite(GRP, cur ^ flip);
        acc.tc += 1;
        rng.next_lut(gid, acc.tc, seed_lo, seed_hi, act_lut, act);
#pragma unroll
        for (int i = 0; i < P; ++i) {                           // CyTronGrid.pyx:15-62 with deaths as lane masks
            const uint64_t run = ~dead[i];                      // :16 (may have been killed head-on by j < i)
            uint64_t on_head[P];
#pragma unroll
            for (int q = 0; q < P; ++q) on_head[q] = (q != i) ? __builtin_amdgcn_ballot_w64(np[i] == pos[q]) : 0ull;
            const uint32_t bit = 1u << (np[i] & 31);
            uint64_t occ = __builtin_amdgcn_ballot_w64((roww[i] & bit) != 0);     // trail or wall
#pragma unroll
            for (int j = 0; j < i; ++j) occ |= on_head[j];      // j moved there earlier in this very step
            const uint64_t moved_m = run & ~occ;
            dead[i] |= run & occ;
#pragma unroll
            for (int q = 0; q < P; ++q)
                if (q != i) dead[q] |= run & on_head[q];        // :56-57 head-on: the owner dies too
            const bool moved = __builtin_amdgcn_inverse_ballot_w64(moved_m);
            dir_[i] = __builtin_amdgcn_inverse_ballot_w64(run) ? nd[i] : dir_[i];
            pos[i] = moved ? np[i] : pos[i];
            atomicOr((unsigned int *)(lds + ((moved ? ra[i] : junk_row) - lds0)), bit);
        }
        // per-lane view of who is alive (0 / 1), shared by the step count, the terminal test and the winners
        uint32_t alive01[P], alive = 0, alive_bits = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            alive01[p] = __builtin_amdgcn_inverse_ballot_w64(~dead[p]) ? 1u : 0u;
            alive_steps[p] += alive01[p];
            alive += alive01[p];
            alive_bits |= alive01[p] << p;
        }
        const uint64_t term_m = __builtin_amdgcn_ballot_w64(alive <= 1) & valid_m;      // TronGridEnvironment.py:309-321
        acc.ts += 1;
        if (term_m) {                                           // some game of the wave ended
            const bool me = __builtin_amdgcn_inverse_ballot_w64(term_m);
            // The spare has been rewritten completely iff four steps have passed since this lane's last swap (acc.ts,
            // the length of the episode that just ended) -- or no swap happened in this launch yet (the set-up left it
            // fresh).  An episode of fewer than four steps: rewrite all of it now (rare).
            if (__builtin_amdgcn_ballot_w64(me && acc.ts < 4u && acc.n_ep > 0u)) {
                if (me && acc.ts < 4u && acc.n_ep > 0u) {
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) rewrite(grp, cur ^ flip);
                }
            }
            if (me) {                                           // new_state: swap the slabs
                last_alive = (int)alive_bits;
                cur ^= flip;
                acc.n_ep += 1;
                acc.last_len = (int)acc.ts;
                acc.ts = 0;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    acc.wins[p] += alive01[p];
                    pos[p] = 8 * cur + fresh_pos[p]; dir_[p] = fresh_dir[p];
                }
            }
#pragma unroll
            for (int p = 0; p < P; ++p) dead[p] &= ~term_m;
        }
    
    };
    for (int t = 0;;) {                                         // four steps per trip, one per quarter of the spare slab
        if (t >= T) break;
        one_step(std::integral_constant<int, 0>{}); ++t;
        if (t >= T) break;
        one_step(std::integral_constant<int, 1>{}); ++t;
        if (t >= T) break;
        one_step(std::integral_constant<int, 2>{}); ++t;
        if (t >= T) break;
        one_step(std::integral_constant<int, 3>{}); ++t;
    }
    acc.len_sum = ts_at_entry + (uint32_t)T - acc.ts;
    if (acc.n_ep > 0) acc.last_w = last_alive;
#pragma unroll
    for (int p = 0; p < P; ++p) acc.ret[p] = 2 * (int)alive_steps[p] - T + 9 * (int)acc.wins[p];
    if (valid) acc.store(st, B, b);

    // ---- replay of the unfinished episode on byte slabs: rebuilds board / heads / dirs / deaths (helpers above)
    const bool from_start = acc.n_ep > 0;                       // else: from the state the launch came in with
    const int steps_r = valid ? (from_start ? (int)acc.ts : T) : 0;
    int replay_len = steps_r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) replay_len = max(replay_len, __shfl_xor(replay_len, off, CRL_WAVE));
    __syncthreads();                                            // bit slabs are dead from here; byte slabs reuse the LDS
    // The LDS holds the byte slabs of all 4 waves (boards up to 20x20) or of one wave at a time (LARGE).  Per turn:
    // the whole workgroup lays out fresh boards, the owning wave(s) replay, the whole workgroup copies the boards out.
    constexpr int kTurns = LARGE ? 4 : 1;
    constexpr int kSlabs = LARGE ? CRL_WAVE : 256;
    const int slot = LARGE ? lane : (int)threadIdx.x;
    const int bmine = lds0 + slot * pad.stride;
    for (int turn = 0; turn < kTurns; ++turn) {
        tron_replay_fresh_slabs<RS>(lds0, pad, N, kSlabs, (int)threadIdx.x, 256);
        __syncthreads();
        if ((int)threadIdx.x < kSlabs) tron_replay_stamp_heads<P, RS>(cfg, g, lds0 + (int)threadIdx.x * pad.stride);
        __syncthreads();
        if (!LARGE || wave == turn) {
            if (__builtin_amdgcn_ballot_w64(valid && !from_start)) {            // somebody resumes from the incoming board
                tron_replay_copy_in<RS>(board + env0 * NN, n_env, bmine - lane * pad.stride, pad, g, lane);
                if (from_start) tron_replay_refresh_slab<P, RS>(cfg, g, bmine);  // this lane's slab was overwritten too
            }
            tron_replay_lane<P, RS>(cfg, g, pad, bmine, act_lut, valid, from_start, steps_r, replay_len, acc.tc, gid,
                                    seed_lo, seed_hi, b, B, heads, dirs, deaths);
        }
        __syncthreads();
        {
            const int64_t gbase = (int64_t)blockIdx.x * blockDim.x + (LARGE ? turn * CRL_WAVE : 0);
            const int64_t rem = B - gbase;
            const int n_out = (int)(rem < 0 ? 0 : (rem > kSlabs ? kSlabs : rem));
            tron_replay_copy_out<RS>(board + gbase * NN, n_out, lds0, pad, g, (int)threadIdx.x, 256);
        }
        if (LARGE) __syncthreads();
    }
}

// ---- quad bitboard rollout: one lane per PLAYER on occupancy bitboards (boards up to 40x40, P <= 4) ------------------
// The lane-per-game bitboard kernel above keeps a 40x40 game in 528 bytes of LDS, but 65,536 games are still only
// 1,024 waves: one per SIMD, issuing an instruction every ~5.9 cycles.  This kernel is to it what
// tron_rollout_quad_kernel is to the byte kernel: a lane plays ONE player (probe a bit, die or move, set a bit), the
// quad shares what is per game (alive count by two DPP adds, the random stream, the reset), the reference's sequential
// order (CyTronGrid.pyx:15-62) is resolved -- redundantly in the four lanes, on DPP-gathered copies -- only on the
// 1.3 % of wave-steps where a target meets another player's head or target.  64 games per workgroup (34 KB of LDS), four
// workgroups per CU: 4 waves per SIMD.
//   * Two bit slabs per game as above: a reset swaps them; the retired one is rewritten with the start layout by the
//     quad, 16 bytes per lane and step (group t & 3 of the slab, an immediate offset: no address arithmetic), so it is
//     fresh again four steps later; an episode shorter than that rewrites it at once.
//   * Owners are not tracked: as above the state handed back is rebuilt by REPLAYING the unfinished episode on byte
//     slabs.  The bit slabs are dead by then, and their LDS takes the byte slabs of 16 games at a time: four turns, each
//     laid out and copied out by the whole workgroup and replayed by one wave with a lane per game (the same
//     tron_resolve_lds as everywhere).
template <int P, bool LARGE>
__global__ void __launch_bounds__(256, 4)
tron_rollout_qbits_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const TronBits bits, const int64_t B,
                          const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                          int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                          int8_t *__restrict__ deaths, const crl_tron_stats st, const int split_replay)
{
    static_assert(P <= 4, "one lane per player, four lanes per game");
    constexpr int kGames = 64, kWaveGames = 16;
    constexpr int kMaxN = LARGE ? kLdsMaxNLarge : kLdsMaxNSmall;
    constexpr int kMaxW = (((kMaxN + 2) * (kMaxN + 1) + 31) / 32 + 3) & ~3;     // pattern words, a multiple of 4
    constexpr int kSlab = (kMaxW * 4 + 127) & ~127;                             // 256 / 128 bytes
    constexpr int kGroups = kSlab / 64;                                         // 64 bytes per quad and step: 4 / 2 groups
    constexpr int RS = LARGE ? kRowBytesLarge : kRowBytesSmall;                 // byte slabs of the replay
    static_assert(kGroups <= 4, "a slab is rewritten within four steps");
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    __shared__ uint32_t wall_words[kSlab / 4];
    __shared__ uint32_t r_tc[kGames];                           // hand-over to the replay: step counter at the end,
    __shared__ int r_steps[kGames];                             // steps to replay (-1: T steps from the incoming state)
    QUAD_STAMP(0);
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN, S = N + 1;
    const uint32_t bstep4 = (uint32_t)((-S) & 0xff) | (1u << 8) | ((uint32_t)S << 16) | (0xffu << 24);
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int wave = threadIdx.x >> 6;
    const int p = lane & 3;                                     // my player
    const int slot = threadIdx.x >> 2;                          // my game's slabs in this workgroup
    const int64_t b = (int64_t)blockIdx.x * kGames + slot;
    const bool gvalid = b < B;
    const bool pvalid = gvalid && p < P;
    const int64_t bb = gvalid ? b : 0;
    const int64_t env0 = (int64_t)blockIdx.x * kGames + wave * kWaveGames;   // first game of this wave
    const int n_env = (int)((B - env0) < kWaveGames ? (B - env0 > 0 ? B - env0 : 0) : kWaveGames);
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;
    const int mine = lds0 + slot * bits.stride;                 // my game: slab A, slab B, four junk words
    const bool wide = (N & 3) == 0 && N >= 8;
    const uint32_t gid = (uint32_t)(first_env_id + (uint64_t)bb);

    // the empty board (walls only), one word per thread, shared through LDS
    if (threadIdx.x < kSlab / 4) {
        uint32_t w = 0;
        if (threadIdx.x < kMaxW)
            for (int k = 0; k < 32; ++k) {
                const uint32_t bit = threadIdx.x * 32 + k;
                const uint32_t row = __umulhi(bit, bits.inv_s), col = bit - row * S;
                w |= (uint32_t)((row == 0) | (row > (uint32_t)N) | (col == (uint32_t)N)) << k;
            }
        wall_words[threadIdx.x] = w;
    }
    // my player's state and the running totals: loads in flight while the slabs are laid out
    // (unconditional, from a clamped index: under `pvalid ? load : 0` every load sits in its own branch with its own wait)
    const int64_t pb = (int64_t)(p < P ? p : 0) * B + bb;
    const int h_in = heads[pb];
    const int d_in = dirs[pb];
    const int k_in = pvalid ? (int)deaths[pb] : 1;
    const int old_ret = st.ret_sum[pb];
    const uint32_t old_wins = st.win_count[pb];
    const uint32_t tc_in = st.tcount[bb], ts_at_entry = st.tstep[bb];
    const uint32_t old_n_ep = st.n_episodes[bb], old_len_sum = st.len_sum[bb];
    const uint32_t old_last_w = st.last_winners[bb];
    __syncthreads();
    // my quarter of every 64-byte group of a slab: words 16 g + 4 p .. + 3.  `fresh` is the start layout (walls + start
    // heads), kept in registers for the rewrites.
    int fh = cfg.start_heads[0], fd = cfg.start_dirs[0];
    fh = (p == 1) ? cfg.start_heads[1] : fh; fd = (p == 1) ? cfg.start_dirs[1] : fd;
    fh = (p == 2) ? cfg.start_heads[2] : fh; fd = (p == 2) ? cfg.start_dirs[2] : fd;
    fh = (p == 3) ? cfg.start_heads[3] : fh; fd = (p == 3) ? cfg.start_dirs[3] : fd;
    u32x4 fresh[kGroups];
#pragma unroll
    for (int grp = 0; grp < kGroups; ++grp) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = 16 * grp + 4 * p + i;
            const uint32_t wall = wall_words[j];
            uint32_t v = wall;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int sh = cfg.start_heads[q];
                const int sy = (int)__umulhi((uint32_t)sh, g.inv_n);
                const int fb = (sy + 1) * S + (sh - sy * N);
                v |= ((fb >> 5) == j) ? 1u << (fb & 31) : 0u;
            }
            w[i] = v;
            *(lds_u32 *)(uintptr_t)(uint32_t)(mine + 4 * j) = wall;             // slab A: the empty board
            *(lds_u32 *)(uintptr_t)(uint32_t)(mine + kSlab + 4 * j) = v;        // slab B, the spare: the start layout
        }
        fresh[grp] = (u32x4){w[0], w[1], w[2], w[3]};
        asm volatile("" : "+v"(fresh[grp]));
    }
    *(lds_u32 *)(uintptr_t)(uint32_t)(mine + 2 * kSlab + 4 * p) = 0u;           // my junk word
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- copy in: the wave ORs the occupied cells of its 16 boards into their slabs A
    {
        const int8_t *gslab = board + env0 * NN;
        const int slab0 = lds0 + wave * kWaveGames * bits.stride;
        if (wide) {
            const int bytes = n_env * NN;
#pragma unroll 5
            for (int off = lane * 16; off < bytes; off += CRL_WAVE * 16) {      // unrolled: several loads in flight
                const uint4 v = *reinterpret_cast<const uint4 *>(gslab + off);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                const int e = (int)__umulhi((uint32_t)off, pad.inv_nn);
                const int cq = (off - e * NN) >> 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // one bit per non-zero byte of the dword, gathered to a nibble (see tron_rollout_bits_kernel)
                    const uint32_t nz = (((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[q]) & 0x80808080u;
                    const uint64_t nib = ((nz >> 7) * 0x10204080u) >> 28;
                    const int y = (int)__umulhi((uint32_t)(cq + q), pad.inv_nq);
                    const int bit = (y + 1) * S + 4 * (cq + q - y * (N >> 2));   // 4 cells of one row: contiguous bits,
                    const uint64_t two = nib << (bit & 31);                       // possibly across a word boundary
                    unsigned int *wp = (unsigned int *)(lds + (slab0 - lds0) + e * bits.stride + ((bit >> 5) << 2));
                    atomicOr(wp, (unsigned int)two);
                    atomicOr(wp + 1, (unsigned int)(two >> 32));
                }
            }
        } else {
            // rows that are not whole dwords (round 5): the wave's 16 boards are still ONE run of bytes that starts on a 16-byte
            // boundary (env0 is a multiple of 16), so it is read in aligned 16-byte chunks; bit (y + 1) S + x = cell + y + S, so the
            // four cells of a dword are four contiguous bits -- split by one extra bit where the dword runs into the next row --,
            // and only a dword that straddles two GAMES (16 per wave) goes cell by cell.  (Byte by byte this copy took 120 us of
            // a launch at 39x39, against 20 us at 40x40.)
            const int bytes = n_env * NN, full = N >= 4 ? bytes & ~15 : 0;     // (N < 4: a dword could cross two rows -- cell by cell)
            auto slab_word = [&](const int e, const int bit) {
                return (unsigned int *)(lds + (slab0 - lds0) + e * bits.stride + ((bit >> 5) << 2));
            };
#pragma unroll 4
            for (int off = lane * 16; off < full; off += CRL_WAVE * 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(gslab + off);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                const int e0 = (int)__umulhi((uint32_t)off, pad.inv_nn);
                const int r = off - e0 * NN;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int c = r + 4 * q, e = e0;
                    if (c >= NN) { c -= NN; e += 1; }
                    if (w[q] == 0u) continue;
                    if (c + 3 < NN) {
                        const uint32_t nz = (((w[q] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[q]) & 0x80808080u;
                        const uint32_t nib = (uint32_t)(((nz >> 7) * 0x10204080u) >> 28);
                        const int y = (int)__umulhi((uint32_t)c, g.inv_n), x = c - y * N;
                        const int left = N - x;                  // cells of this dword that are still in row y
                        const uint32_t lo = left >= 4 ? 0xfu : ((1u << left) - 1u);
                        const int bit = (y + 1) * S + x;
                        const uint64_t two = ((uint64_t)(nib & lo) << (bit & 31)) | ((uint64_t)(nib & ~lo) << ((bit & 31) + 1));
                        unsigned int *wp = slab_word(e, bit);
                        atomicOr(wp, (unsigned int)two);
                        atomicOr(wp + 1, (unsigned int)(two >> 32));
                    } else {
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) {
                            int cc = c + k2, ee = e;
                            if (cc >= NN) { cc -= NN; ee += 1; }
                            if ((w[q] >> (8 * k2)) & 0xffu) {
                                const int y = (int)__umulhi((uint32_t)cc, g.inv_n);
                                const int bit = (y + 1) * S + (cc - y * N);
                                atomicOr(slab_word(ee, bit), 1u << (bit & 31));
                            }
                        }
                    }
                }
            }
            for (int cc0 = full + lane; cc0 < bytes; cc0 += CRL_WAVE) {     // (a wave with fewer than 16 games: the last bytes of its run)
                const int e = (int)__umulhi((uint32_t)cc0, pad.inv_nn), c = cc0 - e * NN;
                const int y = (int)__umulhi((uint32_t)c, g.inv_n);
                const int bit = (y + 1) * S + (c - y * N);
                if (gslab[cc0] != 0) atomicOr(slab_word(e, bit), 1u << (bit & 31));
            }
        }
    }
    // ---- my player: the head as a BIT address (8 * slab address + bit index; a step is +-1 or +-S)
    const int jaddr = mine + 2 * kSlab + 4 * p;                 // my junk word: dead players probe and "move" there
    const int jpos = 8 * jaddr;
    const int fy = (int)__umulhi((uint32_t)(fh < 0 ? 0 : fh), g.inv_n);
    // my start cell as a bit index inside a slab; a seat without a player sits on a padding bit behind the pattern
    // words (no target ever equals it, and it never moves)
    static_assert(kSlab * 8 - 8 >= kMaxW * 32 || kSlab * 8 - 8 >= (kMaxN + 2) * (kMaxN + 1), "padding bits behind the board");
    const int fresh_off = (p < P) ? (fy + 1) * S + (fh - fy * N) : kSlab * 8 - 8 + p;
    const int fresh_d8 = fd << 3;
    const int fresh_a = (p < P) ? 1 : 0;
    int cur8 = 8 * mine;                                        // 8 * (slab in play)
    const int flip8 = (8 * mine) ^ (8 * (mine + kSlab));
    int spare_p = mine + kSlab + 16 * p;                        // my 16 bytes of group 0 of the spare slab
    const int both_p = (mine + 16 * p) + (mine + kSlab + 16 * p);   // a swap is both_p - spare_p (slabs are not aligned
                                                                    // to their size: no xor)
    int pos, d8 = (d_in & 3) << 3;
    {
        const int hc = min(max(h_in, 0), NN - 1);
        const int y = (int)__umulhi((uint32_t)hc, g.inv_n);
        pos = cur8 + (pvalid ? (y + 1) * S + (hc - y * N) : fresh_off);
    }
    uint32_t alive_steps = 0, wn = 0, marks = 0, ok_at2 = 0;    // as in tron_rollout_quad_kernel; ok_at2: the spare is
                                                                // completely fresh once 2 * (steps done) reaches it
    __syncthreads();                                            // action table
    QUAD_STAMP(1);
    uint32_t a_lo = 0, a_hi = 0;
    auto refill = [&](const uint32_t group) { tron_quad_actions(gid, group, p, seed_lo, seed_hi, act_lut, a_lo, a_hi); };
    refill(tc_in >> 5);
    uint32_t acts = ((tc_in & 16u) ? a_hi : a_lo) >> ((tc_in & 15u) * 2u);
    uint32_t dry2 = 32u - 2u * (tc_in & 15u);
    int neg2 = -(int)dry2;
    // my player is alive: carried as a 0 / 1 vector register (the compiler would not keep a lane mask across the unrolled
    // loop and its reset branches; one compare per step turns it into one)
    int a = (pvalid && k_in == 0) ? 1 : 0;
    auto store_group = [&](const int grp) {                     // my 16 bytes of group grp of the spare slab
#ifndef CRL_DIAG_NO_REWRITE     /* diagnostic builds only (WRONG results): what the rolling rewrite of the spare slab costs */
        CRL_BOUNDS_IN(spare_p + 64 * grp, mine, mine + bits.stride - 15, 112);   // the rolling rewrite's 16 bytes
        *(lds_u128 *)(uintptr_t)(uint32_t)(spare_p + 64 * grp) = fresh[grp];
#endif
    };
    // When every game of the wave enters with its step counter a multiple of four (launches of 4 k steps keep it so),
    // `acts` can only run dry at the end of a four-step trip: the countdown is then kept per trip, not per step.
    const bool by_trip = __builtin_amdgcn_ballot_w64((tc_in & 3u) != 0u) == 0ull;
    // ... and when the whole wave shares ONE step counter (the usual case) the countdown is a scalar: add, compare and
    // branch on the scalar unit, every step (see tron_rollout_quad_kernel).
    const bool uni_wave = __builtin_amdgcn_ballot_w64(tc_in != (uint32_t)__builtin_amdgcn_readfirstlane((int)tc_in)) == 0ull;
    int sneg2 = __builtin_amdgcn_readfirstlane(neg2);
    uint32_t sdry2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dry2);
    auto one_step = [&](auto grp_tag, auto uni_tag, const bool trips) {   // trips: wave-uniform, this step is part of such a trip
        constexpr int GRP = decltype(grp_tag)::value;
        constexpr bool UNI = decltype(uni_tag)::value;
        const bool run = a != 0;
        const int dir8 = (int)((acts << 3) + (uint32_t)d8);
        const int tgt = pos + __builtin_amdgcn_sbfe((int)bstep4, dir8, 8);
        const int tq = run ? tgt : jpos;                        // a dead player probes its own junk word
        const int wa = (tq >> 3) & ~3;                          // the word holding the cell; its bit is tq & 31
        uint32_t bit;                                           // 1 << (tq & 31): the shift reads five bits by itself
        asm("v_lshlrev_b32 %0, %1, 1" : "=v"(bit) : "v"(tq));
        // Probe and trail in ONE LDS operation (round 3, second half): an atomic OR that returns the word as it was.  A player
        // that finds its bit set dies and has changed nothing; one that finds it clear has entered the cell.  (Round 2 read
        // the word, decided, and then issued a second, non-returning atomic OR on the word -- or on the junk word for a player
        // that had not moved: two of the step's LDS operations, both on random words, i.e. at ~3.5 LDS cycles per conflict-free
        // one; the LDS array is what this kernel waits for.)  Two players of a game entering one cell in one step see each
        // other's bit, whichever the LDS serves first: that is an interaction, and the fix-up below sorts it out.
        uint32_t word;
        CRL_BOUNDS_IN(wa, mine, mine + bits.stride, 111);       // the probed word lies in my game's two slabs + junk words
#ifdef CRL_QBITS_PROBE_THEN_OR     /* the round-2 form, kept for A/B builds */
        asm volatile("ds_read_b32 %0, %1" : "=v"(word) : "v"(wa) : "memory");
#else
        asm volatile("ds_or_rtn_b32 %0, %1, %2" : "=v"(word) : "v"(wa), "v"(bit) : "memory");
#endif
        // right behind the probe: the LDS serves a wave's requests in order, so the store runs while the wave waits
        if constexpr (GRP < kGroups) store_group(GRP);
        const int x1 = tq ^ tron_quad<0x39>(pos), x2 = tq ^ tron_quad<0x4E>(pos), x3 = tq ^ tron_quad<0x93>(pos);
        const int y1 = tq ^ tron_quad<0x39>(tq), y2 = tq ^ tron_quad<0x4E>(tq);
        uint32_t near = min(min(min(min((uint32_t)x1, (uint32_t)x2), (uint32_t)x3), (uint32_t)y1), (uint32_t)y2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(word), "+v"(near) : : "memory");
        // the common path, for every lane and without a branch: each player on its own (see tron_rollout_quad_kernel)
        const bool dead = run & ((word & bit) != 0u);           // :47-57 trail or wall
        const bool moved = run ^ dead;                          // :60-62
        const int pos_was = pos;
        d8 = run ? dir8 : d8;                                   // :44 the direction is committed even if the move dies
        pos = moved ? tgt : pos;
#ifdef CRL_QBITS_PROBE_THEN_OR
        atomicOr((unsigned int *)(lds + ((moved ? wa : jaddr) - lds0)), bit);
#endif
        bool alive_now = moved;
#if defined(CRL_DIAG_NO_SLOW)      /* diagnostic builds only (WRONG results): what the interaction path costs the common one */
        if (false && near == 0u) {
#elif defined(CRL_DIAG_DETECT_ONLY) /* ... and with the detection kept but nothing behind it */
        if (__builtin_amdgcn_ballot_w64(near == 0u) != 0ull) asm volatile("s_nop 0");
        if (false) {
#else
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(near == 0u) != 0ull, 0)) {
#endif
            // rare fix-up: take back the bits the common path set (those cells were empty), then the quad's four players
            // in the reference's order from the pre-step state, redundantly in its four lanes
            if (moved) atomicAnd((unsigned int *)(lds + (wa - lds0)), ~bit);
            const int d = (run ? (dir8 - (int)(acts << 3)) >> 3 : d8 >> 3) & 3, dir = (dir8 >> 3) & 3;
            const int al = run ? 1 : 0;
            int oc = (word & bit) ? 1 : 0;
#ifndef CRL_QBITS_PROBE_THEN_OR
            {   // the probe set its bit in the same operation: a player that found the bit of an EMPTY cell set by a player of
                // its game that entered the cell in this very step must see the cell as it was before the step (had the cell
                // been occupied before, nobody would have entered it)
                // (every DPP read in ALL lanes before any lane selects: under a short-circuit && the second read would run
                //  only in the lanes whose targets match, and a DPP read of a lane that sits the branch out returns 0)
                const int mvi = moved ? 1 : 0;
                const int t1 = tron_quad<0x39>(tq), t2 = tron_quad<0x4E>(tq), t3 = tron_quad<0x93>(tq);
                const int m1 = tron_quad<0x39>(mvi), m2 = tron_quad<0x4E>(mvi), m3 = tron_quad<0x93>(mvi);
                const int entered = (tq == t1 ? m1 : 0) | (tq == t2 ? m2 : 0) | (tq == t3 ? m3 : 0);
                oc = entered ? 0 : oc;
            }
#endif
            int ps[4] = {tron_quad<0x00>(pos_was), tron_quad<0x55>(pos_was), tron_quad<0xAA>(pos_was), tron_quad<0xFF>(pos_was)};
            int ds[4] = {tron_quad<0x00>(d), tron_quad<0x55>(d), tron_quad<0xAA>(d), tron_quad<0xFF>(d)};
            int al4[4] = {tron_quad<0x00>(al), tron_quad<0x55>(al), tron_quad<0xAA>(al), tron_quad<0xFF>(al)};
            const int tg[4] = {tron_quad<0x00>(tgt), tron_quad<0x55>(tgt), tron_quad<0xAA>(tgt), tron_quad<0xFF>(tgt)};
            const int oq[4] = {tron_quad<0x00>(oc), tron_quad<0x55>(oc), tron_quad<0xAA>(oc), tron_quad<0xFF>(oc)};
            const int nd[4] = {tron_quad<0x00>(dir), tron_quad<0x55>(dir), tron_quad<0xAA>(dir), tron_quad<0xFF>(dir)};
            const int jbase = mine + 2 * kSlab;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                       // CyTronGrid.pyx:15-62
                const bool runi = al4[i] != 0;                   // :16 (may have been killed head-on by j < i)
                bool on_head[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) on_head[q] = (q != i) && (tg[i] == ps[q]);   // ps[q] already moved for q < i
                bool occ = oq[i] != 0;                          // trail or wall
#pragma unroll
                for (int j = 0; j < i; ++j) occ |= on_head[j];  // j moved there earlier in this very step
                const bool mv = runi & !occ;
                al4[i] = (runi & occ) ? 0 : al4[i];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q != i) al4[q] = (runi & on_head[q]) ? 0 : al4[q];                // :56-57 head-on: the owner dies too
                ds[i] = runi ? nd[i] : ds[i];                   // :44
                ps[i] = mv ? tg[i] : ps[i];
                atomicOr((unsigned int *)(lds + ((mv ? ((tg[i] >> 3) & ~3) : jbase) - lds0)), 1u << (tg[i] & 31));
            }
            int pS = ps[0], dS = ds[0], aS = al4[0];
            pS = (p == 1) ? ps[1] : pS; dS = (p == 1) ? ds[1] : dS; aS = (p == 1) ? al4[1] : aS;
            pS = (p == 2) ? ps[2] : pS; dS = (p == 2) ? ds[2] : dS; aS = (p == 2) ? al4[2] : aS;
            pS = (p == 3) ? ps[3] : pS; dS = (p == 3) ? ds[3] : dS; aS = (p == 3) ? al4[3] : aS;
            pos = pS;
            d8 = dS << 3;
            alive_now = (p < P) && aS != 0;
        }
        // TronGridEnvironment.py:309-321 for the game: alive players over the quad
        a = alive_now ? 1 : 0;
        int alive = a + tron_quad<0xB1>(a);
        alive += tron_quad<0x4E>(alive);
        alive_steps += (uint32_t)a;
        if (alive <= 1) {                                       // new_state: swap the slabs
            // 2 * (launch steps done): the countdown moves on below; inside a trip it is as of the trip's start
            uint32_t done2;
            if constexpr (UNI) done2 = sdry2 + (uint32_t)sneg2 + 2u;
            else done2 = dry2 + (uint32_t)neg2 + (uint32_t)(trips ? 2 * (GRP + 1) : 2);
            // the spare is completely fresh four steps after it was retired -- or at once, for a shorter episode (rare)
            if (done2 < ok_at2) {
#pragma unroll
                for (int grp = 0; grp < kGroups; ++grp) store_group(grp);
            }
            ok_at2 = done2 + 8u;
            cur8 ^= flip8;
            spare_p = both_p - spare_p;
            asm("v_add3_u32 %0, %0, %1, %2" : "+v"(wn) : "v"(a), "s"(0x10000u));       // (see tron_rollout_quad_kernel)
            asm("v_lshl_add_u32 %0, %0, 16, %1\n\tv_add_u32 %0, %0, %2" : "+v"(marks) : "v"(done2), "v"(a));
            pos = cur8 + fresh_off;
            d8 = fresh_d8;
            a = fresh_a;
        }
        acts >>= 2;                                             // (after the fix-up, which reads this step's action)
        if constexpr (UNI) {
            sneg2 += 2;
            if (sneg2 == 0) {
                const uint32_t c = tc_in + (sdry2 >> 1);
                if ((c & 16u) == 0u) refill(c >> 5);
                acts = (c & 16u) ? a_hi : a_lo;
                sdry2 += 32u;
                sneg2 = -32;
            }
        } else {
            if (!trips || GRP == 3) neg2 += trips ? 8 : 2;
            if ((!trips || GRP == 3) && neg2 == 0) {
                const uint32_t c = tc_in + (dry2 >> 1);
                if ((c & 16u) == 0u) refill(c >> 5);
                acts = (c & 16u) ? a_hi : a_lo;
                dry2 += 32u;
                neg2 = -32;
            }
        }
    };
    // four steps per trip, one per group of the spare slab; the odd steps afterwards
    if (uni_wave) {
        for (int trip = T >> 2; trip > 0; --trip) {
            one_step(std::integral_constant<int, 0>{}, std::true_type{}, false);
            one_step(std::integral_constant<int, 1>{}, std::true_type{}, false);
            one_step(std::integral_constant<int, 2>{}, std::true_type{}, false);
            one_step(std::integral_constant<int, 3>{}, std::true_type{}, false);
        }
        if ((T & 3) > 0) one_step(std::integral_constant<int, 0>{}, std::true_type{}, false);
        if ((T & 3) > 1) one_step(std::integral_constant<int, 1>{}, std::true_type{}, false);
        if ((T & 3) > 2) one_step(std::integral_constant<int, 2>{}, std::true_type{}, false);
    } else {
        for (int trip = T >> 2; trip > 0; --trip) {
            one_step(std::integral_constant<int, 0>{}, std::false_type{}, by_trip);
            one_step(std::integral_constant<int, 1>{}, std::false_type{}, by_trip);
            one_step(std::integral_constant<int, 2>{}, std::false_type{}, by_trip);
            one_step(std::integral_constant<int, 3>{}, std::false_type{}, by_trip);
        }
        if ((T & 3) > 0) one_step(std::integral_constant<int, 0>{}, std::false_type{}, false);
        if ((T & 3) > 1) one_step(std::integral_constant<int, 1>{}, std::false_type{}, false);
        if ((T & 3) > 2) one_step(std::integral_constant<int, 2>{}, std::false_type{}, false);
    }
    QUAD_STAMP(2);
    // ---- statistics (my player's columns; the game's by lane 0 of the quad) and the hand-over to the replay
    const uint32_t n_ep = wn >> 16, wins = wn & 0xffffu;
    const int done_last = (int)((marks & 0xffffu) >> 1), done_prev = (int)(marks >> 17), last_alive = (int)(marks & 1u);
    const uint32_t tc = tc_in + (uint32_t)T;
    const uint32_t ts = n_ep ? (uint32_t)(T - done_last) : ts_at_entry + (uint32_t)T;
    const int last_len = (n_ep > 1u) ? done_last - done_prev : (int)ts_at_entry + done_last;
    int lw = (last_alive & 1) << p;
    lw |= tron_quad<0xB1>(lw);
    lw |= tron_quad<0x4E>(lw);
    const int ret = 2 * (int)alive_steps - T + 9 * (int)wins;
    int32_t *row = st.results ? st.results + b * (3 + 2 * P) : nullptr;
    uint16_t *pk = st.packed ? st.packed + b * kTronPackedRow<P> : nullptr;
    if (pvalid) {
        const int rs = old_ret + ret;
        const uint32_t wc = old_wins + wins;
        st.ret_sum[p * B + b] = rs;
        st.win_count[p * B + b] = wc;
        if (row) { row[3 + p] = (int32_t)wc; row[3 + P + p] = rs; }
        if (pk) pk[4 + p] = (uint16_t)rs;
    }
    if (p == 0) {
        r_tc[slot] = tc;
        r_steps[slot] = gvalid ? (n_ep ? (int)ts : -1) : 0;
    }
    if (gvalid && p == 0) {
        const uint32_t ne = old_n_ep + n_ep, ls = old_len_sum + (ts_at_entry + (uint32_t)T - ts);
        st.tcount[b] = tc;
        st.tstep[b] = ts;
        st.n_episodes[b] = ne;
        st.len_sum[b] = ls;
        if (n_ep > 0) {
            st.last_winners[b] = (uint8_t)lw;
            st.last_len[b] = (uint16_t)last_len;
        }
        if (row) { row[0] = (int32_t)ne; row[1] = (int32_t)ls; row[2] = n_ep > 0 ? lw : (int32_t)old_last_w; }
        if (pk) { pk[0] = (uint16_t)ne; pk[1] = (uint16_t)ls; pk[2] = (uint16_t)(n_ep > 0 ? (uint32_t)lw : old_last_w); pk[3] = (uint16_t)ts; }
    }
    // the replay as a kernel of its own behind this one (tron_replay_kernel: what it needs is in tcount / tstep by now)
    if (split_replay) { QUAD_STAMP(3); return; }
    __syncthreads();                                            // bit slabs are dead from here; byte slabs reuse the LDS

    // ---- replay of the unfinished episodes on byte slabs: rebuilds board / heads / dirs / deaths (helpers above).  Four
    // turns of 16 games: the workgroup lays out fresh boards, wave `turn` replays with a lane per game, the workgroup
    // copies out.
    for (int turn = 0; turn < kGames / kWaveGames; ++turn) {
        tron_replay_fresh_slabs<RS>(lds0, pad, N, kWaveGames, (int)threadIdx.x, 256);
        __syncthreads();
        if ((int)threadIdx.x < kWaveGames) tron_replay_stamp_heads<P, RS>(cfg, g, lds0 + (int)threadIdx.x * pad.stride);
        __syncthreads();
        const int64_t gbase = (int64_t)blockIdx.x * kGames + turn * kWaveGames;     // first game of this turn
        const int n_turn = (int)((B - gbase) < kWaveGames ? (B - gbase > 0 ? B - gbase : 0) : kWaveGames);
        if (wave == turn) {                                     // a lane per game; lanes 16.. idle along on slabs 0..15
            const int gl = lane & (kWaveGames - 1);
            const bool lvalid = lane < n_turn;                  // (n_turn <= 16)
            const int rsteps = lvalid ? r_steps[turn * kWaveGames + gl] : 0;
            const bool from_start = rsteps >= 0;                // else: from the state the launch came in with
            const int steps_r = lvalid ? (from_start ? rsteps : T) : 0;
            int replay_len = steps_r;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) replay_len = max(replay_len, __shfl_xor(replay_len, off, CRL_WAVE));
            const int bmine = lds0 + gl * pad.stride;
            if (__builtin_amdgcn_ballot_w64(lvalid && !from_start)) {            // somebody resumes from the incoming board
                tron_replay_copy_in<RS>(board + gbase * NN, n_turn, lds0, pad, g, lane);
                if (lvalid && from_start) tron_replay_refresh_slab<P, RS>(cfg, g, bmine);
            }
            tron_replay_lane<P, RS>(cfg, g, pad, bmine, act_lut, lvalid, from_start, steps_r, replay_len,
                                    r_tc[turn * kWaveGames + gl], (uint32_t)(first_env_id + (uint64_t)(lvalid ? gbase + gl : 0)),
                                    seed_lo, seed_hi, gbase + gl, B, heads, dirs, deaths);
        }
        __syncthreads();
        tron_replay_copy_out<RS>(board + gbase * NN, n_turn, lds0, pad, g, (int)threadIdx.x, 256);
        __syncthreads();
    }
#ifdef CRL_QUAD_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    QUAD_STAMP(3);
}

// (2c) the same replay with ONE LANE PER PLAYER: the wave's 64 lanes are 16 games x 4 seats, as in the lane-per-player kernels.
// With a lane per game a wave replays its 16 games on a quarter of its lanes, ~300 vector instructions per step (four players'
// probes, resolution and trail stores unrolled) -- 38 us of pure instruction issue for 65,536 games of ~30 replayed steps;
// here a lane plays its own player (one probe, die or move, one store) and the reference's order is resolved, on DPP-gathered
// copies, only in the wave-steps where a target meets another player's head or target.  A single episode: no tags, no reset,
// cells hold the plain owner.  Every lane runs `replay_len` iterations and takes part in the last steps_r of them.
template <int RS>
__device__ __forceinline__ void tron_replay_quad(const crl_tron_cfg &cfg, const TronGeom &g, const TronPad &pad, const int P,
                                                 const int bmine, const uint8_t *act_lut, const bool gvalid, const bool from_start,
                                                 const int steps_r, const int replay_len, const uint32_t tc_end, const uint32_t gid,
                                                 const uint32_t seed_lo, const uint32_t seed_hi, const int64_t bg, const int64_t B,
                                                 int16_t *__restrict__ heads, int8_t *__restrict__ dirs, int8_t *__restrict__ deaths)
{
    constexpr uint32_t step4 = (uint32_t)((-RS) & 0xff) | (1u << 8) | ((uint32_t)RS << 16) | (0xffu << 24);
    const int N = g.N, NN = g.NN;
    const int p = (int)(threadIdx.x & 3);
    const bool seat = p < P, pvalid = gvalid && seat;
    const int junk = bmine + pad.junk + p;                      // four junk bytes per slab: one per lane of the quad
    const int64_t bbg = gvalid ? bg : 0;
    const int64_t pb = (int64_t)(seat ? p : 0) * B + bbg;
    int fh = cfg.start_heads[0], fd = cfg.start_dirs[0];
    fh = (p == 1) ? cfg.start_heads[1] : fh; fd = (p == 1) ? cfg.start_dirs[1] : fd;
    fh = (p == 2) ? cfg.start_heads[2] : fh; fd = (p == 2) ? cfg.start_dirs[2] : fd;
    fh = (p == 3) ? cfg.start_heads[3] : fh; fd = (p == 3) ? cfg.start_dirs[3] : fd;
    int hc = seat ? fh : 0, d = fd & 3, k = pvalid ? 0 : 1;
    if (!from_start) {                                          // (whole quads: a game's four lanes share from_start)
        hc = min(max((int)heads[pb], 0), NN - 1);
        d = dirs[pb] & 3;
        k = pvalid ? (int)deaths[pb] : 1;
    }
    const int y0 = (int)__umulhi((uint32_t)hc, g.inv_n);
    int h = pvalid ? bmine + (y0 + 1) * RS + (hc - y0 * N) : junk;     // the head as an LDS address
    const int first_r = replay_len - steps_r;                   // this game joins at iteration first_r
    uint32_t c = tc_end - (uint32_t)replay_len;
    uint32_t a_lo = 0, a_hi = 0, acts = 0;
    const uint32_t stamp = (uint32_t)(p + 1);
    for (int t = 0; t < replay_len; ++t, ++c) {
        if (t == 0 || (c & 15u) == 0u) {                        // (whole quads take this branch together: they share c)
            if (t == 0 || (c & 31u) == 0u) tron_quad_actions(gid, c >> 5, p, seed_lo, seed_hi, act_lut, a_lo, a_hi);
            acts = ((c & 16u) ? a_hi : a_lo) >> ((c & 15u) * 2u);
        }
        const bool on = gvalid && t >= first_r;
        const bool run = on && k == 0;
        const int dir = (d + (int)(acts & 3u)) & 3;
        const int tgt = h + __builtin_amdgcn_sbfe((int)step4, dir << 3, 8);
        const int tq = run ? tgt : junk;
        CRL_BOUNDS_IN(tq, bmine, bmine + pad.stride, 141);      // the probe stays inside my game's slab (junk bytes included)
        const uint32_t raw = *(const lds_u8 *)(uintptr_t)(uint32_t)tq;
        const int x1 = tq ^ tron_quad<0x39>(h), x2 = tq ^ tron_quad<0x4E>(h), x3 = tq ^ tron_quad<0x93>(h);
        const int y1 = tq ^ tron_quad<0x39>(tq), y2 = tq ^ tron_quad<0x4E>(tq);
        const uint32_t near = min(min(min(min((uint32_t)x1, (uint32_t)x2), (uint32_t)x3), (uint32_t)y1), (uint32_t)y2);
        const int h_was = h, d_was = d, k_was = k;
        // the common path: each player on its own (correct unless players interact)
        const bool dead = run & (raw != 0u);                    // :47-57 (a wall reads 0xff)
        const bool moved = run ^ dead;                          // :60-62
        k = dead ? (raw == (uint32_t)kWallCell ? p + 1 : (int)raw) : k;
        d = run ? dir : d;                                      // :44
        h = moved ? tgt : h;
        CRL_BOUNDS_IN(moved ? h : junk, bmine, bmine + pad.stride, 142);
        *(lds_u8 *)(uintptr_t)(uint32_t)(moved ? h : junk) = (uint8_t)stamp;
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(near == 0u) != 0ull, 0)) {
            // rare: take the common path's stamp back (the cell was empty), then the quad's four players in the reference's order
            // from the pre-step state, redundantly in its four lanes (tron_rollout_quad_kernel)
            if (moved) *(lds_u8 *)(uintptr_t)(uint32_t)tgt = (uint8_t)0;
            TronRegs<4> s;
            TronProbe<4> pr;
            uint32_t stamp4[4];
            const int kk = on ? k_was : 1;                      // a game that has not joined yet: nobody runs
            s.h[0] = tron_quad<0x00>(h_was); s.h[1] = tron_quad<0x55>(h_was); s.h[2] = tron_quad<0xAA>(h_was); s.h[3] = tron_quad<0xFF>(h_was);
            s.d[0] = tron_quad<0x00>(d_was); s.d[1] = tron_quad<0x55>(d_was); s.d[2] = tron_quad<0xAA>(d_was); s.d[3] = tron_quad<0xFF>(d_was);
            s.k[0] = tron_quad<0x00>(kk); s.k[1] = tron_quad<0x55>(kk); s.k[2] = tron_quad<0xAA>(kk); s.k[3] = tron_quad<0xFF>(kk);
            pr.tgt[0] = tron_quad<0x00>(tgt); pr.tgt[1] = tron_quad<0x55>(tgt); pr.tgt[2] = tron_quad<0xAA>(tgt); pr.tgt[3] = tron_quad<0xFF>(tgt);
            pr.raw[0] = tron_quad<0x00>((int)raw); pr.raw[1] = tron_quad<0x55>((int)raw); pr.raw[2] = tron_quad<0xAA>((int)raw); pr.raw[3] = tron_quad<0xFF>((int)raw);
            pr.ndir[0] = tron_quad<0x00>(dir); pr.ndir[1] = tron_quad<0x55>(dir); pr.ndir[2] = tron_quad<0xAA>(dir); pr.ndir[3] = tron_quad<0xFF>(dir);
#pragma unroll
            for (int q = 0; q < 4; ++q) stamp4[q] = (uint32_t)(q + 1);
            LdsBoard<3> bd{0u};
            bd.within(bmine, bmine + pad.stride);
            const int junk0 = bmine + pad.junk;
            tron_resolve_lds<4>(bd, s, pr, stamp4, junk0);      // trail writes of all four players from every lane: identical
            int hS = s.h[0], dS = s.d[0], kS = s.k[0];
            hS = (p == 1) ? s.h[1] : hS; dS = (p == 1) ? s.d[1] : dS; kS = (p == 1) ? s.k[1] : kS;
            hS = (p == 2) ? s.h[2] : hS; dS = (p == 2) ? s.d[2] : dS; kS = (p == 2) ? s.k[2] : kS;
            hS = (p == 3) ? s.h[3] : hS; dS = (p == 3) ? s.d[3] : dS; kS = (p == 3) ? s.k[3] : kS;
            h = (on && seat) ? hS : h_was;
            d = on ? dS : d_was;
            k = on ? kS : k_was;
        }
        acts >>= 2;
    }
    if (pvalid) {
        const int rel = h - bmine;
        const int row = rel / RS;
        heads[p * B + bg] = (int16_t)((row - 1) * N + (rel - row * RS));
        dirs[p * B + bg] = (int8_t)d;
        deaths[p * B + bg] = (int8_t)k;
    }
}

// The replay of tron_rollout_qbits_kernel as a kernel of its own (round 5).  Inside the bitboard kernel the replay runs in four
// turns per workgroup -- one wave replays 16 games on byte slabs that take the place of the workgroup's bit slabs while the other
// three wait --, and the phase stamps (tools/debug/quad_phases.py <T> 40 qbits) put it at 50-75 us of a launch at 40x40 whatever
// T is, against 20 us for the copy in and 0.27 us per step.  Here every wave of 16 games is its own workgroup with its own byte
// slabs (29.6 KB at 40x40: five per CU), so the chip replays 1,280 waves at a time instead of 1,024 quarter-time ones -- with a
// lane per PLAYER (tron_replay_quad): a lane per game left the replay bound by instruction issue on a quarter of the lanes.  What the
// replay needs it finds where the bitboard kernel has just put it: tcount (the step counter at the end) and tstep (steps into
// the unfinished episode: below T exactly when the game was reset during the launch, else it resumes from the incoming state).
template <int P, bool LARGE>
__global__ void __launch_bounds__(64)
tron_replay_kernel(const crl_tron_cfg cfg, const TronGeom g, const TronPad pad, const int64_t B,
                   const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id, const int T,
                   int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                   int8_t *__restrict__ deaths, const crl_tron_stats st)
{
    constexpr int kWaveGames = 16;
    constexpr int RS = LARGE ? kRowBytesLarge : kRowBytesSmall;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint8_t act_lut[84];
    tron_fill_action_lut(act_lut);
    const int N = g.N, NN = g.NN;
    const int lane = threadIdx.x;
    const int lds0 = (int)(uint32_t)(uintptr_t)(lds_u8 *)lds;
    const int64_t gbase = (int64_t)blockIdx.x * kWaveGames;     // first game of this wave
    const int n_turn = (int)((B - gbase) < kWaveGames ? (B - gbase > 0 ? B - gbase : 0) : kWaveGames);
    const int gl = lane >> 2;                                   // a lane per player: game gl of the wave, seat lane & 3
    const bool lvalid = gl < n_turn;
    const int64_t bg = gbase + gl;
    const uint32_t tc_end = lvalid ? st.tcount[bg] : 0u;
    const int ts = lvalid ? (int)st.tstep[bg] : 0;
    tron_replay_fresh_slabs<RS>(lds0, pad, N, kWaveGames, lane, CRL_WAVE);
    __syncthreads();
    if (lane < kWaveGames) tron_replay_stamp_heads<P, RS>(cfg, g, lds0 + lane * pad.stride);
    __syncthreads();
    const bool from_start = ts < T;                             // else: from the state the launch came in with
    const int steps_r = lvalid ? (from_start ? ts : T) : 0;
    int replay_len = steps_r;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) replay_len = max(replay_len, __shfl_xor(replay_len, off, CRL_WAVE));
    const int bmine = lds0 + gl * pad.stride;
    if (__builtin_amdgcn_ballot_w64(lvalid && !from_start)) {   // somebody resumes from the incoming board
        tron_replay_copy_in<RS>(board + gbase * NN, n_turn, lds0, pad, g, lane);
        // (a slab of a game that was reset was overwritten too: one lane of its quad lays it out again)
        if (lvalid && from_start && (lane & 3) == 0) tron_replay_refresh_slab<P, RS>(cfg, g, bmine);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    tron_replay_quad<RS>(cfg, g, pad, P, bmine, act_lut, lvalid, from_start, steps_r, replay_len, tc_end,
                         (uint32_t)(first_env_id + (uint64_t)(lvalid ? bg : 0)), seed_lo, seed_hi, bg, B, heads, dirs, deaths);
    __syncthreads();
    tron_replay_copy_out<RS>(board + gbase * NN, n_turn, lds0, pad, g, lane, CRL_WAVE);
}

// state sanity for hand-made states: heads inside the board and on their owner's cell
__global__ void __launch_bounds__(256)
tron_check_state_kernel(const int P, const int NN, const int64_t B, const int8_t *__restrict__ board,
                        const int16_t *__restrict__ heads, int32_t *__restrict__ n_bad)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    bool bad = false;
    for (int p = 0; p < P; ++p) {
        const int h = heads[p * B + b];
        bad |= h < 0 || h >= NN || board[b * NN + (h < 0 || h >= NN ? 0 : h)] != p + 1;
    }
    if (bad) atomicAdd(n_bad, 1);
}

// the rollout's random agent for one step (same digits as TronRng, straight from the contract): actions of step c of
// global game g in the crl_tron_step encoding (0 forward, +1 right, -1 left)
template <int P>
__device__ __forceinline__ void tron_sample_actions(const uint32_t g, const uint32_t c, const uint32_t seed_lo,
                                                    const uint32_t seed_hi, int (&act)[P])
{
    const uint32_t j = c & 7u;
#pragma unroll
    for (int q = 0; q < (P + 3) / 4; ++q) {
        const philox_out r = philox4x32_10(g, c >> 3, (uint32_t)q, CRL_TAG_TRON, seed_lo, seed_hi);
        uint32_t v = (j >> 1) == 0 ? r.w[0] : (j >> 1) == 1 ? r.w[1] : (j >> 1) == 2 ? r.w[2] : r.w[3];
        v *= (j & 1u) ? 81u : 1u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (4 * q + i < P) {
                const uint32_t a3 = __umulhi(v, 3u);
                v *= 3u;
                act[4 * q + i] = a3 == 2u ? -1 : (int)a3;
            }
        }
    }
}

template <int P>
__global__ void __launch_bounds__(256)
tron_sample_kernel(const int64_t B, const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id,
                   uint32_t *__restrict__ tcount, const int advance, int8_t *__restrict__ actions)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t c = tcount[b];
    int act[P];
    tron_sample_actions<P>((uint32_t)(first_env_id + (uint64_t)b), c, seed_lo, seed_hi, act);
#pragma unroll
    for (int p = 0; p < P; ++p) actions[(int64_t)p * B + b] = (int8_t)act[p];
    if (advance) tcount[b] = c + 1u;
}

// observation: board relabel is a pure streaming pass, 16 cells per thread.  For P <= 7 the relabelling of observer pl is
// an 8-entry byte table (cell value -> relative id); the P tables sit in LDS, a thread fetches its game's table (one
// ds_read_b64) and relabels 4 cells per v_perm_b32, the board bytes being the selector.
__global__ void __launch_bounds__(256)
tron_observe_board_kernel(const int NN, const int P, const uint32_t inv_cp, const bool nt, const int64_t B, const int8_t *__restrict__ board,
                          const int8_t *__restrict__ player, int8_t *__restrict__ obs, const int mode, const uint64_t inv_nn64)
{
    // mode 1: boards of whole 16-byte chunks (a chunk belongs to one game); mode 2 (round 5): any other board -- the reference's
    // default 19 x 19 --, the batch as ONE byte stream in aligned 16-byte chunks, a chunk that straddles two games relabelled for
    // both observers and merged by a byte mask, the last bytes of a stream that is not whole chunks one by one; mode 0: byte by byte (unaligned buffers)
    __shared__ uint2 lut[CRL_TRON_MAX_P];
    if (threadIdx.x < (unsigned)P) {
        uint32_t lo = 0, hi = 0;
        for (int v = 1; v < 8; ++v) {
            const int n = (v - ((int)threadIdx.x + 1) + P) % P;            // CyTronGrid.pyx:70-71 (every v > 0, also beyond P)
            const uint32_t r = (uint32_t)(n + 1);
            if (v < 4) lo |= r << (8 * v); else hi |= r << (8 * (v - 4));
        }
        lut[threadIdx.x] = make_uint2(lo, hi);
    }
    __syncthreads();
    const int64_t total = B * (int64_t)NN;
    const int64_t n_items = mode ? total / 16 : total;
    // an observer id outside 0..P-1 is not refused by the reference: the rolled vectors use numpy's modulo
    // (TronGridEnvironment.py:393), the board goes through C's remainder (CyTronGrid.pyx:1 cdivision=True, :71).
    // Up to pr = P the operand v - (pr + 1) + P stays >= 0 and both agree (= observer pr mod P: the table);
    // beyond, low trail ids come out <= 0 -- reproduced by the arithmetic branch
    auto relabel = [&](const uint32_t (&w)[4], const int pr, uint32_t (&o)[4]) {
        const int pl = pr + 1;
        if (P <= 7 && pr <= P) {
            int pm = pr % P;
            pm = pm < 0 ? pm + P : pm;
            const uint2 t = lut[pm];
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = __builtin_amdgcn_perm(t.y, t.x, w[q]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t r = 0;
#pragma unroll
                for (int s8 = 0; s8 < 32; s8 += 8) {
                    const int c = (int)(int8_t)((w[q] >> s8) & 0xffu);
                    const int n = (c - pl + P) % P;                        // CyTronGrid.pyx:70-71, C remainder
                    r |= (uint32_t)((c > 0 ? n + 1 : c) & 0xff) << s8;
                }
                o[q] = r;
            }
        }
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (mode == 1) {
            const int64_t off = i * 16;
            // the game of chunk i: i / (chunks per board) -- an exact 32-bit multiply-high where the chunk index fits
            const int64_t game = (i >> 32) == 0 && inv_cp ? (int64_t)__umulhi((uint32_t)i, inv_cp) : off / NN;
            const uint4 v = *reinterpret_cast<const uint4 *>(board + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
            relabel(w, player[game], o);
            crl_stream_store16(obs + off, make_uint4(o[0], o[1], o[2], o[3]), nt);
        } else if (mode == 2) {
            const int64_t off = i * 16;
            const int64_t e0 = (int64_t)__umul64hi((uint64_t)off, inv_nn64);    // off / NN (exact: off * NN < 2^64)
            const int first = NN - (int)(off - e0 * NN);                        // bytes of the chunk that belong to game e0
            const uint4 v = *reinterpret_cast<const uint4 *>(board + off);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            uint32_t o[4];
            const int pr0 = player[e0];
            relabel(w, pr0, o);
            if (first < 16) {
                const int pr1 = player[e0 + 1];
                if (pr1 != pr0) {
                    uint32_t o1[4];
                    relabel(w, pr1, o1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t m = 0;                                         // the bytes of dword q that belong to e0
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) m |= (4 * q + k2 < first) ? (0xffu << (8 * k2)) : 0u;
                        o[q] = (o[q] & m) | (o1[q] & ~m);
                    }
                }
            }
            crl_stream_store16(obs + off, make_uint4(o[0], o[1], o[2], o[3]), nt);
        } else {
            const int pl = (int)player[i / NN] + 1;
            const int c = board[i];
            const int n = (c - pl + P) % P;                                    // C remainder, as above
            obs[i] = (int8_t)(c > 0 ? n + 1 : c);
        }
    }
    // mode 2 on a stream that is not whole chunks (a batch that is not a multiple of 16 games): its last bytes one by one
    if (mode == 2 && blockIdx.x == 0 && (int64_t)threadIdx.x < total - n_items * 16) {
        const int64_t i = n_items * 16 + threadIdx.x;
        const int pl = (int)player[i / NN] + 1;
        const int c = board[i];
        const int n = (c - pl + P) % P;
        obs[i] = (int8_t)(c > 0 ? n + 1 : c);
    }
}

template <int P>
__global__ void __launch_bounds__(256)
tron_observe_players_kernel(const int64_t B, const int16_t *__restrict__ heads, const int8_t *__restrict__ dirs,
                            const int8_t *__restrict__ deaths, const int8_t *__restrict__ player,
                            int16_t *__restrict__ oh, int8_t *__restrict__ od, int8_t *__restrict__ ok)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int pl = (int)player[b] % P;                                     // rolled_idx = (arange + player) % P with numpy's modulo
    pl = pl < 0 ? pl + P : pl;                                       // (TronGridEnvironment.py:393): any integer is an observer
    int h[P], d[P], k[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { h[p] = heads[p * B + b]; d[p] = dirs[p * B + b]; k[p] = deaths[p * B + b]; }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        int hh = 0, dd = 0, kk = 0;
        int src = i + pl;                                // TronGridEnvironment.py:392-396
        src = src >= P ? src - P : src;
#pragma unroll
        for (int q = 0; q < P; ++q) { hh = (q == src) ? h[q] : hh; dd = (q == src) ? d[q] : dd; kk = (q == src) ? k[q] : kk; }
        oh[i * B + b] = (int16_t)hh;
        od[i * B + b] = (int8_t)dd;
        ok[i * B + b] = (int8_t)kk;
    }
}

// compute_ranking (TronGridEnvironment.py:483-508): one wave per game.  Lanes count the cells of every player
// in a strided sweep of the board, a wave reduction gives the trail lengths, then the mutual-kill tie rule
// (including its read of deaths[-1] for alive players and the missing-Counter-key 0) and competition ranks.
template <int P>
__global__ void __launch_bounds__(256)
tron_ranking_kernel(const int NN, const int64_t B, const int8_t *__restrict__ board, const int8_t *__restrict__ deaths,
                    int8_t *__restrict__ rank)
{
    const int lane = threadIdx.x & (CRL_WAVE - 1);
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / CRL_WAVE) + (threadIdx.x >> 6);
    if (b >= B) return;
    int cnt[P];
#pragma unroll
    for (int p = 0; p < P; ++p) cnt[p] = 0;
    for (int c = lane; c < NN; c += CRL_WAVE) {
        const int v = board[b * NN + c];
#pragma unroll
        for (int p = 0; p < P; ++p) cnt[p] += (v == p + 1);
    }
    int score[P], k[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int v = cnt[p];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, CRL_WAVE);
        score[p] = v;
        k[p] = deaths[p * B + b];
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {                                // :492-495, ascending like np.where
        const int via = k[i] > 0 ? k[i] - 1 : P - 1;            // python index -1 = the last player
        int kvia = 0, other = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            kvia = (q == via) ? k[q] : kvia;
            other = (k[i] > 0 && q == k[i] - 1) ? score[q] : other;   // alive: scores[-1] is a missing key -> 0
        }
        if (kvia == i + 1 && other < score[i]) score[i] = other;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < P; ++i) {                            // :497-506 competition ranking
            int higher = 0;
#pragma unroll
            for (int q = 0; q < P; ++q) higher += score[q] > score[i];
            rank[i * B + b] = (int8_t)higher;
        }
    }
}

// compute_ranking for boards made of whole 16-byte chunks: a workgroup takes 64 games, streams their boards with
// coalesced 16-byte loads (the one-wave-per-game kernel above reads 64 single bytes per load), counts the cells of every
// player per chunk with byte-parallel compares (bit 7 of ((x & 0x7f..) + 0x7f..) | x is set iff byte x != 0, x = cell ^ id),
// adds the chunk counts into per-game LDS counters, and one lane per game then applies the same tie rule and ranking.
template <int P>
__global__ void __launch_bounds__(256)
tron_ranking_wide_kernel(const int NN, const uint32_t inv_cp, const int64_t B, const int8_t *__restrict__ board,
                         const int8_t *__restrict__ deaths, int8_t *__restrict__ rank)
{
    constexpr int G = 64;
    __shared__ uint32_t cnt[G][CRL_TRON_MAX_P];
    for (int i = threadIdx.x; i < G * CRL_TRON_MAX_P; i += 256) (&cnt[0][0])[i] = 0u;
    __syncthreads();
    const int cp = NN >> 4;
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int n_game = (int)((B - g0) < G ? (B - g0) : G);
    const int total = n_game * cp;
    for (int c = threadIdx.x; c < total; c += 256) {
        const int e = cp == 1 ? c : (int)__umulhi((uint32_t)c, inv_cp);
        const uint4 v = *reinterpret_cast<const uint4 *>(board + g0 * NN + (int64_t)c * 16);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint32_t n = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = w[q] ^ (0x01010101u * (uint32_t)(p + 1));
                const uint32_t nz = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;   // bit 7 per NON-matching byte
                n += 4u - (uint32_t)__popc(nz);
            }
            if (n) atomicAdd(&cnt[e][p], n);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x >= n_game) return;
    const int64_t b = g0 + threadIdx.x;
    int score[P], k[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { score[p] = (int)cnt[threadIdx.x][p]; k[p] = deaths[p * B + b]; }
#pragma unroll
    for (int i = 0; i < P; ++i) {                                // :492-495, ascending like np.where
        const int via = k[i] > 0 ? k[i] - 1 : P - 1;            // python index -1 = the last player
        int kvia = 0, other = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            kvia = (q == via) ? k[q] : kvia;
            other = (k[i] > 0 && q == k[i] - 1) ? score[q] : other;   // alive: scores[-1] is a missing key -> 0
        }
        if (kvia == i + 1 && other < score[i]) score[i] = other;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {                                // :497-506 competition ranking
        int higher = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) higher += score[q] > score[i];
        rank[i * B + b] = (int8_t)higher;
    }
}

// The same for boards that are NOT whole 16-byte chunks (19 x 19, the reference's default): a workgroup's 64 boards are one
// contiguous, aligned run of bytes and go 16 at a time as a flat stream; a chunk that straddles two games splits its counts
// by a byte mask.  (Round 5: 32 -> 8 us at 65,536 games of 19 x 19.)
template <int P>
__global__ void __launch_bounds__(256)
tron_ranking_flat_kernel(const int NN, const uint32_t inv_nn, const int64_t B, const int8_t *__restrict__ board,
                         const int8_t *__restrict__ deaths, int8_t *__restrict__ rank)
{
    constexpr int G = 64;
    __shared__ uint32_t cnt[G + 1][CRL_TRON_MAX_P];
    for (int i = threadIdx.x; i < (G + 1) * CRL_TRON_MAX_P; i += 256) (&cnt[0][0])[i] = 0u;
    __syncthreads();
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int n_game = (int)((B - g0) < G ? (B - g0) : G);
    const int bytes = n_game * NN, chunks = bytes >> 4;
    const int8_t *src = board + g0 * NN;                        // 64 * NN bytes per workgroup: a 16-byte boundary
    for (int c = threadIdx.x; c < chunks; c += 256) {
        const int byte0 = c << 4;
        const int e0 = (int)__umulhi((uint32_t)byte0, inv_nn);
        const int first = NN - (byte0 - e0 * NN);               // bytes of the chunk that belong to game e0 (>= 16: all)
        const uint4 v = *reinterpret_cast<const uint4 *>(src + byte0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t m[4];                                          // bit 7 of the bytes that belong to e0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t mq = 0;
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) mq |= (4 * q + k2 < first) ? (0x80u << (8 * k2)) : 0u;
            m[q] = mq;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint32_t n0 = 0, n1 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t x = w[q] ^ (0x01010101u * (uint32_t)(p + 1));
                const uint32_t eq = ~((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x)) & 0x80808080u;   // bit 7 per MATCHING byte
                n0 += (uint32_t)__popc(eq & m[q]);
                n1 += (uint32_t)__popc(eq & ~m[q]);
            }
            if (n0) atomicAdd(&cnt[e0][p], n0);
            if (n1) atomicAdd(&cnt[e0 + 1][p], n1);
        }
    }
    for (int i = (chunks << 4) + threadIdx.x; i < bytes; i += 256) {   // the last workgroup's odd bytes
        const int e = (int)__umulhi((uint32_t)i, inv_nn);
        const int c = src[i];
        if (c >= 1 && c <= P) atomicAdd(&cnt[e][c - 1], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x >= n_game) return;
    const int64_t b = g0 + threadIdx.x;
    int score[P], k[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { score[p] = (int)cnt[threadIdx.x][p]; k[p] = deaths[p * B + b]; }
#pragma unroll
    for (int i = 0; i < P; ++i) {                                // :492-495, ascending like np.where
        const int via = k[i] > 0 ? k[i] - 1 : P - 1;            // python index -1 = the last player
        int kvia = 0, other = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            kvia = (q == via) ? k[q] : kvia;
            other = (k[i] > 0 && q == k[i] - 1) ? score[q] : other;   // alive: scores[-1] is a missing key -> 0
        }
        if (kvia == i + 1 && other < score[i]) score[i] = other;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {                                // :497-506 competition ranking
        int higher = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) higher += score[q] > score[i];
        rank[i * B + b] = (int8_t)higher;
    }
}

// state_to_observation for ALL P observers in one pass over the boards: read 16 cells once, write them P times
// relabelled (observer p sees itself as 1: CyTronGrid.pyx:65-71).  For P <= 7 the relabelling of four cells is ONE
// v_perm_b32: the 8-byte table lut_p[v] = (v == 0 ? 0 : (v - (p+1) + P) % P + 1) is the permute source and the four
// board bytes (values 0..7) are its selector.  Pure streaming: N*N bytes in, P*N*N bytes out per game.
template <int P>
__global__ void __launch_bounds__(256)
tron_observe_all_kernel(const int NN, const int64_t B, const int8_t *__restrict__ board, int8_t *__restrict__ obs, const bool nt,
                        const int wide)
{
    // `wide`: the relabelling does not depend on the game a cell belongs to, so the boards of the batch are ONE byte stream and
    // go 16 bytes at a time whenever the stream (B * N * N bytes, hence every observer's plane) is a whole number of aligned
    // chunks (1) -- also for boards that are not, like the reference's default 19 x 19 (round 5: 36 -> 20 us at 65,536 games).
    // (2): aligned buffers, but a stream that is NOT whole chunks (a batch that is not a multiple of 16 games of such a board):
    // observer p's plane then starts p * total bytes in, off the 16-byte grid by its own amount.  Every plane is still written in
    // aligned chunks: the chunk at plane offset d_p + 16 i (d_p = the plane's bytes up to its first boundary) is bytes
    // 16 i + d_p .. + 15 of the input, i.e. input chunks i and i + 1 funnelled by d_p bytes; the < 16 leading and < 32 trailing
    // bytes of each plane go one by one (workgroup 0).  0: byte by byte (unaligned buffers).
    const int64_t total = B * (int64_t)NN;
    const int64_t n_full = total / 16;
    const int64_t n_items = wide ? n_full : total;
    uint32_t lut_lo[P], lut_hi[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        lut_lo[p] = 0; lut_hi[p] = 0;
#pragma unroll
        for (int v = 1; v < 8; ++v) {
            int n = v - (p + 1);
            n = n < 0 ? n + P : n;
            const uint32_t r = (v <= P) ? (uint32_t)(n + 1) : (uint32_t)v;
            if (v < 4) lut_lo[p] |= r << (8 * v); else lut_hi[p] |= r << (8 * (v - 4));
        }
    }
    auto relabel_byte = [&](const int c, const int p) -> int {
        int n = c - (p + 1);
        n = n < 0 ? n + P : n;
        return c > 0 ? n + 1 : c;
    };
    auto relabel16 = [&](const uint4 v, const int p) -> uint4 {
        uint4 o;
        if (P <= 7) {
            o.x = __builtin_amdgcn_perm(lut_hi[p], lut_lo[p], v.x);
            o.y = __builtin_amdgcn_perm(lut_hi[p], lut_lo[p], v.y);
            o.z = __builtin_amdgcn_perm(lut_hi[p], lut_lo[p], v.z);
            o.w = __builtin_amdgcn_perm(lut_hi[p], lut_lo[p], v.w);
        } else {
            uint32_t w[4] = {v.x, v.y, v.z, v.w}, r4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t r = 0;
#pragma unroll
                for (int s8 = 0; s8 < 32; s8 += 8) r |= (uint32_t)(relabel_byte((int)((w[q] >> s8) & 0xffu), p) & 0xff) << s8;
                r4[q] = r;
            }
            o = make_uint4(r4[0], r4[1], r4[2], r4[3]);
        }
        return o;
    };
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * blockDim.x) {
        if (wide == 1) {
            const int64_t off = i * 16;
            const uint4 v = *reinterpret_cast<const uint4 *>(board + off);
#pragma unroll
            for (int p = 0; p < P; ++p) crl_stream_store16(obs + (int64_t)p * total + off, relabel16(v, p), nt);
        } else if (wide == 2) {
            const int64_t off = i * 16;
            const bool more = i + 1 < n_full;                    // (the next chunk is a whole one too)
            const uint4 v0 = *reinterpret_cast<const uint4 *>(board + off);
            const uint4 v1 = *reinterpret_cast<const uint4 *>(board + (more ? off + 16 : off));
            const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int d = (int)((16u - (uint32_t)(((int64_t)p * total) & 15)) & 15u);     // (uniform)
                if (d == 0) {
                    crl_stream_store16(obs + (int64_t)p * total + off, relabel16(v0, p), nt);
                } else if (more) {
                    const int dq = d >> 2;
                    const uint32_t r = (uint32_t)(d & 3);
                    uint32_t x[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) x[j] = dq == 0 ? w[j] : dq == 1 ? w[j + 1] : dq == 2 ? w[j + 2] : w[j + 3];
                    const uint4 f = make_uint4(__builtin_amdgcn_alignbyte(x[1], x[0], r), __builtin_amdgcn_alignbyte(x[2], x[1], r),
                                               __builtin_amdgcn_alignbyte(x[3], x[2], r), __builtin_amdgcn_alignbyte(x[4], x[3], r));
                    crl_stream_store16(obs + (int64_t)p * total + off + d, relabel16(f, p), nt);
                }
            }
        } else {
            const int c = board[i];
#pragma unroll
            for (int p = 0; p < P; ++p) obs[(int64_t)p * total + i] = (int8_t)relabel_byte(c, p);
        }
    }
    if (wide == 2 && blockIdx.x == 0) {
        // what the chunks left of every plane: its first d_p bytes and the bytes behind its last whole chunk
        for (int t = threadIdx.x; t < P * 48; t += blockDim.x) {
            const int p = t / 48, j = t - p * 48;
            const int d = (int)((16u - (uint32_t)(((int64_t)p * total) & 15)) & 15u);
            const int64_t stored = d == 0 ? n_full : (n_full > 0 ? n_full - 1 : 0);
            const int64_t idx = j < d ? (int64_t)j : (int64_t)d + 16 * stored + (j - d);
            if (idx < total && (j < d || idx >= (int64_t)d + 16 * stored))
                obs[(int64_t)p * total + idx] = (int8_t)relabel_byte((int)board[idx], p);
        }
    }
}

// per-player vectors of all P observations: out[p][i][b] = in[(i + p) % P][b]
template <int P>
__global__ void __launch_bounds__(256)
tron_observe_all_players_kernel(const int64_t B, const int16_t *__restrict__ heads, const int8_t *__restrict__ dirs,
                                const int8_t *__restrict__ deaths, int16_t *__restrict__ oh, int8_t *__restrict__ od,
                                int8_t *__restrict__ ok)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int h[P], d[P], k[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { h[p] = heads[p * B + b]; d[p] = dirs[p * B + b]; k[p] = deaths[p * B + b]; }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int src = (i + p) % P;                 // compile-time after unrolling
            oh[((int64_t)p * P + i) * B + b] = (int16_t)h[src];
            od[((int64_t)p * P + i) * B + b] = (int8_t)d[src];
            ok[((int64_t)p * P + i) * B + b] = (int8_t)k[src];
        }
}

// ---- fused per-step call: [sample ->] next_state (auto-reset) -> state_to_observation of ALL P observers ------------
// What a self-play learner runs every step (TronGridEnvironment.next_state :265-323 + state_to_observation :363-420 for
// every player).  As three launches (crl_tron_sample, crl_tron_step, crl_tron_observe_all) the board is probed with
// scattered byte loads, rewritten, and then read again in full; here a workgroup reads its G boards ONCE, coalesced,
// into LDS, one lane per game plays the step there (the same tron_step_core as crl_tron_step), and all 256 threads
// stream the P relabelled copies out of LDS (one v_perm_b32 per 4 cells and observer).  HBM sees N*N bytes in and
// P*N*N bytes out per game, plus the <= P trail bytes (byte stores) and, for the games that were reset, one fresh board.
struct DualBoard {                  // step on the LDS copy, mirror the trail writes to the canonical HBM board
    uint8_t *l;
    int8_t *g;
    CRL_CELLS_MEMBER
    __device__ __forceinline__ int raw(const int c) const { CRL_CELLS_CHECK(c, 132); return l[c]; }
    __device__ __forceinline__ int owner(const int r) const { return r; }
    __device__ __forceinline__ void put(const int c, const int who) const { CRL_CELLS_CHECK(c, 133); l[c] = (uint8_t)who; g[c] = (int8_t)who; }
};

// four cells of an observation for observer p: ONE v_perm_b32 through the 8-entry table when P <= 7; with eight players the
// cell values 0..8 do not fit it and the bytes are relabelled by arithmetic (CyTronGrid.pyx:70-71)
template <int P>
__device__ __forceinline__ uint32_t tron_relabel4(const uint32_t w, const int p, const uint32_t lut_lo, const uint32_t lut_hi)
{
    if constexpr (P <= 7) {
        return __builtin_amdgcn_perm(lut_hi, lut_lo, w);
    } else {
        uint32_t r = 0;
#pragma unroll
        for (int s8 = 0; s8 < 32; s8 += 8) {
            const int c = (int)((w >> s8) & 0xffu);
            int n = c - (p + 1);
            n = n < 0 ? n + P : n;
            r |= (uint32_t)((c > 0 ? n + 1 : c) & 0xff) << s8;
        }
        return r;
    }
}

constexpr uint32_t kStepObserveNT = 0x80000000u;            // kernel-side flag bit next to CRL_STEP_AUTO_RESET

template <int P, int G>
__global__ void __launch_bounds__(256)
tron_step_observe_kernel(const crl_tron_cfg cfg, const TronGeom g, const uint32_t inv_cp, const int64_t B,
                         const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id,
                         int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                         int8_t *__restrict__ deaths, const int8_t *__restrict__ actions, uint32_t *__restrict__ tcount,
                         int8_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                         int8_t *__restrict__ obs_board, int16_t *__restrict__ oh, int8_t *__restrict__ od,
                         int8_t *__restrict__ ok, const uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int NN = g.NN, SLAB = NN + 16;                        // NN % 16 == 0: 16-byte LDS accesses stay aligned
    const int cp = NN >> 4;                                     // 16-byte chunks per board
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int n_game = (int)((B - g0) < G ? (B - g0) : G);
    const int total = n_game * cp;
    uint8_t *rflag = lds + G * SLAB;                            // [G] this game was reset by the step
    const bool nt = (flags & kStepObserveNT) != 0;              // nontemporal observation stores (see crl_stream_nt)
    // the stepping lanes' player vectors (and step counter): issued FIRST, in flight together with the board loads of
    // phase A instead of a second round of memory latency behind the barrier (round 5: -4 % out of cache)
    const int e = threadIdx.x;
    const bool stepper = e < G, valid = stepper && e < n_game;
    const int64_t b = g0 + (valid ? e : 0);
    TronRegs<P> s;
    int act[P], rew[P];
    uint32_t c_in = 0u;
    auto load_players = [&]() {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            s.h[p] = valid ? heads[p * B + b] : 0;
            s.d[p] = valid ? dirs[p * B + b] : 0;
            s.k[p] = valid ? deaths[p * B + b] : 1;
            act[p] = (valid && actions) ? actions[p * B + b] : 0;
        }
        if (!actions) c_in = valid ? tcount[b] : 0u;
    };
    if (CRL_SO_HOIST && stepper) load_players();
    // ---- phase A: boards HBM -> LDS, coalesced 16-byte loads, up to 4 in flight per thread
    for (int base = threadIdx.x; base < total; base += 4 * 256) {
        uint4 v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * 256;
            const int cc = c < total ? c : 0;
            const int ee = cp == 1 ? cc : (int)__umulhi((uint32_t)cc, inv_cp);
            const int off = (cc - ee * cp) << 4;
            dst[u] = c < total ? ee * SLAB + off : -1;
            v[u] = *reinterpret_cast<const uint4 *>(board + (g0 + ee) * NN + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4 *>(lds + dst[u]) = v[u];
    }
    __syncthreads();
    // ---- phase B: one lane per game plays the step on its LDS board
    if (stepper) {
        if (!CRL_SO_HOIST) load_players();
        if (!actions) {                                         // the rollout's random agent at this game's step counter
            tron_sample_actions<P>((uint32_t)(first_env_id + (uint64_t)b), c_in, seed_lo, seed_hi, act);
            if (valid) tcount[b] = c_in + 1u;
        }
        tron_split_heads<P>(g, s);
        int term, wm;
        const DualBoard bd{lds + e * SLAB, board + b * NN CRL_CELLS_INIT(NN)};
        tron_step_core<P>(g, bd, valid, s, act, rew, term, wm);
        const bool do_reset = valid && term && (flags & CRL_STEP_AUTO_RESET);
        rflag[e] = do_reset ? 1 : 0;
        if (do_reset) tron_regs_to_start<P>(cfg, g, s);         // phase C writes the fresh board (LDS slab is ignored)
        if (valid) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                rewards[p * B + b] = (int8_t)rew[p];
                heads[p * B + b] = (int16_t)s.h[p];
                dirs[p * B + b] = (int8_t)s.d[p];
                deaths[p * B + b] = (int8_t)s.k[p];
            }
            terminal[b] = (uint8_t)term;
            winners[b] = (uint8_t)wm;
#pragma unroll
            for (int p = 0; p < P; ++p)                         // TronGridEnvironment.py:392-396: rolled so index 0 is the observer
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const int src = (i + p) % P;                // compile-time after unrolling
                    oh[((int64_t)p * P + i) * B + b] = (int16_t)s.h[src];
                    od[((int64_t)p * P + i) * B + b] = (int8_t)s.d[src];
                    ok[((int64_t)p * P + i) * B + b] = (int8_t)s.k[src];
                }
        }
    }
    __syncthreads();
    // ---- phase C: stream the P relabelled copies out (CyTronGrid.pyx:65-71 as one v_perm_b32 per 4 cells)
    uint32_t lut_lo[P], lut_hi[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        lut_lo[p] = 0; lut_hi[p] = 0;
#pragma unroll
        for (int v = 1; v < 8; ++v) {
            int n = v - (p + 1);
            n = n < 0 ? n + P : n;
            const uint32_t r = (v <= P) ? (uint32_t)(n + 1) : (uint32_t)v;
            if (v < 4) lut_lo[p] |= r << (8 * v); else lut_hi[p] |= r << (8 * (v - 4));
        }
    }
    const int64_t plane = B * (int64_t)NN;                      // one observer's boards
    for (int c = threadIdx.x; c < total; c += 256) {
        const int e = cp == 1 ? c : (int)__umulhi((uint32_t)c, inv_cp);
        const int off = (c - e * cp) << 4;
        uint4 v = *reinterpret_cast<const uint4 *>(lds + e * SLAB + off);
        const int64_t gofs = (g0 + e) * NN + off;
        if (rflag[e]) {                                         // new_state: the fresh board replaces the finished one
            v = tron_fresh_chunk16<P>(cfg, off);
            *reinterpret_cast<uint4 *>(board + gofs) = v;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint4 o;
            o.x = tron_relabel4<P>(v.x, p, lut_lo[p], lut_hi[p]);
            o.y = tron_relabel4<P>(v.y, p, lut_lo[p], lut_hi[p]);
            o.z = tron_relabel4<P>(v.z, p, lut_lo[p], lut_hi[p]);
            o.w = tron_relabel4<P>(v.w, p, lut_lo[p], lut_hi[p]);
            crl_stream_store16(obs_board + (int64_t)p * plane + gofs, o, nt);
        }
    }
}

// The same fused call for boards that are NOT whole 16-byte chunks -- the reference's default 19 x 19 (361 cells) among them --,
// round 5.  A workgroup's G boards are ONE contiguous run of G * N * N bytes, a whole number of aligned chunks whatever N is
// (G = 64 or 16), so they go HBM -> LDS -> HBM 16 bytes at a time as a flat stream: the relabelling of phase C does not care
// which game a cell belongs to, only the stepping lanes (phase B, byte accesses at e * N * N) and the boards of the games that
// were reset do -- a chunk may straddle two games there, and takes its fresh bytes per half.  Needs B % 16 == 0 (every
// workgroup's run, and every observer's plane of the output, then starts on a 16-byte boundary); other batches keep the
// one-game-per-workgroup kernel below.  65,536 games of 19 x 19: 161 us -> the fused kernel's rate.
template <int P>
__device__ __forceinline__ uint4 tron_fresh_flat16(const crl_tron_cfg &cfg, const int r, const int NN)
{
    // bytes j = 0..15 of a fresh-board stream starting at cell r of a game: cell (r + j) mod NN
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < P; ++p) {
        int d = (int)cfg.start_heads[p] - r;
        d += d < 0 ? NN : 0;
        if (d < 16) {
            const uint32_t byte = (uint32_t)(p + 1) << ((d & 3) * 8);
            w[0] |= (d >> 2) == 0 ? byte : 0u;
            w[1] |= (d >> 2) == 1 ? byte : 0u;
            w[2] |= (d >> 2) == 2 ? byte : 0u;
            w[3] |= (d >> 2) == 3 ? byte : 0u;
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <int P, int G, bool RAGGED = false>
__global__ void __launch_bounds__(256)
tron_step_observe_flat_kernel(const crl_tron_cfg cfg, const TronGeom g, const uint32_t inv_nn, const int64_t B,
                              const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id,
                              int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                              int8_t *__restrict__ deaths, const int8_t *__restrict__ actions, uint32_t *__restrict__ tcount,
                              int8_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                              int8_t *__restrict__ obs_board, int16_t *__restrict__ oh, int8_t *__restrict__ od,
                              int8_t *__restrict__ ok, const uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int NN = g.NN;
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int n_game = (int)((B - g0) < G ? (B - g0) : G);      // a multiple of 16 (B % 16 == 0) unless RAGGED
    const int run_bytes = n_game * NN;
    const int chunks = run_bytes >> 4;                          // ... so this is exact (RAGGED: the whole chunks of the last workgroup's run)
    uint8_t *rflag = lds + ((G * NN + 15) & ~15);               // [G + 1] this game was reset by the step ([n_game] = 0: read, never set)
    const bool nt = (flags & kStepObserveNT) != 0;
    const int e = threadIdx.x;
    const bool stepper = e < G, valid = stepper && e < n_game;
    const int64_t b = g0 + (valid ? e : 0);
    TronRegs<P> s;
    int act[P], rew[P];
    uint32_t c_in = 0u;
    if (stepper) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            s.h[p] = valid ? heads[p * B + b] : 0;
            s.d[p] = valid ? dirs[p * B + b] : 0;
            s.k[p] = valid ? deaths[p * B + b] : 1;
            act[p] = (valid && actions) ? actions[p * B + b] : 0;
        }
        if (!actions) c_in = valid ? tcount[b] : 0u;
    }
    if (threadIdx.x == 0) rflag[G] = 0;
    // ---- phase A: the workgroup's boards as one stream, HBM -> LDS, up to 4 loads in flight per thread
    const int8_t *src = board + g0 * NN;
    for (int base = threadIdx.x; base < chunks; base += 4 * 256) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * 256;
            v[u] = *reinterpret_cast<const uint4 *>(src + ((c < chunks ? c : 0) << 4));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * 256;
            if (c < chunks) *reinterpret_cast<uint4 *>(lds + (c << 4)) = v[u];
        }
    }
    if constexpr (RAGGED)                                       // the bytes behind the run's last whole chunk
        for (int i = (chunks << 4) + (int)threadIdx.x; i < run_bytes; i += 256) lds[i] = (uint8_t)src[i];
    __syncthreads();
    // ---- phase B: one lane per game plays the step on its LDS board (byte accesses at e * NN)
    if (stepper) {
        if (!actions) {
            tron_sample_actions<P>((uint32_t)(first_env_id + (uint64_t)b), c_in, seed_lo, seed_hi, act);
            if (valid) tcount[b] = c_in + 1u;
        }
        tron_split_heads<P>(g, s);
        int term, wm;
        const DualBoard bd{lds + (valid ? e : 0) * NN, board + b * NN CRL_CELLS_INIT(NN)};
        tron_step_core<P>(g, bd, valid, s, act, rew, term, wm);
        const bool do_reset = valid && term && (flags & CRL_STEP_AUTO_RESET);
        rflag[e] = do_reset ? 1 : 0;
        if (do_reset) tron_regs_to_start<P>(cfg, g, s);
        if (valid) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                rewards[p * B + b] = (int8_t)rew[p];
                heads[p * B + b] = (int16_t)s.h[p];
                dirs[p * B + b] = (int8_t)s.d[p];
                deaths[p * B + b] = (int8_t)s.k[p];
            }
            terminal[b] = (uint8_t)term;
            winners[b] = (uint8_t)wm;
#pragma unroll
            for (int p = 0; p < P; ++p)                         // TronGridEnvironment.py:392-396: rolled so index 0 is the observer
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const int src_p = (i + p) % P;
                    oh[((int64_t)p * P + i) * B + b] = (int16_t)s.h[src_p];
                    od[((int64_t)p * P + i) * B + b] = (int8_t)s.d[src_p];
                    ok[((int64_t)p * P + i) * B + b] = (int8_t)s.k[src_p];
                }
        }
    }
    __syncthreads();
    // ---- phase C: stream the P relabelled copies out; the boards of reset games are replaced on the way
    uint32_t lut_lo[P], lut_hi[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        lut_lo[p] = 0; lut_hi[p] = 0;
#pragma unroll
        for (int v = 1; v < 8; ++v) {
            int n = v - (p + 1);
            n = n < 0 ? n + P : n;
            const uint32_t r = (v <= P) ? (uint32_t)(n + 1) : (uint32_t)v;
            if (v < 4) lut_lo[p] |= r << (8 * v); else lut_hi[p] |= r << (8 * (v - 4));
        }
    }
    const int64_t plane = B * (int64_t)NN;
    if constexpr (RAGGED) {
        // A batch that is not a multiple of 16 games: observer p's plane starts p * B * N * N bytes in, off the 16-byte grid by
        // its own amount, and the last workgroup's run is not whole chunks.  C1: the fresh boards of the games that were reset
        // are patched into the LDS image (and stored to `board`, whose chunks ARE aligned); C2: every plane goes out in ALIGNED
        // chunks all the same -- the chunk at run offset d_p + 16 k is LDS chunks k and k + 1 funnelled by d_p bytes -- and the
        // < 16 bytes at either end of the run one by one.
        const int chunks_up = (run_bytes + 15) >> 4;
        for (int c = threadIdx.x; c < chunks_up; c += 256) {
            const int byte0 = c << 4;
            const int e0 = (int)__umulhi((uint32_t)byte0, inv_nn);
            const int r = byte0 - e0 * NN, first = NN - r;
            const bool f0 = rflag[e0] != 0, f1 = first < 16 && e0 + 1 < n_game && rflag[e0 + 1] != 0;
            if (f0 | f1) {
                uint4 v = *reinterpret_cast<const uint4 *>(lds + byte0);
                const uint4 fr = tron_fresh_flat16<P>(cfg, r, NN);
                uint32_t m[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t mq = 0;
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) mq |= ((4 * q + k2 < first) ? f0 : f1) ? (0xffu << (8 * k2)) : 0u;
                    m[q] = mq;
                }
                v.x = (v.x & ~m[0]) | (fr.x & m[0]); v.y = (v.y & ~m[1]) | (fr.y & m[1]);
                v.z = (v.z & ~m[2]) | (fr.z & m[2]); v.w = (v.w & ~m[3]) | (fr.w & m[3]);
                *reinterpret_cast<uint4 *>(lds + byte0) = v;    // (the LDS image is whole chunks long)
                if (byte0 + 16 <= run_bytes) {
                    *reinterpret_cast<uint4 *>(board + g0 * NN + byte0) = v;
                } else {
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                    for (int j = 0; byte0 + j < run_bytes; ++j) board[g0 * NN + byte0 + j] = (int8_t)((w[j >> 2] >> (8 * (j & 3))) & 0xffu);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int8_t *dst = obs_board + (int64_t)p * plane + g0 * NN;                     // this run in observer p's plane
            const int d = (int)((16u - (uint32_t)(((int64_t)p * plane) & 15)) & 15u);   // its bytes up to the first 16-byte boundary
            const int lead = d < run_bytes ? d : run_bytes;
            const int K = (run_bytes - lead) >> 4;                                      // aligned chunks inside the run
            const int dq = d >> 2;
            const uint32_t rr = (uint32_t)(d & 3);
            for (int k = threadIdx.x; k < K; k += 256) {
                const uint4 v0 = *reinterpret_cast<const uint4 *>(lds + (k << 4));
                uint4 f = v0;
                if (d != 0) {                                   // (uniform)
                    const uint4 v1 = *reinterpret_cast<const uint4 *>(lds + ((k + 1) << 4));
                    const uint32_t w[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    uint32_t x[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) x[j] = dq == 0 ? w[j] : dq == 1 ? w[j + 1] : dq == 2 ? w[j + 2] : w[j + 3];
                    f = make_uint4(__builtin_amdgcn_alignbyte(x[1], x[0], rr), __builtin_amdgcn_alignbyte(x[2], x[1], rr),
                                   __builtin_amdgcn_alignbyte(x[3], x[2], rr), __builtin_amdgcn_alignbyte(x[4], x[3], rr));
                }
                uint4 o;
                o.x = tron_relabel4<P>(f.x, p, lut_lo[p], lut_hi[p]);
                o.y = tron_relabel4<P>(f.y, p, lut_lo[p], lut_hi[p]);
                o.z = tron_relabel4<P>(f.z, p, lut_lo[p], lut_hi[p]);
                o.w = tron_relabel4<P>(f.w, p, lut_lo[p], lut_hi[p]);
                crl_stream_store16(dst + lead + (k << 4), o, nt);
            }
            // the run's bytes in front of its first and behind its last aligned chunk (< 16 each)
            const int tail0 = lead + (K << 4);
            const int t = (int)threadIdx.x;
            const int idx = t < lead ? t : (t - lead < run_bytes - tail0 ? tail0 + t - lead : -1);
            if (idx >= 0) dst[idx] = (int8_t)(tron_relabel4<P>((uint32_t)lds[idx], p, lut_lo[p], lut_hi[p]) & 0xffu);
        }
        return;
    }
    for (int c = threadIdx.x; c < chunks; c += 256) {
        const int byte0 = c << 4;
        uint4 v = *reinterpret_cast<const uint4 *>(lds + byte0);
        const int e0 = (int)__umulhi((uint32_t)byte0, inv_nn);  // game of the chunk's first byte (byte0 < 2^16 * ...: exact)
        const int r = byte0 - e0 * NN;                          // ... and its cell there
        const int first = NN - r;                               // bytes of the chunk that still belong to e0 (>= 16: all of them)
        const bool f0 = rflag[e0] != 0, f1 = first < 16 && rflag[e0 + 1] != 0;
        const int64_t gofs = g0 * NN + byte0;
        if (f0 | f1) {                                          // new_state: the fresh board replaces the finished one
            const uint4 fr = tron_fresh_flat16<P>(cfg, r, NN);
            uint32_t m[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t mq = 0;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) mq |= ((4 * q + k2 < first) ? f0 : f1) ? (0xffu << (8 * k2)) : 0u;
                m[q] = mq;
            }
            v.x = (v.x & ~m[0]) | (fr.x & m[0]); v.y = (v.y & ~m[1]) | (fr.y & m[1]);
            v.z = (v.z & ~m[2]) | (fr.z & m[2]); v.w = (v.w & ~m[3]) | (fr.w & m[3]);
            // (all 16 bytes are this workgroup's: the bytes of a neighbour that was not reset are rewritten with what phase B
            //  left in LDS and in HBM alike)
            *reinterpret_cast<uint4 *>(board + gofs) = v;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            uint4 o;
            o.x = tron_relabel4<P>(v.x, p, lut_lo[p], lut_hi[p]);
            o.y = tron_relabel4<P>(v.y, p, lut_lo[p], lut_hi[p]);
            o.z = tron_relabel4<P>(v.z, p, lut_lo[p], lut_hi[p]);
            o.w = tron_relabel4<P>(v.w, p, lut_lo[p], lut_hi[p]);
            crl_stream_store16(obs_board + (int64_t)p * plane + gofs, o, nt);
        }
    }
}

// crl_tron_step through LDS: the workgroup reads its G boards ONCE, coalesced 16-byte loads (phase A of the fused kernel
// above), one lane per game plays the step on the LDS copy and mirrors the <= P trail bytes to HBM, and the boards of the
// games that were reset are rewritten by all threads.  Against tron_step_kernel's byte probes (each a 64-byte sector, ~10x
// the algorithmic bytes: profiles/traffic_step_api.json) this streams N*N bytes per game in and probes for free.
template <int P, int G>
__global__ void __launch_bounds__(256)
tron_step_staged_kernel(const crl_tron_cfg cfg, const TronGeom g, const uint32_t inv_cp, const int64_t B,
                        int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                        int8_t *__restrict__ deaths, const int8_t *__restrict__ actions,
                        int8_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                        const uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int NN = g.NN, SLAB = NN + 16;                        // NN % 16 == 0: 16-byte LDS accesses stay aligned
    const int cp = NN >> 4;                                     // 16-byte chunks per board
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int n_game = (int)((B - g0) < G ? (B - g0) : G);
    const int total = n_game * cp;
    uint8_t *rflag = lds + G * SLAB;                            // [G] this game was reset by the step
    __shared__ int any_reset;
    if (threadIdx.x == 0) any_reset = 0;
    // the stepping lanes' player vectors: in flight together with the board loads
    TronRegs<P> s;
    int act[P], rew[P];
    const int e = threadIdx.x;
    const bool stepper = e < G, valid = stepper && e < n_game;
    const int64_t b = g0 + (valid ? e : 0);
    if (stepper) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            s.h[p] = valid ? heads[p * B + b] : 0;
            s.d[p] = valid ? dirs[p * B + b] : 0;
            s.k[p] = valid ? deaths[p * B + b] : 1;
            act[p] = valid ? actions[p * B + b] : 0;
        }
    }
    for (int base = threadIdx.x; base < total; base += 4 * 256) {
        uint4 v[4];
        int dst[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = base + u * 256;
            const int cc = c < total ? c : 0;
            const int ee = cp == 1 ? cc : (int)__umulhi((uint32_t)cc, inv_cp);
            const int off = (cc - ee * cp) << 4;
            dst[u] = c < total ? ee * SLAB + off : -1;
            v[u] = *reinterpret_cast<const uint4 *>(board + (g0 + ee) * NN + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *reinterpret_cast<uint4 *>(lds + dst[u]) = v[u];
    }
    __syncthreads();
    if (stepper) {
        tron_split_heads<P>(g, s);
        int term, wm;
        const DualBoard bd{lds + e * SLAB, board + b * NN CRL_CELLS_INIT(NN)};
        tron_step_core<P>(g, bd, valid, s, act, rew, term, wm);
        const bool do_reset = valid && term && (flags & CRL_STEP_AUTO_RESET);
        rflag[e] = do_reset ? 1 : 0;
        if (do_reset) { tron_regs_to_start<P>(cfg, g, s); any_reset = 1; }
        if (valid) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                rewards[p * B + b] = (int8_t)rew[p];
                heads[p * B + b] = (int16_t)s.h[p];
                dirs[p * B + b] = (int8_t)s.d[p];
                deaths[p * B + b] = (int8_t)s.k[p];
            }
            terminal[b] = (uint8_t)term;
            winners[b] = (uint8_t)wm;
        }
    }
    if (!(flags & CRL_STEP_AUTO_RESET)) return;                 // (uniform)
    __syncthreads();
    if (!any_reset) return;                                     // (uniform: read after the barrier)
    for (int c = threadIdx.x; c < total; c += 256) {            // new_state: the fresh board replaces the finished one
        const int ee = cp == 1 ? c : (int)__umulhi((uint32_t)c, inv_cp);
        const int off = (c - ee * cp) << 4;
        if (rflag[ee]) *reinterpret_cast<uint4 *>(board + (g0 + ee) * NN + off) = tron_fresh_chunk16<P>(cfg, off);
    }
}

// The same fused call for every shape the kernel above cannot take (N*N % 16 != 0 -- the reference's default 19x19 board
// among them --, P = 8, boards too large for 16 slabs of LDS): ONE GAME PER WORKGROUP, byte granularity throughout.  The
// board goes into LDS with coalesced byte loads, thread 0 plays the step there (mirroring the <= P trail bytes to HBM),
// and all threads write the P relabelled copies (CyTronGrid.pyx:70-71 in arithmetic).  Also what a single-state caller
// (B = 1, host-mapped memory) runs: one launch, any board.  Before round 3 these shapes took three launches (sample,
// step, observe_all) and needed explicit actions.
template <int P>
__global__ void __launch_bounds__(256)
tron_step_observe_any_kernel(const crl_tron_cfg cfg, const TronGeom g, const int64_t B,
                             const uint32_t seed_lo, const uint32_t seed_hi, const uint64_t first_env_id,
                             int8_t *__restrict__ board, int16_t *__restrict__ heads, int8_t *__restrict__ dirs,
                             int8_t *__restrict__ deaths, const int8_t *__restrict__ actions, uint32_t *__restrict__ tcount,
                             int8_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                             int8_t *__restrict__ obs_board, int16_t *__restrict__ oh, int8_t *__restrict__ od,
                             int8_t *__restrict__ ok, const uint32_t flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ int s_reset;
    const int NN = g.NN;
    const int64_t b = blockIdx.x;
    int8_t *gb = board + b * NN;
    for (int c = threadIdx.x; c < NN; c += 256) lds[c] = (uint8_t)gb[c];
    __syncthreads();
    if (threadIdx.x == 0) {
        TronRegs<P> s;
        int act[P], rew[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            s.h[p] = heads[p * B + b];
            s.d[p] = dirs[p * B + b];
            s.k[p] = deaths[p * B + b];
            act[p] = actions ? actions[p * B + b] : 0;
        }
        if (!actions) {                                         // the rollout's random agent at this game's step counter
            const uint32_t c = tcount[b];
            tron_sample_actions<P>((uint32_t)(first_env_id + (uint64_t)b), c, seed_lo, seed_hi, act);
            tcount[b] = c + 1u;
        }
        tron_split_heads<P>(g, s);
        int term, wm;
        const DualBoard bd{lds, gb CRL_CELLS_INIT(NN)};
        tron_step_core<P>(g, bd, true, s, act, rew, term, wm);
        const bool do_reset = term && (flags & CRL_STEP_AUTO_RESET);
        s_reset = do_reset ? 1 : 0;
        if (do_reset) tron_regs_to_start<P>(cfg, g, s);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            rewards[p * B + b] = (int8_t)rew[p];
            heads[p * B + b] = (int16_t)s.h[p];
            dirs[p * B + b] = (int8_t)s.d[p];
            deaths[p * B + b] = (int8_t)s.k[p];
        }
        terminal[b] = (uint8_t)term;
        winners[b] = (uint8_t)wm;
#pragma unroll
        for (int p = 0; p < P; ++p)                             // TronGridEnvironment.py:392-396: rolled so index 0 is the observer
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int src = (i + p) % P;
                oh[((int64_t)p * P + i) * B + b] = (int16_t)s.h[src];
                od[((int64_t)p * P + i) * B + b] = (int8_t)s.d[src];
                ok[((int64_t)p * P + i) * B + b] = (int8_t)s.k[src];
            }
    }
    __syncthreads();
    const bool fresh = s_reset != 0;
    const int64_t plane = B * (int64_t)NN;
    for (int c = threadIdx.x; c < NN; c += 256) {
        int v = lds[c];
        if (fresh) {                                            // new_state: the fresh board replaces the finished one
            v = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) v = (cfg.start_heads[p] == c) ? p + 1 : v;
            gb[c] = (int8_t)v;
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            int n = v - (p + 1);                                // CyTronGrid.pyx:70-71, v in 1..P
            n = n < 0 ? n + P : n;
            obs_board[(int64_t)p * plane + b * NN + c] = (int8_t)(v > 0 ? n + 1 : v);
        }
    }
}

// ---- the reference's own native boundary, with its own types ---------------------------------------------------------
// CyTronGrid.pyx:3-7 next_state_inplace(long[:, ::1] board, long[::1] heads, long[::1] directions, long[::1] deaths,
// const long[::1] actions) and :65 relative_player_inplace(long[:, ::1] board, num_players, player): C-contiguous int64
// arrays in the reference's layout, mutated in place.  For callers that HOLD reference states (the single-state drop-in
// class on crl_host_alloc memory): no int64 <-> int8 conversion on the host, and the board is touched where the step
// touches it -- <= P probes and <= P trail stores of 8 bytes.  One workgroup per game (B games lie one behind the other,
// each in the reference layout).  With observation outputs (optional) the whole board is staged into LDS as bytes by all
// threads -- its loads are in flight together with thread 0's player vectors, ONE round trip on mapped memory --, thread 0
// then probes the LDS copy, and all threads write the P relabelled int64 copies (CyTronGrid.pyx:70-71, C remainder).
struct Board64 {                    // in place on the caller's int64 board
    int64_t *p;
    CRL_CELLS_MEMBER
    __device__ __forceinline__ int raw(const int c) const { CRL_CELLS_CHECK(c, 134); return (int)p[c]; }
    __device__ __forceinline__ int owner(const int r) const { return r; }
    __device__ __forceinline__ void put(const int c, const int who) const { CRL_CELLS_CHECK(c, 135); p[c] = (int64_t)who; }
};
struct DualBoard64 {                // step on the LDS byte copy, mirror the trail writes to the caller's int64 board
    uint8_t *l;
    int64_t *g;
    CRL_CELLS_MEMBER
    __device__ __forceinline__ int raw(const int c) const { CRL_CELLS_CHECK(c, 136); return l[c]; }
    __device__ __forceinline__ int owner(const int r) const { return r; }
    __device__ __forceinline__ void put(const int c, const int who) const { CRL_CELLS_CHECK(c, 137); l[c] = (uint8_t)who; g[c] = (int64_t)who; }
};

// the player vectors of ONE game by value (kernel arguments): what the single-state call passes instead of letting the
// kernel fetch them over PCIe before it can probe the board
struct TronVec64 {
    int32_t h[CRL_TRON_MAX_P], d[CRL_TRON_MAX_P], k[CRL_TRON_MAX_P], a[CRL_TRON_MAX_P];
};

template <int P>
__global__ void __launch_bounds__(256)
tron_next_state64_kernel(const TronGeom g, const int64_t B, int64_t *__restrict__ board, int64_t *__restrict__ heads,
                         int64_t *__restrict__ dirs, int64_t *__restrict__ deaths, const int64_t *__restrict__ actions,
                         int64_t *__restrict__ rewards, uint8_t *__restrict__ terminal, uint8_t *__restrict__ winners,
                         int64_t *__restrict__ obs_board, int64_t *__restrict__ oh, int64_t *__restrict__ od,
                         int64_t *__restrict__ ok, const TronVec64 vec, const bool by_value, uint32_t *flag, const uint32_t seq)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int NN = g.NN;
    const int64_t b = blockIdx.x;
    int64_t *gb = board + b * NN;
    const bool want_obs = obs_board != nullptr;
    TronRegs<P> s;
    int act[P], rew[P];
    if (threadIdx.x == 0) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            s.h[p] = by_value ? vec.h[p] : (int)heads[b * P + p];
            s.d[p] = by_value ? vec.d[p] : (int)dirs[b * P + p];
            s.k[p] = by_value ? vec.k[p] : (int)deaths[b * P + p];
            act[p] = by_value ? vec.a[p] : (int)actions[b * P + p];
        }
    }
    if (want_obs) {
        for (int c = threadIdx.x; c < NN; c += 256) lds[c] = (uint8_t)gb[c];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        tron_split_heads<P>(g, s);
        int term, wm;
        if (want_obs) {
            const DualBoard64 bd{lds, gb CRL_CELLS_INIT(NN)};
            tron_step_core<P, DualBoard64, true>(g, bd, true, s, act, rew, term, wm);
        } else {
            const Board64 bd{gb CRL_CELLS_INIT(NN)};
            tron_step_core<P, Board64, true>(g, bd, true, s, act, rew, term, wm);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            heads[b * P + p] = s.h[p];
            dirs[b * P + p] = s.d[p];
            deaths[b * P + p] = s.k[p];
            if (rewards) rewards[b * P + p] = rew[p];           // TronGridEnvironment.py:313,319-321
        }
        if (terminal) terminal[b] = (uint8_t)term;
        if (winners) winners[b] = (uint8_t)wm;
        if (want_obs) {
#pragma unroll
            for (int p = 0; p < P; ++p)                         // TronGridEnvironment.py:393-397: rolled so index 0 is the observer
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const int src = (i + p) % P;
                    oh[(b * P + p) * P + i] = s.h[src];
                    od[(b * P + p) * P + i] = s.d[src];
                    ok[(b * P + p) * P + i] = s.k[src];
                }
        }
    }
    if (!want_obs) {                                            // (single-state call: completion behind thread 0's own stores)
        if (flag && threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < NN; c += 256) {
        const int v = lds[c];
#pragma unroll
        for (int p = 0; p < P; ++p)
            obs_board[(b * P + p) * (int64_t)NN + c] = v > 0 ? (int64_t)((v - (p + 1) + P) % P + 1) : (int64_t)v;
    }
    if (flag) {                                                 // every thread's stores are out system-wide before the flag is
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// CyTronGrid.pyx:65-71 on int64 boards in place; `player` as the reference passes it (observer id + 1), one per game
__global__ void __launch_bounds__(256)
tron_relative_player64_kernel(const int64_t total, const int NN, int64_t *__restrict__ board, const int64_t num_players,
                              const int64_t *__restrict__ player)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = board[i];
        if (v > 0) board[i] = ((v - player[i / NN] + num_players) % num_players) + 1;    // C remainder, as cdivision=True
    }
}


inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

inline TronGeom geom_of(const crl_tron_cfg &cfg)
{
    TronGeom g;
    g.N = cfg.N;
    g.NN = cfg.N * cfg.N;
    g.inv_n = (uint32_t)(0x100000000ull / (uint64_t)cfg.N) + 1u;
    return g;
}

// wall-bordered LDS slab of the rollout kernel (see tron_rollout_lds_kernel)
inline TronPad pad_of(const crl_tron_cfg &cfg, const int RS)
{
    TronPad p;
    const int N = cfg.N;
    p.junk = (N + 2) * RS;
    p.stride = (p.junk + 1 + 3) & ~3;                 // a whole, ODD number of dwords
    if (((p.stride >> 2) & 1) == 0) p.stride += 4;
    const int usable = (cfg.P <= 7 ? 31 : 15) - 2;
    p.sweep_rows = (N + usable - 1) / usable;
    p.inv_nn = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(N * N)) + 1u;
    p.inv_nq = (N >= 4) ? (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(N / 4)) + 1u : 0u;
    return p;
}

constexpr int kLdsDynamic = 160 * 1024 - 1024;  // dynamic part; the kernels also hold small static tables (actions, wall pattern)


} // namespace

#define TRON_DISPATCH_P(P_, CALL)          \
    switch (P_) {                          \
        case 1: { constexpr int PP = 1; CALL; } break; \
        case 2: { constexpr int PP = 2; CALL; } break; \
        case 3: { constexpr int PP = 3; CALL; } break; \
        case 4: { constexpr int PP = 4; CALL; } break; \
        case 5: { constexpr int PP = 5; CALL; } break; \
        case 6: { constexpr int PP = 6; CALL; } break; \
        case 7: { constexpr int PP = 7; CALL; } break; \
        case 8: { constexpr int PP = 8; CALL; } break; \
        default: crl_set_error("tron: P=%d out of range 1..8", P_); return CRL_EINVAL; \
    }

#define TRON_DISPATCH_P4(P_, CALL)         \
    switch (P_) {                          \
        case 1: { constexpr int PP = 1; CALL; } break; \
        case 2: { constexpr int PP = 2; CALL; } break; \
        case 3: { constexpr int PP = 3; CALL; } break; \
        case 4: { constexpr int PP = 4; CALL; } break; \
        default: crl_set_error("tron: P=%d out of range 1..4", P_); return CRL_EINVAL; \
    }

int crl_tron_bounds(unsigned int *out4)
{
    CRL_BOUNDS_READBACK(out4);
    return CRL_OK;
}

extern "C" {

int crl_tron_create(int N, int P, const int16_t *start_heads, const int8_t *start_dirs, crl_ctx **out)
{
    CRL_REQUIRE(out != nullptr, "crl_tron_create: out is NULL");
    CRL_REQUIRE(N >= 2 && N <= 181, "crl_tron_create: N=%d out of range 2..181 (int16 heads)", N);
    CRL_REQUIRE(P >= 1 && P <= CRL_TRON_MAX_P, "crl_tron_create: P=%d out of range 1..%d", P, CRL_TRON_MAX_P);
    CRL_REQUIRE(start_heads && start_dirs, "crl_tron_create: start arrays are NULL");
    for (int p = 0; p < P; ++p) {
        CRL_REQUIRE(start_heads[p] >= 0 && start_heads[p] < N * N, "crl_tron_create: start_heads[%d]=%d outside the board", p, start_heads[p]);
        CRL_REQUIRE(start_dirs[p] >= 0 && start_dirs[p] < 4, "crl_tron_create: start_dirs[%d]=%d not in 0..3", p, start_dirs[p]);
        for (int q = 0; q < p; ++q)
            CRL_REQUIRE(start_heads[p] != start_heads[q], "crl_tron_create: players %d and %d share a start cell", q, p);
    }
    crl_ctx *c = new crl_ctx();
    memset(c, 0, sizeof(*c));
    c->game = CRL_GAME_TRON;
    c->tron.N = N;
    c->tron.P = P;
    for (int p = 0; p < P; ++p) { c->tron.start_heads[p] = start_heads[p]; c->tron.start_dirs[p] = start_dirs[p]; }
    for (int p = P; p < CRL_TRON_MAX_P; ++p) { c->tron.start_heads[p] = -1; c->tron.start_dirs[p] = 0; }
    *out = c;
    return CRL_OK;
}

#define TRON_CTX_CHECK(fn)                                                                  \
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_TRON, fn ": ctx is not a tron context"); \
    CRL_REQUIRE(B > 0 && B <= ((int64_t)1 << 31), fn ": B=%lld out of range", (long long)B)

int crl_tron_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask,
                   int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths, void *stream)
{
    TRON_CTX_CHECK("crl_tron_reset");
    CRL_REQUIRE(board && heads && dirs && deaths, "crl_tron_reset: NULL state pointer");
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (NN % 16 == 0) && (((uintptr_t)board & 15) == 0);
    const bool flat = !wide && NN >= 16 && (((uintptr_t)board & 15) == 0);        // any other board: the batch as one byte stream
    const int64_t items = flat ? (B * (int64_t)NN + 15) / 16 : B * (int64_t)(wide ? NN / 16 : NN);
    const unsigned grid = (unsigned)((items + 255) / 256 > 65536 * 4 ? 65536 * 4 : (items + 255) / 256);
    const uint64_t inv_nn64 = ~(uint64_t)0 / (uint64_t)NN + 1u;                    // floor((2^64 - 1) / NN) + 1: exact quotients for byte offsets < 2^64 / NN
    TRON_DISPATCH_P(cfg.P, {
        if (wide) hipLaunchKernelGGL((tron_reset_board_kernel<PP, true>), dim3(grid), dim3(256), 0, s, cfg, B, mask, board);
        else if (flat) hipLaunchKernelGGL((tron_reset_board_flat_kernel<PP>), dim3(grid), dim3(256), 0, s, cfg, B, mask, board, inv_nn64);
        else hipLaunchKernelGGL((tron_reset_board_kernel<PP, false>), dim3(grid), dim3(256), 0, s, cfg, B, mask, board);
        hipLaunchKernelGGL((tron_reset_players_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, s, cfg, B, mask, heads, dirs, deaths);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_step(const crl_ctx *ctx, int64_t B,
                  int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                  const int8_t *actions, int8_t *rewards, uint8_t *terminal, uint8_t *winners,
                  uint32_t flags, void *stream)
{
    TRON_CTX_CHECK("crl_tron_step");
    CRL_REQUIRE(board && heads && dirs && deaths, "crl_tron_step: NULL state pointer");
    CRL_REQUIRE(actions && rewards && terminal && winners, "crl_tron_step: NULL action/output pointer");
    CRL_REQUIRE((flags & ~(CRL_STEP_AUTO_RESET | CRL_STEP_BYTES | CRL_STEP_STAGED)) == 0, "crl_tron_step: unknown flags 0x%x", flags);
    const crl_tron_cfg &cfg = ctx->tron;
    CRL_REQUIRE(!(flags & CRL_STEP_AUTO_RESET) || ((cfg.N * cfg.N) % 16 != 0) || (((uintptr_t)board & 15) == 0),
                "crl_tron_step: board must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const TronGeom g = geom_of(cfg);
    // Boards of whole 16-byte chunks that fit 64 (or 16) to a workgroup's LDS go through tron_step_staged_kernel (one coalesced
    // read of every board); everything else, or flags & CRL_STEP_BYTES, through the byte probes of tron_step_kernel.  Identical
    // results (tests run both); CRL_STEP_STAGED pins the LDS kernel where the shape allows it.
    const int NN = cfg.N * cfg.N, slab = NN + 16;
    const int G = (64 * slab + 64 <= 48 * 1024) ? 64 : (16 * slab + 16 <= 48 * 1024) ? 16 : 0;
    const bool can_stage = (NN % 16) == 0 && G > 0 && (((uintptr_t)board & 15) == 0);
    const bool staged = can_stage && !(flags & CRL_STEP_BYTES) && (CRL_STEP_DEFAULT_STAGED || (flags & CRL_STEP_STAGED));
    const uint32_t kflags = flags & CRL_STEP_AUTO_RESET;
    if (staged) {
        const uint32_t inv_cp = NN == 16 ? 0u : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(NN / 16)) + 1u;
        const size_t lds_bytes = (size_t)G * slab + G;
        TRON_DISPATCH_P(cfg.P, {
            if (G == 64)
                hipLaunchKernelGGL((tron_step_staged_kernel<PP, 64>), dim3(blocks_for(B, 64)), dim3(256), lds_bytes, s, cfg, g, inv_cp, B,
                                   board, heads, dirs, deaths, actions, rewards, terminal, winners, kflags);
            else
                hipLaunchKernelGGL((tron_step_staged_kernel<PP, 16>), dim3(blocks_for(B, 16)), dim3(256), lds_bytes, s, cfg, g, inv_cp, B,
                                   board, heads, dirs, deaths, actions, rewards, terminal, winners, kflags);
        });
        CRL_LAUNCH_CHECK();
        return CRL_OK;
    }
    TRON_DISPATCH_P(cfg.P, {
        hipLaunchKernelGGL((tron_step_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, s, cfg, g, B,
                           board, heads, dirs, deaths, actions, rewards, terminal, winners, kflags);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

static int tron_rollout_impl(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                             int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                             crl_tron_stats st, uint32_t flags, void *stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    // every launch of the rollout goes through here: with events, the FIRST launch carries the start event and the LAST one
    // the stop event in the dispatch itself (hipExtLaunchKernelGGL) -- no marker packets around the kernels
    bool first_launch = true;
    auto launch = [&](auto kernel, const dim3 grid, const dim3 block, const size_t lds_bytes, const bool last, auto... args) {
        hipEvent_t e0 = first_launch ? ev_start : nullptr, e1 = last ? ev_stop : nullptr;
        first_launch = false;
        if (e0 || e1) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds_bytes, (hipStream_t)stream, e0, e1, 0u, args...);
        else hipLaunchKernelGGL(kernel, grid, block, lds_bytes, (hipStream_t)stream, args...);
    };
    TRON_CTX_CHECK("crl_tron_rollout");
    CRL_REQUIRE(board && heads && dirs && deaths, "crl_tron_rollout: NULL state pointer");
    CRL_REQUIRE(st.tcount && st.tstep && st.n_episodes && st.win_count && st.len_sum && st.ret_sum &&
                st.last_winners && st.last_len, "crl_tron_rollout: NULL stats pointer");
    CRL_REQUIRE(T >= 0 && T <= (1 << 24), "crl_tron_rollout: T=%d out of range", T);
    CRL_REQUIRE((flags & ~(CRL_ROLLOUT_NO_LDS | CRL_ROLLOUT_BYTES | CRL_ROLLOUT_BITS | CRL_ROLLOUT_QUAD | CRL_ROLLOUT_QBITS | CRL_ROLLOUT_GQUAD | CRL_ROLLOUT_PAIR)) == 0, "crl_tron_rollout: unknown flags 0x%x", flags);
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    CRL_REQUIRE((NN % 16 != 0) || (((uintptr_t)board & 15) == 0), "crl_tron_rollout: board must be 16-byte aligned");
    if (T == 0) return CRL_OK;
    const TronGeom g = geom_of(cfg);
    // LDS-resident kernels.  Byte slabs: 256 games per workgroup on boards up to 20x20, 64 (one wave) up to 40x40.
    // Bitboards (T >= 256): 256 games per workgroup either way; the replay takes the byte slabs one wave at a time
    // on the larger boards.
    const bool small = cfg.N <= kLdsMaxNSmall;
    const int RS = small ? kRowBytesSmall : kRowBytesLarge;
    const TronPad pad = pad_of(cfg, RS);
    // The lane-per-player kernel on boards in GLOBAL memory has no fixed cost (no copy in / out, no tags to strip, no replay):
    // a launch costs what its steps cost, 2-4.5 us each at 65,536 games of 20x20..40x40 (a probe pulls a cache line per byte).
    // So, unless a kernel is pinned, it takes (profiles/r5_shape_sweep.txt, `kernels`):
    //  * a ONE-step launch on boards up to 20x20 (11 us against the byte-slab kernel's 13-16);
    //  * short launches on boards 21..40 wide, where the bitboard kernel's copy in + replay kernel is 45-90 us (70-170 us on
    //    boards that are not whole dwords a row): up to 18 steps (32 on the latter);
    //  * boards above 40x40, which nothing else plays out of LDS: with four players always up to 44x44, up to 200 steps up to
    //    56x56 and up to 48 steps above -- beyond that the lane-per-game global kernel, whose episode tags save the rewrite of a
    //    finished board (N * N bytes per reset) at the price of a pass over all boards at the end of the launch, and whose step
    //    gets cheaper with fewer players (a lane per game probes P cells; a lane per player idles, and shorter episodes mean
    //    more rewrites): three players up to 56 / 24 steps, one or two up to 24 / 14 (41x41, 65,536 two-player games: 2.7 us
    //    per step against this kernel's 6.8).
    // More than four players: the lane-per-game kernels.
    const bool no_pin = !(flags & (CRL_ROLLOUT_NO_LDS | CRL_ROLLOUT_BYTES | CRL_ROLLOUT_BITS | CRL_ROLLOUT_QUAD | CRL_ROLLOUT_QBITS | CRL_ROLLOUT_PAIR));
    // (65,536 games of two players, 2048 steps: 499 against 632 us at 20x20, 508 against 642 at 13x13; the pair kernel copies
    //  boards whose rows are not whole dwords byte by byte, which a launch of 256 steps earns back: profiles/r5_shape_sweep.txt)
    const int kPairMinT = ((cfg.N & 3) == 0 && cfg.N >= 8) ? 32 : 256;
    const bool lds_fit = cfg.N <= kLdsMaxNLarge && (((uintptr_t)board & 15) == 0);
    const bool wide_rows = (cfg.N & 3) == 0;
    const bool gquad_pays = small ? T == 1
                          : cfg.N <= kLdsMaxNLarge ? T <= (wide_rows ? 18 : 32)
                          : cfg.P == 4 ? (cfg.N <= 44 || T <= (cfg.N <= 56 ? 200 : 48))
                          : cfg.P == 3 ? T <= (cfg.N <= 56 ? 56 : 24)
                                       : T <= (cfg.N <= 56 ? 24 : 14);
    const bool use_gquad = cfg.P <= 4 && ((flags & CRL_ROLLOUT_GQUAD) || (no_pin && (gquad_pays || (!lds_fit && cfg.N <= kLdsMaxNLarge))));
    const bool lds_ok = !(flags & CRL_ROLLOUT_NO_LDS) && !use_gquad && lds_fit;
    if (use_gquad) {
        constexpr int kQuadMaxT = 16383;                        // (16-bit episode / win / step counts per launch, as below)
        for (int t0 = 0; t0 < T; t0 += kQuadMaxT) {
            TRON_DISPATCH_P4(cfg.P, {
                launch(tron_rollout_gquad_kernel<PP>, dim3(blocks_for(B, 64)), dim3(256), (size_t)0, t0 + kQuadMaxT >= T, cfg, g, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, std::min(kQuadMaxT, T - t0), board, heads, dirs, deaths, st);
            });
            CRL_LAUNCH_CHECK();
        }
        return CRL_OK;
    }
    // default: bitboards where they buy residency (boards above 20x20 fit 4x the games per CU); on small boards the
    // byte kernel already has every game resident and no replay to pay for
    const bool use_bits = lds_ok && !(flags & CRL_ROLLOUT_BYTES) && ((flags & CRL_ROLLOUT_BITS) || (!small && T >= 256));
    TronBits bits;
    {
        const int max_n = small ? kLdsMaxNSmall : kLdsMaxNLarge;
        const int max_w = (((max_n + 2) * (max_n + 1) + 31) / 32 + 3) & ~3;
        bits.stride = 2 * ((max_w * 4 + 127) & ~127) + 16;              // two 128-byte-aligned slabs of pattern words + a junk word
        bits.inv_s = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(cfg.N + 1)) + 1u;
    }
    // one lane per player, four per game: boards up to 20x20 with at most 4 players (see tron_rollout_quad_kernel)
    const bool quad_ok = lds_ok && small && cfg.P <= 4 && pad.sweep_rows == 1;
    // the default wherever it applies: 1.42e11 vs 1.24e11 env-steps/s for the lane-per-game byte kernel at 20x20, P = 4
    const bool use_quad = quad_ok && !(flags & (CRL_ROLLOUT_BYTES | CRL_ROLLOUT_BITS | CRL_ROLLOUT_QBITS));
    // the same on bitboards (+ replay): boards above 20x20, where byte slabs would leave a CU with one wave per SIMD.
    // Launches of any length: at 40x40 even a 16-step launch takes 0.15 ms against 0.25 ms on byte slabs (the replay is
    // cheaper than moving 1,852-byte slabs through a lone wave per SIMD); tools/README.md
    const bool qbits_ok = lds_ok && cfg.P <= 4;
    const bool use_qbits = qbits_ok && !use_quad &&
                           ((flags & CRL_ROLLOUT_QBITS) || (!small && !(flags & (CRL_ROLLOUT_BYTES | CRL_ROLLOUT_BITS))));
    if (use_qbits) {
        // four lanes of a game store 64 contiguous bytes per 16-byte rewrite store, which the LDS serves eight lanes (two
        // games) at a time: a game stride of 16 dwords mod 32 banks keeps the two games on different banks (the
        // lane-per-game kernel's 4 mod 32 would overlap them)
        TronBits qb = bits;
#ifndef CRL_QBITS_PAD
#define CRL_QBITS_PAD 64            /* bytes behind a game's two slabs: the four junk words + bank padding (tools/lib_variant.sh) */
#endif
        qb.stride = bits.stride - 16 + CRL_QBITS_PAD;
        const size_t lds_q = std::max((size_t)64 * qb.stride, (size_t)16 * pad.stride);
        constexpr int kQuadMaxT = 16383;                        // (16-bit episode / win / step counts per launch, see below)
#ifndef CRL_QBITS_SPLIT_REPLAY
#define CRL_QBITS_SPLIT_REPLAY 1    /* the replay as a kernel of its own behind the bitboard kernel (0: inside it, four turns per workgroup) */
#endif
        const int split = CRL_QBITS_SPLIT_REPLAY;
        for (int t0 = 0; t0 < T; t0 += kQuadMaxT) {
            const int tt = std::min(kQuadMaxT, T - t0);
            TRON_DISPATCH_P4(cfg.P, {
                const bool last = t0 + kQuadMaxT >= T;
                if (small)
                    launch(tron_rollout_qbits_kernel<PP, false>, dim3(blocks_for(B, 64)), dim3(256), lds_q, last && !split, cfg, g, pad, qb, B,
                           (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, tt, board, heads, dirs, deaths, st, split);
                else
                    launch(tron_rollout_qbits_kernel<PP, true>, dim3(blocks_for(B, 64)), dim3(256), lds_q, last && !split, cfg, g, pad, qb, B,
                           (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, tt, board, heads, dirs, deaths, st, split);
                if (split) {
                    if (small)
                        launch(tron_replay_kernel<PP, false>, dim3(blocks_for(B, 16)), dim3(64), (size_t)16 * pad.stride, last, cfg, g, pad, B,
                               (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, tt, board, heads, dirs, deaths, st);
                    else
                        launch(tron_replay_kernel<PP, true>, dim3(blocks_for(B, 16)), dim3(64), (size_t)16 * pad.stride, last, cfg, g, pad, B,
                               (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, tt, board, heads, dirs, deaths, st);
                }
            });
            CRL_LAUNCH_CHECK();
        }
        return CRL_OK;
    }
    // one or two players: two lanes per game (tron_rollout_pair_kernel) once the launch is long enough for its plainer copies
    // (its slabs leave a SIMD two waves at 20x20, three from 15x15 down: with two it wins while the quad kernel's waves do not
    //  fill the chip either -- up to ~100,000 games -- and ties beyond: 262,144 games of 20x20, 512 steps: 607 against 583 us)
    const bool pair_pays = (size_t)pad.stride * 64 * 6 <= (size_t)160 * 1024 || B <= 98304;
    const bool use_pair = quad_ok && cfg.P <= 2 && use_quad && ((flags & CRL_ROLLOUT_PAIR) || (no_pin && T >= kPairMinT && pair_pays));
    if (use_pair) {
        constexpr int kQuadMaxT = 16383;
        for (int t0 = 0; t0 < T; t0 += kQuadMaxT) {
            launch(tron_rollout_pair_kernel<kRowBytesSmall>, dim3(blocks_for(B, 64)), dim3(128), (size_t)64 * pad.stride,
                   t0 + kQuadMaxT >= T, cfg, g, pad, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, std::min(kQuadMaxT, T - t0),
                   board, heads, dirs, deaths, st);
            CRL_LAUNCH_CHECK();
        }
        return CRL_OK;
    }
    if (use_quad) {
        // (the kernel keeps a launch's episode, win and step counts in 14..16 bits: longer rollouts go out as several launches,
        //  which is the same rollout -- the state and the step counters carry over)
        constexpr int kQuadMaxT = 16383;
        for (int t0 = 0; t0 < T; t0 += kQuadMaxT) {
#if CRL_QUAD_WG > 256
            {
                static thread_local int opted = 0;
                if (!opted) {
                    CRL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tron_rollout_quad_kernel<kRowBytesSmall>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDynamic));
                    opted = 1;
                }
            }
#endif
            launch(tron_rollout_quad_kernel<kRowBytesSmall>, dim3(blocks_for(B, CRL_QUAD_WG / 4)), dim3(CRL_QUAD_WG), (size_t)(CRL_QUAD_WG / 4) * pad.stride,
                   t0 + kQuadMaxT >= T, cfg, g, pad, B, (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, std::min(kQuadMaxT, T - t0),
                   board, heads, dirs, deaths, st);
            CRL_LAUNCH_CHECK();
        }
        return CRL_OK;
    }
    const int threads = (use_bits || small) ? 256 : 64;
    const size_t byte_slabs = (size_t)(small ? 256 : 64) * pad.stride;
    const size_t lds_bytes = use_bits ? std::max(byte_slabs, (size_t)256 * bits.stride) : byte_slabs;
    CRL_REQUIRE(!lds_ok || lds_bytes <= (size_t)kLdsDynamic, "crl_tron_rollout: internal: %zu bytes of LDS", lds_bytes);
    TRON_DISPATCH_P(cfg.P, {
        if (lds_ok) {
            // opt in to > 64 KiB of dynamic LDS once per kernel instance and device (not per launch)
            static thread_local int opted_in[4][64] = {{0}};
            int dev = 0;
            CRL_HIP(hipGetDevice(&dev));
            const int which = (use_bits ? 2 : 0) + (small ? 1 : 0);
            const void *fn = use_bits ? (small ? reinterpret_cast<const void *>(&tron_rollout_bits_kernel<PP, false>)
                                               : reinterpret_cast<const void *>(&tron_rollout_bits_kernel<PP, true>))
                                      : (small ? reinterpret_cast<const void *>(&tron_rollout_lds_kernel<PP, kRowBytesSmall>)
                                               : reinterpret_cast<const void *>(&tron_rollout_lds_kernel<PP, kRowBytesLarge>));
            if (dev < 0 || dev >= 64 || !opted_in[which][dev]) {
                CRL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsDynamic));
                if (dev >= 0 && dev < 64) opted_in[which][dev] = 1;
            }
            const dim3 grid(blocks_for(B, threads));
            const dim3 block(threads);
            if (use_bits && small)
                launch(tron_rollout_bits_kernel<PP, false>, grid, block, lds_bytes, true, cfg, g, pad, bits, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, board, heads, dirs, deaths, st);
            else if (use_bits)
                launch(tron_rollout_bits_kernel<PP, true>, grid, block, lds_bytes, true, cfg, g, pad, bits, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, board, heads, dirs, deaths, st);
            else if (small)
                launch(tron_rollout_lds_kernel<PP, kRowBytesSmall>, grid, block, lds_bytes, true, cfg, g, pad, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, board, heads, dirs, deaths, st);
            else
                launch(tron_rollout_lds_kernel<PP, kRowBytesLarge>, grid, block, lds_bytes, true, cfg, g, pad, B,
                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, board, heads, dirs, deaths, st);
        } else {
            launch(tron_rollout_kernel<PP>, dim3(blocks_for(B, 256)), dim3(256), (size_t)0, true, cfg, g, B,
                   (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, T, board, heads, dirs, deaths, st);
        }
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                     int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                     crl_tron_stats st, uint32_t flags, void *stream)
{
    return tron_rollout_impl(ctx, B, seed, first_env_id, T, board, heads, dirs, deaths, st, flags, stream, nullptr, nullptr);
}

int crl_tron_rollout_timed(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                           int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                           crl_tron_stats st, uint32_t flags, void *stream, void *start_event, void *stop_event)
{
    CRL_REQUIRE(T > 0 || (!start_event && !stop_event), "crl_tron_rollout_timed: no launch to attach the events to (T = 0)");
    return tron_rollout_impl(ctx, B, seed, first_env_id, T, board, heads, dirs, deaths, st, flags, stream,
                             (hipEvent_t)start_event, (hipEvent_t)stop_event);
}

#ifdef CRL_QUAD_STAMPS
int crl_diag_quad_stamps(uint64_t *out, int n)      /* diagnostic builds only: copies n stamps (4 per wave) to the host */
{
    CRL_HIP(hipDeviceSynchronize());
    CRL_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_quad_stamps), (size_t)n * sizeof(uint64_t)));
    return CRL_OK;
}
#endif

int crl_tron_check_state(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads, int32_t *n_bad,
                         void *stream)
{
    TRON_CTX_CHECK("crl_tron_check_state");
    CRL_REQUIRE(board && heads && n_bad, "crl_tron_check_state: NULL pointer");
    hipLaunchKernelGGL(tron_check_state_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, ctx->tron.P,
                       ctx->tron.N * ctx->tron.N, B, board, heads, n_bad);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, uint32_t *tcount, int advance,
                    int8_t *actions, void *stream)
{
    TRON_CTX_CHECK("crl_tron_sample");
    CRL_REQUIRE(tcount && actions, "crl_tron_sample: NULL pointer");
    TRON_DISPATCH_P(ctx->tron.P, {
        hipLaunchKernelGGL((tron_sample_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, (hipStream_t)stream, B,
                           (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, tcount, advance, actions);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_observe(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads,
                     const int8_t *dirs, const int8_t *deaths, const int8_t *player,
                     int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths, void *stream)
{
    TRON_CTX_CHECK("crl_tron_observe");
    CRL_REQUIRE(board && heads && dirs && deaths && player, "crl_tron_observe: NULL input pointer");
    CRL_REQUIRE(obs_board && obs_heads && obs_dirs && obs_deaths, "crl_tron_observe: NULL output pointer");
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    CRL_REQUIRE((NN % 16 != 0) || ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0), "crl_tron_observe: boards must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool aligned16 = ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0);
    const int mode = (NN % 16 == 0) ? 1 : (aligned16 && NN >= 16) ? 2 : 0;
    const int64_t items = mode ? B * (int64_t)NN / 16 : B * (int64_t)NN;
    const unsigned grid = (unsigned)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256 < 1 ? 1 : (items + 255) / 256);
    // m = floor(2^32 / c) + 1 for c = chunks per board: umulhi(i, m) == i / c for every chunk id i with i * c < 2^32 (the
    // error term m * c - 2^32 is at most c); batches beyond that keep the 64-bit division
    const uint32_t cpb = (uint32_t)(NN / 16);
    const uint32_t inv_cp = (mode == 1 && cpb > 1 && (uint64_t)items * cpb < ((uint64_t)1 << 32)) ? (uint32_t)(((uint64_t)1 << 32) / cpb) + 1u : 0u;
    hipLaunchKernelGGL(tron_observe_board_kernel, dim3(grid), dim3(256), 0, s, NN, cfg.P, inv_cp, crl_stream_nt((int64_t)2 * NN * B, true), B,
                       board, player, obs_board, mode, ~(uint64_t)0 / (uint64_t)NN + 1u);
    TRON_DISPATCH_P(cfg.P, {
        hipLaunchKernelGGL((tron_observe_players_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, s, B,
                           heads, dirs, deaths, player, obs_heads, obs_dirs, obs_deaths);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_observe_all(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads, const int8_t *dirs,
                         const int8_t *deaths, int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths,
                         void *stream)
{
    TRON_CTX_CHECK("crl_tron_observe_all");
    CRL_REQUIRE(board && heads && dirs && deaths, "crl_tron_observe_all: NULL input pointer");
    CRL_REQUIRE(obs_board && obs_heads && obs_dirs && obs_deaths, "crl_tron_observe_all: NULL output pointer");
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    CRL_REQUIRE((NN % 16 != 0) || ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0), "crl_tron_observe_all: boards must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0);
    const int wide = !aligned ? 0 : (B * (int64_t)NN) % 16 == 0 ? 1 : 2;     // 2: the observers' planes are off the 16-byte grid
    const int64_t items = wide ? B * (int64_t)NN / 16 : B * (int64_t)NN;
    const unsigned grid = (unsigned)((items + 255) / 256 > 16384 ? 16384 : (items + 255) / 256 < 1 ? 1 : (items + 255) / 256);
    TRON_DISPATCH_P(cfg.P, {
        hipLaunchKernelGGL((tron_observe_all_kernel<PP>), dim3(grid), dim3(256), 0, s, NN, B, board, obs_board,
                           crl_stream_nt((int64_t)(PP + 1) * NN * B, true), wide);
        hipLaunchKernelGGL((tron_observe_all_players_kernel<PP>), dim3(blocks_for(B, 256)), dim3(256), 0, s, B,
                           heads, dirs, deaths, obs_heads, obs_dirs, obs_deaths);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                          int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                          const int8_t *actions, uint32_t *tcount,
                          int8_t *rewards, uint8_t *terminal, uint8_t *winners,
                          int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths,
                          uint32_t flags, void *stream)
{
    TRON_CTX_CHECK("crl_tron_step_observe");
    CRL_REQUIRE(board && heads && dirs && deaths, "crl_tron_step_observe: NULL state pointer");
    CRL_REQUIRE(actions || tcount, "crl_tron_step_observe: actions and tcount are both NULL (nothing to play)");
    CRL_REQUIRE(rewards && terminal && winners, "crl_tron_step_observe: NULL step output pointer");
    CRL_REQUIRE(obs_board && obs_heads && obs_dirs && obs_deaths, "crl_tron_step_observe: NULL observation pointer");
    CRL_REQUIRE((flags & ~CRL_STEP_AUTO_RESET) == 0, "crl_tron_step_observe: unknown flags 0x%x", flags);
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    CRL_REQUIRE((NN % 16 != 0) || ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0), "crl_tron_step_observe: boards must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const TronGeom g = geom_of(cfg);
    // the fused kernel needs whole 16-byte chunks per board, the 8-entry relabelling table (P <= 7) and G boards in LDS;
    // everything else takes the same three kernels the separate entry points launch (identical results either way)
    const int slab = NN + 16;
    const int G = (64 * slab + 64 <= 48 * 1024) ? 64 : (16 * slab + 16 <= 48 * 1024) ? 16 : 0;
    if ((NN % 16) == 0 && cfg.P <= 8 && G > 0) {
        // floor(2^32 / chunks per board) + 1: exact quotients for chunk ids < 2^16 (one chunk per board: unused)
        const uint32_t inv_cp = NN == 16 ? 0u : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(NN / 16)) + 1u;
        const size_t lds_bytes = (size_t)G * slab + G;
        const dim3 grid(blocks_for(B, G));
        const uint32_t kflags = flags | (crl_stream_nt((int64_t)(cfg.P + 1) * NN * B, false) ? kStepObserveNT : 0u);
        switch (cfg.P) {
#define CRL_SO_CASE(P_)                                                                                                   \
        case P_:                                                                                                          \
            if (G == 64)                                                                                                  \
                hipLaunchKernelGGL((tron_step_observe_kernel<P_, 64>), grid, dim3(256), lds_bytes, s, cfg, g, inv_cp, B,   \
                                   (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, board, heads, dirs, deaths,     \
                                   actions, tcount, rewards, terminal, winners, obs_board, obs_heads, obs_dirs, obs_deaths, kflags); \
            else                                                                                                          \
                hipLaunchKernelGGL((tron_step_observe_kernel<P_, 16>), grid, dim3(256), lds_bytes, s, cfg, g, inv_cp, B,   \
                                   (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, board, heads, dirs, deaths,     \
                                   actions, tcount, rewards, terminal, winners, obs_board, obs_heads, obs_dirs, obs_deaths, kflags); \
            break;
            CRL_SO_CASE(1) CRL_SO_CASE(2) CRL_SO_CASE(3) CRL_SO_CASE(4) CRL_SO_CASE(5) CRL_SO_CASE(6) CRL_SO_CASE(7) CRL_SO_CASE(8)
#undef CRL_SO_CASE
            default: crl_set_error("tron: P=%d out of range", cfg.P); return CRL_EINVAL;
        }
        CRL_LAUNCH_CHECK();
        return CRL_OK;
    }
    // boards that are not whole 16-byte chunks (19 x 19, the reference's default, among them): the flat-stream kernel, when
    // the batch is a multiple of 16 games and the buffers are aligned
    {
        const int Gf = (64 * NN + 64 + 32 <= 48 * 1024) ? 64 : (16 * NN + 64 + 32 <= 48 * 1024) ? 16 : 0;
        if ((NN % 16) != 0 && cfg.P <= 8 && Gf > 0 && ((((uintptr_t)board | (uintptr_t)obs_board) & 15) == 0)) {
            const bool ragged = B % 16 != 0;                    // (round 5: the planes off the 16-byte grid, the last run not whole chunks)
            const uint32_t inv_nn = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)NN) + 1u;     // exact for byte offsets < 64 * NN
            const size_t lds_bytes = (size_t)((Gf * NN + 15) & ~15) + Gf + 16;
            const dim3 grid(blocks_for(B, Gf));
            const uint32_t kflags = flags | (crl_stream_nt((int64_t)(cfg.P + 1) * NN * B, false) ? kStepObserveNT : 0u);
            switch (cfg.P) {
#define CRL_SOF_LAUNCH(P_, G_, R_)                                                                                        \
                    hipLaunchKernelGGL((tron_step_observe_flat_kernel<P_, G_, R_>), grid, dim3(256), lds_bytes, s, cfg, g, inv_nn, B, \
                                       (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, board, heads, dirs, deaths,   \
                                       actions, tcount, rewards, terminal, winners, obs_board, obs_heads, obs_dirs, obs_deaths, kflags)
#define CRL_SOF_CASE(P_)                                                                                                   \
            case P_:                                                                                                      \
                if (Gf == 64) { if (ragged) CRL_SOF_LAUNCH(P_, 64, true); else CRL_SOF_LAUNCH(P_, 64, false); }           \
                else { if (ragged) CRL_SOF_LAUNCH(P_, 16, true); else CRL_SOF_LAUNCH(P_, 16, false); }                    \
                break;
                CRL_SOF_CASE(1) CRL_SOF_CASE(2) CRL_SOF_CASE(3) CRL_SOF_CASE(4) CRL_SOF_CASE(5) CRL_SOF_CASE(6) CRL_SOF_CASE(7) CRL_SOF_CASE(8)
#undef CRL_SOF_CASE
#undef CRL_SOF_LAUNCH
                default: crl_set_error("tron: P=%d out of range", cfg.P); return CRL_EINVAL;
            }
            CRL_LAUNCH_CHECK();
            return CRL_OK;
        }
    }
    // every other shape: one game per workgroup, bytes (tron_step_observe_any_kernel)
    CRL_REQUIRE(B < ((int64_t)1 << 31), "crl_tron_step_observe: B too large for the one-game-per-workgroup kernel");
    CRL_REQUIRE(NN <= 60 * 1024, "crl_tron_step_observe: board too large for LDS");
    TRON_DISPATCH_P(cfg.P, {
        hipLaunchKernelGGL((tron_step_observe_any_kernel<PP>), dim3((unsigned)B), dim3(256), (size_t)((NN + 15) & ~15), s, cfg, g, B,
                           (uint32_t)seed, (uint32_t)(seed >> 32), first_env_id, board, heads, dirs, deaths,
                           actions, tcount, rewards, terminal, winners, obs_board, obs_heads, obs_dirs, obs_deaths, flags);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

static int tron_next_state64_launch(const char *fn, const crl_ctx *ctx, int64_t B, int64_t *board, int64_t *heads, int64_t *directions,
                                    int64_t *deaths, const int64_t *actions, int64_t *rewards, uint8_t *terminal, uint8_t *winners,
                                    int64_t *obs_board, int64_t *obs_heads, int64_t *obs_directions, int64_t *obs_deaths, void *stream,
                                    const bool by_value, uint32_t *flag, const uint32_t seq)
{
    CRL_REQUIRE(ctx != nullptr && ctx->game == CRL_GAME_TRON, "%s: ctx is not a tron context", fn);
    CRL_REQUIRE(B > 0 && B <= ((int64_t)1 << 31), "%s: B=%lld out of range", fn, (long long)B);
    CRL_REQUIRE(board && heads && directions && deaths && actions, "%s: NULL state / action pointer", fn);
    CRL_REQUIRE(!obs_board == !obs_heads && !obs_board == !obs_directions && !obs_board == !obs_deaths,
                "%s: the four observation pointers go together (all or none)", fn);
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    CRL_REQUIRE(!obs_board || NN <= 60 * 1024, "%s: board too large for LDS", fn);
    TronVec64 vec = {};
    if (by_value)                                               // host-visible state (crl_host_alloc): read here, passed as arguments
        for (int p = 0; p < cfg.P; ++p) {
            vec.h[p] = (int32_t)heads[p]; vec.d[p] = (int32_t)directions[p]; vec.k[p] = (int32_t)deaths[p]; vec.a[p] = (int32_t)actions[p];
        }
    hipStream_t s = (hipStream_t)stream;
    const TronGeom g = geom_of(cfg);
    const size_t lds_bytes = obs_board ? (size_t)((NN + 15) & ~15) : 0;
    TRON_DISPATCH_P(cfg.P, {
        hipLaunchKernelGGL((tron_next_state64_kernel<PP>), dim3((unsigned)B), dim3(256), lds_bytes, s, g, B, board, heads,
                           directions, deaths, actions, rewards, terminal, winners, obs_board, obs_heads, obs_directions, obs_deaths,
                           vec, by_value, flag, seq);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_next_state_inplace64(const crl_ctx *ctx, int64_t B, int64_t *board, int64_t *heads, int64_t *directions,
                                  int64_t *deaths, const int64_t *actions, int64_t *rewards, uint8_t *terminal, uint8_t *winners,
                                  int64_t *obs_board, int64_t *obs_heads, int64_t *obs_directions, int64_t *obs_deaths, void *stream)
{
    return tron_next_state64_launch("crl_tron_next_state_inplace64", ctx, B, board, heads, directions, deaths, actions, rewards, terminal,
                                    winners, obs_board, obs_heads, obs_directions, obs_deaths, stream, false, nullptr, 0u);
}

int crl_tron_next_state_inplace64_host(const crl_ctx *ctx, int64_t *board, int64_t *heads, int64_t *directions,
                                       int64_t *deaths, const int64_t *actions, int64_t *rewards, uint8_t *terminal, uint8_t *winners,
                                       int64_t *obs_board, int64_t *obs_heads, int64_t *obs_directions, int64_t *obs_deaths, void *stream,
                                       uint32_t *flag, uint32_t seq, double timeout_s)
{
    CRL_REQUIRE(flag != nullptr, "crl_tron_next_state_inplace64_host: flag is NULL");
    const int rc = tron_next_state64_launch("crl_tron_next_state_inplace64_host", ctx, 1, board, heads, directions, deaths, actions, rewards,
                                            terminal, winners, obs_board, obs_heads, obs_directions, obs_deaths, stream, true, flag, seq);
    if (rc != CRL_OK) return rc;
    return crl_spin_mapped(stream, flag, seq, timeout_s, "crl_tron_next_state_inplace64_host");
}

int crl_tron_relative_player_inplace64(const crl_ctx *ctx, int64_t B, int64_t *board, int64_t num_players, const int64_t *player,
                                       void *stream)
{
    TRON_CTX_CHECK("crl_tron_relative_player_inplace64");
    CRL_REQUIRE(board && player, "crl_tron_relative_player_inplace64: NULL pointer");
    CRL_REQUIRE(num_players > 0, "crl_tron_relative_player_inplace64: num_players=%lld (the reference divides by it)", (long long)num_players);
    const int NN = ctx->tron.N * ctx->tron.N;
    const int64_t total = B * (int64_t)NN;
    hipLaunchKernelGGL(tron_relative_player64_kernel, dim3(std::min<unsigned>(blocks_for(total, 256), 8192u)), dim3(256), 0,
                       (hipStream_t)stream, total, NN, board, num_players, player);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_tron_ranking(const crl_ctx *ctx, int64_t B, const int8_t *board, const int8_t *deaths, int8_t *rank, void *stream)
{
    TRON_CTX_CHECK("crl_tron_ranking");
    CRL_REQUIRE(board && deaths && rank, "crl_tron_ranking: NULL pointer");
    const crl_tron_cfg &cfg = ctx->tron;
    const int NN = cfg.N * cfg.N;
    const bool wide = (NN % 16) == 0 && (((uintptr_t)board & 15) == 0);
    const uint32_t inv_cp = (!wide || NN == 16) ? 0u : (uint32_t)(((uint64_t)1 << 32) / (uint64_t)(NN / 16)) + 1u;
    TRON_DISPATCH_P(cfg.P, {
        if (wide)
            hipLaunchKernelGGL((tron_ranking_wide_kernel<PP>), dim3(blocks_for(B, 64)), dim3(256), 0, (hipStream_t)stream,
                               NN, inv_cp, B, board, deaths, rank);
        else if (NN >= 16 && NN <= 4096 && (((uintptr_t)board & 15) == 0))      // any other board up to 64 x 64: the flat stream
            hipLaunchKernelGGL((tron_ranking_flat_kernel<PP>), dim3(blocks_for(B, 64)), dim3(256), 0, (hipStream_t)stream,
                               NN, (uint32_t)(((uint64_t)1 << 32) / (uint64_t)NN) + 1u, B, board, deaths, rank);
        else
            hipLaunchKernelGGL((tron_ranking_kernel<PP>), dim3(blocks_for(B, 4)), dim3(256), 0, (hipStream_t)stream,
                               NN, B, board, deaths, rank);
    });
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

} // extern "C"
