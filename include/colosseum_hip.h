/* colosseum_hip.h -- C ABI of libcolosseum_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for the colosseumrl hot path: the batched equivalents of
 *   BaseEnvironment.new_state / next_state / valid_actions (+ the winners output)
 * for the Tron, TicTacToe and Blokus environments.  Plain pointers and sizes, no
 * torch types.  The reference's only native boundary is the Cython module
 *   colosseumrl/envs/tron/CyTronGrid.pyx:3-7   next_state_inplace(board, heads, directions, deaths, actions)
 *   colosseumrl/envs/tron/CyTronGrid.pyx:65    relative_player_inplace(board, num_players, player)
 * (C-contiguous buffers mutated in place, caller owns everything, no error path);
 * the entry points below keep those conventions and add what the Python layers
 * around it compute (rewards / terminal / winners, resets, legal-move sets).
 *
 * Conventions
 *   - every `void *`/typed pointer marked DEVICE is a device (HBM) address owned by the caller
 *     (the Python host passes torch-ROCm tensors' data_ptr()); kernels mutate state in place;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are
 *     asynchronous on it, never synchronise the host, and are graph-capturable
 *     (crl_*_create / crl_destroy are the only calls that allocate or copy synchronously);
 *   - every function returns 0 on success and a negative CRL_E* code on error; nothing throws
 *     across the ABI; crl_last_error() returns a thread-local message for the last failure;
 *   - contexts are immutable after creation: any number of host threads may use one context
 *     concurrently on different buffers/streams (several env instances live under the GIL in
 *     the reference's MatchmakingServer.py:128-135);
 *   - env b of a batch of B owns board[b*cells .. (b+1)*cells) and element [p*B + b] of every
 *     per-player array ("[P][B]" struct-of-arrays: consecutive lanes touch consecutive bytes).
 *
 * There is NO CPU implementation behind this ABI.  The CPU restatement used by the tests lives
 * in oracle/ and is a different library.
 */
#ifndef COLOSSEUM_HIP_H
#define COLOSSEUM_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRL_OK            0
#define CRL_EINVAL       -1   /* bad argument (null pointer, size out of range, ...) */
#define CRL_EHIP         -2   /* HIP runtime error (message has hipGetErrorString) */
#define CRL_ENODEV       -3   /* no gfx950 device visible */
#define CRL_EUNSUPPORTED -4

#define CRL_TRON_MAX_P   8
#define CRL_TTT_MAX_P    8
#define CRL_TTT_MAX_LINES 256

/* flags for *_step */
#define CRL_STEP_AUTO_RESET 1u  /* after writing the step outputs, reset every env that just became terminal */
/* crl_tron_step only: pin one of its two interchangeable kernels (identical results; the default is the faster one) */
#define CRL_STEP_BYTES      2u  /* byte probes in HBM, one lane per game, nothing staged */
#define CRL_STEP_STAGED     4u  /* boards read once, coalesced, into LDS (boards of whole 16-byte chunks that fit; else ignored) */
/* flags for crl_tron_rollout */
#define CRL_ROLLOUT_NO_LDS  2u  /* force the lane-per-game global-memory kernel even when the boards would fit in LDS */
#define CRL_ROLLOUT_BYTES   4u  /* force the lane-per-game byte-per-cell LDS kernel */
#define CRL_ROLLOUT_BITS    8u  /* force the lane-per-game bitboard LDS kernel with replay epilogue (default for boards 21..40 wide with P > 4 and T >= 256) */
#define CRL_ROLLOUT_QUAD   16u  /* the lane-per-player kernel (four lanes per game): boards up to 20x20 with at most 4 players.
                                 * It is the default there; elsewhere the flag is ignored */
#define CRL_ROLLOUT_QBITS  32u  /* the lane-per-player bitboard kernel with replay epilogue (at most 4 players, boards up to
                                 * 40x40): the default for boards 21..40 wide in launches of more than 18 steps (shorter ones
                                 * stay in global memory: the kernel's fixed cost is worth that many steps there); with more
                                 * than 4 players the flag is ignored */

#define CRL_ROLLOUT_PAIR  128u  /* the byte-slab kernel with TWO lanes per game (boards up to 20x20, one or two players; elsewhere the
                                 * flag is ignored): the default there for launches of 32 steps and more (256 where a row is not whole dwords) */
#define CRL_ROLLOUT_GQUAD  64u  /* one lane per player on boards in GLOBAL memory (any board size, at most 4 players; with more
                                 * the flag is ignored): the default wherever the boards are not played out of LDS */

typedef struct crl_ctx crl_ctx;   /* opaque, immutable after creation */

const char *crl_last_error(void);
/* ABI revision of this header: bumped whenever a struct passed by value (crl_*_stats), an argument list or the RNG
 * contract of the sampled agents changes.  crl_version() returns the revision the LIBRARY was built from; a binding must
 * refuse a library whose revision differs from the header it was written against (colosseumrl_amd/_native.py does).
 * 100: round 1.  101: crl_tron_stats gained `packed`.  102: TicTacToe sampled agent draws 8 plies per Philox block.
 * 104: round 4.  105: crl_stream_wait_mapped.  106: crl_diag_issue_probe.  107: crl_blokus_fits.  108: crl_diag_bounds.
 * 109: crl_blokus_step / _step_observe place ANY action as the reference's next_state does (numpy index rules, extended ids,
 *      CRL_BLOKUS_*_ERROR codes in the reward slot).  110: crl_tron_next_state_inplace64 (+ _host) / _relative_player_inplace64.  111: crl_ttt_step_board_host. */
#define CRL_ABI_VERSION 111
int crl_version(void);
/* number of visible HIP devices, or a negative code */
int crl_device_count(void);
void crl_destroy(crl_ctx *ctx);

/* ------------------------------------------------------------------ host staging for single-state callers
 * The reference's API is one state per call (BaseEnvironment.next_state, match_server.py:201-203): for such a caller a
 * hipMemcpy each way costs more than the step.  crl_host_alloc returns `bytes` of zeroed, page-locked host memory that
 * the GPU maps (hipHostMalloc, mapped + coherent): every entry point of this header accepts `*device` (+ offsets)
 * wherever it takes a DEVICE pointer, so a B = 1 call reads its state from and writes its results to host memory
 * directly -- no copy, one crl_stream_synchronize per call.  `device` may be NULL when the caller only needs `*host`
 * (both name the same bytes; under ROCm's unified addressing they are usually equal).
 * crl_stream_create gives a non-blocking stream of the current device, so that env instances on different host
 * threads (MatchmakingServer.py:128-135) do not wait for each other's work. */
int crl_host_alloc(size_t bytes, void **host, void **device);
int crl_host_free(void *host);
int crl_stream_create(void **stream);
/* Only for streams crl_stream_create returned, and only once NOTHING else refers to them: a stream that was handed to another
 * runtime object (e.g. wrapped as torch.cuda.ExternalStream, or with events / graphs of another library recorded on it) stays
 * referenced there, and destroying it under that object crashes the process at ITS next use of the handle, not inside this
 * library (seen in round 4: a segmentation fault at interpreter exit).  Let the other owner go first, or give it a stream it
 * owns itself and pass THAT stream's handle to the crl_* calls instead. */
int crl_stream_destroy(void *stream);
/* the blocking calls of the ABI besides create / destroy: */
int crl_stream_synchronize(void *stream);
/* ... and the same wait done on MAPPED MEMORY: a one-thread kernel is queued behind whatever `stream` holds and stores `seq`
 * (system-scope release) into a 32-bit word of crl_host_alloc memory -- `flag_device` / `flag_host` name that word from the
 * two sides --, and the host spins on the word.  For the short launch chains of a single-state caller this returns ~3.5 us
 * earlier than hipStreamSynchronize (9.7 against 13.2 us for launch + wait, tools/ubench/mailbox_rtt.hip).  Results the
 * chain wrote to mapped memory are visible when the call returns.  The caller uses a fresh `seq` per call (any value other
 * than the word's current one).  After `timeout_s` seconds without the flag the call falls back to hipStreamSynchronize and
 * reports what that reports (a faulted stream never delivers). */
int crl_stream_wait_mapped(void *stream, uint32_t *flag_device, const volatile uint32_t *flag_host, uint32_t seq, double timeout_s);

/* ------------------------------------------------------------------ diagnostics
 * How fast does THIS device, as it runs now, issue vector instructions?  `blocks` workgroups of 256 threads (4 waves: one
 * per SIMD of a CU) each run `iters` loop trips of 64 independent 32-bit integer VALU instructions; out (DEVICE, blocks x 256
 * dwords) keeps the work alive, clk (DEVICE, 2 x uint64) receives what wave 0 counted around its loop: shader-clock ticks
 * and 100 MHz wall-clock ticks (ratio x 100 = MHz under this load).  Asynchronous; time the launch with events: wave64
 * instructions per second = blocks x 4 x iters x 64 / time.  bench.py runs it at 4 waves per SIMD next to the rollouts, so
 * that a box whose clocks are capped shows in the record (the rollout kernels are bound by instruction issue). */
int crl_diag_issue_probe(uint32_t *out, uint64_t *clk, int blocks, int iters, void *stream);
/* The bounds asserts of a -DCRL_BOUNDS build (tools/gpu_bounds.sh; GPU AddressSanitizer is not available on the pool): out12
 * (HOST) = {failed checks, code of the first, its value, its limit} for tron.hip, ttt.hip and blokus.hip -- the range checks
 * on the kernels' data-dependent LDS / table accesses.  Synchronises the device.  Returns 1 from a build with the asserts
 * (after a self-test: a check that must fail is run and must be recorded), 0 from the shipped build, which compiles no
 * check and reports zeros; negative on error. */
int crl_diag_bounds(uint32_t *out12);

/* ------------------------------------------------------------------ RNG (exposed for parity tests) */
/* out[i*4..i*4+3] = Philox-4x32-10(ctr[i*4..], key); n counters; DEVICE pointers */
int crl_philox4x32(const uint32_t *ctr, uint32_t key0, uint32_t key1, uint32_t *out, int64_t n, void *stream);

/* ------------------------------------------------------------------ Tron
 * State (reference TronGridEnvironment.py:256-263, int64 there):
 *   board  int8  [B][N*N]   0 empty, p+1 = trail/head of player p
 *   heads  int16 [P][B]     flat y*N+x
 *   dirs   int8  [P][B]     0 N(y-1) 1 E(x+1) 2 S(y+1) 3 W(x-1)
 *   deaths int8  [P][B]     0 alive, else 1-based killer id
 */
/* replaces the per-instance set-up of TronGridEnvironment.__init__/generate_start_positions
 * (TronGridEnvironment.py:92-118,183-226); start_* are HOST arrays of length P */
int crl_tron_create(int N, int P, const int16_t *start_heads, const int8_t *start_dirs, crl_ctx **out);

/* replaces TronGridEnvironment.new_state (TronGridEnvironment.py:228-263) for B envs.
 * mask (DEVICE, uint8 [B]) may be NULL = reset all; else only envs with mask[b] != 0 */
int crl_tron_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask,
                   int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths, void *stream);

/* replaces CyTronGrid.next_state_inplace (CyTronGrid.pyx:3-62) + the tail of
 * TronGridEnvironment.next_state (TronGridEnvironment.py:309-323) for B envs.
 *   actions  int8 [P][B] in {0 forward, +1 right, -1 left} (dead players' entries ignored; not validated, exactly
 *            like the reference's Cython kernel: other values are taken mod 4, so 2 would reverse)
 *   rewards  int8 [P][B] out: -1 dead, +1 alive, +10 surviving winner
 *   terminal uint8 [B]   out: alive <= 1
 *   winners  uint8 [B]   out: bitmask of alive players if terminal, else 0 (reference: None) */
int crl_tron_step(const crl_ctx *ctx, int64_t B,
                  int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                  const int8_t *actions, int8_t *rewards, uint8_t *terminal, uint8_t *winners,
                  uint32_t flags, void *stream);

/* per-env rollout bookkeeping, all DEVICE arrays (none may be NULL except `results`) */
typedef struct {
    uint32_t *tcount;       /* [B] rollout steps this env has taken so far = the RNG counter */
    uint32_t *tstep;        /* [B] steps taken in the current episode */
    uint32_t *n_episodes;   /* [B] episodes finished */
    uint32_t *win_count;    /* [P][B] */
    uint32_t *len_sum;      /* [B] */
    int32_t  *ret_sum;      /* [P][B] sum of rewards */
    uint8_t  *last_winners; /* [B] */
    uint16_t *last_len;     /* [B] */
    int32_t  *results;      /* [B][3+2P] or NULL: the per-game row the end-of-rollout gather ships, rewritten by every
                             * launch from the running totals above: n_episodes, len_sum, last_winners, win_count[P],
                             * ret_sum[P] (no separate packing pass over the SoA arrays) */
    uint16_t *packed;       /* [B][CRL_TRON_PACKED_ROW(P)] or NULL: the same row in 16-bit fields for a latency-bound
                             * gather (winners mask + episode length + P int16 returns; 16 bytes per game at
                             * P = 4 instead of 44): n_episodes, len_sum, last_winners, tstep (steps into the unfinished
                             * episode), ret_sum[P] as int16 -- the LOW 16 bits of the running totals, hence exact while
                             * the totals span at most 3,276 steps (|ret_sum| <= 10 per step); the host side ships this
                             * row only then (colosseumrl_amd/parallel.py) and the int32 row otherwise */
} crl_tron_stats;
#define CRL_TRON_PACKED_ROW(P) ((4 + (P) + 1) & ~1)   /* u16 entries per game: a whole number of dwords */

/* T fused env-steps per env with a uniform random agent and auto-reset (the benchmark loop of
 * BASELINE.md section 3; no reference counterpart -- the reference steps one env per Python call).
 * RNG contract, keyed by (seed, global env g = first_env_id + b, the env's step count c = tcount, player p);
 * one Philox call serves 8 consecutive steps of up to 4 players:
 *   W = Philox4x32-10(ctr = {g, c >> 3, p >> 2, 0x54520000}, key = {seed lo, seed hi})
 *   j = c & 7;  v = W[j >> 1] * 3^((j & 1) * 4 + (p & 3))  (mod 2^32)
 *   a = mulhi32(v, 3): 0 -> forward, 1 -> right, 2 -> left        (base-3 digits of the fraction W/2^32)
 * Boards up to 40x40 are played out of LDS (one copy in / one copy out per launch): boards up to 20x20 on a
 * byte-per-cell slab (one lane per player when P <= 4, else one lane per game), larger ones on an occupancy bitboard
 * (one lane per player when P <= 4; else one lane per game and T >= 256) whose unfinished episode is replayed with
 * owners at the end of the launch (P <= 4: by a second kernel queued behind the first).  Boards above 40x40 stay in global memory -- and so do, with P <= 4, launches too short
 * to earn an LDS kernel's copies back (one step on boards up to 20x20, up to 18 steps on boards 21..40 wide, 32 where a row
 * is not whole dwords): one lane per player (CRL_ROLLOUT_GQUAD; no fixed cost per launch) or one lane per game
 * (CRL_ROLLOUT_NO_LDS; episode tags, one pass over the boards per launch: P > 4, and launches on boards above 40x40 that are
 * long enough to earn that pass back -- from 15..57 steps on with one to three players, from 49..201 with four above 44x44).
 * CRL_ROLLOUT_BYTES / _BITS /
 * _QUAD / _QBITS pin one of the LDS kernels.  All give identical results.  The LDS kernels rely on the invariant of every state
 * produced by crl_tron_reset / crl_tron_step / crl_tron_rollout: board[heads[p]] == p + 1 for every player; callers
 * that upload hand-made states run crl_tron_check_state first (heads outside the board are clamped onto it, so a
 * broken state gives wrong results, never a wild access). */
int crl_tron_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                     int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                     crl_tron_stats stats, uint32_t flags, void *stream);

/* crl_tron_rollout with HIP events ATTACHED TO THE DISPATCHES: start_event (hipEvent_t as void*, may be NULL) is updated
 * with the start of the rollout's first kernel, stop_event (may be NULL) with the end of its last one
 * (hipExtLaunchKernelGGL) -- the kernel time of a short launch without the marker packets and the ~3.6 us of host time
 * two hipEventRecord calls around it cost.  The events must have been created (and, for hipEventElapsedTime, are complete
 * once the stream is synchronised).  No reference counterpart. */
int crl_tron_rollout_timed(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                           int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                           crl_tron_stats stats, uint32_t flags, void *stream, void *start_event, void *stop_event);

/* Counts (adds to *n_bad, a device int32 the caller zeroes) the games whose state breaks what every state produced by
 * crl_tron_reset / _step / _rollout satisfies and the LDS rollout kernels rely on: every head inside the board and
 * board[heads[p]] == p + 1.  For callers that upload hand-made states before a rollout.  No reference counterpart. */
int crl_tron_check_state(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads, int32_t *n_bad,
                         void *stream);

/* The random agent of crl_tron_rollout as a stand-alone call: actions[p*B+b] (0 forward, +1 right, -1 left, the
 * crl_tron_step encoding) of step tcount[b] of env first_env_id + b under the RNG contract above; advance != 0 also
 * increments tcount.  T x (crl_tron_sample; crl_tron_step with CRL_STEP_AUTO_RESET) leaves the same state as
 * crl_tron_rollout(T); callers overwrite the rows of the players they control.  No reference counterpart. */
int crl_tron_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, uint32_t *tcount, int advance,
                    int8_t *actions, void *stream);

/* replaces CyTronGrid.relative_player_inplace (CyTronGrid.pyx:65-71) + the rolls of
 * TronGridEnvironment.state_to_observation (TronGridEnvironment.py:385-405), fully observable branch.
 * player int8 [B]: observer of env b.  ANY id is an observer, as in the reference (ABI 109): the rolled vectors use numpy's
 * modulo ((arange + player) % P, :393), the board C's remainder (CyTronGrid.pyx:1 cdivision=True) -- ids -128..P observe as
 * player mod P, beyond P low trail ids come out <= 0 exactly as there.  Outputs have the shapes of the state arrays. */
int crl_tron_observe(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads,
                     const int8_t *dirs, const int8_t *deaths, const int8_t *player,
                     int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths, void *stream);

/* The same observation for ALL P observers in one pass (what a self-play learner consumes every step):
 * obs_board int8 [P][B][N*N], obs_heads int16 [P][P][B], obs_dirs / obs_deaths int8 [P][P][B]; slice [p] equals
 * crl_tron_observe with player[b] = p.  Streams N*N bytes in and P*N*N bytes out per game. */
int crl_tron_observe_all(const crl_ctx *ctx, int64_t B, const int8_t *board, const int16_t *heads, const int8_t *dirs,
                         const int8_t *deaths, int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths,
                         void *stream);

/* One launch for what a self-play learner does every step: [sample ->] next_state -> state_to_observation of every
 * player, i.e. TronGridEnvironment.next_state (TronGridEnvironment.py:265-323, called per tick from match_server.py:201-203)
 * followed by state_to_observation (:363-420, match_server.py:218) for all P observers.  Exactly equivalent to
 *   [crl_tron_sample(..., advance = 1);]  crl_tron_step(..., flags);  crl_tron_observe_all(...)
 * with actions == NULL meaning "draw them with the rollout's random agent at tcount[b] and advance tcount" (tcount may be
 * NULL when actions are given; it is not touched then).  The boards are read from HBM once (coalesced, into LDS),
 * stepped there, and the P relabelled copies streamed out: N*N bytes in + P*N*N bytes out per game.  Boards with
 * N*N % 16 != 0 (the reference's default 19x19 among them) take the same route as one flat byte stream per workgroup when the
 * batch is a multiple of 16 games and the board buffers are 16-byte aligned; other batches of such boards, P = 8, or boards too
 * large for 16 LDS slabs take a kernel with one game per workgroup and byte accesses: still one launch, same results,
 * actions == NULL allowed. */
int crl_tron_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                          int8_t *board, int16_t *heads, int8_t *dirs, int8_t *deaths,
                          const int8_t *actions, uint32_t *tcount,
                          int8_t *rewards, uint8_t *terminal, uint8_t *winners,
                          int8_t *obs_board, int16_t *obs_heads, int8_t *obs_dirs, int8_t *obs_deaths,
                          uint32_t flags, void *stream);

/* The Cython module's OWN functions with their own types (the reference's only native boundary), for callers that hold
 * reference states:
 *   CyTronGrid.pyx:3-7   next_state_inplace(long[:, ::1] board, long[::1] heads, long[::1] directions, long[::1] deaths,
 *                                           const long[::1] actions)
 *   CyTronGrid.pyx:65    relative_player_inplace(long[:, ::1] board, const long num_players, const long player)
 * int64, C-contiguous, the reference's layout (board [N][N]; heads / directions / deaths / actions [P]; actions in
 * {0 forward, +1 right, -1 left}), MUTATED IN PLACE, caller owns every buffer, no error path for the contents (as there).
 * The direction is computed as there, (directions[i] + action + 4) % 4 with C's remainder: sums below -4 give a negative
 * direction, the player then runs into the cell it stands on and the negative direction is stored -- 240 direct calls of the
 * reference function with such arguments are a golden fixture (values are read as 32-bit integers; heads must lie on the board).  B games lie one behind the other ([B][N*N], [B][P]); the
 * single-state drop-in class calls it with B = 1 on crl_host_alloc memory, so TronGridEnvironment.next_state neither
 * converts its four arrays to the batched steppers' int8 / int16 struct-of-arrays nor back.  The step touches the board where the
 * reference does: <= P probes, <= P trail stores.  Optional outputs (NULL to skip), what TronGridEnvironment.next_state
 * computes around the call (TronGridEnvironment.py:309-321): rewards int64 [B][P] (-1 dead, +1 alive, +10 surviving
 * winner), terminal uint8 [B], winners uint8 [B] (bitmask of the alive players if terminal, else 0).  Optional
 * observations of the NEW state for all P observers (all four pointers or none; TronGridEnvironment.py:385-405):
 * obs_board int64 [B][P][N*N] (relative_player_inplace of a copy with player = p + 1), obs_heads / obs_directions /
 * obs_deaths int64 [B][P][P] (rolled so index 0 is the observer). */
int crl_tron_next_state_inplace64(const crl_ctx *ctx, int64_t B, int64_t *board, int64_t *heads, int64_t *directions,
                                  int64_t *deaths, const int64_t *actions, int64_t *rewards, uint8_t *terminal, uint8_t *winners,
                                  int64_t *obs_board, int64_t *obs_heads, int64_t *obs_directions, int64_t *obs_deaths, void *stream);
/* The single-state form of the call above, ONE BLOCKING call for a caller whose state lies in crl_host_alloc memory (B = 1;
 * every pointer is such memory, which host and GPU address alike under ROCm's unified addressing -- a binding checks that
 * crl_host_alloc returned equal host / device addresses before it uses this entry): launch + completion.  Because the host can
 * read the state, the player vectors (heads, directions, deaths, actions) travel BY VALUE in the kernel arguments -- the
 * kernel's first data access is then the board probe, one PCIe round trip instead of two -- and the kernel publishes
 * completion itself: `seq` into *flag (a 32-bit word of crl_host_alloc memory; a fresh value per call) behind its last
 * store, on which the host spins (timeout_s and the fallback as crl_stream_wait_mapped).  No signal kernel, one call through
 * the binding instead of two.  Results as above, visible when the call returns. */
int crl_tron_next_state_inplace64_host(const crl_ctx *ctx, int64_t *board, int64_t *heads, int64_t *directions,
                                       int64_t *deaths, const int64_t *actions, int64_t *rewards, uint8_t *terminal, uint8_t *winners,
                                       int64_t *obs_board, int64_t *obs_heads, int64_t *obs_directions, int64_t *obs_deaths, void *stream,
                                       uint32_t *flag, uint32_t seq, double timeout_s);
/* board[b][i][j] > 0  ->  ((board - player[b] + num_players) % num_players) + 1 with C's remainder (cdivision=True), in place;
 * player int64 [B] as the reference passes it (TronGridEnvironment.py:390: the observer's id + 1). */
int crl_tron_relative_player_inplace64(const crl_ctx *ctx, int64_t B, int64_t *board, int64_t num_players, const int64_t *player,
                                       void *stream);

/* replaces TronGridEnvironment.compute_ranking (TronGridEnvironment.py:483-508) for B games:
 * rank int8 [P][B], 0 = best; trail-length scores, the mutual-kill tie rule (with its deaths[-1] read for
 * alive players) and competition ranking, exactly as the reference computes them */
int crl_tron_ranking(const crl_ctx *ctx, int64_t B, const int8_t *board, const int8_t *deaths, int8_t *rank, void *stream);

/* ------------------------------------------------------------------ TicTacToe (D0 x D1 x D2, K in a row, P players)
 * State (reference tictactoe_2p_env.py:165-169: (board int8, winner)):
 *   occ     uint32 [P][B]  bit c set = flat cell c (row-major) holds player p's mark
 *   winner  int8   [B]     -1 = None (sticky once set)
 *   to_move int8   [B]
 * For boards of at most 16 cells (the reference's 3x3 and 3x5) crl_ttt_create also puts an 8 KB table of the winning masks
 * on the device that is current at the call; crl_ttt_rollout uses it when it runs on that device (and computes the win test
 * as everywhere else when it does not, or when no device was there at create time; the environment variable
 * CRL_TTT_NO_WIN_TABLE, read by crl_ttt_create, skips the table: tests of that path).  crl_destroy frees it.
 */
int crl_ttt_create(int D0, int D1, int D2, int K, int P, crl_ctx **out);
/* number of win lines and a HOST copy of them (for tests); lines may be NULL */
int crl_ttt_lines(const crl_ctx *ctx, uint32_t *lines, int cap);
int crl_ttt_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask,
                  uint32_t *occ, int8_t *winner, int8_t *to_move, void *stream);
/* replaces TicTacToe{2,3,4}PlayerEnv.next_state (tictactoe_2p_env.py:240-315).
 *   action int8 [B]: flat cell index, or -1 for '' ; an occupied cell is a no-op that still passes the turn
 *   reward int8 [B] (the mover's), terminal uint8 [B], winners int8 [B] (-1 = None, else the winner) */
int crl_ttt_step(const crl_ctx *ctx, int64_t B, uint32_t *occ, int8_t *winner, int8_t *to_move,
                 const int8_t *action, int8_t *reward, uint8_t *terminal, int8_t *winners,
                 uint32_t flags, void *stream);
/* replaces TicTacToe*.valid_actions (tictactoe_2p_env.py:317-348): empties bitmask uint32 [B] */
int crl_ttt_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, uint32_t *valid, void *stream);
/* int8 [B][cells] board (4-byte aligned) in the reference encoding (-1 empty), relative to `player` when player != NULL
 * (reference _relative_player_id, tictactoe_2p_env.py:26-27, modulus given by rel_mod) */
int crl_ttt_board(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const int8_t *player, int rel_mod,
                  int8_t *board, void *stream);
/* One launch for a ply of every game, as a learner / vector env needs it: [sample ->] next_state -> valid_actions and
 * state_to_observation of the player to move NEXT.  Exactly equivalent to
 *   [crl_ttt_sample(..., advance = 1);]  crl_ttt_step(..., flags);  crl_ttt_valid(...);  crl_ttt_board(..., player = to_move, rel_mod)
 * with action == NULL meaning "draw it with the rollout's random agent at tcount[b] and advance tcount" (tcount may be NULL
 * when actions are given).  obs_board int8 [B][cells] (4-byte aligned), valid uint32 [B]. */
int crl_ttt_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                         uint32_t *occ, int8_t *winner, int8_t *to_move, const int8_t *action, uint32_t *tcount,
                         int8_t *reward, uint8_t *terminal, int8_t *winners, int8_t *obs_board, uint32_t *valid,
                         int rel_mod, uint32_t flags, void *stream);
/* The same calls on states in the REFERENCE's own layout (tictactoe_2p_env.py:165-169): board int8 [B][cells], -1 = empty,
 * else the owner's id; winner / to_move as above.  For callers that hold reference states (the single-state drop-in
 * classes, on crl_host_alloc memory: one launch per next_state) and never see the occupancy masks.
 * crl_ttt_step_board == crl_ttt_step on the equivalent masks, the board updated in place (the one cell the move filled;
 * all -1 after an auto-reset), plus -- each optional, NULL to skip -- valid uint32 [B] = empties mask of the NEW state
 * (valid_actions of the player to move next) and obs_board int8 [B][cells] = the new board with ids relative to the
 * player to move next (state_to_observation, _relative_player_id modulo rel_mod). */
int crl_ttt_step_board(const crl_ctx *ctx, int64_t B, int8_t *board, int8_t *winner, int8_t *to_move, const int8_t *action,
                       int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t *valid, int8_t *obs_board, int rel_mod,
                       uint32_t flags, void *stream);
/* The single-state form of crl_ttt_step_board, ONE BLOCKING call for a caller whose state lies in crl_host_alloc memory
 * (B = 1; as crl_tron_next_state_inplace64_host: every pointer is such memory with one address for host and GPU): the
 * state -- board, winner, mover, action -- travels BY VALUE in the kernel arguments (no PCIe read before the step) and the
 * kernel publishes completion itself (`seq` into *flag, on which the host spins; timeout_s as crl_stream_wait_mapped).
 * Results as crl_ttt_step_board, visible when the call returns. */
int crl_ttt_step_board_host(const crl_ctx *ctx, int8_t *board, int8_t *winner, int8_t *to_move, const int8_t *action,
                            int8_t *reward, uint8_t *terminal, int8_t *winners, uint32_t *valid, int8_t *obs_board, int rel_mod,
                            uint32_t flags, void *stream, uint32_t *flag, uint32_t seq, double timeout_s);
/* valid_actions (valid uint32 [B] empties mask; may be NULL) and / or state_to_observation (obs_board int8 [B][cells]
 * relative to player[b] modulo rel_mod, absolute ids when player == NULL; may be NULL) of reference-layout boards */
int crl_ttt_observe_board(const crl_ctx *ctx, int64_t B, const int8_t *board, const int8_t *player, int rel_mod,
                          int8_t *obs_board, uint32_t *valid, void *stream);
typedef struct {
    uint32_t *tcount, *tstep, *n_episodes;   /* tcount = the env's rollout step count (RNG counter) */
    uint32_t *win_count;   /* [P][B] */
    uint32_t *draw_count;  /* [B] */
    uint32_t *len_sum;     /* [B] */
    int32_t  *results;     /* [B][3+P] or NULL: packed row n_episodes, len_sum, draw_count, win_count[P] (as crl_tron_stats) */
} crl_ttt_stats;
/* random agent, one Philox call per 8 plies, one 32-bit word per 2: with c = tcount and n = the number of empty cells,
 *   w = Philox(ctr={g, c >> 3, 0, 0x54540000}, seed)[(c >> 1) & 3]
 *   draw = w when c is even, lo32(w * (n + 1)) when c is odd  (what the even ply's extraction hi32(w * (n + 1)) left over,
 *          when that ply was the one before in the same game: the two choices are w's leading digits in the mixed radix
 *          (n + 1, n); the definition does not look back, any ply follows from (seed, g, c, board) alone)
 *   r = mulhi32(draw, n); the ply marks the r-th empty cell in row-major order */
int crl_ttt_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                    uint32_t *occ, int8_t *winner, int8_t *to_move, crl_ttt_stats stats, void *stream);

/* ------------------------------------------------------------------ Blokus (20 x 20, 4 players)
 * State (reference BlokusEnvironment.py:283-289: (Board, round_count, [AI x 4])):
 *   occ     uint32 [B][4][20]  bit x of occ[b][c][y] set = cell (x, y) holds colour c+1 (Board.board_contents[y][x])
 *   inv     uint32 [B][4]      bit i set = piece i (inventory order of ai.py:12-22) still held by that player
 *   score   int32  [B][4]      AI.player_score
 *   round   int32  [B]         round_count
 *   to_move int32  [B]
 * Action id = ((piece*400 + y*20 + x)*8 + orientation)*5 + shift for the reference string
 * "{piece};({x}, {y});{orientation}{shift}" (BlokusEnvironment.py:55-80; orientation order board.py:47);
 * ascending ids are exactly the order of BlokusEnvironment.valid_actions; -1 = '' (pass).
 * The step entries also take EXTENDED ids, for indices valid_actions never emits but next_state accepts (it places whatever
 * string_to_action parsed, BlokusEnvironment.py:417-419, and numpy wraps negative indices, board.py:103):
 *   CRL_BLOKUS_EXT_BASE + ((piece*1600 + (y+20)*40 + (x+20))*8 + orientation)*5 + shift,   x, y in [-20, 20).       */
#define CRL_BLOKUS_ACTION_IDS 336000
#define CRL_BLOKUS_MASK_WORDS 10500
#define CRL_BLOKUS_EXT_BASE   336000
#define CRL_BLOKUS_EXT_IDS    1344000                 /* 21 * 1600 * 8 * 5 */
/* per-game codes crl_blokus_step / crl_blokus_step_observe put into reward[b] where the reference's next_state RAISES; the
 * game is left exactly as it was (next_state works on copies, BlokusEnvironment.py:408-409), terminal[b] = winners[b] = 0 */
#define CRL_BLOKUS_INDEX_ERROR -1   /* IndexError: a cell outside numpy's index range [-20, 20) (board.py:103), or a shift id
                                     * that names no cell of the piece (computation.py:218) */
#define CRL_BLOKUS_VALUE_ERROR -2   /* ValueError: piece not in the mover's inventory (ai.py:47 list.remove); the board update
                                     * runs first in the reference, so an IndexError takes precedence */
#define CRL_BLOKUS_BAD_ACTION  -3   /* the id is neither a dense nor an extended id (the reference's KeyError for an unknown
                                     * piece name is the closest relative) */
int crl_blokus_create(crl_ctx **out);
/* The random agent of crl_ttt_rollout as a stand-alone call: action[b] = the r-th empty cell (flat index, row-major
 * np.where order) with r as crl_ttt_rollout's RNG contract defines it for step counter tcount[b], -1 on a full board; advance != 0
 * also increments tcount.  T x (crl_ttt_sample; crl_ttt_step with CRL_STEP_AUTO_RESET) == crl_ttt_rollout(T). */
int crl_ttt_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, const uint32_t *occ,
                   uint32_t *tcount, int advance, int8_t *action, void *stream);

/* DIAGNOSTIC: per-phase shader-cycle sums of crl_blokus_rollout (prep, count, rng, select, apply, exists, rest, -).
 * Returns 1 and the sums only in a library built with -DBLK_STAMPS, else 0 and zeros (the shipped build). */
int crl_blokus_stamps(uint64_t *out8, int reset);
/* HOST helper: cells (dx,dy pairs, 2*n int8) of (piece, orientation, shift) relative to the anchor; returns n.
 * Restates computation.py:184-246 (rotate_default_piece / shift_offsets). */
int crl_blokus_placement(int piece, int orient, int shift, int8_t *cells_xy);
/* replaces BlokusEnvironment.new_state (BlokusEnvironment.py:248-289); mask as for crl_tron_reset */
int crl_blokus_reset(const crl_ctx *ctx, int64_t B, const uint8_t *mask, uint32_t *occ, uint32_t *inv, int32_t *score,
                     int32_t *round, int32_t *to_move, void *stream);
/* replaces BlokusEnvironment.next_state (BlokusEnvironment.py:357-451).  No legality check (the reference
 * leaves that to its caller, match_server.py:193): as there, a placement on occupied cells OVERWRITES them (other colours
 * included) and still scores, cells at x or y in -20..-1 WRAP to the far edge (numpy), and where the reference raises the
 * game stays untouched and reward[b] carries CRL_BLOKUS_INDEX_ERROR / _VALUE_ERROR / _BAD_ACTION (see above).
 * action int32 [B]: dense or extended id, < 0 = '' (pass).  reward int8 [B]: the mover's rank in the ascending score
 * order at terminal, else 0; terminal uint8 [B]: no player has a move on the PRE-move board with the
 * post-move inventories; winners uint8 [B]: bitmask of players whose score equals max(0, best), 0 unless terminal */
int crl_blokus_step(const crl_ctx *ctx, int64_t B, uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                    const int32_t *action, int8_t *reward, uint8_t *terminal, uint8_t *winners, uint32_t flags, void *stream);
/* replaces BlokusEnvironment.valid_actions / board.get_all_valid_moves (BlokusEnvironment.py:453-500, board.py:170-193)
 * for `player` (int8 [B]; NULL = the player to move): count int32 [B] (may be NULL) and/or the dense id bitmap
 * mask uint32 [B][CRL_BLOKUS_MASK_WORDS] (may be NULL; 42 KB per game, meant for small B). */
int crl_blokus_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                     const int32_t *round, const int32_t *to_move, const int8_t *player, int32_t *count, uint32_t *mask, void *stream);
/* replaces BlokusEnvironment.state_to_observation (BlokusEnvironment.py:721-768) for observer player[b]:
 * obs_board int8 [B][20][20], 4-byte aligned (-1 empty, else (owner - observer) % 4, rotated by np.rot90(k=-observer)),
 * obs_pieces uint8 [B][4][21] (row r = player (r + observer) % 4), obs_score int32 [B][4] (rolled by -observer) */
int crl_blokus_observe(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                       const int8_t *player, int8_t *obs_board, uint8_t *obs_pieces, int32_t *obs_score, void *stream);
/* The ordered legal-action LIST, compacted: what BlokusEnvironment.valid_actions returns (BlokusEnvironment.py:453-500,
 * board.py:170-193) in a form a policy can consume for a whole batch.  count int32 [B] = number of legal actions of `player`
 * (int8 [B]; NULL = the player to move); ids int32 [B][cap]: ids[b][0 .. min(count[b], cap)) = their dense ids in
 * ascending order = the reference's order (piece -> anchor row-major -> orientation -> shift); entries beyond are left
 * untouched.  Either output may be NULL; cap in 1..2^28 when ids is given.  (Observed maximum in reference-played games:
 * 1,693 actions; cap = 2048 is safe for play from the empty board; hand-made boards reach 12,952; count tells when a list
 * was cut -- the kernel counts the whole list whatever cap is.) */
int crl_blokus_valid_list(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                          const int32_t *round, const int32_t *to_move, const int8_t *player, int32_t *ids, int32_t *count,
                          int cap, void *stream);
/* action[b] = dense id of the rank[b]-th (0-based) entry of that list without materialising it, -1 when rank[b] is outside
 * [0, count): "play the r-th legal action" for caller-chosen ranks (crl_blokus_sample draws the rank itself).
 * count (may be NULL) receives the list length. */
int crl_blokus_select(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                      const int32_t *round, const int32_t *to_move, const int8_t *player, const int32_t *rank,
                      int32_t *action, int32_t *count, void *stream);
/* replaces BlokusEnvironment.is_valid_action (BlokusEnvironment.py:667-719, called per move from match_server.py:193):
 * ok uint8 [B] = 1 iff action[b] (dense id) is in valid_actions(player) -- tested directly (piece held, the shift names a
 * cell of the piece, that cell's target is an anchor, every cell lands on an allowed cell), no enumeration */
int crl_blokus_is_valid(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const uint32_t *inv, const int32_t *score,
                        const int32_t *round, const int32_t *to_move, const int8_t *player, const int32_t *action,
                        uint8_t *ok, void *stream);
/* One step of Board.check_orientation_shifts (board.py:156-168 -> computation.py:144-180 check_shifted): ok uint8 [B] = 1 iff
 * every cell of the placement action[b] names lies on the board, is empty and has no orthogonal neighbour of colour player[b] + 1
 * -- whether or not the index is an anchor and whether or not the piece is held (so: crl_blokus_is_valid minus those two
 * conditions; it needs the row bitboards only). */
int crl_blokus_fits(const crl_ctx *ctx, int64_t B, const uint32_t *occ, const int8_t *player, const int32_t *action, uint8_t *ok,
                    void *stream);
/* occ rows from Board.board_contents as int8 [B][20][20] (0 empty, else colour): the inverse of crl_blokus_board, for
 * callers that hold reference-layout boards */
int crl_blokus_pack(const crl_ctx *ctx, int64_t B, const int8_t *board, uint32_t *occ, void *stream);
/* Board.board_contents as int8 [B][20][20] (0 empty, else colour) */
int crl_blokus_board(const crl_ctx *ctx, int64_t B, const uint32_t *occ, int8_t *board, void *stream);
/* One launch for a ply of every game, as a learner / vector env needs it: [sample ->] next_state -> number of legal actions and
 * state_to_observation of the player to move NEXT.  Exactly equivalent to
 *   [crl_blokus_sample(..., advance = 1);]  crl_blokus_step(..., flags);  crl_blokus_valid(..., player = NULL, count, NULL);
 *   crl_blokus_observe(..., player = to_move, ...)
 * with action == NULL meaning "play the rollout's random agent at tcount[b] and advance tcount" (tcount may be NULL when
 * actions are given).  n_valid int32 [B]; obs_* as crl_blokus_observe (obs_board 4-byte aligned); obs_player int8 [B] = the
 * observer = the player to move. */
int crl_blokus_step_observe(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id,
                            uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                            const int32_t *action, uint32_t *tcount, int8_t *reward, uint8_t *terminal, uint8_t *winners,
                            int32_t *n_valid, int8_t *obs_board, uint8_t *obs_pieces, int32_t *obs_score, int8_t *obs_player,
                            uint32_t flags, void *stream);
typedef struct {
    uint32_t *tcount, *tstep, *n_episodes;
    uint32_t *win_count;   /* [4][B] */
    uint32_t *len_sum;     /* [B] */
    int32_t  *score_sum;   /* [4][B] final scores summed over finished episodes */
    int32_t  *results;     /* [B][10] or NULL: packed row n_episodes, len_sum, win_count[4], score_sum[4] (as crl_tron_stats) */
} crl_blokus_stats;
/* random agent: the mover plays the r-th action of valid_actions() (reference order), r = mulhi32(w, n), '' if n = 0;
 * w = Philox(ctr={g, c >> 2, 0, 0x424c0000}, seed)[c & 3] with c = tcount; auto-reset on terminal */
int crl_blokus_rollout(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, int T,
                       uint32_t *occ, uint32_t *inv, int32_t *score, int32_t *round, int32_t *to_move,
                       crl_blokus_stats stats, void *stream);

/* The random agent of crl_blokus_rollout as a stand-alone call: action[b] = dense id of the r-th legal action of the
 * player to move, in reference order, r = mulhi32(Philox(...)[tcount & 3], number of legal actions); -1 (pass, '')
 * when there is none; advance != 0 also increments tcount.
 * T x (crl_blokus_sample; crl_blokus_step with CRL_STEP_AUTO_RESET) == crl_blokus_rollout(T). */
int crl_blokus_sample(const crl_ctx *ctx, int64_t B, uint64_t seed, uint64_t first_env_id, const uint32_t *occ,
                      const uint32_t *inventory, const int32_t *score, const int32_t *round, const int32_t *to_move,
                      uint32_t *tcount, int advance, int32_t *action, void *stream);

#ifdef __cplusplus
}
#endif
#endif
