// crl_common.hpp -- shared pieces of libcolosseum_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/colosseum_hip.h"

#define CRL_WAVE 64

enum crl_game { CRL_GAME_TRON = 1, CRL_GAME_TTT = 2, CRL_GAME_BLOKUS = 3 };

struct crl_tron_cfg {
    int32_t N, P;
    int16_t start_heads[CRL_TRON_MAX_P];
    int8_t start_dirs[CRL_TRON_MAX_P];
};

struct crl_ttt_cfg {
    int32_t D0, D1, D2, K, P, n_cells, n_lines;
    uint32_t full;
};

// win-line directions of a TicTacToe board as kernel arguments (ttt.hip)
struct ttt_dirs {
    int32_t n_dirs, K, P, n_cells;
    uint32_t full;
    int32_t stride[13];
    uint32_t start[13];   // bit c set: the K-window starting at cell c along this direction is on the board
};

struct crl_ctx {
    int game;
    crl_tron_cfg tron;
    crl_ttt_cfg ttt;
    ttt_dirs ttt_dd;
    uint32_t ttt_lines_host[CRL_TTT_MAX_LINES];
    uint32_t *ttt_lines_dev;   // DEVICE copy of the win-line table
    uint32_t *ttt_win_dev;     // boards of <= 16 cells: DEVICE bit table, bit m = "the mask m holds a K-line" (8 KB), or NULL
    int ttt_win_device;        // the device that table lives on
    void *blokus;              // blokus tables (blokus.hip)
};

void crl_set_error(const char *fmt, ...);

#define CRL_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) { crl_set_error(__VA_ARGS__); return CRL_EINVAL; } \
    } while (0)

#define CRL_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            crl_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CRL_EHIP;                                                            \
        }                                                                               \
    } while (0)

#define CRL_LAUNCH_CHECK() CRL_HIP(hipGetLastError())

// ---------------------------------------------------------------- bounds asserts (diagnostic build: -DCRL_BOUNDS)
// GPU AddressSanitizer is not available on this pool, so the data-dependent LDS / table accesses of the kernels carry
// explicit range checks in a build of their own (tools/gpu_bounds.sh): a failing check does not fault, it counts itself
// and keeps the first offender -- {failures, code, value, limit} per translation unit, read by crl_diag_bounds().  The
// shipped build compiles none of it (the macros expand to nothing).  Codes: 1xx tron.hip, 2xx ttt.hip, 3xx blokus.hip.
#ifdef CRL_BOUNDS
namespace {
__device__ unsigned int g_crl_bounds[4];
__device__ __forceinline__ void crl_bounds_fail(const unsigned code, const unsigned v, const unsigned lim)
{
    if (atomicAdd(&g_crl_bounds[0], 1u) == 0u) { g_crl_bounds[1] = code; g_crl_bounds[2] = v; g_crl_bounds[3] = lim; }
}
} // namespace
#define CRL_BOUNDS_LT(v, lim, code) do { if (!((unsigned)(v) < (unsigned)(lim))) crl_bounds_fail((code), (unsigned)(v), (unsigned)(lim)); } while (0)
#define CRL_BOUNDS_IN(a, lo, hi, code) CRL_BOUNDS_LT((unsigned)(a) - (unsigned)(lo), (unsigned)(hi) - (unsigned)(lo), (code))
#define CRL_BOUNDS_READBACK(out4)                                                                       \
    do {                                                                                                \
        CRL_HIP(hipDeviceSynchronize());                                                                \
        CRL_HIP(hipMemcpyFromSymbol((out4), HIP_SYMBOL(g_crl_bounds), 4 * sizeof(unsigned int)));       \
    } while (0)
#else
#define CRL_BOUNDS_LT(v, lim, code) do { } while (0)
#define CRL_BOUNDS_IN(a, lo, hi, code) do { } while (0)
#define CRL_BOUNDS_READBACK(out4) do { (out4)[0] = (out4)[1] = (out4)[2] = (out4)[3] = 0u; } while (0)
#endif
// {failures, first code, first value, first limit} of one translation unit (zeros in the shipped build)
int crl_tron_bounds(unsigned int *out4);
int crl_ttt_bounds(unsigned int *out4);
int crl_blokus_bounds(unsigned int *out4);

// host side of a completion flag in mapped memory (capi.hip): spin until *flag_host == seq, fall back to hipStreamSynchronize
// after timeout_s; `who` names the caller in error messages
int crl_spin_mapped(void *stream, const volatile uint32_t *flag_host, uint32_t seq, double timeout_s, const char *who);

// ---------------------------------------------------------------- Philox-4x32-10
// Salmon et al., SC'11 (Random123 constants). 10 rounds of two 32 x 32 -> 64-bit products.
// WIDE: each product is ONE v_mad_u64_u32 instead of the v_mul_lo_u32 + v_mul_hi_u32 pair the compiler picks (it splits
// a 64-bit product of zero-extended operands into that very pair, hence inline asm).  Measured on gfx950 a v_mad_u64_u32
// costs what ONE of the pair does (4.4 cycles per wave and SIMD with >= 2 waves, against 2.2 for plain integer VALU:
// profiles/r2_valu_issue_calibration.json) but has the longer latency, and a Philox round waits for its products: in an
// A/B on one box (tools/README.md) WIDE gained 3 % on the TicTacToe rollouts and 0.5 % on the
// lane-per-player Tron kernel (>= 4 waves per SIMD cover the latency) and LOST 1-4 % on the lone-wave Tron kernels and
// 5 % on Blokus.  So it is a per-kernel choice.
struct philox_out { uint32_t w[4]; };

template <bool WIDE>
__device__ __forceinline__ void crl_mul_wide(const uint32_t a, const uint32_t b, uint32_t &lo, uint32_t &hi)
{
    if constexpr (WIDE) {
        uint64_t p;
        asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "s"(a), "v"(b) : "vcc");   // a: a constant, in an SGPR
        lo = (uint32_t)p;
        hi = (uint32_t)(p >> 32);
    } else {
        lo = a * b;
        hi = __umulhi(a, b);
    }
}

template <bool WIDE = false>
__device__ __forceinline__ philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                    uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t lo0, hi0, lo1, hi1;
        crl_mul_wide<WIDE>(0xD2511F53u, c0, lo0, hi0);
        crl_mul_wide<WIDE>(0xCD9E8D57u, c2, lo1, hi1);
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    philox_out o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

#define CRL_TAG_TRON   0x54520000u
#define CRL_TAG_TTT    0x54540000u
#define CRL_TAG_BLOKUS 0x424c0000u
