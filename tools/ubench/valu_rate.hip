// Calibration micro-benchmark for the `valu_issue` roofline of bench.py (gfx950 / MI355X).
//
// Measures, at 1 / 2 / 4 / 8 waves per SIMD on every CU, how many wave64 instructions per second the chip issues for
//   mix "valu"  : independent 32-bit integer VALU instructions only (v_add_u32 / v_xor_b32 on 8 chains);
//   mix "tron"  : the instruction mix of the Tron rollout loop per PMC (profiles/r1_tron_n20: 175 VALU : 30 SALU : 16 LDS
//                 per wave-step): per block 11 VALU (2 of them v_cmp -> SGPR pair, 2 v_cndmask reading one), 2 SALU
//                 (s_and_b64 / s_or_b64 on lane masks) and 1 ds_read_u8, the LDS result consumed at the block's end.
//   mix "ttt"   : the TicTacToe rollout ply per PMC (profiles/r4_ttt_5x5: 58 VALU : 6 SALU : 1 LDS per wave-ply; 7.8 % of the
//                 5x5 rollout kernel's VALU instructions are full-width 32-bit multiplies -- Philox rounds, the mulhi draws --,
//                 which issue at half the rate): per block 53 plain VALU + 5 multiplies + 6 SALU + 1 ds_read_b32;
//   mix "blokus": the Blokus rollout ply per PMC (profiles/r4_blokus: 438 VALU : 261 SALU : 99 LDS per wave-ply): per block
//                 22 VALU (2 v_cmp -> SGPR pair, 2 v_cndmask, 2 v_bcnt among them), 13 SALU (lane-mask logic, s_bcnt1,
//                 address arithmetic) and 5 ds_read_b64 consumed at the block's end.
// bench.py prices a kernel's PMC VALU rate against the rate of ITS mix at ITS occupancy (`valu_frac_of_mix`) next to the
// pure-integer peak (`valu_frac`): with the scalar unit, the LDS and quarter-rate multiplies in the stream, the pure peak
// is not reachable by that instruction stream whatever the schedule.
// The shader clock is measured in the kernel (s_memtime ticks per s_memrealtime tick x 100 MHz), so the result is
// quoted both as wave-instructions per second (what bench.py divides by) and as cycles per instruction per SIMD.
// Everything is inline asm so the compiler can neither fuse nor drop instructions.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Prints one JSON object (committed as profiles/r<round>_issue_calibration.json; bench.py reads the newest).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kValuPerIter = 64;      // mix "valu": 64 VALU per loop iteration
constexpr int kBlocksPerIter = 8;     // mix "tron": 8 blocks of (11 VALU + 2 SALU + 1 LDS) per loop iteration

// `half` != 0: only lanes 0..31 of every wave execute the loop (EXEC = 0x00000000ffffffff) -- does a half-empty wave64
// issue faster on the SIMD-32 (i.e. is the second 32-lane pass skipped)?
__global__ void __launch_bounds__(256) valu_kernel(uint32_t *out, uint64_t *clk, int iters, int half)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t c = blockIdx.x | 1u;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (half && (threadIdx.x & 32)) iters = 0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept 4\n"
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
            "v_xor_b32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
            "v_xor_b32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
            ".endr\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// 32x32 -> 64-bit products (Philox rounds): a v_mul_lo_u32 + v_mul_hi_u32 pair against one v_mad_u64_u32.
// `wide` = 0: 32 pairs per iteration (64 instructions); 1: 32 v_mad_u64_u32 per iteration.
__global__ void __launch_bounds__(256) mul_kernel(uint32_t *out, uint64_t *clk, int iters, int wide)
{
    uint32_t a0 = threadIdx.x | 1u, a1 = a0 + 2, a2 = a0 + 4, a3 = a0 + 6;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
    const uint32_t c = 0xD2511F53u;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (wide) {
        uint64_t p0 = a0, p1 = a1, p2 = a2, p3 = a3;
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                ".rept 8\n"
                "v_mad_u64_u32 %0, vcc, %4, %8, 0\n v_mad_u64_u32 %1, vcc, %5, %8, 0\n"
                "v_mad_u64_u32 %2, vcc, %6, %8, 0\n v_mad_u64_u32 %3, vcc, %7, %8, 0\n"
                ".endr\n"
                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(c) : "vcc");
        }
        a0 = (uint32_t)p0; a1 = (uint32_t)p1; a2 = (uint32_t)p2; a3 = (uint32_t)p3;
        h0 = (uint32_t)(p0 >> 32); h1 = (uint32_t)(p1 >> 32); h2 = (uint32_t)(p2 >> 32); h3 = (uint32_t)(p3 >> 32);
    } else {
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                ".rept 8\n"
                "v_mul_hi_u32 %4, %0, %8\n v_mul_lo_u32 %0, %0, %8\n v_mul_hi_u32 %5, %1, %8\n v_mul_lo_u32 %1, %1, %8\n"
                "v_mul_hi_u32 %6, %2, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_hi_u32 %7, %3, %8\n v_mul_lo_u32 %3, %3, %8\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(c));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ h0 ^ h1 ^ h2 ^ h3;
}

__global__ void __launch_bounds__(256) tron_mix_kernel(uint32_t *out, uint64_t *clk, int iters)
{
    __shared__ uint32_t cells[256 * 5];              // odd dword stride per lane, like the rollout's slabs
    for (int j = 0; j < 5; ++j) cells[threadIdx.x * 5 + j] = threadIdx.x + j;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = 0, a7 = 0;
    const uint32_t c = blockIdx.x | 1u;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)cells + threadIdx.x * 20;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept 8\n"
            "ds_read_u8 %6, %9\n"                                            // 1 LDS (byte probe)
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n"      // 7 plain VALU
            "v_bfe_i32 %3, %3, 3, 8\n v_xor_b32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_and_b32 %3, 0xff, %3\n"
            "v_cmp_eq_u32 s[20:21], %0, %1\n"                                // 2 compares into SGPR pairs
            "v_cmp_lt_u32 s[22:23], %2, %4\n"
            "s_and_b64 s[20:21], s[20:21], s[22:23]\n"                       // 2 SALU on lane masks
            "s_or_b64 s[22:23], s[22:23], exec\n"
            "v_cndmask_b32 %5, %5, %0, s[20:21]\n"                           // 2 selects reading them
            "s_waitcnt lgkmcnt(0)\n"
            "v_cndmask_b32 %7, %7, %6, s[22:23]\n"                           // consumes the LDS byte
            ".endr\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(c), "v"(addr)
            : "s20", "s21", "s22", "s23", "scc", "memory");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// TicTacToe ply mix: 53 plain VALU on 8 independent chains (2 of them consume the LDS word), 5 full-width multiplies
// (2 v_mul_hi_u32 + v_mul_lo_u32 pairs and one v_mul_lo_u32: the static share of the 5x5 rollout kernel's code, 7.8 % of its
// VALU instructions), 6 SALU, 1 LDS dword read -- see the header.
__global__ void __launch_bounds__(256) ttt_mix_kernel(uint32_t *out, uint64_t *clk, int iters)
{
    __shared__ uint32_t table[256 * 3];
    for (int j = 0; j < 3; ++j) table[threadIdx.x * 3 + j] = threadIdx.x + j;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 | 1u, a7 = a0 + 9, l0 = 0;
    uint32_t m0 = a0 | 1u, m1 = a0 + 11, m2 = a0 + 13, h0 = 0, h1 = 0;
    const uint32_t c = blockIdx.x | 1u;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)table + threadIdx.x * 12;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept 4\n"
            "ds_read_b32 %8, %18\n"                                                       // 1 LDS (table lookup)
            ".rept 3\n"                                                                   // 48 plain VALU, 8 independent chains
            "v_add_u32 %0, %0, %17\n v_xor_b32 %1, %1, %17\n v_add_u32 %2, %2, %17\n v_xor_b32 %3, %3, %17\n"
            "v_add_u32 %4, %4, %17\n v_xor_b32 %5, %5, %17\n v_add_u32 %6, %6, %17\n v_xor_b32 %7, %7, %17\n"
            "v_lshlrev_b32 %0, 1, %0\n v_and_b32 %1, %1, %17\n v_lshrrev_b32 %2, 1, %2\n v_or_b32 %3, %3, %17\n"
            "v_xor_b32 %4, %4, %17\n v_add_u32 %5, %5, %17\n v_xor_b32 %6, %6, %17\n v_add_u32 %7, %7, %17\n"
            ".endr\n"
            "v_mul_hi_u32 %12, %9, %17\n v_mul_lo_u32 %9, %9, %17\n"                       // 5 full-width multiplies ...
            "s_add_u32 %14, %14, %19\n s_xor_b32 %15, %15, %19\n s_lshl_b32 %16, %16, 1\n" // ... 6 SALU between them
            "v_mul_hi_u32 %13, %10, %17\n v_mul_lo_u32 %10, %10, %17\n"
            "s_add_u32 %15, %15, %19\n s_xor_b32 %14, %14, %19\n s_or_b32 %16, %16, %19\n"
            "v_mul_lo_u32 %11, %11, %17\n"
            "v_add_u32 %0, %0, %12\n v_xor_b32 %1, %1, %13\n v_add_u32 %2, %2, %17\n"      // 3 more plain VALU
            "s_waitcnt lgkmcnt(0)\n"
            "v_xor_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %8, %4\n"                           // 2 VALU consuming the table word
            ".endr\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(l0), "+v"(m0), "+v"(m1), "+v"(m2),
              "+v"(h0), "+v"(h1), "+s"(s0), "+s"(s1), "+s"(s2)
            : "v"(c), "v"(addr), "s"(c) : "scc", "memory");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ l0 ^ m0 ^ m1 ^ m2 ^ h0 ^ h1 ^ s0 ^ s1 ^ s2;
}

// Blokus ply mix: 22 VALU (2 v_cmp -> SGPR pair, 2 v_cndmask, 2 v_bcnt), 13 SALU, 5 ds_read_b64 -- see the header.
__global__ void __launch_bounds__(256) blokus_mix_kernel(uint32_t *out, uint64_t *clk, int iters)
{
    __shared__ uint64_t rows[256 * 5 + 8];
    for (int j = 0; j < 5; ++j) rows[threadIdx.x * 5 + j] = threadIdx.x * 0x9E3779B97F4A7C15ull + j;
    __syncthreads();
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5;
    const uint32_t c = blockIdx.x | 1u;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint64_t *)rows + threadIdx.x * 40;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            ".rept 4\n"
            "ds_read_b64 v[40:41], %12\n ds_read_b64 v[42:43], %12 offset:8\n ds_read_b64 v[44:45], %12 offset:16\n"   // 5 LDS
            "ds_read_b64 v[46:47], %12 offset:24\n ds_read_b64 v[48:49], %12 offset:32\n"   // (fixed registers: the low halves are consumed below)
            "v_add_u32 %0, %0, %10\n v_xor_b32 %1, %1, %10\n s_add_u32 %6, %6, %11\n"
            "v_lshlrev_b32 %2, 1, %2\n v_and_b32 %3, %3, %0\n s_xor_b32 %7, %7, %11\n"
            "v_or_b32 %4, %4, %1\n v_add_u32 %5, %5, %10\n s_lshl_b32 %8, %8, 1\n"
            "v_cmp_eq_u32 s[20:21], %0, %1\n s_bcnt1_i32_b64 %9, s[20:21]\n"
            "v_cmp_lt_u32 s[22:23], %2, %4\n s_and_b64 s[20:21], s[20:21], s[22:23]\n"
            "v_xor_b32 %2, %2, %10\n v_lshrrev_b32 %3, 1, %3\n s_or_b64 s[22:23], s[22:23], exec\n"
            "v_cndmask_b32 %5, %5, %0, s[20:21]\n s_add_u32 %6, %6, %9\n"
            "v_cndmask_b32 %4, %4, %1, s[22:23]\n s_xor_b32 %7, %7, %11\n"
            "v_add_u32 %0, %0, %2\n v_xor_b32 %1, %1, %3\n s_add_u32 %8, %8, %11\n"
            "v_and_b32 %2, %2, %4\n v_or_b32 %3, %3, %5\n s_lshr_b32 %9, %9, 1\n"
            "s_waitcnt lgkmcnt(0)\n"
            "v_and_b32 %0, %0, v40\n v_bcnt_u32_b32 %1, v42, %1\n s_or_b32 %6, %6, %11\n"       // consume the five rows
            "v_xor_b32 %2, %2, v44\n v_bcnt_u32_b32 %3, v46, %3\n s_xor_b32 %8, %8, %7\n"
            "v_or_b32 %4, %4, v48\n s_and_b32 %7, %7, %6\n v_add_u32 %5, %5, %10\n"
            ".endr\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
            : "v"(c), "s"(c), "v"(addr)
            : "s20", "s21", "s22", "s23", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49");
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ s0 ^ s1 ^ s2 ^ s3;
}

// Scalar issue rate (round 3: the Blokus rollout issues 408 SALU next to 529 VALU instructions per wave-step -- is the
// scalar unit a bound?).  `with_valu` = 0: 64 independent s_add_u32 / s_xor_b32 per iteration on 8 chains;
// 1: the same 64 SALU interleaved one to one with 64 independent VALU instructions (do the two pipes issue side by side?).
__global__ void __launch_bounds__(256) salu_kernel(uint32_t *out, uint64_t *clk, int iters, int with_valu)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    const uint32_t c = blockIdx.x | 1u;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    uint64_t t0 = 0, r0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (with_valu) {
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                ".rept 8\n"
                "s_add_u32 %4, %4, %12\n v_add_u32 %0, %0, %13\n s_xor_b32 %5, %5, %12\n v_xor_b32 %1, %1, %13\n"
                "s_add_u32 %6, %6, %12\n v_add_u32 %2, %2, %13\n s_xor_b32 %7, %7, %12\n v_xor_b32 %3, %3, %13\n"
                "s_add_u32 %8, %8, %12\n v_xor_b32 %0, %0, %13\n s_xor_b32 %9, %9, %12\n v_add_u32 %1, %1, %13\n"
                "s_add_u32 %10, %10, %12\n v_xor_b32 %2, %2, %13\n s_xor_b32 %11, %11, %12\n v_add_u32 %3, %3, %13\n"
                ".endr\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7)
                : "s"(c), "v"(c) : "scc");
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            asm volatile(
                ".rept 8\n"
                "s_add_u32 %0, %0, %8\n s_xor_b32 %1, %1, %8\n s_add_u32 %2, %2, %8\n s_xor_b32 %3, %3, %8\n"
                "s_add_u32 %4, %4, %8\n s_xor_b32 %5, %5, %8\n s_add_u32 %6, %6, %8\n s_xor_b32 %7, %7, %8\n"
                ".endr\n"
                : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "s"(c) : "scc");
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
}

int main()
{
    uint32_t *out;
    uint64_t *clk;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t)));
    CHECK(hipMalloc(&clk, 2 * sizeof(uint64_t)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = 40000;
    printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"note\": \"wave64 instructions per second, whole chip; "
           "cycles = shader cycles per instruction per SIMD at the in-kernel clock\", \"mixes\": {", prop.gcnArchName, cus, cus * 4);
    for (int mix = 0; mix < 9; ++mix) {
        static const char *const names[9] = {"valu", "tron", "valu_half_exec", "mul_lo_hi_pairs", "mad_u64_u32", "salu", "salu_valu_1to1", "ttt", "blokus"};
        printf("%s\"%s\": [", mix ? ", " : "", names[mix]);
        for (int wps = 1; wps <= 8; wps *= 2) {
            const int blocks = cus * wps;                          // wps blocks of 4 waves per CU = wps waves per SIMD
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {                    // rep 0 warms up (clock ramp), rep 1 is timed
                CHECK(hipEventRecord(e0));
                if (mix == 7) hipLaunchKernelGGL(ttt_mix_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters / 4);
                else if (mix == 8) hipLaunchKernelGGL(blokus_mix_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters / 2);
                else if (mix >= 5) hipLaunchKernelGGL(salu_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters, mix == 6);
                else if (mix >= 3) hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters / 4, mix == 4);
                else if (mix != 1) hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters, mix == 2);
                else hipLaunchKernelGGL(tron_mix_kernel, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
            }
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2];
            CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
            const double ghz = (double)h[0] / (double)h[1] * 0.1;
            // (mul mixes: iters / 4 iterations of 64 multiplies, or of 32 v_mad_u64_u32 = the same 32 full products)
            // (salu mixes: 64 SALU per iteration, plus 64 VALU in the 1:1 mix; "valu" columns then count the VALU part only)
            // (ttt: iters / 4 iterations of 4 blocks of 58 VALU + 6 SALU + 1 LDS; blokus: iters / 2 iterations of 4 blocks of 22 + 13 + 5)
            const double valu_per_wave = mix == 7 ? (double)(iters / 4) * 4 * 58 : mix == 8 ? (double)(iters / 2) * 4 * 22 : mix == 5 ? 0.0 : mix == 6 ? (double)iters * 64 : mix >= 3 ? (double)(iters / 4) * (mix == 4 ? 32 : 64)
                                                  : (double)iters * (mix == 1 ? kBlocksPerIter * 11 : kValuPerIter);
            const double all_per_wave = mix == 7 ? (double)(iters / 4) * 4 * 65 : mix == 8 ? (double)(iters / 2) * 4 * 40 : mix == 5 ? (double)iters * 64 : mix == 6 ? (double)iters * 128 : mix >= 3 ? valu_per_wave
                                                  : (double)iters * (mix == 1 ? kBlocksPerIter * 14 : kValuPerIter);
            const double waves = (double)blocks * 4;
            const double s = ms * 1e-3;
            printf("%s{\"waves_per_simd\": %d, \"ms\": %.3f, \"clock_ghz\": %.3f, \"valu_wave_insts_per_s\": %.4e, "
                   "\"all_wave_insts_per_s\": %.4e, \"cycles_per_valu_per_simd\": %.3f, \"cycles_per_inst_per_simd\": %.3f}",
                   wps > 1 ? ", " : "", wps, ms, ghz, valu_per_wave * waves / s, all_per_wave * waves / s,
                   valu_per_wave > 0 ? s * ghz * 1e9 / (valu_per_wave * wps) : 0.0, s * ghz * 1e9 / (all_per_wave * wps));
        }
        printf("]");
    }
    printf("}}\n");
    return 0;
}
