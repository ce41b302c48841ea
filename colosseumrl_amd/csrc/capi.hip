// capi.hip -- context lifetime, error reporting and the RNG test entry point of libcolosseum_hip.so.
#include "crl_common.hpp"
#include <stdarg.h>
#include <atomic>
#include <chrono>

static thread_local char g_err[512] = "";

void crl_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void crl_blokus_free(void *tables);   // blokus.hip

namespace {
__global__ void __launch_bounds__(256)
philox_kernel(const uint32_t *__restrict__ ctr, const uint32_t k0, const uint32_t k1, uint32_t *__restrict__ out, const int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 c = reinterpret_cast<const uint4 *>(ctr)[i];
    const philox_out r = philox4x32_10(c.x, c.y, c.z, c.w, k0, k1);
    reinterpret_cast<uint4 *>(out)[i] = make_uint4(r.w[0], r.w[1], r.w[2], r.w[3]);
}
// crl_diag_issue_probe: 64 independent integer VALU instructions per loop trip on eight chains (the "valu" mix of
// tools/ubench/valu_rate.hip); wave 0 of block 0 stamps the shader clock counter and the 100 MHz wall clock around its loop
__global__ void __launch_bounds__(256) issue_probe_kernel(uint32_t *out, uint64_t *clk, const int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t c = blockIdx.x | 1u;
    uint64_t t0 = 0, r0 = 0;
    const bool stamp = blockIdx.x == 0 && threadIdx.x == 0;
    if (stamp) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
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
    if (stamp) {
        clk[0] = __builtin_amdgcn_s_memtime() - t0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

#ifdef CRL_BOUNDS
// self-test of the bounds asserts (this translation unit's own counters): one check that must fail
__global__ void bounds_selftest_kernel()
{
    CRL_BOUNDS_LT(5, 3, 999);
    CRL_BOUNDS_IN(7, 4, 8, 998);                                // ... and one that must hold
}
#endif

// publishes `seq` in host-mapped memory behind everything the stream ran before it (crl_stream_wait_mapped)
__global__ void __launch_bounds__(64) signal_kernel(uint32_t *flag, const uint32_t seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
} // namespace

// spin on a word of mapped memory until it reads `seq`: 3.5 us less than hipStreamSynchronize for a short launch chain
// (tools/ubench/mailbox_rtt.hip).  The clock is read every 256 polls only; a stream that faulted never delivers -> after
// `timeout_s` the runtime's own wait reports it.
int crl_spin_mapped(void *stream, const volatile uint32_t *flag_host, uint32_t seq, double timeout_s, const char *who)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 1;; ++spins) {
        if (*flag_host == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            return CRL_OK;
        }
        if ((spins & 255u) == 0u &&
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            break;
    }
    CRL_HIP(hipStreamSynchronize((hipStream_t)stream));
    CRL_REQUIRE(*flag_host == seq, "%s: the stream drained but the flag reads %u, not %u", who, (unsigned)*flag_host, (unsigned)seq);
    return CRL_OK;
}

extern "C" {

const char *crl_last_error(void) { return g_err; }

int crl_version(void) { return CRL_ABI_VERSION; }

int crl_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        crl_set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return CRL_ENODEV;
    }
    return n;
}

void crl_destroy(crl_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->ttt_lines_dev) (void)hipFree(ctx->ttt_lines_dev);
    if (ctx->ttt_win_dev) (void)hipFree(ctx->ttt_win_dev);
    if (ctx->blokus) crl_blokus_free(ctx->blokus);
    delete ctx;
}

// ---- host staging for single-state callers (the reference's one-state-per-call API, see colosseum_hip.h)
int crl_host_alloc(size_t bytes, void **host, void **device)
{
    CRL_REQUIRE(host != nullptr && bytes > 0, "crl_host_alloc: bad argument");
    void *p = nullptr;
    CRL_HIP(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(p, 0, bytes);
    void *d = nullptr;
    hipError_t e = hipHostGetDevicePointer(&d, p, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(p);
        crl_set_error("hipHostGetDevicePointer failed: %s", hipGetErrorString(e));
        return CRL_EHIP;
    }
    *host = p;
    if (device) *device = d;
    return CRL_OK;
}

int crl_host_free(void *host)
{
    if (host) CRL_HIP(hipHostFree(host));
    return CRL_OK;
}

int crl_stream_create(void **stream)
{
    CRL_REQUIRE(stream != nullptr, "crl_stream_create: stream is NULL");
    hipStream_t s = nullptr;
    CRL_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return CRL_OK;
}

int crl_stream_destroy(void *stream)
{
    if (stream) CRL_HIP(hipStreamDestroy((hipStream_t)stream));
    return CRL_OK;
}

int crl_stream_synchronize(void *stream)
{
    CRL_HIP(hipStreamSynchronize((hipStream_t)stream));
    return CRL_OK;
}

int crl_stream_wait_mapped(void *stream, uint32_t *flag_device, const volatile uint32_t *flag_host, uint32_t seq, double timeout_s)
{
    CRL_REQUIRE(flag_device != nullptr && flag_host != nullptr, "crl_stream_wait_mapped: NULL flag pointer");
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag_device, seq);
    CRL_LAUNCH_CHECK();
    return crl_spin_mapped(stream, flag_host, seq, timeout_s, "crl_stream_wait_mapped");
}

int crl_diag_bounds(uint32_t *out12)
{
    CRL_REQUIRE(out12 != nullptr, "crl_diag_bounds: out12 is NULL");
#ifdef CRL_BOUNDS
    {   // a build with asserts proves on every call that a failing assert is recorded
        unsigned int st[4] = {0, 0, 0, 0};
        CRL_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_crl_bounds), st, sizeof(st)));
        hipLaunchKernelGGL(bounds_selftest_kernel, dim3(1), dim3(1), 0, 0);
        CRL_LAUNCH_CHECK();
        CRL_BOUNDS_READBACK(st);
        CRL_REQUIRE(st[0] == 1u && st[1] == 999u && st[2] == 5u && st[3] == 3u,
                    "crl_diag_bounds: self-test failed (%u failures, first code %u value %u limit %u)", st[0], st[1], st[2], st[3]);
    }
#endif
    int rc = crl_tron_bounds(out12);
    if (rc == CRL_OK) rc = crl_ttt_bounds(out12 + 4);
    if (rc == CRL_OK) rc = crl_blokus_bounds(out12 + 8);
    if (rc != CRL_OK) return rc;
#ifdef CRL_BOUNDS
    return 1;                                                    // the asserts are compiled in
#else
    return 0;
#endif
}

int crl_diag_issue_probe(uint32_t *out, uint64_t *clk, int blocks, int iters, void *stream)
{
    CRL_REQUIRE(out != nullptr && clk != nullptr, "crl_diag_issue_probe: NULL pointer");
    CRL_REQUIRE(blocks > 0 && blocks <= (1 << 16) && iters > 0 && iters <= (1 << 24), "crl_diag_issue_probe: blocks / iters out of range");
    hipLaunchKernelGGL(issue_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, clk, iters);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

int crl_philox4x32(const uint32_t *ctr, uint32_t key0, uint32_t key1, uint32_t *out, int64_t n, void *stream)
{
    CRL_REQUIRE(ctr && out, "crl_philox4x32: NULL pointer");
    CRL_REQUIRE(n > 0, "crl_philox4x32: n must be positive");
    CRL_REQUIRE(((((uintptr_t)ctr) | ((uintptr_t)out)) & 15) == 0, "crl_philox4x32: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(philox_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ctr, key0, key1, out, n);
    CRL_LAUNCH_CHECK();
    return CRL_OK;
}

} // extern "C"
