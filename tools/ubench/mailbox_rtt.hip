// Round trip of a host <-> GPU MAILBOX on host-mapped memory against launch + synchronise (gfx950 / MI355X).
//
// The single-state drop-in calls (B = 1) cost 15 us of launch + synchronise whatever the kernel does.  The alternative: ONE
// resident wave polls a request counter in page-locked host memory the GPU maps; the host writes its request, bumps the
// counter and spins on the reply counter.  This benchmark measures that round trip (payload: `words` dwords read from mapped
// memory, summed, result written back) and, for comparison, the same work as launch + hipStreamSynchronize.
// SAFETY: the resident wave leaves its loop when told to (quit flag), after `max_requests`, and unconditionally after
// 50 ms of wall clock (s_memrealtime) -- it can never outlive the process by more than that.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/mailbox_rtt.hip -o /tmp/mailbox_rtt && /tmp/mailbox_rtt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Mailbox {
    volatile uint32_t req;        // host -> GPU: request number
    volatile uint32_t quit;       // host -> GPU
    uint32_t pad0[14];
    volatile uint32_t ack;        // GPU -> host: last request served
    volatile uint32_t result;
    volatile uint32_t exited;     // GPU -> host: the wave has left its loop (1 quit, 2 request budget, 3 wall-clock cap)
    uint32_t pad1[13];
    uint32_t payload[1024];
};

__global__ void __launch_bounds__(64) server_kernel(Mailbox *mb, int words, uint32_t max_requests)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint32_t served = 0, why = 0;
    while (true) {
        const uint32_t r = __hip_atomic_load(&mb->req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (r != served) {
            uint32_t sum = 0;
            for (int i = threadIdx.x; i < words; i += 64)
                sum += __hip_atomic_load(&mb->payload[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            if (threadIdx.x == 0) {
                __hip_atomic_store(&mb->result, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&mb->ack, r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            served = r;
            if (served >= max_requests) { why = 2; break; }
        }
        if (__hip_atomic_load(&mb->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) { why = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000ull) { why = 3; break; }       // 50 ms at 100 MHz: unconditional
    }
    if (threadIdx.x == 0) __hip_atomic_store(&mb->exited, why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// `seq` != 0: the kernel also publishes `seq` in the mailbox's ack word behind its result (system-scope release), so that the host can
// spin on mapped memory instead of calling hipStreamSynchronize
__global__ void __launch_bounds__(64) oneshot_kernel(Mailbox *mb, int words, uint32_t seq)
{
    uint32_t sum = 0;
    for (int i = threadIdx.x; i < words; i += 64) sum += mb->payload[i];
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (threadIdx.x == 0) {
        mb->result = sum;
        if (seq) __hip_atomic_store(&mb->ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// a kernel of its own that only publishes the sequence number: what a generic "signal" call behind ANY launch chain would be
__global__ void __launch_bounds__(64) signal_kernel(Mailbox *mb, uint32_t seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(&mb->ack, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    Mailbox *host = nullptr, *dev = nullptr;
    CHECK(hipHostMalloc((void **)&host, sizeof(Mailbox), hipHostMallocMapped | hipHostMallocCoherent));
    CHECK(hipHostGetDevicePointer((void **)&dev, host, 0));
    hipStream_t stream;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    printf("{\"rows\": [\n");
    for (int words : {1, 128, 512}) {
        memset((void *)host, 0, sizeof(Mailbox));
        for (int i = 0; i < 1024; ++i) host->payload[i] = (uint32_t)i;
        const uint32_t n = 2000;
        hipLaunchKernelGGL(server_kernel, dim3(1), dim3(64), 0, stream, dev, words, n);
        std::vector<double> rt;
        bool lost = false;
        for (uint32_t k = 1; k <= n && !lost; ++k) {
            host->payload[0] = k;                                   // the request's content changes every time
            const double t0 = now_us();
            std::atomic_thread_fence(std::memory_order_release);
            host->req = k;
            while (host->ack != k) {
                if (now_us() - t0 > 20000.0) { lost = true; break; }  // the server is gone (wall-clock cap)
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            rt.push_back(now_us() - t0);
            uint32_t want = k;
            for (int i = 1; i < words; ++i) want += (uint32_t)i;
            if (!lost && host->result != want) { fprintf(stderr, "wrong result %u != %u\n", host->result, want); return 1; }
        }
        host->quit = 1;
        CHECK(hipStreamSynchronize(stream));
        std::sort(rt.begin(), rt.end());
        // the same payload as a launch + synchronise per request
        std::vector<double> ls;
        for (int k = 0; k < 500; ++k) {
            const double t0 = now_us();
            hipLaunchKernelGGL(oneshot_kernel, dim3(1), dim3(64), 0, stream, dev, words, 0u);
            CHECK(hipStreamSynchronize(stream));
            ls.push_back(now_us() - t0);
        }
        std::sort(ls.begin(), ls.end());
        // ... and as a launch whose kernel publishes a sequence number in mapped memory, the host spinning on it
        std::vector<double> lf;
        host->ack = 0;
        for (uint32_t k = 1; k <= 500; ++k) {
            const double t0 = now_us();
            hipLaunchKernelGGL(oneshot_kernel, dim3(1), dim3(64), 0, stream, dev, words, k);
            while (host->ack != k) {
                if (now_us() - t0 > 20000.0) { fprintf(stderr, "flag never came\n"); return 1; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            lf.push_back(now_us() - t0);
        }
        CHECK(hipStreamSynchronize(stream));
        std::sort(lf.begin(), lf.end());
        std::vector<double> l2;
        host->ack = 0;
        for (uint32_t k = 1; k <= 500; ++k) {
            const double t0 = now_us();
            hipLaunchKernelGGL(oneshot_kernel, dim3(1), dim3(64), 0, stream, dev, words, 0u);
            hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, stream, dev, k);
            while (host->ack != k) {
                if (now_us() - t0 > 20000.0) { fprintf(stderr, "flag never came\n"); return 1; }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            l2.push_back(now_us() - t0);
        }
        CHECK(hipStreamSynchronize(stream));
        std::sort(l2.begin(), l2.end());
        printf("  {\"payload_dwords\": %d, \"mailbox_round_trip_us\": {\"median\": %.2f, \"p10\": %.2f, \"p90\": %.2f}, \"served\": %zu, \"exit_reason\": %u, "
               "\"launch_plus_sync_us\": {\"median\": %.2f, \"p10\": %.2f}, \"launch_plus_spin_on_mapped_flag_us\": {\"median\": %.2f, \"p10\": %.2f}, \"launch_plus_signal_kernel_plus_spin_us\": {\"median\": %.2f, \"p10\": %.2f}}%s\n",
               words, rt[rt.size() / 2], rt[rt.size() / 10], rt[rt.size() * 9 / 10], rt.size(), host->exited, ls[ls.size() / 2], ls[ls.size() / 10],
               lf[lf.size() / 2], lf[lf.size() / 10], l2[l2.size() / 2], l2[l2.size() / 10], words == 512 ? "" : ",");
    }
    printf(" ]}\n");
    CHECK(hipStreamDestroy(stream));
    CHECK(hipHostFree(host));
    return 0;
}
