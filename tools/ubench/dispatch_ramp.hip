// How long does the dispatcher need to START a grid of short waves (gfx950 / MI355X)?
//
// Every per-step kernel of this library that maps one wave to one game (Blokus: 16,384 waves per call) and the 20-step Tron
// launch (4,096 waves) carry this floor.  A grid of W waves is launched whose waves do (a) nothing, or (b) a fixed ~2 us of
// dependent VALU work; each wave stamps the 100 MHz wall clock (s_memrealtime) at entry; reported per configuration: the
// kernel's duration by HIP events around ONE isolated launch (median of 30) and the spread of the entry stamps
// (last entry - first entry) = the time the dispatcher took to start the grid.
// Configurations: waves 1,024 .. 16,384 x workgroup 256 / 512 threads x 0 / 34 KB of LDS per 256 threads.
//
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/ubench/dispatch_ramp.hip -o /tmp/dispatch_ramp && /tmp/dispatch_ramp
// Prints one JSON object (committed as profiles/r4_dispatch_ramp.json).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the shape of the 20-step Tron launch: every lane first pulls 7 x 16 bytes of a 26 MB array (its share of 16 boards of 400
// bytes), waits for them, then works; `done` gets the wave's exit stamp
template <bool BIG_VGPR>
__global__ void __launch_bounds__(256, 4) ramp_loads_kernel(uint64_t *entry, uint64_t *done, const uint4 *__restrict__ src, uint32_t *sink, int work)
{
    __shared__ uint32_t pad[34816 / 4];
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    if (BIG_VGPR) asm volatile("v_mov_b32 v120, 0" ::: "v120");      // a 121-register allocation per lane (4 waves per SIMD still fit)
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, wave = gid >> 6;
    if ((threadIdx.x & 63) == 0) entry[wave] = t;
    uint4 v[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) v[k] = src[(size_t)wave * 448 + k * 64 + (threadIdx.x & 63)];
    uint32_t a = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 7; ++k) a += v[k].x ^ v[k].w;
    pad[threadIdx.x] = a;
    for (int i = 0; i < work; ++i) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(a));
    if (a == 0xdeadbeefu) sink[0] = a + pad[0];
    if ((threadIdx.x & 63) == 0) done[wave] = __builtin_amdgcn_s_memrealtime();
}

template <bool BIG_VGPR>
static int run_loads(int waves, int work, bool cold_l2)
{
    uint64_t *entry = nullptr, *done = nullptr;
    uint4 *src = nullptr, *flush = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(&entry, sizeof(uint64_t) * waves));
    CHECK(hipMalloc(&done, sizeof(uint64_t) * waves));
    CHECK(hipMalloc(&src, (size_t)waves * 448 * 16));
    CHECK(hipMalloc(&flush, 512u << 20));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(src, 1, (size_t)waves * 448 * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<float> ms;
    std::vector<double> spread, last_done, med_done;
    std::vector<uint64_t> he(waves), hd(waves);
    for (int rep = 0; rep < 34; ++rep) {
        if (cold_l2) CHECK(hipMemset(flush, rep, 512u << 20));         // push the array out of the L2s and the Infinity Cache
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ramp_loads_kernel<BIG_VGPR>, dim3(waves / 4), dim3(256), 0, 0, entry, done, src, sink, work);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        CHECK(hipMemcpy(he.data(), entry, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hd.data(), done, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost));
        if (rep < 4) continue;
        const uint64_t first = *std::min_element(he.begin(), he.end());
        ms.push_back(t);
        spread.push_back((double)(*std::max_element(he.begin(), he.end()) - first) * 0.01);
        std::sort(hd.begin(), hd.end());
        last_done.push_back((double)(hd.back() - first) * 0.01);
        med_done.push_back((double)(hd[waves / 2] - first) * 0.01);
    }
    for (auto *v : {&spread, &last_done, &med_done}) std::sort(v->begin(), v->end());
    std::sort(ms.begin(), ms.end());
    printf("  {\"config\": \"7 x 16 B loads per lane (%s), then work, 34 KB LDS, %s\", \"waves\": %d, \"work_iters\": %d, \"kernel_us_median\": %.2f, "
           "\"entry_spread_us_median\": %.2f, \"median_wave_done_us\": %.2f, \"last_wave_done_us\": %.2f},\n",
           cold_l2 ? "array flushed out of the caches" : "array cache-resident", BIG_VGPR ? "121 VGPRs" : "few VGPRs", waves, work, ms[ms.size() / 2] * 1e3, spread[spread.size() / 2],
           med_done[med_done.size() / 2], last_done[last_done.size() / 2]);
    CHECK(hipFree(entry)); CHECK(hipFree(done)); CHECK(hipFree(src)); CHECK(hipFree(flush)); CHECK(hipFree(sink));
    return 0;
}

template <int LDS_BYTES>
__global__ void ramp_kernel(uint64_t *entry, uint32_t *sink, int work)
{
    extern __shared__ uint32_t dyn[];
    __shared__ uint32_t pad[LDS_BYTES > 0 ? LDS_BYTES / 4 : 1];
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if ((threadIdx.x & 63) == 0) entry[wave] = t;
    uint32_t a = threadIdx.x;
    for (int i = 0; i < work; ++i) asm volatile("v_mad_u32_u24 %0, %0, %0, %0" : "+v"(a));
    if (LDS_BYTES > 0) pad[threadIdx.x] = a;                      // keep the allocation
    if (a == 0xdeadbeefu) sink[0] = a + (LDS_BYTES > 0 ? pad[0] : 0);
}

template <int LDS_BYTES>
static int run(const char *name, int waves, int threads, int work, bool last)
{
    uint64_t *entry = nullptr;
    uint32_t *sink = nullptr;
    CHECK(hipMalloc(&entry, sizeof(uint64_t) * waves));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = waves * 64 / threads;
    std::vector<float> ms;
    std::vector<double> spread;
    std::vector<uint64_t> host(waves);
    for (int rep = 0; rep < 34; ++rep) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(ramp_kernel<LDS_BYTES>, dim3(blocks), dim3(threads), 0, 0, entry, sink, work);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        CHECK(hipMemcpy(host.data(), entry, sizeof(uint64_t) * waves, hipMemcpyDeviceToHost));
        if (rep < 4) continue;
        const auto mm = std::minmax_element(host.begin(), host.end());
        ms.push_back(t);
        spread.push_back((double)(*mm.second - *mm.first) * 0.01);     // 100 MHz ticks -> us
    }
    std::sort(ms.begin(), ms.end());
    std::sort(spread.begin(), spread.end());
    printf("  {\"config\": \"%s\", \"waves\": %d, \"threads_per_group\": %d, \"lds_bytes_per_group\": %d, \"work_iters\": %d, "
           "\"kernel_us_median\": %.2f, \"entry_spread_us_median\": %.2f, \"ns_per_wave\": %.3f}%s\n",
           name, waves, threads, LDS_BYTES * (threads / 256), work, ms[ms.size() / 2] * 1e3, spread[spread.size() / 2],
           spread[spread.size() / 2] * 1e3 / waves, last ? "" : ",");
    CHECK(hipFree(entry));
    CHECK(hipFree(sink));
    return 0;
}

int main()
{
    printf("{\"note\": \"entry_spread = last wave's entry stamp - first wave's (s_memrealtime, 100 MHz); kernel_us = HIP events around one isolated launch; "
           "work_iters = dependent v_mad per lane (0: empty waves; 600: ~2 us of work)\",\n \"rows\": [\n");
    for (int work : {0, 600, 1500})
        for (int cold = 0; cold < 2; ++cold)
            if (run_loads<false>(4096, work, cold != 0) || run_loads<true>(4096, work, cold != 0)) return 1;
    const int waves[] = {1024, 2048, 4096, 8192, 16384};
    for (int w : waves) {
        if (run<0>("empty, no LDS", w, 256, 0, false)) return 1;
        if (run<34816>("empty, 34 KB LDS per group", w, 256, 0, false)) return 1;
        if (run<0>("2 us of work, no LDS", w, 256, 600, false)) return 1;
        if (run<34816>("2 us of work, 34 KB LDS per group", w, 256, 600, false)) return 1;
        if (run<0>("empty, no LDS, 512 threads", w, 512, 0, w == 16384)) return 1;
    }
    printf(" ]}\n");
    return 0;
}
