// Micro-benchmark: sustained integer VALU issue rate per SIMD vs waves per SIMD on gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) valu_kernel(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {   // 64 independent-ish 32-bit adds/xors per iteration (8 chains)
            a0 = (a0 ^ a1) + 0x9e3779b9u; a1 = (a1 ^ a2) + 0x7f4a7c15u; a2 = (a2 ^ a3) + 0x85ebca6bu; a3 = (a3 ^ a4) + 0xc2b2ae35u;
            a4 = (a4 ^ a5) + 0x27d4eb2fu; a5 = (a5 ^ a6) + 0x165667b1u; a6 = (a6 ^ a7) + 0xd3a2646cu; a7 = (a7 ^ a0) + 0xfd7046c5u;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

int main()
{
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(uint32_t));
    const int iters = 20000;
    const double ops_per_thread = (double)iters * 8 * 8 * 2;   // xor + add
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks = 256 * wps;                          // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(256), 0, 0, out, 100);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(valu_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instr_per_simd = ops_per_thread * wps;                 // each SIMD runs wps waves
        const double cycles = ms * 1e-3 * 2.4e9;
        printf("waves/SIMD %d: %.3f ms, %.2f cycles per wave64 VALU instruction per SIMD (at 2.4 GHz)\n", wps, ms, cycles / wave_instr_per_simd);
    }
    return 0;
}
