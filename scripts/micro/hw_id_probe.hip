// Where do the four waves of a 256-thread workgroup land?  (HW_ID: wave slot, SIMD, CU, SE; XCC_ID)  One row per wave.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/hw_id_probe scripts/micro/hw_id_probe.hip ; run: /tmp/hw_id_probe [blocks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void probe(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // stay resident for a while so that the whole grid is co-resident like the persistent loop
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(100);
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + threadIdx.x / 64;
        out[2 * w] = hw; out[2 * w + 1] = xcc;
    }
}
int main(int argc, char** argv)
{
    const int nb = argc > 1 ? atoi(argv[1]) : 2048;
    unsigned* d; hipMalloc(&d, nb * 4 * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 200);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13] ...
    int simd_of_wave[4][4] = {};
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 4; ++w) simd_of_wave[w][(h[2 * (b * 4 + w)] >> 4) & 3]++;
    for (int w = 0; w < 4; ++w) printf("wave %d of a block lands on SIMD 0..3: %d %d %d %d\n", w, simd_of_wave[w][0], simd_of_wave[w][1], simd_of_wave[w][2], simd_of_wave[w][3]);
    for (int b = 0; b < nb; b += nb / 16) {
        printf("block %4d:", b);
        for (int w = 0; w < 4; ++w) {
            const unsigned x = h[2 * (b * 4 + w)], xc = h[2 * (b * 4 + w) + 1];
            printf("  [xcc %u se %u cu %2u simd %u slot %2u]", xc & 15, (x >> 13) & 7, (x >> 8) & 15, (x >> 4) & 3, x & 15);
        }
        printf("\n");
    }
    // blocks per (xcc, se, cu) and the order they arrived in
    return 0;
}
