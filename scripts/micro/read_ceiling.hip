// What can a kernel with flow_check's STREAMS reach on this device, gathers taken out?  (is stand-alone
// flow_check -- 17 bytes per pixel: 8 F + 8 B read, 1 byte written -- short of the memory system, or at it?)
//   v0  read one array (16 B per lane, block-strided chunks), tiny write per block
//   v1  read two arrays at the same index (F and B as streams)
//   v2  v1 + one byte written per pixel (flow_check's exact byte counts, perfectly coalesced)
//   v3  v2 with the B index shifted by a per-pixel pseudo-flow of a few pixels (the gather pattern without the arithmetic)
// each with the chunk -> block mapping in id order (x = 0) or XCD-banded (x = 1).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/read_ceiling scripts/micro/read_ceiling.hip ; run: /tmp/read_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ F, const float4* __restrict__ B, uint8_t* __restrict__ O,
                                                     float* __restrict__ sink, int64_t n4_per_pair, int W2, int xcd_per, int nchunks)
{
    // one chunk = 512 float4 = 1024 pixels; a thread takes float4 #tid and #tid + 256 of its chunk
    int bx = blockIdx.x;
    if (xcd_per > 0) { bx = (bx & 7) * xcd_per + (bx >> 3); if (bx >= nchunks) return; }
    const int64_t base = (int64_t)blockIdx.y * n4_per_pair;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int64_t i = (int64_t)bx * 512 + k * 256 + threadIdx.x;
        if (i >= n4_per_pair) break;
        const float4 f = F[base + i];
        acc += f.x + f.z;
        if (V >= 1) {
            int64_t j = i;
            if (V >= 3) {   // two rows of B near p + flow: offsets of a few pixels, like the bilinear taps
                const int dx = ((int)(f.x * 7.0f)) % 4, dy = ((int)(f.y * 5.0f)) % 3;
                j = i + dx + (int64_t)dy * W2;
                j = j < 0 ? 0 : (j >= n4_per_pair - W2 ? n4_per_pair - W2 - 1 : j);
                const float4 b2 = B[base + j + W2];
                acc += b2.y;
            }
            const float4 b = B[base + j];
            acc += b.y + b.w;
        }
        if (V >= 2) *(uchar2*)(O + (base + i) * 2) = make_uchar2(acc > 1e30f, acc < -1e30f);
    }
    if (acc == 123.456f) sink[blockIdx.x] = acc;
}

int main()
{
    const int H = 1080, W = 1920, NP = 100;
    const int64_t P = (int64_t)H * W, n4 = P / 2;          // float4 = two pixels
    float4 *F, *B; uint8_t* O; float* sink;
    CHK(hipMalloc(&F, NP * n4 * 16)); CHK(hipMalloc(&B, NP * n4 * 16)); CHK(hipMalloc(&O, NP * P)); CHK(hipMalloc(&sink, 1 << 20));
    CHK(hipMemset(F, 0x3c, NP * n4 * 16)); CHK(hipMemset(B, 0x3c, NP * n4 * 16));
    const int nchunks = (int)((n4 + 511) / 512);
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int v = 0; v < 4; ++v)
        for (int x = 0; x < 2; ++x) {
            const int per = x ? (nchunks + 7) / 8 : 0;
            dim3 grid(x ? 8 * per : nchunks, NP);
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CHK(hipEventRecord(e0));
                switch (v) {
                    case 0: hipLaunchKernelGGL(stream_kernel<0>, grid, dim3(256), 0, 0, F, B, O, sink, n4, W / 2, per, nchunks); break;
                    case 1: hipLaunchKernelGGL(stream_kernel<1>, grid, dim3(256), 0, 0, F, B, O, sink, n4, W / 2, per, nchunks); break;
                    case 2: hipLaunchKernelGGL(stream_kernel<2>, grid, dim3(256), 0, 0, F, B, O, sink, n4, W / 2, per, nchunks); break;
                    default: hipLaunchKernelGGL(stream_kernel<3>, grid, dim3(256), 0, 0, F, B, O, sink, n4, W / 2, per, nchunks); break;
                }
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double bytes = (double)NP * P * (v == 0 ? 8.0 : (v == 1 ? 16.0 : 17.0));
            printf("{\"variant\": %d, \"xcd_banded\": %d, \"ms\": %.4f, \"algorithmic_GBs\": %.1f}\n", v, x, best, bytes / (best * 1e-3) / 1e9);
        }
    return 0;
}
