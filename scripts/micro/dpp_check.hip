// dpp_check.hip -- does the exchange tree of pc_block_sums (csrc/psfm_pc_reduce.h: DPP quad_perm / row_ror, ds_swizzle) count every
// thread's term of every sum exactly once, and is SUM_GMAX the maximum?  Integer-valued doubles: every order of summation gives
// the same bits, so a wrong control word or lane mapping shows as a wrong number.
//   hipcc --offload-arch=gfx950 -O2 -I particle-sfm_amd/csrc -o scripts/micro/dpp_check.bin scripts/micro/dpp_check.hip && scripts/micro/dpp_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "psfm_pc_reduce.h"

// test t: thread i contributes v(t, k, i) to slot k
__device__ __host__ inline double term(int t, int k, int i)
{
    if (t < 256) return i == t ? (double)(k + 1) : 0.0;                  // one thread at a time: every thread counted once, in the right slot
    if (t == 256) return (double)((i * (k + 3) + 7 * k) % 1021);        // everybody, different per slot
    return (double)(((i * 37 + k * 11) % 256) * (k == SUM_GMAX ? 1 : 1));
}

template <int NS_>
__global__ void check(double* out, int t0, int nt)
{
    __shared__ double s_out[PC_NSUM];
    for (int t = t0; t < t0 + nt; ++t) {
        double acc[PC_NSUM];
        for (int k = 0; k < PC_NSUM; ++k) acc[k] = term(t, k, (int)threadIdx.x);
        pc_block_sums<NS_>(acc, s_out);
        if (threadIdx.x < NS_) out[(size_t)t * PC_NSUM + threadIdx.x] = s_out[threadIdx.x];
        __syncthreads();
    }
}

template <int NS_>
static int run(const char* name)
{
    const int NT = 258;
    double* d;
    static double h[NT * PC_NSUM];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("dpp_check: no device\n"); return 100; }
    hipLaunchKernelGGL(check<NS_>, dim3(1), dim3(PC_BLOCK), 0, 0, d, 0, NT);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("dpp_check: copy failed\n"); return 100; }
    int bad = 0;
    for (int t = 0; t < NT; ++t)
        for (int k = 0; k < NS_; ++k) {
            double want = 0.0;
            for (int i = 0; i < PC_BLOCK; ++i) { const double v = term(t, k, i); want = k == SUM_GMAX ? (v > want ? v : want) : want + v; }
            if (h[t * PC_NSUM + k] != want) { if (bad < 12) printf("%s: test %d slot %d: got %.1f want %.1f\n", name, t, k, h[t * PC_NSUM + k], want); ++bad; }
        }
    printf("%s: %s (%d wrong)\n", name, bad ? "FAILED" : "ok", bad);
    hipFree(d);
    return bad;
}

int main()
{
    const int bad = run<PC_NSUM>("pc_block_sums<13>") + run<11>("pc_block_sums<11>");
    printf(bad ? "dpp_check: FAILED\n" : "dpp_check: ok\n");
    return bad ? 1 : 0;
}
