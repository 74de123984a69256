// dpp_check.hip -- does the DPP exchange tree of pc_wave_total (csrc/psfm_solver.hip) count every lane of a wave exactly once?
// Integer-valued doubles: every order of summation gives the same bits, so a wrong control word shows as a wrong number.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/micro/dpp_check.bin scripts/micro/dpp_check.hip && scripts/micro/dpp_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CTRL>
__device__ __forceinline__ double pc_dpp(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    int lo = (int)(unsigned)b, hi = (int)(unsigned)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
__device__ __forceinline__ double pc_readlane(double v, int lane)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)lo));
}
template <bool MAX>
__device__ __forceinline__ double pc_wave_total(double v)
{
#define PC_OP(a_, b_) (MAX ? fmax((a_), (b_)) : (a_) + (b_))
    v = PC_OP(v, pc_dpp<0xB1>(v));
    v = PC_OP(v, pc_dpp<0x4E>(v));
    v = PC_OP(v, pc_dpp<0x141>(v));
    v = PC_OP(v, pc_dpp<0x140>(v));
    const double r0 = pc_readlane(v, 0), r1 = pc_readlane(v, 16), r2 = pc_readlane(v, 32), r3 = pc_readlane(v, 48);
    return PC_OP(PC_OP(PC_OP(r0, r1), r2), r3);
#undef PC_OP
}

// out[0..63]: indicator of lane j summed; out[64]: sum of 2^lane (lanes < 52) ; out[65]: sum of lane ids; out[66]: max of (lane * 7) % 64
__global__ void check(double* out)
{
    const int lane = threadIdx.x;
    for (int j = 0; j < 64; ++j) {
        const double t = pc_wave_total<false>(lane == j ? 1.0 : 0.0);
        if (lane == 0) out[j] = t;
    }
    const double a = pc_wave_total<false>(lane < 52 ? (double)(1ull << lane) : 0.0);
    const double b = pc_wave_total<false>((double)lane);
    const double c = pc_wave_total<true>((double)((lane * 7) % 64));
    // every lane must hold the same total
    const double a2 = pc_wave_total<true>(a), a3 = -pc_wave_total<true>(-a);
    if (lane == 0) { out[64] = a; out[65] = b; out[66] = c; out[67] = a2 == a3 ? 1.0 : 0.0; }
}

int main()
{
    double* d;
    double h[68];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("dpp_check: no device\n"); return 2; }
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("dpp_check: copy failed\n"); return 2; }
    int bad = 0;
    for (int j = 0; j < 64; ++j) if (h[j] != 1.0) { printf("lane %d counted %g times\n", j, h[j]); ++bad; }
    if (h[64] != (double)((1ull << 52) - 1ull)) { printf("sum of 2^lane wrong: %.0f\n", h[64]); ++bad; }
    if (h[65] != 2016.0) { printf("sum of lane ids wrong: %g\n", h[65]); ++bad; }
    if (h[66] != 63.0) { printf("max wrong: %g\n", h[66]); ++bad; }
    if (h[67] != 1.0) { printf("lanes disagree on the total\n"); ++bad; }
    printf(bad ? "dpp_check: FAILED (%d)\n" : "dpp_check: ok\n", bad);
    return bad ? 1 : 0;
}
