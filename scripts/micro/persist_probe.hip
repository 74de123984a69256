// Micro-benchmark (debug, not part of the product): what does one frame of a PERSISTENT chain loop cost on MI355X?
//   mode 0: grid barrier only                      (arrive atomic + spin)
//   mode 1: + coherent byte load after the barrier and a write-through byte store before it
//   mode 2: + per-thread bilinear-ish gathers (4 x 8 B + 4 x 1 B) from a fresh 1080p frame and a 16 B log store
//   mode 3: as 2, plus a SECOND dependent gather after the coherent byte load (stands for the newborns' first step)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define NSH 64
struct Args {
    const float2* flow; const uint8_t* occ; double2* log; uint8_t* map; unsigned* arrive; int* abort_flag;
    unsigned* shard_cnt;   // NSH counters, one per 128-byte line (monotonic)
    unsigned* top_cnt;     // one counter (monotonic)
    unsigned* flags;       // NSH release flags, one per 128-byte line: last completed frame + 1
    int H, W, GW, G, frames, mode, nblk; long long spin_limit;
};

__device__ __forceinline__ bool wait_barrier(unsigned* ctr, unsigned target, long long limit, int* abort_flag)
{
    long long n = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++n > limit) { *abort_flag = 1; return false; }
        if ((n & 255) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void persist(Args a)
{
    __shared__ int s_ok;
    const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
    const int gy = i / a.GW, gx = i - gy * a.GW;
    double2 p = make_double2(gx * 2.0, gy * 2.0);
    const bool live = i < a.G;
    float acc = 0.f;
    const size_t P = (size_t)a.H * a.W;
    for (int t = 0; t < a.frames; ++t) {
        const float2* F = a.flow + (size_t)t * P;
        const uint8_t* O = a.occ + (size_t)t * P;
        uint8_t* map_cur = a.map + (size_t)((t + 1) & 1) * a.G;
        const uint8_t* map_prev = a.map + (size_t)(t & 1) * a.G;
        float2 f0, f1, f2, f3; unsigned o0 = 0, o1 = 0, o2 = 0, o3 = 0;
        int x0 = 0, y0 = 0;
        if (a.mode >= 2) {
            x0 = min(max((int)p.x, 0), a.W - 2); y0 = min(max((int)p.y, 0), a.H - 2);
            const size_t k = (size_t)y0 * a.W + x0;
            f0 = F[k]; f1 = F[k + 1]; f2 = F[k + a.W]; f3 = F[k + a.W + 1];
            o0 = O[k]; o1 = O[k + 1]; o2 = O[k + a.W]; o3 = O[k + a.W + 1];
        }
        // wait for the previous frame's barrier
        if (t > 0) {
            if (tid == 0) s_ok = wait_barrier(a.flags + (blockIdx.x % NSH) * 32, (unsigned)t, a.spin_limit, a.abort_flag) ? 1 : 0;
            __syncthreads();
            if (!s_ok) return;
        }
        unsigned b = 0;
        if (a.mode >= 1 && live) b = __hip_atomic_load(map_prev + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (a.mode >= 2) {
            const float fx = 0.25f * (f0.x + f1.x + f2.x + f3.x), fy = 0.25f * (f0.y + f1.y + f2.y + f3.y);
            acc += (float)(o0 + o1 + o2 + o3);
            p.x += fx; p.y += fy;
            if (live) a.log[(size_t)(t + 1) * a.G + i] = p;
        }
        if (a.mode >= 3) {
            // dependent on the coherent byte: a second gather round trip
            const int xx = min(max(gx * 2 + (int)(b & 1), 0), a.W - 2), yy = min(max(gy * 2, 0), a.H - 2);
            const size_t k = (size_t)yy * a.W + xx;
            const float2 g0 = F[k], g1 = F[k + 1], g2 = F[k + a.W], g3 = F[k + a.W + 1];
            acc += g0.x + g1.x + g2.x + g3.x;
        }
        if (a.mode >= 1 && live) {
            const int px = min(max((int)p.x, 0), a.W - 1) >> 1, py = min(max((int)p.y, 0), a.H - 1) >> 1;
            uint8_t* m = map_cur + (size_t)py * a.GW + px;
            const uint8_t val = (uint8_t)(t + 1 + (b & 0));
            if (a.mode <= 3) {
                __hip_atomic_store(m, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else if (a.mode == 4) {        // 4 write-through byte stores per track
                __hip_atomic_store(m, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (px + 1 < a.GW) __hip_atomic_store(m + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (py + 1 < a.H / 2) {
                    __hip_atomic_store(m + a.GW, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (px + 1 < a.GW) __hip_atomic_store(m + a.GW + 1, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            } else if (a.mode == 5) {        // one device-scope atomicOr per track on a bit map (no aggregation)
                const int cell = py * a.GW + px;
                __hip_atomic_fetch_or((unsigned*)map_cur + (cell >> 5), 1u << (cell & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (a.mode == 6) {        // 4 plain byte stores per track, made visible by a release fence before the arrival
                m[0] = val;
                if (px + 1 < a.GW) m[1] = val;
                if (py + 1 < a.H / 2) { m[a.GW] = val; if (px + 1 < a.GW) m[a.GW + 1] = val; }
            } else if (a.mode == 7) {        // 4 device-scope atomicOr per track on a bit map
                const int cell = py * a.GW + px;
                unsigned* w = (unsigned*)map_cur;
                __hip_atomic_fetch_or(w + (cell >> 5), 1u << (cell & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_or(w + ((cell + 1) >> 5), 1u << ((cell + 1) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (py + 1 < a.H / 2) {
                    __hip_atomic_fetch_or(w + ((cell + a.GW) >> 5), 1u << ((cell + a.GW) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_or(w + ((cell + a.GW + 1) >> 5), 1u << ((cell + a.GW + 1) & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        if (a.mode == 6) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_s_waitcnt(0);   // all of this wave's memory operations acknowledged
        __syncthreads();
        if (tid < 64) {
            // two-level arrival: shard counter -> top counter -> the last arriver's wave publishes the 64 release flags
            int last = 0;
            if (tid == 0) {
                const int sh = blockIdx.x % NSH;
                const unsigned members = (unsigned)((a.nblk - sh + NSH - 1) / NSH);
                const unsigned old = __hip_atomic_fetch_add(a.shard_cnt + sh * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == members * (unsigned)(t + 1)) {
                    const unsigned nsh = (unsigned)(a.nblk < NSH ? a.nblk : NSH);
                    const unsigned o2 = __hip_atomic_fetch_add(a.top_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (o2 + 1 == nsh * (unsigned)(t + 1)) last = 1;
                }
            }
            last = __builtin_amdgcn_readfirstlane(last);
            if (last) __hip_atomic_store(a.flags + tid * 32, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (acc == 123.456f) a.log[i] = p;
}

int main(int argc, char** argv)
{
    const int H = 1080, W = 1920, GW = 960, GH = 540, G = GW * GH, frames = 100;
    const size_t P = (size_t)H * W;
    Args a; a.H = H; a.W = W; a.GW = GW; a.G = G; a.frames = frames; a.spin_limit = 2000000;
    float2* flow; uint8_t* occ; double2* log; uint8_t* map; unsigned* arrive; int* abortf;
    CK(hipMalloc(&flow, P * 8 * frames)); CK(hipMalloc(&occ, P * frames)); CK(hipMalloc(&log, (size_t)G * 16 * (frames + 1)));
    CK(hipMalloc(&map, 2 * G)); CK(hipMalloc(&arrive, 4 * frames)); CK(hipMalloc(&abortf, 4));
    unsigned* bar; CK(hipMalloc(&bar, 128 * (2 * NSH + 1)));
    a.shard_cnt = bar; a.top_cnt = bar + 32 * NSH; a.flags = bar + 32 * (NSH + 1);
    {
        std::vector<float2> h(P);
        for (size_t k = 0; k < P; ++k) { h[k].x = 1.5f * sinf(k * 1e-3f); h[k].y = 1.5f * cosf(k * 1.3e-3f); }
        for (int t = 0; t < frames; ++t) CK(hipMemcpy(flow + t * P, h.data(), P * 8, hipMemcpyHostToDevice));
        CK(hipMemset(occ, 0, P * frames));
    }
    a.flow = flow; a.occ = occ; a.log = log; a.map = map; a.arrive = arrive; a.abort_flag = abortf;
    int nb_per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_per_cu, persist, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d, blocks/CU %d\n", prop.multiProcessorCount, nb_per_cu);
    const int nblk = (G + 255) / 256;
    if (nblk > nb_per_cu * prop.multiProcessorCount) { printf("grid %d does not fit\n", nblk); return 1; }
    a.nblk = nblk;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 8; ++mode) {
        a.mode = mode;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(arrive, 0, 4 * frames)); CK(hipMemset(bar, 0, 128 * (2 * NSH + 1))); CK(hipMemset(abortf, 0, 4)); CK(hipMemset(map, 0, 2 * G));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            void* params[] = { &a };
            CK(hipLaunchCooperativeKernel((const void*)persist, dim3(nblk), dim3(256), params, 0, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            int ab = 0; CK(hipMemcpy(&ab, abortf, 4, hipMemcpyDeviceToHost));
            printf("mode %d rep %d: %.3f ms total, %.2f us per frame%s\n", mode, rep, ms, ms * 1e3 / frames, ab ? "  ABORTED (spin limit)" : "");
        }
    }
    return 0;
}
