// HOST stand-in for <hip/hip_runtime.h>, only for tests/host/*: lets g++ compile the device headers of
// particle-sfm_amd/csrc (psfm_device.h, psfm_chain.h) so that the CPU suite can run the DEVICE's arithmetic -- the fp32
// sampler, the flow_check verdict, a chain step, the folded EDT respawn rule -- against the reference's golden vectors
// without a GPU.  Every *_rn intrinsic is the IEEE operation it names (the test builds with -ffp-contract=off, no fast-math,
// so a + b is one correctly rounded addition; fmaf is a correctly rounded fused multiply-add).  Test infrastructure.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { float2 v; v.x = x; v.y = y; return v; }
static inline double2 make_double2(double x, double y) { double2 v; v.x = x; v.y = y; return v; }

static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline unsigned __lane_id() { return 0u; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }

#define __HIP_MEMORY_SCOPE_SYSTEM 0
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
