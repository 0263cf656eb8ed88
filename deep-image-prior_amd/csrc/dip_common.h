// Shared device/host helpers for the gfx950 kernels of libdip_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dip_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DIP_WAVE 64

extern "C" void dip_set_error(const char* msg);

#define DIP_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            dip_set_error(hipGetErrorString(e__));           \
            return (int)e__;                                 \
        }                                                    \
    } while (0)

#define DIP_FAIL(msg)          \
    do {                       \
        dip_set_error(msg);    \
        return -1;             \
    } while (0)

static inline int dip_round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int dip_cdiv(int a, int b) { return (a + b - 1) / b; }

// max(t, slope*t) == LeakyReLU for slope in (0,1]; slope == 1 -> identity
__device__ __forceinline__ float dip_act(float t, float slope) { return fmaxf(t, slope * t); }

// mirror index v into [0, n) (ReflectionPad semantics: no edge repeat); requires |overshoot| < n
__device__ __forceinline__ int dip_reflect(int v, int n) {
    if (v < 0) v = -v;
    if (v >= n) v = 2 * (n - 1) - v;
    return v;
}

// Chan et al. pairwise combination of (count, mean, M2)
__device__ __forceinline__ void dip_chan(float& na, float& ma, float& Ma, float nb, float mb, float Mb) {
    float n = na + nb;
    if (n > 0.f) {
        float d = mb - ma;
        float f = nb / n;
        ma = ma + d * f;
        Ma = Ma + Mb + d * d * na * f;
    }
    na = n;
}
__device__ __forceinline__ void dip_chan_d(double& na, double& ma, double& Ma, double nb, double mb, double Mb) {
    double n = na + nb;
    if (n > 0.0) {
        double d = mb - ma;
        double f = nb / n;
        ma = ma + d * f;
        Ma = Ma + Mb + d * d * na * f;
    }
    na = n;
}

// XCD-aware bijective remap of a linear workgroup id: consecutive ids land on different XCDs
// (id % 8), so give each XCD a contiguous range of tiles (neighbouring tiles share halo rows
// and the packed weights in that XCD's L2).
__device__ __forceinline__ int dip_xcd_remap(int bid, int nwg) {
    const int nx = 8;
    int xcd = bid % nx, idx = bid / nx;
    int q = nwg / nx, r = nwg % nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
