// Shared device/host helpers for the gfx950 kernels of libdip_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "dip_hip.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DIP_WAVE 64

extern "C" void dip_set_error(const char* msg);
// grouped execution (dip_group.h): 1 when a launch since the last call refused to run (a pointer outside the slab, ...);
// the message is in dip_last_error()
extern "C" int dip_group_take_fault(void);

#define DIP_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) {                             \
            dip_set_error(hipGetErrorString(e__));           \
            return (int)e__;                                 \
        }                                                    \
        if (dip_group_take_fault()) return -1;               \
    } while (0)

#define DIP_FAIL(msg)          \
    do {                       \
        dip_set_error(msg);    \
        return -1;             \
    } while (0)

static inline int dip_round_up(int x, int m) { return (x + m - 1) / m * m; }
// "has this one-time per-device set-up (hipFuncSetAttribute, ...) been done on the CURRENT device?": the flag array is
// owned by the caller (one per kernel instantiation); a process that drives several GPUs sets every one of them up
static inline bool dip_once_per_device(bool (&done)[16]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return true;      // (unknown device: redo the set-up)
    const bool first = !done[dev];
    done[dev] = true;
    return first;
}
static inline int dip_cdiv(int a, int b) { return (a + b - 1) / b; }

// Activation behind a BatchNorm, encoded in DipTransform.slope (models/common.py:76-92 of the reference):
//   slope in (0, 1]          LeakyReLU(slope): max(t, slope*t); slope == 1 -> identity ('none')
//   slope == DIP_ACT_SWISH   Swish: t * sigmoid(t)                       (models/common.py:62-73)
//   slope == DIP_ACT_ELU     nn.ELU(alpha = 1): t > 0 ? t : expm1(t)
//   slope == DIP_ACT_RELU    nn.ReLU: max(t, 0)        (act_fun given as a module class, models/common.py:90-91)
// The branch is wave-uniform.  dip_act_leaky is the LeakyReLU-only form for the LDS-DMA conv kernel,
// whose K loop is scheduled instruction by instruction (other activations take the register-staged kernel).
__device__ __forceinline__ float dip_act_leaky(float t, float slope) { return fmaxf(t, slope * t); }
__device__ __forceinline__ float dip_act(float t, float slope) {
    if (slope > 0.f) return fmaxf(t, slope * t);
    if (slope == DIP_ACT_SWISH) return t / (1.f + expf(-t));
    if (slope == DIP_ACT_RELU) return fmaxf(t, 0.f);
    return t > 0.f ? t : expm1f(t);
}
// a * b rounded to fp32 and NOT contractable into a neighbouring add (hipcc fuses across __fmul_rn under its
// default -ffp-contract=fast): two kernels that must produce the same bits use this
__device__ __forceinline__ float dip_mul_rn(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return p;
}
// d act / d t
__device__ __forceinline__ float dip_act_grad(float t, float slope) {
    if (slope > 0.f) return t > 0.f ? 1.f : slope;
    if (slope == DIP_ACT_SWISH) {
        const float sg = 1.f / (1.f + expf(-t));
        return sg * (1.f + t * (1.f - sg));
    }
    if (slope == DIP_ACT_RELU) return t > 0.f ? 1.f : 0.f;
    return t > 0.f ? 1.f : expf(t);
}

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

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N, class F>
__device__ __forceinline__ void dip_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        dip_static_for<I + 1, N>(f);
    }
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

// ---------------------------------------------------------------------------------------------
// Block-level tree reductions for the NHWC streaming kernels.  Thread (prow, cg) owns 4 channels;
// the `rpi` threads that share a channel group `cg` are combined in log2(rpi) steps through LDS
// (a single thread looping over 255 partners was measured at 80-90 us for thin, 4-channel tensors).
// `sh` holds 12 floats per thread; the result ends up in the registers of the prow == 0 threads.
// Deterministic: fixed pairing order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int dip_pow2_ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

__device__ __forceinline__ void dip_tree_chan4(float* sh, int nc4, int rpi, int prow, int cg, bool active, float& n,
                                               f32x4& mean, f32x4& M2) {
    float* mine = sh + (size_t)threadIdx.x * 12;
    mine[0] = n;
    *reinterpret_cast<f32x4*>(mine + 4) = mean;
    *reinterpret_cast<f32x4*>(mine + 8) = M2;
    for (int s = dip_pow2_ceil(rpi) >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (active && prow < s && prow + s < rpi) {
            const float* q = sh + (size_t)((prow + s) * nc4 + cg) * 12;
            const float nb = q[0];
            float na = n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float ne = n, me = mean[e], Me = M2[e];
                dip_chan(ne, me, Me, nb, q[4 + e], q[8 + e]);
                mean[e] = me; M2[e] = Me; na = ne;
            }
            n = na;
            mine[0] = n;
            *reinterpret_cast<f32x4*>(mine + 4) = mean;
            *reinterpret_cast<f32x4*>(mine + 8) = M2;
        }
    }
}

__device__ __forceinline__ void dip_tree_sum8(float* sh, int nc4, int rpi, int prow, int cg, bool active, f32x4& s1,
                                              f32x4& s2) {
    float* mine = sh + (size_t)threadIdx.x * 8;
    *reinterpret_cast<f32x4*>(mine) = s1;
    *reinterpret_cast<f32x4*>(mine + 4) = s2;
    for (int s = dip_pow2_ceil(rpi) >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (active && prow < s && prow + s < rpi) {
            const float* q = sh + (size_t)((prow + s) * nc4 + cg) * 8;
            s1 += *reinterpret_cast<const f32x4*>(q);
            s2 += *reinterpret_cast<const f32x4*>(q + 4);
            *reinterpret_cast<f32x4*>(mine) = s1;
            *reinterpret_cast<f32x4*>(mine + 4) = s2;
        }
    }
}
