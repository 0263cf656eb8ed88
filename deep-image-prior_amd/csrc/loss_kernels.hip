// Loss head and device-side iteration state.
//
//  * dip_loss_head_fwd / dip_loss_head_bwd: the tail of the closure fused into two launches --
//    output conv (1x1, <= 4 channels) + nn.Sigmoid (models/skip.py:96-98 of the reference) +
//    optional mask multiply + torch.nn.MSELoss (denoising.ipynb:177,219; inpainting.ipynb:310:
//    mse(out * mask, img * mask), mean over ALL elements).  HBM-bound: the 128-channel activation
//    is read once; the scalar loss is reduced wavefront -> LDS tree -> one partial per block, and
//    the last-arriving block sums the partials in a fixed order (deterministic, no float atomics).
//  * DipIterState + dip_adam_tick / dip_adam_step_dev / dip_noise_axpy_dev: Adam's step count and
//    the Philox offset live in device memory, so one optimisation iteration is a STATIC launch list
//    and can be replayed as a hipGraph (nothing changes on the host between iterations).
#include "dip_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// block = 256 threads; LPP lanes (power of two, <= 64) share a pixel, lane cg owns channels 4cg..4cg+3
__global__ __launch_bounds__(256) void loss_head_fwd_kernel(const DipLossHeadDesc d, const int LPP, const int ppb) {
    __shared__ float red[256];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int cg = tid & (LPP - 1), prow = tid / LPP, rpi = 256 / LPP;
    const int nc4 = (d.Cin + 3) >> 2;
    const bool cvalid = cg < nc4;
    float w[4][4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = cg * 4 + e;
            w[o][e] = (o < d.Cout && c < d.Cin) ? d.w[(size_t)o * d.Cin + c] : 0.f;
        }
    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool has_tr = d.tr.a != nullptr;
    if (has_tr && cvalid) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cg * 4 + e < d.Cin) { ta[e] = d.tr.a[cg * 4 + e]; tb[e] = d.tr.b[cg * 4 + e]; }
    }
    float bias[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bias[o] = (d.bias != nullptr && o < d.Cout) ? d.bias[o] : 0.f;

    const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, d.HW);
    float lsum = 0.f;
    for (int pb = p0; pb < p1; pb += rpi) {
        const int p = pb + prow;
        const bool pv = p < p1;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pv && cvalid) {
            v = *reinterpret_cast<const f32x4*>(d.u + (size_t)p * d.Cu + cg * 4);
            if (has_tr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dip_act(fmaf(ta[e], v[e], tb[e]), d.tr.slope);
            }
        }
        float acc[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = v[0] * w[o][0];
            s = fmaf(v[1], w[o][1], s);
            s = fmaf(v[2], w[o][2], s);
            s = fmaf(v[3], w[o][3], s);
            acc[o] = s;
        }
        for (int off = LPP >> 1; off >= 1; off >>= 1) {           // wavefront reduction over the pixel's lanes
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] += __shfl_xor(acc[o], off);
        }
        if (pv && cg == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o < d.Cout) {
                    float y = acc[o] + bias[o];
                    if (d.sigmoid) y = sigmoidf_(y);
                    d.out[(size_t)o * d.HW + p] = y;
                    const float t = d.target[(size_t)o * d.HW + p];
                    float a = y, b = t;
                    if (d.mask != nullptr) {
                        const float m = d.mask[(size_t)(d.mask_c == 1 ? 0 : o) * d.HW + p];
                        a = y * m;                                 // mse(out * mask, img * mask)
                        b = t * m;
                    }
                    const float df = a - b;
                    lsum = fmaf(df, df, lsum);
                }
            }
        }
    }
    // block tree (fixed order) -> one partial per block
    red[tid] = lsum;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) {
        d.partials[blockIdx.x] = red[0];
        __threadfence();
        const unsigned t = atomicAdd(d.ticket, 1u);
        is_last = (t == gridDim.x - 1);
    }
    __syncthreads();
    if (!is_last) return;
    __threadfence();
    // last-arriving block: fixed-order sum of all partials (independent of which block is last)
    double s = 0.0;
    for (int i = tid; i < (int)gridDim.x; i += 256) s += (double)__hip_atomic_load(d.partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ double dred[256];
    dred[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) dred[tid] += dred[tid + st];
        __syncthreads();
    }
    if (tid == 0) {
        *d.loss = (float)(dred[0] / ((double)d.Cout * (double)d.HW));
        *d.ticket = 0u;                                             // re-armed for the next launch / graph replay
    }
}

// dy[p][o] = gscale * 2/N * (out*m - t*m) * m * out*(1-out)       (NHWC, channel stride Cy; pad channels zero)
__global__ __launch_bounds__(256) void loss_head_bwd_kernel(const DipLossHeadDesc d, const float* __restrict__ gscale,
                                                            float* __restrict__ dy, const int Cy) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.HW) return;
    const float gs = gscale != nullptr ? *gscale : 1.f;
    const float k = 2.f / ((float)d.Cout * (float)d.HW);
    for (int c0 = 0; c0 < Cy; c0 += 4) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = c0 + e;
            if (o < d.Cout) {
                const float y = d.out[(size_t)o * d.HW + p];
                const float t = d.target[(size_t)o * d.HW + p];
                float a = y, b = t, m = 1.f;
                if (d.mask != nullptr) {
                    m = d.mask[(size_t)(d.mask_c == 1 ? 0 : o) * d.HW + p];
                    a = y * m;
                    b = t * m;
                }
                float g = (a - b) * k * gs;                        // aten mse_loss_backward: 2/N * (x - t) * grad
                if (d.mask != nullptr) g = g * m;                  // MulBackward
                if (d.sigmoid) g = g * ((1.f - y) * y);            // aten sigmoid_backward
                v[e] = g;
            }
        }
        *reinterpret_cast<f32x4*>(dy + (size_t)p * Cy + c0) = v;
    }
}

// ---------------------------------------------------------------- device-side iteration state
__global__ void adam_tick_kernel(DipIterState* st, double lr, double beta1, double beta2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long step = st->step + 1ull;
    st->step = step;
    // scalar prep as torch/optim/adam.py (_single_tensor_adam), in double
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    st->step_size = (float)(lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
}

__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

}  // namespace

static int lpp_of(int Cin) {
    const int nc4 = (Cin + 3) / 4;
    int l = 1;
    while (l < nc4) l <<= 1;
    return l;
}

extern "C" int dip_loss_head_nblk(int HW, int Cin) {
    const int rpi = 256 / lpp_of(Cin);
    int ppb = dip_cdiv(HW, 1024);
    ppb = dip_round_up(ppb < rpi * 4 ? rpi * 4 : ppb, rpi);
    return dip_cdiv(HW, ppb);
}

extern "C" int dip_loss_head_fwd(const DipLossHeadDesc* dp, void* stream) {
    const DipLossHeadDesc& d = *dp;
    if (d.Cout < 1 || d.Cout > 4) DIP_FAIL("loss_head: 1..4 output channels");
    if (d.Cin < 1 || d.Cin > 256 || (d.Cu & 3)) DIP_FAIL("loss_head: Cin must be <= 256 and the channel stride a multiple of 4");
    if (d.mask != nullptr && d.mask_c != 1 && d.mask_c != d.Cout) DIP_FAIL("loss_head: mask must have 1 or Cout channels");
    const int LPP = lpp_of(d.Cin), rpi = 256 / LPP;
    int ppb = dip_cdiv(d.HW, 1024);
    ppb = dip_round_up(ppb < rpi * 4 ? rpi * 4 : ppb, rpi);
    const int nblk = dip_cdiv(d.HW, ppb);
    if (nblk != d.nblk) DIP_FAIL("loss_head: nblk must come from dip_loss_head_nblk");
    hipLaunchKernelGGL(loss_head_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, d, LPP, ppb);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_loss_head_bwd(const DipLossHeadDesc* dp, const float* gscale, float* dy, int Cy, void* stream) {
    const DipLossHeadDesc& d = *dp;
    if (d.Cout < 1 || d.Cout > 4 || (Cy & 3) || Cy < d.Cout) DIP_FAIL("loss_head_bwd: bad channel counts");
    hipLaunchKernelGGL(loss_head_bwd_kernel, dim3(dip_cdiv(d.HW, 256)), dim3(256), 0, (hipStream_t)stream, d, gscale,
                       dy, Cy);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_adam_tick(DipIterState* st, double lr, double beta1, double beta2, void* stream) {
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, st, lr, beta1, beta2);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(counter), (unsigned long long)inc);
    DIP_CHECK_LAUNCH();
    return 0;
}
