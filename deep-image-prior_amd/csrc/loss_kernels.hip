// Loss head and device-side iteration state.
//
//  * dip_loss_head_fwd / dip_loss_head_bwd: the tail of the closure fused into two launches --
//    output conv (1x1, <= 4 channels) + nn.Sigmoid (models/skip.py:96-98 of the reference) +
//    optional mask multiply + torch.nn.MSELoss (denoising.ipynb:177,219; inpainting.ipynb:310:
//    mse(out * mask, img * mask), mean over ALL elements).  HBM-bound: the 128-channel activation
//    is read once (conv on the 4x4x1 MFMA, one lane per pixel); the scalar loss is reduced per lane
//    -> LDS tree -> one partial per block, and the last-arriving block sums the partials in a fixed
//    order (deterministic, no float atomics).
//  * DipIterState + dip_adam_tick / dip_adam_step_dev / dip_noise_axpy_dev: Adam's step count and
//    the Philox offset live in device memory, so one optimisation iteration is a STATIC launch list
//    and can be replayed as a hipGraph (nothing changes on the host between iterations).
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// One lane = one pixel.  The 1x1 conv towards <= 4 channels runs on v_mfma_f32_4x4x1_16b_f32 (16
// independent 4x4 outer products per instruction, A of block 0 broadcast with CBSZ = 4):
//   A[i][k] = w[o = i][c = k]  (lanes 0..3),   B[k][j] = act(u[pixel(lane)][c = k]),   D[i][lane] = y[o = i][pixel]
// so after Cin K steps every lane holds the 4 outputs of its own pixel: no cross-lane reduction for the
// conv, NCHW stores / target / mask loads are coalesced across the lanes, and the only reduction left
// is the scalar loss: per-lane sums -> LDS tree -> one partial per block -> loss_reduce_kernel.
template <bool GRP = false>
__global__ __launch_bounds__(256) void loss_head_fwd_kernel(const DipLossHeadDesc d_, const int ppb, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipLossHeadDesc, d);
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int oi = tid & 3;                                    // A row of this lane (only lanes 0..3 of a wave are read)
    const int cin4 = (d.Cin + 3) >> 2;
    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;
    float bias[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bias[o] = (d.bias != nullptr && o < d.Cout) ? d.bias[o] : 0.f;
    const float* wrow = d.w + (size_t)(oi < d.Cout ? oi : 0) * d.Cin;
    const bool wvalid = oi < d.Cout;

    const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, d.HW);
    float lsum = 0.f;
    for (int pb = p0; pb < p1; pb += 256) {
        const int p = pb + tid;
        const bool pv = p < p1;
        const float* up = d.u + (size_t)(pv ? p : p0) * d.Cu;
        f32x4 ac4[4];                                          // independent accumulator chains
#pragma unroll
        for (int e = 0; e < 4; ++e) ac4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int k4 = 0; k4 < cin4; ++k4) {
            f32x4 b = *reinterpret_cast<const f32x4*>(up + k4 * 4);
            f32x4 a;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = k4 * 4 + e;
                a[e] = (wvalid && c < d.Cin) ? wrow[c] : 0.f;
                if (has_tr) {
                    const float ta = c < d.Cin ? d.tr.a[c] : 0.f, tb = c < d.Cin ? d.tr.b[c] : 0.f;   // wave-uniform
                    b[e] = dip_act(fmaf(ta, b[e], tb), slope);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ac4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[e], b[e], ac4[e], 4, 0, 0);
        }
        const f32x4 acc = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
        if (pv) {
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
    // block tree (fixed order) -> one partial per block; loss_reduce_kernel sums the partials
    red[tid] = lsum;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    if (tid == 0) d.partials[blockIdx.x] = red[0];
}

// Coalesced variant for Cin/4 = NC4 a power of two <= 32 (every net of the notebooks: 128, 16 or 8 channels in front
// of the output conv): NC4 consecutive lanes share a pixel, each owns 4 channels -- a wave reads 64/NC4 whole pixels
// = one contiguous run per load instruction (the lane-per-pixel kernel above strides 4*Cu bytes between lanes and
// ran at 1.8 TB/s) -- computes its 4 x Cout partial products, and an xor-butterfly over the NC4 lanes (fixed order)
// leaves the sums in every lane; lane o of the group finishes output channel o.
template <int NC4, bool GRP = false>
__global__ __launch_bounds__(256) void loss_head_fwd_coal_kernel(const DipLossHeadDesc d_, const int ppb, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipLossHeadDesc, d);
    __shared__ float red[256];
    constexpr int PW = 64 / NC4;                 // pixels per wave and step
    constexpr int PB = 4 * PW;                   // ... per block and step
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cg = lane % NC4, psub = lane / NC4;
    const bool has_tr = d.tr.a != nullptr;
    const float slope = has_tr ? d.tr.slope : 1.f;
    const bool leaky = slope > 0.f;
    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[4][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = cg * 4 + e;
        if (has_tr && c < d.Cin) { ta[e] = d.tr.a[c]; tb[e] = d.tr.b[c]; }
#pragma unroll
        for (int o = 0; o < 4; ++o) w[o][e] = (o < d.Cout && c < d.Cin) ? d.w[(size_t)o * d.Cin + c] : 0.f;
    }
    // the lane group's sums are formed by a PACKED butterfly: the first two exchanges (distances NC4 / 2, NC4 / 4) also halve
    // the number of values a lane carries (4 -> 2 -> 1: a lane sends the half it will not finish), the others add one value:
    // log2(NC4) + 1 cross-lane moves per pixel step instead of 4 log2(NC4) -- with 32 lanes per pixel 6 instead of 20, which
    // had the LDS crossbar (ds_bpermute), not HBM, pace this kernel (48 us for 134 MB).  Output channel of a lane: 2 bA + bB.
    const bool bA = (cg & (NC4 / 2)) != 0, bB = (cg & (NC4 / 4)) != 0;
    const int myo = 2 * (bA ? 1 : 0) + (bB ? 1 : 0);
    const bool writer = (cg & (NC4 / 4 - 1)) == 0 && myo < d.Cout;
    const float mybias = (d.bias != nullptr && myo < d.Cout) ? d.bias[myo] : 0.f;
    const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, d.HW);
    float lsum = 0.f;
    constexpr int UN = 4;                        // independent loads in flight per thread
    for (int base = p0 + wave * PW; base < p1; base += UN * PB) {      // (wave-uniform trip count: the shuffles below)
        const int pb = base + psub;
        f32x4 u[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int p = pb + q * PB;
            u[q] = *reinterpret_cast<const f32x4*>(d.u + (size_t)(p < p1 ? p : p0) * d.Cu + cg * 4);
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int p = pb + q * PB;
            f32x4 v = u[q];
            if (leaky) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dip_act_leaky(fmaf(ta[e], v[e], tb[e]), slope);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dip_act(fmaf(ta[e], v[e], tb[e]), slope);
            }
            float part[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) part[o] = (w[o][0] * v[0] + w[o][1] * v[1]) + (w[o][2] * v[2] + w[o][3] * v[3]);
            const float k0 = (bA ? part[2] : part[0]) + __shfl_xor(bA ? part[0] : part[2], NC4 / 2);
            const float k1 = (bA ? part[3] : part[1]) + __shfl_xor(bA ? part[1] : part[3], NC4 / 2);
            float mine = (bB ? k1 : k0) + __shfl_xor(bB ? k0 : k1, NC4 / 4);
#pragma unroll
            for (int sft = NC4 / 8; sft >= 1; sft >>= 1) mine += __shfl_xor(mine, sft);
            if (p < p1 && writer) {
                float y = mine + mybias;
                if (d.sigmoid) y = sigmoidf_(y);
                d.out[(size_t)myo * d.HW + p] = y;
                const float t = d.target[(size_t)myo * d.HW + p];
                float a = y, b = t;
                if (d.mask != nullptr) {
                    const float m = d.mask[(size_t)(d.mask_c == 1 ? 0 : myo) * d.HW + p];
                    a = y * m;                                     // mse(out * mask, img * mask)
                    b = t * m;
                }
                const float df = a - b;
                lsum = fmaf(df, df, lsum);
            }
        }
    }
    red[tid] = lsum;
    __syncthreads();
    for (int s2 = 128; s2 >= 1; s2 >>= 1) {
        if (tid < s2) red[tid] += red[tid + s2];
        __syncthreads();
    }
    if (tid == 0) d.partials[blockIdx.x] = red[0];
}

// Fixed-order fp64 sum of the per-block partials -> the scalar loss.  A launch of its own on purpose: a
// "last-arriving block" ticket needs an agent-scope release fence in EVERY block, which on the multi-XCD
// MI355X writes back / invalidates L2 each time (the ticketed version of this head took 110 us instead
// of ~35: DESIGN.md, dead ends).
template <bool GRP = false>
__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ partials_, int n, double scale,
                                                          float* __restrict__ loss_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, partials);
    DIP_GRP_PTR(float*, loss);
    __shared__ double dred[256];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += (double)partials[i];
    dred[tid] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if (tid < st) dred[tid] += dred[tid + st];
        __syncthreads();
    }
    if (tid == 0) *loss = (float)(dred[0] * scale);
}

// dy[p][o] = gscale * 2/N * (out*m - t*m) * m * out*(1-out)       (NHWC, channel stride Cy; pad channels zero)
template <bool GRP = false>
__global__ __launch_bounds__(256) void loss_head_bwd_kernel(const DipLossHeadDesc d_, const float* __restrict__ gscale_,
                                                            float* __restrict__ dy_, const int Cy, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipLossHeadDesc, d);
    DIP_GRP_PTR(const float*, gscale);
    DIP_GRP_PTR(float*, dy);
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
template <bool GRP = false>
__global__  void adam_tick_kernel(DipIterState* st_, double lr, double beta1, double beta2, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(DipIterState*, st);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long step = st->step + 1ull;
    st->step = step;
    // scalar prep as torch/optim/adam.py (_single_tensor_adam), in double
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    st->step_size = (float)(lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
}

template <bool GRP = false>
__global__  void counter_add_kernel(unsigned long long* c_, unsigned long long inc, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(unsigned long long*, c);
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += inc;
}

}  // namespace

// the coalesced head needs a lane per output channel inside a pixel's lane group
#define NC4_MIN_CHECK(cout, nc4) ((nc4) < 4 || (cout) > (nc4))
static int head_ppb(int HW) {
    int ppb = dip_round_up(dip_cdiv(HW, 1024), 256);      // ~1024 blocks, whole 256-pixel strips
    return ppb < 256 ? 256 : ppb;
}

extern "C" int dip_loss_head_nblk(int HW, int Cin) {
    (void)Cin;
    return dip_cdiv(HW, head_ppb(HW));
}

extern "C" int dip_loss_head_fwd(const DipLossHeadDesc* dp, void* stream) {
    const DipLossHeadDesc& d = *dp;
    if (d.Cout < 1 || d.Cout > 4) DIP_FAIL("loss_head: 1..4 output channels");
    if (d.Cin < 1 || (d.Cu & 3) || dip_round_up(d.Cin, 4) > d.Cu) DIP_FAIL("loss_head: channel stride must be a multiple of 4 covering Cin");
    if (d.mask != nullptr && d.mask_c != 1 && d.mask_c != d.Cout) DIP_FAIL("loss_head: mask must have 1 or Cout channels");
    const int ppb = head_ppb(d.HW);
    const int nblk = dip_cdiv(d.HW, ppb);
    if (nblk != d.nblk) DIP_FAIL("loss_head: nblk must come from dip_loss_head_nblk");
    const int nc4 = (d.Cin + 3) / 4;
    static const bool no_coal = getenv("DIP_LOSS_HEAD_NO_COAL") != nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (no_coal || nc4 > 32 || (nc4 & (nc4 - 1)) != 0 || (NC4_MIN_CHECK(d.Cout, nc4))) {
        dip_launch_pair<DIP_FAM_LOSS>(loss_head_fwd_kernel<false>, loss_head_fwd_kernel<true>, dim3(nblk), dim3(256), 0, st, d, ppb);
    } else {
        switch (nc4) {
            case 32: dip_launch_pair<DIP_FAM_LOSS>(loss_head_fwd_coal_kernel<32>, loss_head_fwd_coal_kernel<32, true>, dim3(nblk), dim3(256), 0, st, d, ppb); break;
            case 16: dip_launch_pair<DIP_FAM_LOSS>(loss_head_fwd_coal_kernel<16>, loss_head_fwd_coal_kernel<16, true>, dim3(nblk), dim3(256), 0, st, d, ppb); break;
            case 8: dip_launch_pair<DIP_FAM_LOSS>(loss_head_fwd_coal_kernel<8>, loss_head_fwd_coal_kernel<8, true>, dim3(nblk), dim3(256), 0, st, d, ppb); break;
            default: dip_launch_pair<DIP_FAM_LOSS>(loss_head_fwd_coal_kernel<4>, loss_head_fwd_coal_kernel<4, true>, dim3(nblk), dim3(256), 0, st, d, ppb); break;
        }
    }
    DIP_CHECK_LAUNCH();
    dip_launch_pair<DIP_FAM_LOSS>(loss_reduce_kernel<false>, loss_reduce_kernel<true>, dim3(1), dim3(256), 0, (hipStream_t)stream,
                                  (const float*)d.partials, nblk, 1.0 / ((double)d.Cout * (double)d.HW), d.loss);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_loss_head_bwd(const DipLossHeadDesc* dp, const float* gscale, float* dy, int Cy, void* stream) {
    const DipLossHeadDesc& d = *dp;
    if (d.Cout < 1 || d.Cout > 4 || (Cy & 3) || Cy < d.Cout) DIP_FAIL("loss_head_bwd: bad channel counts");
    dip_launch_pair<DIP_FAM_LOSS>(loss_head_bwd_kernel<false>, loss_head_bwd_kernel<true>, dim3(dip_cdiv(d.HW, 256)), dim3(256), 0, (hipStream_t)stream,
                                  d, gscale, dy, Cy);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_adam_tick(DipIterState* st, double lr, double beta1, double beta2, void* stream) {
    dip_launch_pair<DIP_FAM_LOSS>(adam_tick_kernel<false>, adam_tick_kernel<true>, dim3(1), dim3(1), 0, (hipStream_t)stream, st, lr, beta1, beta2);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_counter_add(uint64_t* counter, uint64_t inc, void* stream) {
    dip_launch_pair<DIP_FAM_LOSS>(counter_add_kernel<false>, counter_add_kernel<true>, dim3(1), dim3(1), 0, (hipStream_t)stream,
                                  reinterpret_cast<unsigned long long*>(counter), (unsigned long long)inc);
    DIP_CHECK_LAUNCH();
    return 0;
}
