// Boundary layout conversion, sigmoid head, weight repacking, fused Adam, Philox reg-noise and
// the depth-wise Lanczos down-sampler.  All HBM/latency-bound helpers around the MFMA kernels.
#include "dip_common.h"
#include "dip_group.h"

namespace {

// ---------------------------------------------------------------- layout / head
// tile transpose through LDS so both sides are coalesced: block handles 64 pixels x all channels
template <bool GRP = false>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src_, float* __restrict__ dst_,
                                                           int C, int HW, int Cs, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, src);
    DIP_GRP_PTR(float*, dst);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    // C is small at this boundary (<= 32 for every reference config); each lane walks the
    // channels of its pixel: reads are coalesced across lanes per channel plane, writes are
    // 16-B vectors.
    for (int c = 0; c < Cs; c += 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (c + e < C) ? src[(size_t)(c + e) * HW + p] : 0.f;
        *reinterpret_cast<f32x4*>(dst + (size_t)p * Cs + c) = v;
    }
}

// <= 32 channels (every notebook's net_input): transpose through LDS so that BOTH sides are coalesced -- the planes are
// read 256 pixels at a time, the [256 pixels][Cs] block leaves as one contiguous run of 16-byte stores (the per-lane
// walk above scatters its stores over 64 cache lines per instruction: 33 us for the 32 x 512^2 input, 2 TB/s)
template <bool GRP = false>
__global__ __launch_bounds__(256) void nchw_to_nhwc_lds_kernel(const float* __restrict__ src_, float* __restrict__ dst_,
                                                               int C, int HW, int Cs, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, src);
    DIP_GRP_PTR(float*, dst);
    __shared__ float t[256][33];
    const int p0 = blockIdx.x * 256, tid = threadIdx.x;
    const int np = min(256, HW - p0);
    if (tid < np) {
#pragma unroll 8
        for (int c = 0; c < C; ++c) t[tid][c] = src[(size_t)c * HW + p0 + tid];
        for (int c = C; c < Cs; ++c) t[tid][c] = 0.f;
    }
    __syncthreads();
    const int n4 = np * Cs / 4;
    for (int k = tid; k < n4; k += 256) {
        const int f = k * 4, px = f / Cs, c = f - px * Cs;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = t[px][c + e];
        *reinterpret_cast<f32x4*>(dst + (size_t)p0 * Cs + f) = v;
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int Cs, int accumulate) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < C; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * Cs + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < C) {
                float* o = dst + (size_t)(c + e) * HW + p;
                *o = accumulate ? (*o + v[e]) : v[e];
            }
    }
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ y, float* __restrict__ out, int C,
                                                       int HW, int Cs, int sigmoid) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < C; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)p * Cs + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < C) out[(size_t)(c + e) * HW + p] = sigmoid ? 1.f / (1.f + expf(-v[e])) : v[e];
    }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                       float* __restrict__ dy, int C, int HW, int Cs, int sigmoid) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < Cs; c += 4) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float g = 0.f;
            if (c + e < C) {
                g = gout[(size_t)(c + e) * HW + p];
                if (sigmoid) {
                    const float o = out[(size_t)(c + e) * HW + p];
                    g = g * ((1.f - o) * o);     // aten sigmoid_backward: grad * (1 - y) * y
                }
            }
            v[e] = g;
        }
        *reinterpret_cast<f32x4*>(dy + (size_t)p * Cs + c) = v;
    }
}

// ---------------------------------------------------------------- weight repack (one launch, all convs)
template <bool GRP = false>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ params_, float* __restrict__ packed_,
                                                           const DipPackRec* __restrict__ recs_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, params);
    DIP_GRP_PTR(float*, packed);
    DIP_GRP_PTR(const DipPackRec*, recs);
    const DipPackRec r = recs[blockIdx.y];
    const int KK = r.KS * r.KS;
    const int nf = KK * r.CinP4 * r.CoutP32;
    const int nd = r.dgrad_off >= 0 ? KK * r.CoutP4 * r.CinP32 : 0;
    const float* w = params + r.w_off;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += gridDim.x * 256) {
        if (i < nf) {
            // Wf[tap][c/4][o][c%4]
            const int c_lo = i & 3;
            const int o = (i >> 2) % r.CoutP32;
            const int rest = (i >> 2) / r.CoutP32;
            const int c4 = rest % (r.CinP4 >> 2);
            const int tap = rest / (r.CinP4 >> 2);
            const int c = c4 * 4 + c_lo;
            float v = 0.f;
            if (o < r.Cout && c < r.Cin) v = w[((size_t)o * r.Cin + c) * KK + tap];
            packed[r.fwd_off + i] = v;
        } else {
            // Wd[tap][o/4][c][o%4] = W[o][c][KK-1-tap]
            const int j = i - nf;
            const int o_lo = j & 3;
            const int c = (j >> 2) % r.CinP32;
            const int rest = (j >> 2) / r.CinP32;
            const int o4 = rest % (r.CoutP4 >> 2);
            const int tap = rest / (r.CoutP4 >> 2);
            const int o = o4 * 4 + o_lo;
            float v = 0.f;
            if (o < r.Cout && c < r.Cin) v = w[((size_t)o * r.Cin + c) * KK + (KK - 1 - tap)];
            packed[r.dgrad_off + j] = v;
        }
    }
}

// ---------------------------------------------------------------- Adam (torch 2.x _single_tensor_adam order)
// The update is evaluated with exactly the roundings of ATen's CPU kernels (the oracle):
//   exp_avg.lerp_(g, 1-b1)                 -> fma(w, g - m, m)                       (Lerp.h, vec::fmadd)
//   exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2) -> (v*b2) + ((1-b2)*g)*g   each rounded  (PointwiseOpsKernel.cpp)
//   denom = sqrt(v)/bc2_sqrt + eps         -> correctly rounded sqrt / div
//   p.addcdiv_(m, denom, -step_size)       -> p + ((-step_size)*m)/denom
// __fmul_rn / __fadd_rn / __fdiv_rn keep hipcc from contracting them into other FMAs.
// DEV: step_size / bc2_sqrt come from the DipIterState in device memory (graph-replayable).
template <bool DEV, bool GRP = false>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p_, const float* __restrict__ g_,
                                                   float* __restrict__ m_, float* __restrict__ v_, int64_t n, float w1,
                                                   float beta2, float omb2, float step_size, float bc2_sqrt,
                                                   float eps, const DipIterState* __restrict__ st_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(float*, p);
    DIP_GRP_PTR(const float*, g);
    DIP_GRP_PTR(float*, m);
    DIP_GRP_PTR(float*, v);
    DIP_GRP_PTR(const DipIterState*, st);
    if constexpr (DEV) {
        step_size = st->step_size;
        bc2_sqrt = st->bc2_sqrt;
    }
    const float neg_step = -step_size;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float gi = g[i];
        const float mi = fmaf(w1, __fsub_rn(gi, m[i]), m[i]);
        const float vi = __fadd_rn(__fmul_rn(v[i], beta2), __fmul_rn(__fmul_rn(omb2, gi), gi));
        const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), bc2_sqrt), eps);
        p[i] = __fadd_rn(p[i], __fdiv_rn(__fmul_rn(neg_step, mi), denom));
        m[i] = mi;
        v[i] = vi;
    }
}

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

// SEED_DEV: the seed, too, lives in device memory (offset_dev[1]): grouped fits with one stream per instance
template <bool GRP = false, bool SEED_DEV = false>
__global__ __launch_bounds__(256) void noise_axpy_kernel(const float* __restrict__ z_, float* __restrict__ out_,
                                                         int64_t n, float sigma, uint64_t seed, uint64_t offset,
                                                         const uint64_t* __restrict__ offset_dev_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, z);
    DIP_GRP_PTR(float*, out);
    DIP_GRP_PTR(const uint64_t*, offset_dev);
    if (offset_dev != nullptr) offset = *offset_dev;              // device-side stream position (graph replay)
    if constexpr (SEED_DEV) seed = offset_dev[1];
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;   // one Philox block = 4 normals
    const int64_t i0 = q * 4;
    if (i0 >= n) return;
    uint32_t c[4] = {(uint32_t)(q + offset), (uint32_t)((uint64_t)(q + offset) >> 32), 0u, 0u};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    float nrm[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
        const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float rad = sqrtf(-2.0f * __logf(u1));
        float s, co;
        __sincosf(6.28318530717958647692f * u2, &s, &co);
        nrm[2 * h] = rad * co;
        nrm[2 * h + 1] = rad * s;
    }
    if (i0 + 3 < n && ((reinterpret_cast<uintptr_t>(z + i0) | reinterpret_cast<uintptr_t>(out + i0)) & 15) == 0) {
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + i0);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(sigma, nrm[e], zv[e]);
        *reinterpret_cast<f32x4*>(out + i0) = o;
    } else {
        for (int e = 0; e < 4 && i0 + e < n; ++e) out[i0 + e] = fmaf(sigma, nrm[e], z[i0 + e]);
    }
}

// ---------------------------------------------------------------- depth-wise Lanczos down-sampler (NCHW)
__global__ __launch_bounds__(256) void lanczos_fwd_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                          float* __restrict__ y, int C, int H, int W, int k, int f,
                                                          int pad, int Ho, int Wo) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * Ho * Wo) return;
    const int ox = id % Wo, oy = (id / Wo) % Ho, c = id / (Wo * Ho);
    const float* xc = x + (size_t)c * H * W;
    float acc = 0.f;
    for (int i = 0; i < k; ++i) {
        const int sy = min(max(oy * f + i - pad, 0), H - 1);       // ReplicationPad2d
        for (int j = 0; j < k; ++j) {
            const int sx = min(max(ox * f + j - pad, 0), W - 1);
            acc = fmaf(taps[i * k + j], xc[(size_t)sy * W + sx], acc);
        }
    }
    y[id] = acc;
}

// gather form of the adjoint: gx[c][sy][sx] = sum over (oy,i),(ox,j) with clamp(oy*f+i-pad)==sy ...
__global__ __launch_bounds__(256) void lanczos_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ taps,
                                                          float* __restrict__ gx, int C, int H, int W, int k, int f,
                                                          int pad, int Ho, int Wo) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * H * W) return;
    const int sx = id % W, sy = (id / W) % H, c = id / (W * H);
    const float* g = gy + (size_t)c * Ho * Wo;
    // padded-domain rows that clamp onto sy: sy itself, plus everything above/below at the edges
    const int ylo = (sy == 0) ? -pad : sy, yhi = (sy == H - 1) ? H - 1 + pad : sy;
    const int xlo = (sx == 0) ? -pad : sx, xhi = (sx == W - 1) ? W - 1 + pad : sx;
    float acc = 0.f;
    for (int py = ylo; py <= yhi; ++py) {
        // oy*f + i - pad == py  ->  i = py + pad - oy*f in [0,k)
        const int t = py + pad;
        const int oy_hi = min(t / f, Ho - 1);
        const int oy_lo = max((t - k + 1 + f - 1) / f, 0);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const int i = t - oy * f;
            if (i < 0 || i >= k) continue;
            for (int px = xlo; px <= xhi; ++px) {
                const int u = px + pad;
                const int ox_hi = min(u / f, Wo - 1);
                const int ox_lo = max((u - k + 1 + f - 1) / f, 0);
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    const int j = u - ox * f;
                    if (j < 0 || j >= k) continue;
                    acc = fmaf(taps[i * k + j], g[(size_t)oy * Wo + ox], acc);
                }
            }
        }
    }
    gx[id] = acc;
}

// ---------------------------------------------------------------- dense down-sampler (NCHW): opt_over='down'
// The reference's Downsampler IS a dense Conv2d(n, n, k, stride=f) behind ReplicationPad2d; get_params('down', ...)
// hands its weight to the optimiser (utils/common_utils.py:44-46).  n is 3 (or 1) image planes, k = 4f or 6f: a few
// MFLOP per call, so these are plain gather kernels -- one thread per output element, fixed summation order.
__global__ __launch_bounds__(256) void down_dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int C, int H, int W, int k, int f, int pad, int Ho, int Wo) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * Ho * Wo) return;
    const int ox = id % Wo, oy = (id / Wo) % Ho, o = id / (Wo * Ho);
    float acc = bias != nullptr ? bias[o] : 0.f;
    for (int c = 0; c < C; ++c) {
        const float* xc = x + (size_t)c * H * W;
        const float* wc = w + ((size_t)o * C + c) * k * k;
        for (int i = 0; i < k; ++i) {
            const int sy = min(max(oy * f + i - pad, 0), H - 1);
            for (int j = 0; j < k; ++j) {
                const int sx = min(max(ox * f + j - pad, 0), W - 1);
                acc = fmaf(wc[i * k + j], xc[(size_t)sy * W + sx], acc);
            }
        }
    }
    y[id] = acc;
}

// gradient wrt the input: the gather form of lanczos_bwd_kernel with the sum over output planes added
__global__ __launch_bounds__(256) void down_dense_bwd_data_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                  float* __restrict__ gx, int C, int H, int W, int k,
                                                                  int f, int pad, int Ho, int Wo) {
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= C * H * W) return;
    const int sx = id % W, sy = (id / W) % H, c = id / (W * H);
    const int ylo = (sy == 0) ? -pad : sy, yhi = (sy == H - 1) ? H - 1 + pad : sy;
    const int xlo = (sx == 0) ? -pad : sx, xhi = (sx == W - 1) ? W - 1 + pad : sx;
    float acc = 0.f;
    for (int o = 0; o < C; ++o) {
        const float* g = gy + (size_t)o * Ho * Wo;
        const float* wc = w + ((size_t)o * C + c) * k * k;
        for (int py = ylo; py <= yhi; ++py) {
            const int t = py + pad;
            const int oy_hi = min(t / f, Ho - 1);
            const int oy_lo = max((t - k + 1 + f - 1) / f, 0);
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                const int i = t - oy * f;
                if (i < 0 || i >= k) continue;
                for (int px = xlo; px <= xhi; ++px) {
                    const int u = px + pad;
                    const int ox_hi = min(u / f, Wo - 1);
                    const int ox_lo = max((u - k + 1 + f - 1) / f, 0);
                    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                        const int j = u - ox * f;
                        if (j < 0 || j >= k) continue;
                        acc = fmaf(wc[i * k + j], g[(size_t)oy * Wo + ox], acc);
                    }
                }
            }
        }
    }
    gx[id] = acc;
}

// gradient wrt weight and bias: block (o, c, i, j) sums gy[o][p] * xpad[c][p*f + (i, j)] over the output pixels p
// (thread t takes p = t, t + 256, ...; fixed-order tree over the 256 partial sums); blocks >= C*C*k*k sum gy[o] alone
__global__ __launch_bounds__(256) void down_dense_bwd_weight_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                    float* __restrict__ dw, float* __restrict__ db, int C,
                                                                    int H, int W, int k, int f, int pad, int Ho, int Wo) {
    __shared__ float sh[256];
    const int nw = C * C * k * k;
    const int b = blockIdx.x;
    const bool is_bias = b >= nw;
    int o, c = 0, i = 0, j = 0;
    if (is_bias) {
        o = b - nw;
    } else {
        j = b % k; i = (b / k) % k; c = (b / (k * k)) % C; o = b / (k * k * C);
    }
    const float* g = gy + (size_t)o * Ho * Wo;
    const float* xc = x + (size_t)c * H * W;
    float acc = 0.f;
    for (int p = threadIdx.x; p < Ho * Wo; p += 256) {
        const int oy = p / Wo, ox = p - oy * Wo;
        float xv = 1.f;
        if (!is_bias) {
            const int sy = min(max(oy * f + i - pad, 0), H - 1), sx = min(max(ox * f + j - pad, 0), W - 1);
            xv = xc[(size_t)sy * W + sx];
        }
        acc = fmaf(g[p], xv, acc);
    }
    sh[threadIdx.x] = acc;
    for (int s = 128; s >= 1; s >>= 1) {
        __syncthreads();
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    }
    if (threadIdx.x == 0) {
        if (is_bias) { if (db != nullptr) db[o] = sh[0]; }
        else dw[b] = sh[0];
    }
}

}  // namespace

extern "C" int dip_nchw_to_nhwc(const float* src, float* dst, int C, int HW, int Cs, void* stream) {
    if (Cs <= 32 && (Cs & 3) == 0 && C <= Cs) {
        dip_launch_pair<DIP_FAM_MISC>(nchw_to_nhwc_lds_kernel<false>, nchw_to_nhwc_lds_kernel<true>, dim3(dip_cdiv(HW, 256)), dim3(256), 0,
                                      (hipStream_t)stream, src, dst, C, HW, Cs);
        DIP_CHECK_LAUNCH();
        return 0;
    }
    dip_launch_pair<DIP_FAM_MISC>(nchw_to_nhwc_kernel<false>, nchw_to_nhwc_kernel<true>, dim3(dip_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream,
                                  src, dst, C, HW, Cs);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_nhwc_to_nchw(const float* src, float* dst, int C, int HW, int Cs, int accumulate, void* stream) {
    dip_launch(nhwc_to_nchw_kernel, dim3(dip_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, Cs, accumulate);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_head_fwd(const float* y, float* out, int C, int HW, int Cs, int sigmoid, void* stream) {
    dip_launch(head_fwd_kernel, dim3(dip_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, y, out, C, HW, Cs, sigmoid);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_head_bwd(const float* gout, const float* out, float* dy, int C, int HW, int Cs, int sigmoid,
                            void* stream) {
    dip_launch(head_bwd_kernel, dim3(dip_cdiv(HW, 256)), dim3(256), 0, (hipStream_t)stream, gout, out, dy, C, HW, Cs, sigmoid);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_pack_weights(const float* params, float* packed, const DipPackRec* recs_dev, int nrec,
                                int max_elems, void* stream) {
    if (nrec <= 0) return 0;
    int gx = dip_cdiv(max_elems, 256 * 4);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    dip_launch_pair<DIP_FAM_MISC>(pack_weights_kernel<false>, pack_weights_kernel<true>, dim3(gx, nrec), dim3(256), 0, (hipStream_t)stream, params,
                                  packed, recs_dev);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1,
                             double beta2, double eps, int step, void* stream) {
    if (n <= 0) return 0;
    // scalar prep exactly as torch/optim/adam.py (_single_tensor_adam, python doubles)
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dip_launch_pair<DIP_FAM_LOSS>(adam_kernel<false>, adam_kernel<false, true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                                  n, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)step_size, (float)bc2_sqrt, (float)eps,
                                  (const DipIterState*)nullptr);
    DIP_CHECK_LAUNCH();
    return 0;
}

// same update with the step-dependent scalars read from a DipIterState (dip_adam_tick advances it)
extern "C" int dip_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, double beta1, double beta2,
                                 double eps, const DipIterState* st, void* stream) {
    if (n <= 0) return 0;
    if (st == nullptr) DIP_FAIL("adam_step_dev: iteration state is NULL");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dip_launch_pair<DIP_FAM_LOSS>(adam_kernel<true>, adam_kernel<true, true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                                  (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 0.f, 1.f, (float)eps, st);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_noise_axpy(const float* z, float* out, int64_t n, float sigma, uint64_t seed, uint64_t offset,
                              void* stream) {
    if (n <= 0) return 0;
    const int64_t quads = (n + 3) / 4;
    dip_launch_pair<DIP_FAM_LOSS>(noise_axpy_kernel<false>, noise_axpy_kernel<true>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                                  (hipStream_t)stream, z, out, n, sigma, seed, offset, (const uint64_t*)nullptr);
    DIP_CHECK_LAUNCH();
    return 0;
}

// Philox offset read from (and advanced in) device memory: every call continues the stream
extern "C" int dip_counter_add(uint64_t* counter, uint64_t inc, void* stream);
extern "C" int dip_noise_axpy_dev(const float* z, float* out, int64_t n, float sigma, uint64_t seed,
                                  uint64_t* offset_dev, void* stream) {
    if (n <= 0) return 0;
    if (offset_dev == nullptr) DIP_FAIL("noise_axpy_dev: offset pointer is NULL");
    const int64_t quads = (n + 3) / 4;
    dip_launch_pair<DIP_FAM_LOSS>(noise_axpy_kernel<false>, noise_axpy_kernel<true>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                                  (hipStream_t)stream, z, out, n, sigma, seed, (uint64_t)0, (const uint64_t*)offset_dev);
    DIP_CHECK_LAUNCH();
    return dip_counter_add(offset_dev, (uint64_t)quads, stream);
}

// the same with the seed in device memory next to the offset: state_dev = {offset, seed}.  Grouped fits (dip_group_begin)
// give every instance a stream of its own this way -- a by-value seed would be shared by all of them.
extern "C" int dip_noise_axpy_dev2(const float* z, float* out, int64_t n, float sigma, uint64_t* state_dev, void* stream) {
    if (n <= 0) return 0;
    if (state_dev == nullptr) DIP_FAIL("noise_axpy_dev2: state pointer is NULL");
    const int64_t quads = (n + 3) / 4;
    dip_launch_pair<DIP_FAM_LOSS>(noise_axpy_kernel<false, true>, noise_axpy_kernel<true, true>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                                  (hipStream_t)stream, z, out, n, sigma, (uint64_t)0, (uint64_t)0, (const uint64_t*)state_dev);
    DIP_CHECK_LAUNCH();
    return dip_counter_add(state_dev, (uint64_t)quads, stream);
}

extern "C" int dip_lanczos_down_fwd(const float* x, const float* taps, float* y, int C, int H, int W, int k,
                                    int factor, int pad, void* stream) {
    const int Ho = (H + 2 * pad - k) / factor + 1, Wo = (W + 2 * pad - k) / factor + 1;
    dip_launch(lanczos_fwd_kernel, dim3(dip_cdiv(C * Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       taps, y, C, H, W, k, factor, pad, Ho, Wo);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_lanczos_down_bwd(const float* gy, const float* taps, float* gx, int C, int H, int W, int k,
                                    int factor, int pad, void* stream) {
    const int Ho = (H + 2 * pad - k) / factor + 1, Wo = (W + 2 * pad - k) / factor + 1;
    dip_launch(lanczos_bwd_kernel, dim3(dip_cdiv(C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, gy,
                       taps, gx, C, H, W, k, factor, pad, Ho, Wo);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_down_dense_fwd(const float* x, const float* w, const float* bias, float* y, int C, int H, int W, int k,
                                  int factor, int pad, void* stream) {
    const int Ho = (H + 2 * pad - k) / factor + 1, Wo = (W + 2 * pad - k) / factor + 1;
    if (Ho < 1 || Wo < 1) DIP_FAIL("down_dense_fwd: empty output");
    dip_launch(down_dense_fwd_kernel, dim3(dip_cdiv(C * Ho * Wo, 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                       bias, y, C, H, W, k, factor, pad, Ho, Wo);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_down_dense_bwd_data(const float* gy, const float* w, float* gx, int C, int H, int W, int k, int factor,
                                       int pad, void* stream) {
    const int Ho = (H + 2 * pad - k) / factor + 1, Wo = (W + 2 * pad - k) / factor + 1;
    dip_launch(down_dense_bwd_data_kernel, dim3(dip_cdiv(C * H * W, 256)), dim3(256), 0, (hipStream_t)stream, gy,
                       w, gx, C, H, W, k, factor, pad, Ho, Wo);
    DIP_CHECK_LAUNCH();
    return 0;
}
extern "C" int dip_down_dense_bwd_weight(const float* gy, const float* x, float* dw, float* db, int C, int H, int W, int k,
                                         int factor, int pad, void* stream) {
    const int Ho = (H + 2 * pad - k) / factor + 1, Wo = (W + 2 * pad - k) / factor + 1;
    dip_launch(down_dense_bwd_weight_kernel, dim3(C * C * k * k + C), dim3(256), 0, (hipStream_t)stream, gy, x, dw,
                       db, C, H, W, k, factor, pad, Ho, Wo);
    DIP_CHECK_LAUNCH();
    return 0;
}
