// 2x upsample (bilinear align_corners=False / nearest) + channel concat + BatchNorm partial
// statistics (forward), and the adjoint gather fused with LeakyReLU backward + BatchNorm backward
// phase 1 (backward).  HBM-bound, float4 per lane, NHWC.
#include "dip_common.h"
#include "dip_group.h"
#include "bn_ticket.h"
#include "dip_gradsrc.h"
#include <stdlib.h>

namespace {


// PyTorch upsample_bilinear2d source index, align_corners=False, scale 0.5 (src per dst)
__device__ __forceinline__ void bil_src(int dst, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float real = 0.5f * ((float)dst + 0.5f) - 0.5f;
    if (real < 0.f) real = 0.f;
    i0 = (int)real;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = real - (float)i0;
    l0 = 1.f - l1;
}

struct RowLayout {
    int nc4, rpi, prow, cg;
    bool active;
};
__device__ __forceinline__ RowLayout row_layout(int C) {
    RowLayout L;
    L.nc4 = (C + 3) >> 2;
    L.rpi = 256 / L.nc4;
    if (L.rpi < 1) L.rpi = 1;
    L.prow = threadIdx.x / L.nc4;
    L.cg = threadIdx.x - L.prow * L.nc4;
    L.active = (int)threadIdx.x < L.rpi * L.nc4;
    return L;
}

// partial row of this block -> d.stats; with fin.state != NULL the last block to arrive finalises the BatchNorm (bn_ticket.h)
__device__ __forceinline__ void upcat_rows_out(const DipUpcatDesc& d, const DipBnFin& fin, const RowLayout& L, float n,
                                               const f32x4& mean, const f32x4& M2, int C, double* shd, unsigned* flag) {
    if (L.active && L.prow == 0) {
        float* o = d.stats + (size_t)blockIdx.x * 3 * d.Cs_cat + L.cg * 4;
        if (fin.state != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dip_st_sc1(o + e, n);
                dip_st_sc1(o + d.Cs_cat + e, mean[e]);
                dip_st_sc1(o + 2 * d.Cs_cat + e, M2[e]);
            }
        } else {
            st4(o, f32x4{n, n, n, n});
            st4(o + d.Cs_cat, mean);
            st4(o + 2 * d.Cs_cat, M2);
        }
    }
    if (fin.state != nullptr) {
        if (dip_ticket_last(fin.ticket, gridDim.x, flag)) {
            dip_bn_fin_rows(d.stats, gridDim.x, d.Cs_cat, 0, C, fin, shd);
            dip_ticket_reset(fin.ticket);
        }
    }
}

// One thread = 4 channels of a 2x2 block of output pixels (low-resolution pixel (i, j) -> outputs (2i..2i+1, 2j..2j+1)):
// the 3x3 low-resolution neighbourhood is loaded and put through the producer's BatchNorm+activation ONCE (2.25
// transforms and loads per output instead of 4), the column blends are shared by the two output rows.  Scale-2
// bilinear weights (align_corners = False): odd outputs (0.75, 0.25) on (i, i+1), even outputs (0.25, 0.75) on
// (i-1, i), except output 0 = (1, 0) -- exactly upsample_bilinear2d's lambdas (bil_src above).
template <bool GRP = false>
__global__ __launch_bounds__(256) void upcat_fwd_kernel(const DipUpcatDesc d_, int qpb, const DipBnFin fin_, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipUpcatDesc, d);
    DIP_GRP_DESC(DipBnFin, fin);
    __shared__ __attribute__((aligned(16))) double shd[DIP_TICKET_SH_DOUBLES];      // float trees, then the fp64 finalisation
    __shared__ unsigned flag;
    float* sh = reinterpret_cast<float*>(shd);
    const int C = d.ns + d.nd;
    const RowLayout L = row_layout(C);
    // shifted sums per thread, 4 channels
    f32x4 cnt = f32x4{0.f, 0.f, 0.f, 0.f}, K = cnt, s1 = cnt, s2 = cnt;
    float n = 0.f;
    if (L.active) {
        const int ch = L.cg * 4;
        // low-resolution size: (H+1)/2 -- for an odd size the x2 up-sampled tensor is one row / column larger
        // than the skip branch and Concat's centre crop (models/common.py:29-37, offset (1)//2 = 0) drops its last one
        const int Hl = (d.H + 1) >> 1, Wl = (d.W + 1) >> 1;
        const int nq = Hl * Wl;
        const int q0 = blockIdx.x * qpb, q1 = min(q0 + qpb, nq);
        // this thread's channel group never changes: its BatchNorm+activation coefficients are loaded once
        const bool skip_side = ch < d.ns;
        const DipTransform& tt = skip_side ? d.ts : d.td;
        const int tch = skip_side ? ch : ch - d.ns;
        const bool has_t = tt.a != nullptr;
        const f32x4 tA = has_t ? ld4(tt.a + tch) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 tB = has_t ? ld4(tt.b + tch) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float tS = has_t ? tt.slope : 1.f;
        const bool leaky = tS > 0.f;                 // (block-uniform: the branch is outside the element loops)
        auto trr = [&](f32x4 x) {
            f32x4 o;
            if (leaky) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = dip_act_leaky(fmaf(tA[e], x[e], tB[e]), tS);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = dip_act(fmaf(tA[e], x[e], tB[e]), tS);
            }
            return o;
        };
        auto emit = [&](size_t p, const f32x4& v) {
            st4(d.cat + p * d.Cs_cat + ch, v);
            if (n == 0.f) K = v;
            n += 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[e] - K[e];
                s1[e] += dv;
                s2[e] += dv * dv;
            }
        };
        for (int q = q0 + L.prow; q < q1; q += L.rpi) {
            const int i = q / Wl, j = q - i * Wl;
            const size_t p00 = (size_t)(2 * i) * d.W + 2 * j;
            const bool r1 = 2 * i + 1 < d.H, c1 = 2 * j + 1 < d.W;     // second row / column of the block exists
            if (skip_side) {
                const float* sp = d.s + p00 * d.Cs_s + ch;
                // (clamped addresses: every load is issued, the ragged ones are not emitted)
                const f32x4 a0 = ld4(sp), a1 = ld4(sp + (c1 ? d.Cs_s : 0)), a2 = ld4(sp + (r1 ? (size_t)d.W * d.Cs_s : 0)),
                            a3 = ld4(sp + (r1 ? (size_t)d.W * d.Cs_s : 0) + (c1 ? d.Cs_s : 0));
                emit(p00, trr(a0));
                if (c1) emit(p00 + 1, trr(a1));
                if (r1) emit(p00 + d.W, trr(a2));
                if (r1 && c1) emit(p00 + d.W + 1, trr(a3));
            } else if (d.mode == DIP_UP_NEAREST) {
                const f32x4 v = trr(ld4(d.d + (size_t)q * d.Cs_d + (ch - d.ns)));
                emit(p00, v);
                if (c1) emit(p00 + 1, v);
                if (r1) emit(p00 + d.W, v);
                if (r1 && c1) emit(p00 + d.W + 1, v);
            } else {
                const int cd = ch - d.ns;
                const int rr[3] = {max(i - 1, 0), i, min(i + 1, Hl - 1)};
                const int cc[3] = {max(j - 1, 0), j, min(j + 1, Wl - 1)};
                f32x4 t[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) t[a][b] = ld4(d.d + ((size_t)rr[a] * Wl + cc[b]) * d.Cs_d + cd);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) t[a][b] = trr(t[a][b]);
                const float le0 = i > 0 ? 0.25f : 1.f, le1 = i > 0 ? 0.75f : 0.f;
                const float ce0 = j > 0 ? 0.25f : 1.f, ce1 = j > 0 ? 0.75f : 0.f;
                f32x4 E[3], O[3];                    // column blends of the three low-resolution rows
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        E[a][e] = ce0 * t[a][0][e] + ce1 * t[a][1][e];
                        O[a][e] = 0.75f * t[a][1][e] + 0.25f * t[a][2][e];
                    }
                f32x4 v00, v01, v10, v11;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v00[e] = le0 * E[0][e] + le1 * E[1][e];
                    v01[e] = le0 * O[0][e] + le1 * O[1][e];
                    v10[e] = 0.75f * E[1][e] + 0.25f * E[2][e];
                    v11[e] = 0.75f * O[1][e] + 0.25f * O[2][e];
                }
                emit(p00, v00);
                if (c1) emit(p00 + 1, v01);
                if (r1) emit(p00 + d.W, v10);
                if (r1 && c1) emit(p00 + d.W + 1, v11);
            }
        }
    }
    // per-thread (n, mean, M2) -> LDS -> Chan-combine over prow
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, M2 = mean;
    if (n > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = K[e] + s1[e] / n;
            M2[e] = s2[e] - s1[e] * s1[e] / n;
        }
    }
    dip_tree_chan4(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, n, mean, M2);
    upcat_rows_out(d, fin, L, n, mean, M2, C, shd, &flag);
}

// General centre crop (models/common.py:29-37): one thread = 4 channels of ONE output pixel (r, c); the skip branch is
// read at (r + os_y, c + os_x) of its [Hs][Ws] tensor, the deeper branch at the up-sampled coordinate (r + od_y, c + od_x)
// of its [2*Hd][2*Wd] image (upsample_bilinear2d's own source-index rule, bil_src, or nearest).  Used only for the
// geometries the 2x2-block kernel above does not cover (pooling nets / skip-less scales at non-divisible sizes).
template <bool GRP = false>
__global__ __launch_bounds__(256) void upcat_fwd_crop_kernel(const DipUpcatDesc d_, int ppb, const DipBnFin fin_, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipUpcatDesc, d);
    DIP_GRP_DESC(DipBnFin, fin);
    __shared__ __attribute__((aligned(16))) double shd[DIP_TICKET_SH_DOUBLES];
    __shared__ unsigned flag;
    float* sh = reinterpret_cast<float*>(shd);
    const int C = d.ns + d.nd;
    const RowLayout L = row_layout(C);
    f32x4 K = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = K, s2 = K;
    float n = 0.f;
    if (L.active) {
        const int ch = L.cg * 4;
        const bool skip_side = ch < d.ns;
        const DipTransform& tt = skip_side ? d.ts : d.td;
        const int tch = skip_side ? ch : ch - d.ns;
        const bool has_t = tt.a != nullptr;
        const f32x4 tA = has_t ? ld4(tt.a + tch) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 tB = has_t ? ld4(tt.b + tch) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float tS = has_t ? tt.slope : 1.f;
        auto trr = [&](f32x4 x) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = dip_act(fmaf(tA[e], x[e], tB[e]), tS);
            return o;
        };
        const int npix = d.H * d.W;
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        for (int p = p0 + L.prow; p < p1; p += L.rpi) {
            const int r = p / d.W, c = p - r * d.W;
            f32x4 v;
            if (skip_side) {
                v = trr(ld4(d.s + ((size_t)(r + d.os_y) * d.Ws + (c + d.os_x)) * d.Cs_s + ch));
            } else if (d.mode == DIP_UP_NEAREST) {
                v = trr(ld4(d.d + ((size_t)((r + d.od_y) >> 1) * d.Wd + ((c + d.od_x) >> 1)) * d.Cs_d + (ch - d.ns)));
            } else {
                int i0, i1, j0, j1;
                float li0, li1, lj0, lj1;
                bil_src(r + d.od_y, d.Hd, i0, i1, li0, li1);
                bil_src(c + d.od_x, d.Wd, j0, j1, lj0, lj1);
                const float* base = d.d + (ch - d.ns);
                const f32x4 t00 = trr(ld4(base + ((size_t)i0 * d.Wd + j0) * d.Cs_d)), t01 = trr(ld4(base + ((size_t)i0 * d.Wd + j1) * d.Cs_d)),
                            t10 = trr(ld4(base + ((size_t)i1 * d.Wd + j0) * d.Cs_d)), t11 = trr(ld4(base + ((size_t)i1 * d.Wd + j1) * d.Cs_d));
#pragma unroll
                for (int e = 0; e < 4; ++e)      // ATen's order: rows blended from column blends
                    v[e] = li0 * (lj0 * t00[e] + lj1 * t01[e]) + li1 * (lj0 * t10[e] + lj1 * t11[e]);
            }
            st4(d.cat + (size_t)p * d.Cs_cat + ch, v);
            if (n == 0.f) K = v;
            n += 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[e] - K[e];
                s1[e] += dv;
                s2[e] += dv * dv;
            }
        }
    }
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, M2 = mean;
    if (n > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = K[e] + s1[e] / n;
            M2[e] = s2[e] - s1[e] * s1[e] / n;
        }
    }
    dip_tree_chan4(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, n, mean, M2);
    upcat_rows_out(d, fin, L, n, mean, M2, C, shd, &flag);
}

// low-res pixel (i,j): du = sum over the <=4x4 high-res pixels whose interpolation touches it.  The deeper branch is
// [Hl][Wl]; the gradient dcat is [H][W] and covers rows ody..ody+H-1, columns odx..odx+W-1 of the [2*Hl][2*Wl] up-sampled
// image (Concat's centre crop; default geometry: Hl = (H+1)/2, offsets 0).
template <bool GRP = false>
__global__ __launch_bounds__(256) void upsample_bwd_stats_kernel(const float* __restrict__ dcat_, int Cs_cat, int choff,
                                                                 int H, int W, int Hl, int Wl, int ody, int odx, int mode,
                                                                 const float* __restrict__ y_,
                                                                 int Cy, int C, const float* __restrict__ state_, int Cs,
                                                                 float slope, float* dz_, int Cdz, float* partials_,
                                                                 int ppb, const DipBnbFin fin_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, dcat);
    DIP_GRP_PTR(const float*, y);
    DIP_GRP_PTR(const float*, state);
    DIP_GRP_PTR(float*, dz);
    DIP_GRP_PTR(float*, partials);
    DIP_GRP_DESC(DipBnbFin, fin);
    __shared__ __attribute__((aligned(16))) double shd[256 * 8];
    __shared__ unsigned flag;
    float* sh = reinterpret_cast<float*>(shd);
    const RowLayout L = row_layout(C);
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (L.active) {
        const int ch = L.cg * 4;
        const f32x4 mean = ld4(state + ch), rstd = ld4(state + Cs + ch), a = ld4(state + 2 * Cs + ch),
                    b = ld4(state + 3 * Cs + ch);
        const int npix = Hl * Wl;
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        for (int p = p0 + L.prow; p < p1; p += L.rpi) {
            const int i = p / Wl, j = p - i * Wl;
            const f32x4 du = up_adj_du4(dcat + choff + ch, Cs_cat, H, W, Hl, Wl, ody, odx, mode, i, j);
            const f32x4 yv = ld4(y + (size_t)p * Cy + ch);
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaf(a[e], yv[e], b[e]);
                g[e] = dip_mul_rn(du[e], dip_act_grad(z, slope));      // (rounded product: never contracted into the sums)
                const float xh = (yv[e] - mean[e]) * rstd[e];
                s1[e] += g[e];
                s2[e] += g[e] * xh;
            }
            st4(dz + (size_t)p * Cdz + ch, g);
        }
    }
    dip_tree_sum8(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, s1, s2);
    if (L.active && L.prow == 0) {
        float* o = partials + (size_t)blockIdx.x * 2 * Cs + L.cg * 4;
        if (fin.coef != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { dip_st_sc1(o + e, s1[e]); dip_st_sc1(o + Cs + e, s2[e]); }
        } else {
            st4(o, s1);
            st4(o + Cs, s2);
        }
    }
    if (fin.coef != nullptr) {      // the last block to arrive reduces the rows: dgamma, dbeta, k1, k2 (bn_ticket.h)
        if (dip_ticket_last(fin.ticket, gridDim.x, &flag)) {
            dip_bnb_fin_rows(partials, gridDim.x, Cs, 0, C, fin, shd);
            dip_ticket_reset(fin.ticket);
        }
    }
}

// ------------------------------------------------------------------------------------------
// nn.AvgPool2d(2, 2) after a stride-1 conv (conv(..., downsample_mode='avg'), reference
// models/common.py:101-104) + the {count, mean, M2} partials of the BatchNorm that follows;
// the adjoint spreads dy/4 over the 2x2 window.
// ------------------------------------------------------------------------------------------
template <bool MAXP>       // false: nn.AvgPool2d(2, 2), true: nn.MaxPool2d(2, 2)
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x, int W, int Cx, int C,
                                                           float* __restrict__ y, int Hl, int Wl, int Cy, float* stats,
                                                           int ppb) {
    __shared__ __attribute__((aligned(16))) float sh[256 * 12];
    const RowLayout L = row_layout(C);
    f32x4 K = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = K, s2 = K;
    float n = 0.f;
    if (L.active) {
        const int ch = L.cg * 4;
        const int npix = Hl * Wl;
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        for (int p = p0 + L.prow; p < p1; p += L.rpi) {
            const int r = p / Wl, c = p - r * Wl;
            const float* q = x + ((size_t)(2 * r) * W + 2 * c) * Cx + ch;
            const f32x4 v00 = ld4(q), v01 = ld4(q + Cx), v10 = ld4(q + (size_t)W * Cx), v11 = ld4(q + (size_t)W * Cx + Cx);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = MAXP ? fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]))
                            : ((v00[e] + v01[e]) + (v10[e] + v11[e])) * 0.25f;
            st4(y + (size_t)p * Cy + ch, v);
            if (n == 0.f) K = v;
            n += 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dv = v[e] - K[e];
                s1[e] += dv;
                s2[e] += dv * dv;
            }
        }
    }
    if (stats == nullptr) return;
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, M2 = mean;
    if (n > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = K[e] + s1[e] / n;
            M2[e] = s2[e] - s1[e] * s1[e] / n;
        }
    }
    dip_tree_chan4(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, n, mean, M2);
    if (L.active && L.prow == 0) {
        float* o = stats + (size_t)blockIdx.x * 3 * Cy + L.cg * 4;
        st4(o, f32x4{n, n, n, n});
        st4(o + Cy, mean);
        st4(o + 2 * Cy, M2);
    }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy, int Wl, int Cdy, int C,
                                                           float* __restrict__ dx, int H, int W, int Cdx) {
    const int nc4 = (C + 3) >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * W * nc4) return;
    const int cg = (int)(i % nc4);
    const long long p = i / nc4;
    const int r = (int)(p / W), c = (int)(p - (long long)r * W);
    f32x4 g = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((r >> 1) < (H >> 1) && (c >> 1) < Wl) {
        g = ld4(dy + ((size_t)(r >> 1) * Wl + (c >> 1)) * Cdy + cg * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= 0.25f;
    }
    st4(dx + (size_t)p * Cdx + cg * 4, g);
}

// adjoint of nn.MaxPool2d(2, 2): dy goes to the FIRST maximal element of the 2x2 window in scan order
// (ATen max_pool2d keeps the running maximum on `val > maxval`), every other position gets 0; the
// arg-max is recomputed from the pooled layer's input x instead of being stored by the forward
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           int Wl, int Cdy, int Cx, int C, float* __restrict__ dx, int H,
                                                           int W, int Cdx) {
    const int nc4 = (C + 3) >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int Hl = H >> 1;
    if (i >= (long long)Hl * Wl * nc4) return;
    const int cg = (int)(i % nc4);
    const long long p = i / nc4;
    const int r = (int)(p / Wl), c = (int)(p - (long long)r * Wl);
    const float* q = x + ((size_t)(2 * r) * W + 2 * c) * Cx + cg * 4;
    const f32x4 v[4] = {ld4(q), ld4(q + Cx), ld4(q + (size_t)W * Cx), ld4(q + (size_t)W * Cx + Cx)};
    const f32x4 g = ld4(dy + (size_t)p * Cdy + cg * 4);
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int arg = 0;
        float m = v[0][e];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (v[k][e] > m) { m = v[k][e]; arg = k; }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k][e] = (k == arg) ? g[e] : 0.f;
    }
    float* d0 = dx + ((size_t)(2 * r) * W + 2 * c) * Cdx + cg * 4;
    st4(d0, o[0]);
    st4(d0 + Cdx, o[1]);
    st4(d0 + (size_t)W * Cdx, o[2]);
    st4(d0 + (size_t)W * Cdx + Cdx, o[3]);
    // a floored odd border row / column gets zeros (written by the threads of the last window row / column)
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((W & 1) && c == Wl - 1) {
        st4(dx + ((size_t)(2 * r) * W + W - 1) * Cdx + cg * 4, z);
        st4(dx + ((size_t)(2 * r + 1) * W + W - 1) * Cdx + cg * 4, z);
    }
    if ((H & 1) && r == Hl - 1) {
        st4(dx + ((size_t)(H - 1) * W + 2 * c) * Cdx + cg * 4, z);
        st4(dx + ((size_t)(H - 1) * W + 2 * c + 1) * Cdx + cg * 4, z);
        if ((W & 1) && c == Wl - 1) st4(dx + ((size_t)(H - 1) * W + W - 1) * Cdx + cg * 4, z);
    }
}

int pixels_per_block(int npix, int C, int* nblk) {
    const int nc4 = (C + 3) / 4;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    int ppb = dip_cdiv(npix, 1024);
    if (ppb < rpi * 2) ppb = rpi * 2;      // small tensors: few sequential pixels per thread (latency-bound)
    *nblk = dip_cdiv(npix, ppb);
    return ppb;
}

}  // namespace

extern "C" int dip_upcat_nblk(int H, int W, int C) {
    int nblk;
    pixels_per_block(H * W, C, &nblk);
    return nblk;
}

// rows a launch may finalise itself: the last block reads them all through L2-bypassing loads
static constexpr int FIN_MAX_ROWS = 256;

static int upcat_fwd_impl(const DipUpcatDesc* d, const DipBnFin* finp, void* stream) {
    const int C = d->ns + d->nd;
    DipBnFin fin = {};
    if (finp != nullptr && finp->state != nullptr) {
        fin = *finp;
        if (C > 256 || fin.ticket == nullptr || fin.gamma == nullptr || fin.beta == nullptr || fin.C != C)
            DIP_FAIL("upcat_fwd_fin: needs <= 256 channels, ticket, gamma, beta, C = ns + nd");
        if (d->nblk > FIN_MAX_ROWS) DIP_FAIL("upcat_fwd_fin: too many partial rows (dip_fin_rows_ok)");
    }
    if ((d->ns & 3) || (d->nd & 3)) DIP_FAIL("upcat_fwd: channel counts must be multiples of 4");
    if (C > 1024) DIP_FAIL("upcat_fwd: C > 1024 unsupported");
    int nb;
    const int ppb = pixels_per_block(d->H * d->W, C, &nb);
    if (nb != d->nblk) DIP_FAIL("upcat_fwd: nblk mismatch (use dip_upcat_nblk)");
    const bool general = d->Hs > 0 || d->Hd > 0;
    if (general) {
        DipUpcatDesc g = *d;
        if (g.Hs == 0) { g.Hs = g.H; g.Ws = g.W; g.os_y = g.os_x = 0; }
        if (g.Hd == 0) { g.Hd = (g.H + 1) / 2; g.Wd = (g.W + 1) / 2; g.od_y = g.od_x = 0; }
        if (g.os_y < 0 || g.os_x < 0 || g.os_y + g.H > g.Hs || g.os_x + g.W > g.Ws || g.od_y < 0 || g.od_x < 0 ||
            g.od_y + g.H > 2 * g.Hd || g.od_x + g.W > 2 * g.Wd)
            DIP_FAIL("upcat_fwd: crop window outside a branch");
        const bool dflt = g.Hs == g.H && g.Ws == g.W && g.os_y == 0 && g.os_x == 0 && g.Hd == (g.H + 1) / 2 &&
                          g.Wd == (g.W + 1) / 2 && g.od_y == 0 && g.od_x == 0;
        if (!dflt) {
            dip_launch_pair<DIP_FAM_UPCAT>(upcat_fwd_crop_kernel<false>, upcat_fwd_crop_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, ppb,
                                           fin);
            DIP_CHECK_LAUNCH();
            return 0;
        }
    }
    const int qpb = dip_cdiv(((d->H + 1) / 2) * ((d->W + 1) / 2), nb);       // 2x2 output blocks per workgroup
    dip_launch_pair<DIP_FAM_UPCAT>(upcat_fwd_kernel<false>, upcat_fwd_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *d, qpb, fin);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_upcat_fwd(const DipUpcatDesc* d, void* stream) { return upcat_fwd_impl(d, nullptr, stream); }
extern "C" int dip_upcat_fwd_fin(const DipUpcatDesc* d, const DipBnFin* fin, void* stream) {
    return upcat_fwd_impl(d, fin, stream);
}
// 1 when a launch that writes `rows` partial rows of C channels may finalise them itself (DipBnFin / DipBnbFin)
extern "C" int dip_fin_rows_ok(int rows, int C) {
    static const bool off = getenv("DIP_NO_TICKET_FIN") != nullptr;
    return (!off && rows <= FIN_MAX_ROWS && C <= 256) ? 1 : 0;
}

static int pool2_fwd(bool maxp, const float* x, int H, int W, int Cx, int C, float* y, int Cy, float* stats, int nblk,
                     void* stream) {
    if (C > 1024) DIP_FAIL("pool2_fwd: C > 1024 unsupported");
    if ((Cx & 3) || (Cy & 3)) DIP_FAIL("pool2_fwd: channel strides must be multiples of 4");
    int nb;
    const int ppb = pixels_per_block((H / 2) * (W / 2), C, &nb);
    if (stats != nullptr && nb != nblk) DIP_FAIL("pool2_fwd: nblk mismatch (use dip_upcat_nblk(H/2, W/2, C))");
    if (maxp)
        dip_launch(avgpool2_fwd_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, W, Cx, C, y, H / 2,
                           W / 2, Cy, stats, ppb);
    else
        dip_launch(avgpool2_fwd_kernel<false>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, W, Cx, C, y, H / 2,
                           W / 2, Cy, stats, ppb);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_avgpool2_fwd(const float* x, int H, int W, int Cx, int C, float* y, int Cy, float* stats, int nblk,
                                void* stream) {
    return pool2_fwd(false, x, H, W, Cx, C, y, Cy, stats, nblk, stream);
}

extern "C" int dip_maxpool2_fwd(const float* x, int H, int W, int Cx, int C, float* y, int Cy, float* stats, int nblk,
                                void* stream) {
    return pool2_fwd(true, x, H, W, Cx, C, y, Cy, stats, nblk, stream);
}

extern "C" int dip_maxpool2_bwd(const float* dy, const float* x, int H, int W, int Cdy, int Cx, int C, float* dx, int Cdx,
                                void* stream) {
    if ((Cdx & 3) || (Cdy & 3) || (Cx & 3)) DIP_FAIL("maxpool2_bwd: channel strides must be multiples of 4");
    const long long n = (long long)(H / 2) * (W / 2) * ((C + 3) / 4);
    if (n <= 0) return 0;
    dip_launch(maxpool2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x,
                       W / 2, Cdy, Cx, C, dx, H, W, Cdx);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_avgpool2_bwd(const float* dy, int H, int W, int Cdy, int C, float* dx, int Cdx, void* stream) {
    if ((Cdx & 3) || (Cdy & 3)) DIP_FAIL("avgpool2_bwd: channel strides must be multiples of 4");
    const long long n = (long long)H * W * ((C + 3) / 4);
    dip_launch(avgpool2_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, W / 2,
                       Cdy, C, dx, H, W, Cdx);
    DIP_CHECK_LAUNCH();
    return 0;
}

// fin == NULL (or fin->coef == NULL): partials only; otherwise the last block finalises (dgamma, dbeta, coef)
extern "C" int dip_upsample_bwd_stats_crop_fin(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd,
                                               int od_y, int od_x, int mode, const float* y, int Cy, int C,
                                               const float* state, int Cs, float slope, float* dz, int Cdz,
                                               float* partials, int nblk, const DipBnbFin* finp, void* stream) {
    if (C > 1024) DIP_FAIL("upsample_bwd_stats_crop: C > 1024 unsupported");
    if (od_y < 0 || od_x < 0 || od_y + H > 2 * Hd || od_x + W > 2 * Wd) DIP_FAIL("upsample_bwd_stats_crop: crop window outside the up-sampled image");
    int nb;
    const int ppb = pixels_per_block(Hd * Wd, C, &nb);
    if (nb != nblk) DIP_FAIL("upsample_bwd_stats_crop: nblk mismatch (use dip_bn_bwd_nblk(Hd, Wd, C))");
    DipBnbFin fin = {};
    if (finp != nullptr && finp->coef != nullptr) {
        fin = *finp;
        if (C > 256 || nb > FIN_MAX_ROWS || fin.ticket == nullptr || fin.C != C)
            DIP_FAIL("upsample_bwd_stats_fin: needs <= 256 channels and rows (dip_fin_rows_ok), a ticket, C");
    }
    dip_launch_pair<DIP_FAM_UPCAT>(upsample_bwd_stats_kernel<false>, upsample_bwd_stats_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dcat,
                                   Cs_cat, choff, H, W, Hd, Wd, od_y, od_x, mode, y, Cy, C, state, Cs, slope, dz, Cdz, partials, ppb, fin);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_upsample_bwd_stats_crop(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd,
                                           int od_y, int od_x, int mode, const float* y, int Cy, int C,
                                           const float* state, int Cs, float slope, float* dz, int Cdz, float* partials,
                                           int nblk, void* stream) {
    return dip_upsample_bwd_stats_crop_fin(dcat, Cs_cat, choff, H, W, Hd, Wd, od_y, od_x, mode, y, Cy, C, state, Cs, slope,
                                           dz, Cdz, partials, nblk, nullptr, stream);
}

extern "C" int dip_upsample_bwd_stats(const float* dcat, int Cs_cat, int choff, int H, int W, int mode,
                                      const float* y, int Cy, int C, const float* state, int Cs, float slope,
                                      float* dz, int Cdz, float* partials, int nblk, void* stream) {
    return dip_upsample_bwd_stats_crop_fin(dcat, Cs_cat, choff, H, W, (H + 1) / 2, (W + 1) / 2, 0, 0, mode, y, Cy, C, state,
                                           Cs, slope, dz, Cdz, partials, nblk, nullptr, stream);
}
