// BatchNorm2d (train mode, N=1) statistics finalisation and the three backward phases, fused with
// LeakyReLU backward and the adjoint of ReflectionPad2d.  All HBM-bound: float4 per lane, NHWC.
#include "dip_common.h"
#include "bn_ticket.h"
#include "dip_group.h"
#include "dip_gradsrc.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------
// forward finalise: combine the per-tile {count, mean, M2} partials -> state + running stats.
// grid.x = ceil(C/4); a block = 256 tile rows x 4 channels (one 16-byte load per row and moment).
// One pass in fp64 on means shifted by the first tile's mean K:  N = sum n_i,  S1 = sum n_i (m_i-K),
//   S2 = sum (M2_i + n_i (m_i-K)^2);   mean = K + S1/N,  M2 = S2 - N (S1/N)^2
// followed by a fixed-order tree over the 256 rows (deterministic).  The first version walked the
// partials twice with 4-byte loads, 64 rows per block: 32 dependent trips, 27 us at 512x512.
// ------------------------------------------------------------------------------------------
template <bool GRP = false>
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partials_, int ntiles,
                                                          int Cstride, int C, const float* __restrict__ gamma_,
                                                          const float* __restrict__ beta_, float eps, float momentum,
                                                          float* state_, int Cs, float* running_mean_,
                                                          float* running_var_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, partials);
    DIP_GRP_PTR(const float*, gamma);
    DIP_GRP_PTR(const float*, beta);
    DIP_GRP_PTR(float*, state);
    DIP_GRP_PTR(float*, running_mean);
    DIP_GRP_PTR(float*, running_var);
    __shared__ double sh[256][12];
    const int row = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    const bool vec = (c0 + 3 < Cstride) && ((Cstride & 3) == 0);
    // means are accumulated relative to the first tile's mean K (same for every thread): the final
    // M2 = S2 - N d^2 then cancels only d = mean - K, not the mean itself
    double K[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) K[e] = (c0 + e < Cstride) ? (double)partials[Cstride + c0 + e] : 0.0;
    // the parameters the last four threads need are requested NOW: their round trip overlaps the row loads instead of
    // following the tree (the launch sits in the dependent chain of every BatchNorm: 30 per iteration)
    float pg = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    if (row < 4 && c0 + row < C) {
        pg = gamma[c0 + row];
        pb = beta[c0 + row];
        if (running_mean != nullptr) { prm = running_mean[c0 + row]; prv = running_var[c0 + row]; }
    }
#pragma unroll 8
    for (int t = row; t < ntiles; t += 256) {
        const float* p = partials + (size_t)t * 3 * Cstride + c0;
        float n[4], m[4], q[4];
        if (vec) {
            const f32x4 vn = *reinterpret_cast<const f32x4*>(p), vm = *reinterpret_cast<const f32x4*>(p + Cstride),
                        vq = *reinterpret_cast<const f32x4*>(p + 2 * Cstride);
#pragma unroll
            for (int e = 0; e < 4; ++e) { n[e] = vn[e]; m[e] = vm[e]; q[e] = vq[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = c0 + e < Cstride;
                n[e] = ok ? p[e] : 0.f; m[e] = ok ? p[Cstride + e] : 0.f; q[e] = ok ? p[2 * Cstride + e] : 0.f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double ni = (double)n[e], mi = (double)m[e] - K[e];
            acc[e] += ni;
            acc[4 + e] += ni * mi;
            acc[8 + e] += (double)q[e] + ni * mi * mi;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) sh[row][k] = acc[k];
    for (int s = 128; s >= 1; s >>= 1) {
        __syncthreads();
        if (row < s) {
#pragma unroll
            for (int k = 0; k < 12; ++k) sh[row][k] += sh[row + s][k];
        }
    }
    __syncthreads();
    if (row < 4 && c0 + row < C) {
        const int c = c0 + row;
        const double N = sh[0][row], S1 = sh[0][4 + row], S2 = sh[0][8 + row];
        const double dmean = N > 0.0 ? S1 / N : 0.0;
        const double mean = K[row] + dmean;
        double M2 = S2 - N * dmean * dmean;
        if (M2 < 0.0) M2 = 0.0;
        const double var = N > 0.0 ? M2 / N : 0.0;          // biased (normalisation)
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float a = pg * rstd;
        const float fm = (float)mean;
        state[c] = fm;
        state[Cs + c] = rstd;
        state[2 * Cs + c] = a;
        state[3 * Cs + c] = pb - fm * a;
        if (running_mean != nullptr) {
            const double unb = N > 1.0 ? M2 / (N - 1.0) : var;
            running_mean[c] = (1.f - momentum) * prm + momentum * fm;
            running_var[c] = (1.f - momentum) * prv + momentum * (float)unb;
        }
    }
}

// ------------------------------------------------------------------------------------------
// thread layout shared by the NHWC streaming kernels: a block walks a pixel range; thread
// (prow, cg) handles 4 channels [4cg, 4cg+4) of pixels prow, prow+RPI, ...   (RPI = 256 / NC4)
// ------------------------------------------------------------------------------------------
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

// block-level reduction of per-thread (s1, s2) float4 pairs over prow, written as [blk][2][Cs]
__device__ __forceinline__ void block_reduce_2(const RowLayout& L, f32x4 s1, f32x4 s2, float* partials, int Cs,
                                               float* sh /* 256*8 floats */) {
    dip_tree_sum8(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, s1, s2);
    if (L.active && L.prow == 0) {
        float* o = partials + (size_t)blockIdx.x * 2 * Cs + L.cg * 4;
        st4(o, s1);
        st4(o + Cs, s2);
    }
}

// ------------------------------------------------------------------------------------------
// backward phase 1: dz = du * lrelu'(a*y+b); partial sums S1 = sum dz, S2 = sum dz*xhat
// ------------------------------------------------------------------------------------------
template <bool GRP = false>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const DipGradSrc src_, const float* __restrict__ y_, int H,
                                                           int W, int Cy, int C, const float* __restrict__ state_,
                                                           int Cs, float slope, float* dz_, int Cdz, float* partials_,
                                                           int ppb /*pixels per block*/, const DipBnbFin fin_, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipGradSrc, src);
    DIP_GRP_PTR(const float*, y);
    DIP_GRP_PTR(const float*, state);
    DIP_GRP_PTR(float*, dz);
    DIP_GRP_PTR(float*, partials);
    DIP_GRP_DESC(DipBnbFin, fin);
    __shared__ __attribute__((aligned(16))) double shd[256 * 8];        // float tree, then the fp64 finalisation tree
    __shared__ unsigned flag;
    float* sh = reinterpret_cast<float*>(shd);
    const RowLayout L = row_layout(C);
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (L.active) {
        const int ch = L.cg * 4;
        const f32x4 mean = ld4(state + ch), rstd = ld4(state + Cs + ch), a = ld4(state + 2 * Cs + ch),
                    b = ld4(state + 3 * Cs + ch);
        const int npix = H * W;
        const int p0 = blockIdx.x * ppb;
        const int p1 = min(p0 + ppb, npix);
        const GradSrcThin tw = grad_src_thin(src, ch);
        auto one = [&](int p, const f32x4& du, const f32x4& yv) {
            f32x4 g;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z = fmaf(a[e], yv[e], b[e]);
                g[e] = dip_mul_rn(du[e], dip_act_grad(z, slope));      // (rounded product: never contracted into the sums)
                const float xh = (yv[e] - mean[e]) * rstd[e];
                s1[e] += g[e];
                s2[e] += g[e] * xh;
            }
            if (dz != nullptr) st4(dz + (size_t)p * Cdz + ch, g);
        };
        int p = p0 + L.prow;
        // two pixels per iteration: four 16-byte loads in flight per thread (HBM-bound pass)
        for (; p + L.rpi < p1; p += 2 * L.rpi) {
            const int q = p + L.rpi;
            const int r0 = p / W, c0 = p - r0 * W, r1 = q / W, c1 = q - r1 * W;
            const f32x4 du0 = grad_src4(src, tw, r0, c0, H, W, ch);
            const f32x4 du1 = grad_src4(src, tw, r1, c1, H, W, ch);
            const f32x4 y0 = ld4(y + (size_t)p * Cy + ch);
            const f32x4 y1 = ld4(y + (size_t)q * Cy + ch);
            one(p, du0, y0);
            one(q, du1, y1);
        }
        if (p < p1) {
            const int r = p / W, c = p - r * W;
            one(p, grad_src4(src, tw, r, c, H, W, ch), ld4(y + (size_t)p * Cy + ch));
        }
    }
    if (fin.coef == nullptr) {
        block_reduce_2(L, s1, s2, partials, Cs, sh);
        return;
    }
    // in-launch finalisation (bn_ticket.h): write-through rows, ticket, the last block reduces them all
    dip_tree_sum8(sh, L.nc4, L.rpi, L.prow, L.cg, L.active, s1, s2);
    if (L.active && L.prow == 0) {
        float* o = partials + (size_t)blockIdx.x * 2 * Cs + L.cg * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { dip_st_sc1(o + e, s1[e]); dip_st_sc1(o + Cs + e, s2[e]); }
    }
    if (dip_ticket_last(fin.ticket, gridDim.x, &flag)) {
        dip_bnb_fin_rows(partials, gridDim.x, Cs, 0, C, fin, shd);
        dip_ticket_reset(fin.ticket);
    }
}

// ------------------------------------------------------------------------------------------
// backward phase 2: reduce partials (fp64, fixed order) -> dgamma, dbeta, k1 = S1/N, k2 = S2/N
// ------------------------------------------------------------------------------------------
template <bool GRP = false>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials_, int nblk,
                                                              const float* __restrict__ partials_lo_, int nblk_lo,
                                                              int c_lo, int Cs, int C, int npix, float* dgamma_,
                                                              float* dbeta_, float* coef_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, partials);
    DIP_GRP_PTR(const float*, partials_lo);
    DIP_GRP_PTR(float*, dgamma);
    DIP_GRP_PTR(float*, dbeta);
    DIP_GRP_PTR(float*, coef);
    // block = 256 partial rows x 4 channels (16-byte loads), fp64, fixed-order tree
    __shared__ double sh[256][8];
    const int row = threadIdx.x;
    const int c0 = blockIdx.x * 4;
    if (c0 < c_lo) { partials = partials_lo; nblk = nblk_lo; }      // (channels of the thin-column launch)
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int t = row; t < nblk; t += 256) {
        const float* p = partials + (size_t)t * 2 * Cs + c0;      // Cs % 4 == 0, c0 + 3 < Cs
        const f32x4 v1 = ld4(p), v2 = ld4(p + Cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1[e] += (double)v1[e]; a2[e] += (double)v2[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[row][e] = a1[e]; sh[row][4 + e] = a2[e]; }
    for (int s = 128; s >= 1; s >>= 1) {
        __syncthreads();
        if (row < s) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sh[row][k] += sh[row + s][k];
        }
    }
    __syncthreads();
    if (row < 4 && c0 + row < C) {
        const int c = c0 + row;
        const double s1 = sh[0][row], s2 = sh[0][4 + row];
        if (dbeta != nullptr) dbeta[c] = (float)s1;
        if (dgamma != nullptr) dgamma[c] = (float)s2;
        coef[c] = (float)(s1 / npix);
        coef[Cs + c] = (float)(s2 / npix);
    }
}

// ------------------------------------------------------------------------------------------
// Phase 2 inside the phase-3 launches (round 6).  For a low-resolution BatchNorm the finalisation launch between the
// statistics pass and the apply pass is a dependent ~10 us (bn_bwd_finalize_kernel: 18 per iteration of the 'library' net,
// 191 us of its main chain; 30 per iteration of the default net) that produces 2 x C floats.  With few partial rows
// EVERY block of the apply launch reduces them for itself in its prologue -- thread (prow, cg) sums rows prow, prow + rpi, ..
// of its four channels in fp64, a fixed-order tree over prow through LDS, the same values in every block -- and block 0
// writes dgamma, dbeta and the coefficient block.  No ticket, no fence, no cross-workgroup dependency (the forms of
// bn_ticket.h were measured slower than the launch; this one reads <= 320 rows x C x 8 bytes from L2 per block).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bnb_fin_prologue(const RowLayout& L, const float* __restrict__ partials, int nrows, int Cs,
                                                 int C, int npix, double* shd, float* dgamma, float* dbeta, float* coef,
                                                 f32x4& k1, f32x4& k2) {
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.0;
    if (L.active) {
        const int ch = L.cg * 4;
#pragma unroll 4
        for (int t = L.prow; t < nrows; t += L.rpi) {
            const float* p = partials + (size_t)t * 2 * Cs + ch;
            const f32x4 v1 = ld4(p), v2 = ld4(p + Cs);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += (double)v1[e]; a[4 + e] += (double)v2[e]; }
        }
    }
    double* mine = shd + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) mine[k] = a[k];
    for (int s = dip_pow2_ceil(L.rpi) >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (L.active && L.prow < s && L.prow + s < L.rpi) {
            const double* q = shd + (size_t)((L.prow + s) * L.nc4 + L.cg) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) mine[k] += q[k];
        }
    }
    __syncthreads();
    if (!L.active) return;
    const double* r0 = shd + (size_t)L.cg * 8;          // the prow == 0 thread of this channel group
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        k1[e] = (float)(r0[e] / npix);
        k2[e] = (float)(r0[4 + e] / npix);
    }
    if (blockIdx.x == 0 && L.prow == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = L.cg * 4 + e;
            if (c < C) {
                if (dbeta != nullptr) dbeta[c] = (float)r0[e];
                if (dgamma != nullptr) dgamma[c] = (float)r0[4 + e];
                if (coef != nullptr) { coef[c] = k1[e]; coef[Cs + c] = k2[e]; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward phase 3 (in place): dy = a * (dz - k1 - xhat * k2)
// ------------------------------------------------------------------------------------------
// FIN: phase 2 in the prologue (bnb_fin_prologue): `fin` = {partials, nrows, dgamma, dbeta}, coef is written, not read
struct BnbFinArg {
    const float* partials;
    int nrows;
    float* dgamma;
    float* dbeta;
};
template <class F> __host__ __device__ inline void dip_ptrs(BnbFinArg& b, F& f) { f(b.partials); f(b.dgamma); f(b.dbeta); }

template <bool GRP = false, bool FIN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* dz_, int Cdz, const float* __restrict__ y_, int Cy,
                                                           int npix, int C, const float* __restrict__ state_, int Cs,
                                                           float* coef_, int ppb, const BnbFinArg fin_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(float*, dz);
    DIP_GRP_PTR(const float*, y);
    DIP_GRP_PTR(const float*, state);
    DIP_GRP_PTR(float*, coef);
    DIP_GRP_DESC(BnbFinArg, fin);
    __shared__ __attribute__((aligned(16))) double shd[FIN ? 256 * 8 : 1];
    const RowLayout L = row_layout(C);
    f32x4 k1 = f32x4{0.f, 0.f, 0.f, 0.f}, k2 = k1;
    if constexpr (FIN) bnb_fin_prologue(L, fin.partials, fin.nrows, Cs, C, npix, shd, fin.dgamma, fin.dbeta, coef, k1, k2);
    if (!L.active) return;
    const int ch = L.cg * 4;
    const f32x4 mean = ld4(state + ch), rstd = ld4(state + Cs + ch), a = ld4(state + 2 * Cs + ch);
    if constexpr (!FIN) { k1 = ld4(coef + ch); k2 = ld4(coef + Cs + ch); }
    const int p0 = blockIdx.x * ppb;
    const int p1 = min(p0 + ppb, npix);
    for (int p = p0 + L.prow; p < p1; p += L.rpi) {
        f32x4 g = ld4(dz + (size_t)p * Cdz + ch);
        const f32x4 yv = ld4(y + (size_t)p * Cy + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xh = (yv[e] - mean[e]) * rstd[e];
            g[e] = a[e] * (g[e] - k1[e] - xh * k2[e]);
        }
        st4(dz + (size_t)p * Cdz + ch, g);
    }
}

// backward phase 3 from the gradient source (phase 1 ran with dz == NULL): the masked gradient is
// recomputed instead of being written by phase 1 and re-read here -- 5 tensor passes per BatchNorm
// instead of 6:  dy = a * (du * lrelu'(a*y+b) - k1 - xhat * k2)
template <bool GRP = false, bool FIN = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_src_kernel(const DipGradSrc src_, const float* __restrict__ y_, int H,
                                                               int W, int Cy, int C, const float* __restrict__ state_,
                                                               int Cs, float slope, float* coef_,
                                                               float* __restrict__ dy_, int Cdy, int ppb, const BnbFinArg fin_,
                                                               const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipGradSrc, src);
    DIP_GRP_PTR(const float*, y);
    DIP_GRP_PTR(const float*, state);
    DIP_GRP_PTR(float*, coef);
    DIP_GRP_PTR(float*, dy);
    DIP_GRP_DESC(BnbFinArg, fin);
    __shared__ __attribute__((aligned(16))) double shd[FIN ? 256 * 8 : 1];
    const RowLayout L = row_layout(C);
    f32x4 k1 = f32x4{0.f, 0.f, 0.f, 0.f}, k2 = k1;
    if constexpr (FIN) bnb_fin_prologue(L, fin.partials, fin.nrows, Cs, C, H * W, shd, fin.dgamma, fin.dbeta, coef, k1, k2);
    if (!L.active) return;
    const int ch = L.cg * 4;
    const f32x4 mean = ld4(state + ch), rstd = ld4(state + Cs + ch), a = ld4(state + 2 * Cs + ch),
                b = ld4(state + 3 * Cs + ch);
    if constexpr (!FIN) { k1 = ld4(coef + ch); k2 = ld4(coef + Cs + ch); }
    const int npix = H * W;
    const int p0 = blockIdx.x * ppb;
    const int p1 = min(p0 + ppb, npix);
    const GradSrcThin tw = grad_src_thin(src, ch);
    auto one = [&](int p, const f32x4& du, const f32x4& yv) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(a[e], yv[e], b[e]);
            const float gm = dip_mul_rn(du[e], dip_act_grad(z, slope));   // same rounding as the phase-1 value
            const float xh = (yv[e] - mean[e]) * rstd[e];
            g[e] = a[e] * (gm - k1[e] - xh * k2[e]);
        }
        st4(dy + (size_t)p * Cdy + ch, g);
    };
    int p = p0 + L.prow;
    for (; p + L.rpi < p1; p += 2 * L.rpi) {            // two pixels per iteration (see bn_bwd_stats_kernel)
        const int q = p + L.rpi;
        const int r0 = p / W, c0 = p - r0 * W, r1 = q / W, c1 = q - r1 * W;
        const f32x4 du0 = grad_src4(src, tw, r0, c0, H, W, ch);
        const f32x4 du1 = grad_src4(src, tw, r1, c1, H, W, ch);
        const f32x4 y0 = ld4(y + (size_t)p * Cy + ch);
        const f32x4 y1 = ld4(y + (size_t)q * Cy + ch);
        one(p, du0, y0);
        one(q, du1, y1);
    }
    if (p < p1) {
        const int r = p / W, c = p - r * W;
        one(p, grad_src4(src, tw, r, c, H, W, ch), ld4(y + (size_t)p * Cy + ch));
    }
}

// fold a padded gradient onto the image, NHWC -> NCHW
__global__ __launch_bounds__(256) void fold_to_nchw_kernel(const DipGradSrc src, int H, int W, int C, float* dst) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int r = p / W, c = p - r * W;
    for (int ch = 0; ch < C; ch += 4) {
        const f32x4 g = grad_src4(src, r, c, H, W, ch);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ch + e < C) dst[(size_t)(ch + e) * H * W + p] = g[e];
    }
}

// fold a padded gradient onto the image, NHWC -> NHWC
__global__ __launch_bounds__(256) void fold_to_nhwc_kernel(const DipGradSrc src, int H, int W, int C, float* dst, int Cd) {
    const int nc4 = (C + 3) >> 2;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)H * W * nc4) return;
    const int cg = (int)(i % nc4);
    const long long p = i / nc4;
    const int r = (int)(p / W), c = (int)(p - (long long)r * W);
    st4(dst + (size_t)p * Cd + cg * 4, grad_src4(src, r, c, H, W, cg * 4));
}

// target block counts of the streaming kernels: 1024 where a block writes a row of partials (statistics pass), 4096 for
// the two apply kernels -- 68 VGPRs allow 7 workgroups per CU, 1024 blocks keep 4 resident; measured on the 25 + 5
// launches of an iteration: memory-bound group 1.606 -> 1.573 ms (2048 blocks: 1.595)
__host__ int bn_blocks(bool apply) { return apply ? 4096 : 1024; }

__host__ int pixels_per_block(int npix, int C, int* nblk, bool apply = false) {
    // ~1024 blocks for large tensors; small ones: two pixels per thread
    const int nc4 = (C + 3) / 4;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    int ppb = dip_cdiv(npix, bn_blocks(apply));
    if (ppb < rpi * 2) ppb = rpi * 2;      // small tensors: few sequential pixels per thread (latency-bound)
    *nblk = dip_cdiv(npix, ppb);
    return ppb;
}

}  // namespace

extern "C" int dip_bn_finalize(const float* partials, int ntiles, int Cstride, int C, const float* gamma,
                               const float* beta, float eps, float momentum, float* state, int Cs,
                               float* running_mean, float* running_var, void* stream) {
    dip_launch_pair<DIP_FAM_BN>(bn_finalize_kernel<false>, bn_finalize_kernel<true>, dim3(dip_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partials,
                                ntiles, Cstride, C, gamma, beta, eps, momentum, state, Cs, running_mean, running_var);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_nblk(int H, int W, int C) {
    int nblk;
    pixels_per_block(H * W, C, &nblk);
    return nblk;
}

// fin == NULL (or fin->coef == NULL): partials only (dip_bn_bwd_finalize follows); otherwise phase 2 rides in the launch:
// the last block to arrive writes dgamma, dbeta and coef (<= 256 rows and channels: dip_fin_rows_ok)
extern "C" int dip_bn_bwd_stats_fin(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                                    const float* state, int Cs, float slope, float* dz, int Cdz, float* partials,
                                    int nblk, const DipBnbFin* finp, void* stream) {
    if (C > 1024) DIP_FAIL("bn_bwd_stats: C > 1024 unsupported");
    int nb;
    const int ppb = pixels_per_block(H * W, C, &nb);
    if (nb != nblk) DIP_FAIL("bn_bwd_stats: nblk mismatch (use dip_bn_bwd_nblk)");
    DipBnbFin fin = {};
    if (finp != nullptr && finp->coef != nullptr) {
        fin = *finp;
        if (C > 256 || nb > 256 || fin.ticket == nullptr || fin.C != C)
            DIP_FAIL("bn_bwd_stats_fin: needs <= 256 channels and rows (dip_fin_rows_ok), a ticket, C");
    }
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_stats_kernel<false>, bn_bwd_stats_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *src, y, H, W, Cy,
                                C, state, Cs, slope, dz, Cdz, partials, ppb, fin);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_stats(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                                const float* state, int Cs, float slope, float* dz, int Cdz, float* partials,
                                int nblk, void* stream) {
    return dip_bn_bwd_stats_fin(src, y, H, W, Cy, C, state, Cs, slope, dz, Cdz, partials, nblk, nullptr, stream);
}

extern "C" int dip_bn_bwd_finalize(const float* partials, int nblk, int Cs, int C, int npix, float* dgamma,
                                   float* dbeta, float* coef, void* stream) {
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_finalize_kernel<false>, bn_bwd_finalize_kernel<true>, dim3(dip_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream,
                                partials, nblk, (const float*)nullptr, 0, 0, Cs, C, npix, dgamma, dbeta, coef);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_finalize2(const float* partials, int nblk, const float* partials_lo, int nblk_lo, int c_lo,
                                    int Cs, int C, int npix, float* dgamma, float* dbeta, float* coef, void* stream) {
    if ((c_lo & 3) || (c_lo > 0 && partials_lo == nullptr)) DIP_FAIL("bn_bwd_finalize2: c_lo must be a multiple of 4");
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_finalize_kernel<false>, bn_bwd_finalize_kernel<true>, dim3(dip_cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream,
                                partials, nblk, partials_lo, nblk_lo, c_lo, Cs, C, npix, dgamma, dbeta, coef);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_apply(float* dz, int Cdz, const float* y, int Cy, int npix, int C, const float* state,
                                int Cs, const float* coef, void* stream) {
    int nb;
    const int ppb = pixels_per_block(npix, C, &nb, true);
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_apply_kernel<false>, bn_bwd_apply_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dz, Cdz, y, Cy, npix,
                                C, state, Cs, const_cast<float*>(coef), ppb, BnbFinArg{});
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_apply_src(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C,
                                    const float* state, int Cs, float slope, const float* coef, float* dy, int Cdy,
                                    void* stream) {
    if (C > 1024) DIP_FAIL("bn_bwd_apply_src: C > 1024 unsupported");
    int nb;
    const int ppb = pixels_per_block(H * W, C, &nb, true);
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_apply_src_kernel<false>, bn_bwd_apply_src_kernel<true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, *src, y, H,
                                W, Cy, C, state, Cs, slope, const_cast<float*>(coef), dy, Cdy, ppb, BnbFinArg{});
    DIP_CHECK_LAUNCH();
    return 0;
}

// phase 2 + phase 3 in one launch: every block reduces the `nrows` partial rows [nrows][2][Cs] in its prologue
// (bnb_fin_prologue); block 0 writes dgamma, dbeta (may be NULL) and coef
extern "C" int dip_bn_bwd_fin_rows_ok(int nrows, int C) {
    static const int maxrows = getenv("DIP_BNB_FIN_MAX_ROWS") ? atoi(getenv("DIP_BNB_FIN_MAX_ROWS")) : 320;
    return (nrows >= 1 && nrows <= maxrows && C >= 1 && C <= 1024) ? 1 : 0;
}

extern "C" int dip_bn_bwd_apply_fin(float* dz, int Cdz, const float* y, int Cy, int npix, int C, const float* state, int Cs,
                                    const float* partials, int nrows, float* dgamma, float* dbeta, float* coef, void* stream) {
    if (C > 1024 || (Cs & 3) || partials == nullptr || nrows < 1) DIP_FAIL("bn_bwd_apply_fin: C <= 1024, Cs % 4 == 0, partial rows");
    int nb;
    const int ppb = pixels_per_block(npix, C, &nb, true);
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_apply_kernel<false, true>, bn_bwd_apply_kernel<true, true>, dim3(nb), dim3(256), 0, (hipStream_t)stream, dz, Cdz,
                                y, Cy, npix, C, state, Cs, coef, ppb, BnbFinArg{partials, nrows, dgamma, dbeta});
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_bn_bwd_apply_src_fin(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C, const float* state,
                                        int Cs, float slope, const float* partials, int nrows, float* dgamma, float* dbeta,
                                        float* coef, float* dy, int Cdy, void* stream) {
    if (C > 1024 || (Cs & 3) || partials == nullptr || nrows < 1) DIP_FAIL("bn_bwd_apply_src_fin: C <= 1024, Cs % 4 == 0, partial rows");
    int nb;
    const int ppb = pixels_per_block(H * W, C, &nb, true);
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_apply_src_kernel<false, true>, bn_bwd_apply_src_kernel<true, true>, dim3(nb), dim3(256), 0, (hipStream_t)stream,
                                *src, y, H, W, Cy, C, state, Cs, slope, coef, dy, Cdy, ppb, BnbFinArg{partials, nrows, dgamma, dbeta});
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_fold_to_nhwc(const DipGradSrc* src, int H, int W, int C, float* dst, int Cd, void* stream) {
    if ((Cd & 3) || (src->Cg & 3)) DIP_FAIL("fold_to_nhwc: channel strides must be multiples of 4");
    const long long n = (long long)H * W * ((C + 3) / 4);
    dip_launch(fold_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *src, H, W, C, dst, Cd);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_fold_to_nchw(const DipGradSrc* src, int H, int W, int C, float* dst, void* stream) {
    dip_launch(fold_to_nchw_kernel, dim3(dip_cdiv(H * W, 256)), dim3(256), 0, (hipStream_t)stream, *src, H, W, C, dst);
    DIP_CHECK_LAUNCH();
    return 0;
}
