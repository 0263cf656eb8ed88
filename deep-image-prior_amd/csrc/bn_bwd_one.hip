// BatchNorm2d (train mode, N = 1) + activation backward of a LOW-RESOLUTION activation in ONE launch
// (autograd NativeBatchNormBackward + LeakyReluBackward of models/common.py:82,96 of the reference; with the adjoint of
// nn.Upsample(scale_factor=2), models/skip.py:81, in front of it for the deeper branch of a Concat).
//
// Why.  The three-launch form (bn_kernels.hip: statistics -> finalise -> apply) exists because the two sums
//   S1 = sum dz,  S2 = sum dz * xhat,   dz = du * act'(a y + b)
// run over the whole H x W plane: a grid-wide dependency.  Here a workgroup
// OWNS four channels of the whole plane (NHWC: one 16-byte load per pixel), so the dependency is workgroup-local:
//   pass 1  every thread walks its pixels: du (gradient source, reflection / replication fold, crop window, or the
//           up-sampling adjoint), y -> per-thread fp32 sums, then a fixed-order fp64 reduction over the workgroup's waves;
//   pass 2  the same pixels again (the plane's slice is L2-resident: <= 2 x 16 B x 16 K pixels per workgroup):
//           dy = a * (dz - S1 / N - xhat * S2 / N), written once.
// dgamma, dbeta and the coefficient block [k1, k2] are written as dip_bn_bwd_finalize writes them.  grid = ceil(C / 4)
// workgroups of 1024 threads: 32 CUs for a 128-channel layer.
// MEASURED (round 6, tools/bnone_time.py, profiles/r06_bn_bwd_one.txt): back to back, 8.2 / 10.5 us at 16^2 / 32^2 x 128
// channels against 12.1 / 12.5 us for the three launches -- and 34 / 135 us at 64^2 / 128^2 against 13 / 19: a workgroup
// that owns 4 of 128 channels uses 16 bytes of every 128-byte line it pulls into its L1, so 32 CUs deliver 1/8 of their
// fill bandwidth, and owning 32 channels (whole lines) leaves 4 CUs: ownership needs few CUs, bandwidth needs many.  The
// form therefore serves <= 1024 pixels only (DIP_BNB_ONE_MAX_PIXELS); in the iteration it removes 12 launches of the
// default net (244 -> 232) and 9 of the 'library' net for +-0 / +0.4 % it/s: what the low-resolution walk waits for is
// dependent memory round trips, not launches (DESIGN.md section 3.7 said so; this is the measurement).
// Deterministic (fixed pixel -> thread map, fixed reduction order); the rounded product dz = du * act'(z) is the same
// expression as in bn_kernels.hip.
#include "dip_common.h"
#include "dip_group.h"
#include "dip_gradsrc.h"
#include <stdlib.h>

namespace {

constexpr int ONE_NT = 1024;

struct UpGeom;
template <class F> __host__ __device__ inline void dip_ptrs(UpGeom&, F&) {}      // (dip_group.h: no pointers inside)
struct UpGeom {                 // SRC == 1: src.g = dcat [H][W][Cg], the activation is [Hl][Wl]
    int H, W, ody, odx, mode;
};

// SRC 0: DipGradSrc over the activation's own H x W domain; SRC 1: adjoint of the 2x up-sampling of dcat
template <int SRC, bool GRP = false>
__global__ __launch_bounds__(ONE_NT) void bn_bwd_one_kernel(const DipGradSrc src_, const UpGeom ug, const float* __restrict__ y_,
                                                            int Ha, int Wa, int Cy, int C, const float* __restrict__ state_,
                                                            int Cs, float slope, float* __restrict__ dy_, int Cdy,
                                                            float* dgamma_, float* dbeta_, float* coef_,
                                                            const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipGradSrc, src);
    DIP_GRP_PTR(const float*, y);
    DIP_GRP_PTR(const float*, state);
    DIP_GRP_PTR(float*, dy);
    DIP_GRP_PTR(float*, dgamma);
    DIP_GRP_PTR(float*, dbeta);
    DIP_GRP_PTR(float*, coef);
    __shared__ double red[ONE_NT / 64][8];
    __shared__ float kk[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x * 4;
    const int npix = Ha * Wa;
    const f32x4 mean = ld4(state + ch), rstd = ld4(state + Cs + ch), a = ld4(state + 2 * Cs + ch), b = ld4(state + 3 * Cs + ch);

    auto du_of = [&](int p) -> f32x4 {
        const int r = p / Wa, c = p - r * Wa;
        if constexpr (SRC == 0) return grad_src4(src, r, c, Ha, Wa, ch);
        else return up_adj_du4(src.g + src.choff + ch, src.Cg, ug.H, ug.W, Ha, Wa, ug.ody, ug.odx, ug.mode, r, c);
    };

    // ---- pass 1: S1, S2 ------------------------------------------------------------------------------------
    f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    auto acc1 = [&](const f32x4& du, const f32x4& yv) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(a[e], yv[e], b[e]);
            const float g = dip_mul_rn(du[e], dip_act_grad(z, slope));
            const float xh = (yv[e] - mean[e]) * rstd[e];
            s1[e] += g;
            s2[e] += g * xh;
        }
    };
    int p = tid;
    for (; p + ONE_NT < npix; p += 2 * ONE_NT) {           // two pixels per trip: their loads are in flight together
        const int q = p + ONE_NT;
        const f32x4 du0 = du_of(p), du1 = du_of(q);
        const f32x4 y0 = ld4(y + (size_t)p * Cy + ch), y1 = ld4(y + (size_t)q * Cy + ch);
        acc1(du0, y0);
        acc1(du1, y1);
    }
    if (p < npix) acc1(du_of(p), ld4(y + (size_t)p * Cy + ch));

    // fixed-order fp64 reduction: xor-butterfly inside a wave, then the waves in index order
    double v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = (double)s1[e]; v[4 + e] = (double)s2[e]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], m);
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < 8; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (tid < 8) {
        double s = 0.0;
        for (int w = 0; w < ONE_NT / 64; ++w) s += red[w][tid];
        const int e = tid & 3, c = ch + e;
        kk[tid] = (float)(s / npix);                       // k1 (tid < 4) / k2, as dip_bn_bwd_finalize rounds them
        if (c < C) {
            if (tid < 4) {
                if (dbeta != nullptr) dbeta[c] = (float)s;
                if (coef != nullptr) coef[c] = (float)(s / npix);
            } else {
                if (dgamma != nullptr) dgamma[c] = (float)s;
                if (coef != nullptr) coef[Cs + c] = (float)(s / npix);
            }
        }
    }
    __syncthreads();
    const f32x4 k1 = f32x4{kk[0], kk[1], kk[2], kk[3]}, k2 = f32x4{kk[4], kk[5], kk[6], kk[7]};

    // ---- pass 2: dy = a * (dz - k1 - xhat * k2) ---------------------------------------------------------------
    auto apply = [&](int pp, const f32x4& du, const f32x4& yv) {
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float z = fmaf(a[e], yv[e], b[e]);
            const float gm = dip_mul_rn(du[e], dip_act_grad(z, slope));
            const float xh = (yv[e] - mean[e]) * rstd[e];
            g[e] = a[e] * (gm - k1[e] - xh * k2[e]);
        }
        st4(dy + (size_t)pp * Cdy + ch, g);
    };
    p = tid;
    for (; p + ONE_NT < npix; p += 2 * ONE_NT) {
        const int q = p + ONE_NT;
        const f32x4 du0 = du_of(p), du1 = du_of(q);
        const f32x4 y0 = ld4(y + (size_t)p * Cy + ch), y1 = ld4(y + (size_t)q * Cy + ch);
        apply(p, du0, y0);
        apply(q, du1, y1);
    }
    if (p < npix) apply(p, du_of(p), ld4(y + (size_t)p * Cy + ch));
}

int one_max_pixels() {
    static const int v = [] {
        const char* e = getenv("DIP_BNB_ONE_MAX_PIXELS");
        return e ? atoi(e) : 1024;
    }();
    return v;
}

}  // namespace

// 1 when the engine should run a BatchNorm backward over npix pixels x C channels as ONE launch (DIP_BNB_ONE_MAX_PIXELS,
// default 1024; 0 switches the form off)
extern "C" int dip_bn_bwd_one_ok(int npix, int C) {
    return (npix >= 1 && npix <= one_max_pixels() && C >= 1 && C <= 4096) ? 1 : 0;
}

extern "C" int dip_bn_bwd_one(const DipGradSrc* src, const float* y, int H, int W, int Cy, int C, const float* state, int Cs,
                              float slope, float* dy, int Cdy, float* dgamma, float* dbeta, float* coef, void* stream) {
    if ((Cy & 3) || (Cdy & 3) || (Cs & 3) || (src->Cg & 3) || C > Cs || H < 1 || W < 1)
        DIP_FAIL("bn_bwd_one: channel strides must be multiples of 4, C <= Cs");
    const UpGeom ug = {};
    dip_launch_pair<DIP_FAM_BN>(bn_bwd_one_kernel<0, false>, bn_bwd_one_kernel<0, true>, dim3(dip_cdiv(C, 4)), dim3(ONE_NT), 0,
                                (hipStream_t)stream, *src, ug, y, H, W, Cy, C, state, Cs, slope, dy, Cdy, dgamma, dbeta, coef);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_upsample_bwd_one(const float* dcat, int Cs_cat, int choff, int H, int W, int Hd, int Wd, int od_y, int od_x,
                                    int mode, const float* y, int Cy, int C, const float* state, int Cs, float slope,
                                    float* dy, int Cdy, float* dgamma, float* dbeta, float* coef, void* stream) {
    if ((Cy & 3) || (Cdy & 3) || (Cs & 3) || (Cs_cat & 3) || (choff & 3) || C > Cs)
        DIP_FAIL("upsample_bwd_one: channel strides / offset must be multiples of 4, C <= Cs");
    if (od_y < 0 || od_x < 0 || od_y + H > 2 * Hd || od_x + W > 2 * Wd) DIP_FAIL("upsample_bwd_one: crop window outside the up-sampled image");
    DipGradSrc src = {};
    src.g = dcat; src.Cg = Cs_cat; src.choff = choff;
    const UpGeom ug = {H, W, od_y, od_x, mode};
    dip_launch_pair<DIP_FAM_UPCAT>(bn_bwd_one_kernel<1, false>, bn_bwd_one_kernel<1, true>, dim3(dip_cdiv(C, 4)), dim3(ONE_NT), 0,
                                   (hipStream_t)stream, src, ug, y, Hd, Wd, Cy, C, state, Cs, slope, dy, Cdy, dgamma, dbeta, coef);
    DIP_CHECK_LAUNCH();
    return 0;
}
