// Gradient sources shared by the BatchNorm-backward kernels (bn_kernels.hip, bn_bwd_one.hip) and the adjoint of the 2x
// up-sampling (upcat_kernels.hip): a float4 of four channels of the incoming gradient du for one activation pixel.
#pragma once
#include "dip_common.h"

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// The thin 1x1 conv in front of the activation (DipGradSrc.tw): its <= 4 weight rows for channels [ch, ch+4), loaded once per
// thread (the streaming kernels keep `ch` for their whole pixel range), and du for one pixel from the conv's output gradient --
// a fixed chain of fused multiply-adds, j = 0 .. tn-1, per channel.
struct GradSrcThin {
    f32x4 w[4];
};
__device__ __forceinline__ GradSrcThin grad_src_thin(const DipGradSrc& s, int ch) {
    GradSrcThin t;
#pragma unroll
    for (int j = 0; j < 4; ++j) t.w[j] = (s.tw != nullptr && j < s.tn) ? ld4(s.tw + (size_t)j * s.tcw + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    return t;
}
__device__ __forceinline__ f32x4 grad_src_thin4(const DipGradSrc& s, const GradSrcThin& t, int r, int c, int W) {
    f32x4 gv = ld4(s.g + ((size_t)r * W + c) * s.Cg);
#pragma unroll
    for (int j = 1; j < 4; ++j) gv[j] = j < s.tn ? gv[j] : 0.f;          // (the pad channels of g are not part of the sum)
    f32x4 du;
#pragma unroll
    for (int e = 0; e < 4; ++e) du[e] = fmaf(gv[3], t.w[3][e], fmaf(gv[2], t.w[2][e], fmaf(gv[1], t.w[1][e], gv[0] * t.w[0][e])));
    return du;
}

// incoming gradient for pixel (r,c), channels [ch, ch+4): padded source with optional reflection fold
__device__ __forceinline__ f32x4 grad_src4(const DipGradSrc& s, int r, int c, int H, int W, int ch) {
    if (s.tw != nullptr) return grad_src_thin4(s, grad_src_thin(s, ch), r, c, W);
    const int P = s.pad;
    const float* base = s.g + s.choff + ch;
    if (s.win_h > 0) {                       // adjoint of a centre crop: zero outside the window
        const int wr = r - s.win_y, wc = c - s.win_x;
        if (wr < 0 || wr >= s.win_h || wc < 0 || wc >= s.win_w) return f32x4{0.f, 0.f, 0.f, 0.f};
        return ld4(base + ((size_t)wr * s.win_w + wc) * s.Cg);
    }
    const int Wg = W + 2 * P;
    if (!s.fold || P == 0) return ld4(base + ((size_t)(r + P) * Wg + (c + P)) * s.Cg);
    if (s.fold == 2) {
        // adjoint of nn.ReplicationPad2d: a border pixel collects every ring position that clamps onto it
        const int r0 = r == 0 ? 0 : r + P, r1 = r == H - 1 ? H - 1 + 2 * P : r + P;
        const int c0 = c == 0 ? 0 : c + P, c1 = c == W - 1 ? W - 1 + 2 * P : c + P;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = r0; i <= r1; ++i)
            for (int j = c0; j <= c1; ++j) acc += ld4(base + ((size_t)i * Wg + j) * s.Cg);
        return acc;
    }
    // rows of the padded domain that reflect onto r: r+P itself, P-r (top), 2(H-1)-r+P (bottom)
    int rr[3], nr = 0, cc[3], ncn = 0;
    rr[nr++] = r + P;
    if (r >= 1 && r <= P) rr[nr++] = P - r;
    if (r <= H - 2 && r >= H - 1 - P) rr[nr++] = 2 * (H - 1) - r + P;
    cc[ncn++] = c + P;
    if (c >= 1 && c <= P) cc[ncn++] = P - c;
    if (c <= W - 2 && c >= W - 1 - P) cc[ncn++] = 2 * (W - 1) - c + P;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < ncn; ++j) acc += ld4(base + ((size_t)rr[i] * Wg + cc[j]) * s.Cg);
    return acc;
}

// Adjoint of nn.Upsample(scale_factor=2) (models/skip.py:81 of the reference) for low-res pixel (i, j), channels at
// `base` = dcat + choff + ch: du = sum over the <= 4x4 high-res pixels whose interpolation touches it.  The deeper branch is
// [Hl][Wl]; the gradient dcat is [H][W] and covers rows ody..ody+H-1, columns odx..odx+W-1 of the [2*Hl][2*Wl] up-sampled
// image (Concat's centre crop; default geometry: Hl = (H+1)/2, offsets 0).
__device__ __forceinline__ f32x4 up_adj_du4(const float* base, int Cs_cat, int H, int W, int Hl, int Wl, int ody, int odx,
                                            int mode, int i, int j) {
    f32x4 du = f32x4{0.f, 0.f, 0.f, 0.f};
    if (mode == DIP_UP_NEAREST) {
#pragma unroll
        for (int dr = 0; dr < 2; ++dr)
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                const int hr = 2 * i + dr - ody, hc = 2 * j + dc - odx;          // position inside the crop window
                const f32x4 gq = ld4(base + ((size_t)min(max(hr, 0), H - 1) * W + min(max(hc, 0), W - 1)) * Cs_cat);
                const float wq = (hr >= 0 && hr < H && hc >= 0 && hc < W) ? 1.f : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) du[e] = fmaf(wq, gq[e], du[e]);
            }
        return du;
    }
    // adjoint weights of the scale-2 bilinear up-sampling (align_corners = False) in closed form: high-res
    // rows 2i-1 .. 2i+2 touch low-res row i with (0.25, 0.75, 0.75, 0.25); row 0 gives all of itself to
    // i = 0, the last low-res row also collects the clamped upper neighbour, rows outside [0, H) nothing
    // (weights of the FULL up-sampled image x "is the row inside the crop window")
    auto inw = [](int u, int o, int nwin) { return (u - o >= 0 && u - o < nwin) ? 1.f : 0.f; };
    const float wr[4] = {(i >= 1 ? 0.25f : 0.f) * inw(2 * i - 1, ody, H), (i == 0 ? 1.f : 0.75f) * inw(2 * i, ody, H),
                         (i == Hl - 1 ? 1.f : 0.75f) * inw(2 * i + 1, ody, H),
                         (i + 1 <= Hl - 1 ? 0.25f : 0.f) * inw(2 * i + 2, ody, H)};
    const float wc[4] = {(j >= 1 ? 0.25f : 0.f) * inw(2 * j - 1, odx, W), (j == 0 ? 1.f : 0.75f) * inw(2 * j, odx, W),
                         (j == Wl - 1 ? 1.f : 0.75f) * inw(2 * j + 1, odx, W),
                         (j + 1 <= Wl - 1 ? 0.25f : 0.f) * inw(2 * j + 2, odx, W)};
    // all 16 loads of the 4x4 window are issued unconditionally (clamped address, zero
    // weight outside the image): branching on the weights serialised them
    f32x4 gw[16];
#pragma unroll
    for (int tr = 0; tr < 4; ++tr) {
        const int hr = min(max(2 * i - 1 + tr - ody, 0), H - 1);
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            const int hc = min(max(2 * j - 1 + tc - odx, 0), W - 1);
            gw[tr * 4 + tc] = ld4(base + ((size_t)hr * W + hc) * Cs_cat);
        }
    }
#pragma unroll
    for (int tr = 0; tr < 4; ++tr)
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
            const float w = wr[tr] * wc[tc];
#pragma unroll
            for (int e = 0; e < 4; ++e) du[e] = fmaf(w, gw[tr * 4 + tc][e], du[e]);
        }
    return du;
}

// the same with the thin conv's weights hoisted by the caller (t = grad_src_thin(s, ch) in front of its pixel loop)
__device__ __forceinline__ f32x4 grad_src4(const DipGradSrc& s, const GradSrcThin& t, int r, int c, int H, int W, int ch) {
    if (s.tw != nullptr) return grad_src_thin4(s, t, r, c, W);
    return grad_src4(s, r, c, H, W, ch);
}
