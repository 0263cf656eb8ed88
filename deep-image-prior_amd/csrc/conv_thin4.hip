// 3x3 stride-1 convolution towards <= 4 output channels (vector ALU, LDS-staged halo).
//
// Where it is used: the data gradient of the decoder convs flows into a 132-channel concat tensor
// = 4 skip channels + 128 up-sampled channels (reference models/skip.py:50-53, models/common.py:39).
// On the matrix core the 4 skip columns cost a whole 32-column block (the N = 160 variant of
// conv_igemm_kernel, or a 32-column launch with 28 idle columns); here they cost
// 2*9*Cin*4 FLOP per pixel on the VALU (2.4 GFLOP at 512x512, Cin = 128) and the other 128 columns
// go through conv_igemm_dma_kernel at its full rate.
//
//   y[p][n] = bias[n] + sum_tap sum_c x[src(p, tap)][c] * Wp[(tap*Cin/4 + c/4)*CoutP + n][c%4],  n < 4
//
// One thread = one output pixel of a 16x16 tile, 4 accumulators.  The 18x18-pixel halo of a
// 32-channel chunk is staged in LDS (pixel pitch 36 dwords: the 16 lanes of a ds_read_b128 pass
// fall on 16 distinct 16-byte slots); the weights of a (tap, 4-channel group) are 16 consecutive
// floats of the packed B operand; a chunk's 72 groups (4.6 KB) are staged in LDS next to the halo and
// read as broadcasts (through the scalar cache they missed on every group: 18 KB per workgroup walk
// thrashes it, 85 us per workgroup instead of ~25).
#include "dip_common.h"
#include "dip_group.h"

namespace {

constexpr int T4_TH = 16, T4_TW = 16, T4_HH = T4_TH + 2, T4_HW = T4_TW + 2, T4_NPIX = T4_HH * T4_HW;
constexpr int T4_PITCH = 36;                                   // dwords per halo pixel (32 channels + 4 pad)
constexpr int T4_SLOTS = (T4_NPIX * 8 + 255) / 256;            // 16-byte pieces per thread and chunk

__device__ __forceinline__ int t4_map_src(int v, int n_in, int reflect) {
    if (reflect) v = dip_reflect(v, n_in);
    return (v < 0 || v >= n_in) ? -1 : v;
}

template <bool GRP = false>
__global__ __launch_bounds__(256) void conv_thin4_kernel(const DipConvDesc d_, const int ntx, const int CoutP,
                                                        const int ncols, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    __shared__ __attribute__((aligned(16))) float halo[T4_NPIX * T4_PITCH];
    __shared__ int srcoff[T4_NPIX];
    __shared__ __attribute__((aligned(16))) float wsh[9 * 8 * 16];           // [tap][c4][n][c%4] of the current chunk
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
    for (int hp = tid; hp < T4_NPIX; hp += 256) {
        const int hr = hp / T4_HW, hc = hp - hr * T4_HW;
        const int sr = t4_map_src(ty * T4_TH + hr - d.off, d.Hin, d.pad_mode);
        const int sc = t4_map_src(tx * T4_TW + hc - d.off, d.Win, d.pad_mode);
        srcoff[hp] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
    }
    __syncthreads();
    int soff[T4_SLOTS];                     // this thread's staging pieces: source float offset or -1
#pragma unroll
    for (int i = 0; i < T4_SLOTS; ++i) {
        const int f = tid + i * 256;
        const int hp = f >> 3;
        const int so = (hp < T4_NPIX) ? srcoff[hp] : -2;
        soff[i] = so >= 0 ? so * d.Cx + (f & 7) * 4 : so;
    }
    const int py = tid >> 4, px = tid & 15;
    const float* hbase = halo + (py * T4_HW + px) * T4_PITCH;
    const int cin4 = d.Cin >> 2;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int cb = 0; cb < d.Cin; cb += 32) {
        const int c4n = min(8, (d.Cin - cb) >> 2);          // 4-channel groups in this chunk
        f32x4 st[T4_SLOTS];
#pragma unroll
        for (int i = 0; i < T4_SLOTS; ++i) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (soff[i] >= 0 && ((tid + i * 256) & 7) < c4n) v = *reinterpret_cast<const f32x4*>(d.x + soff[i] + cb);
            st[i] = v;
        }
        f32x4 wst[2];                       // this thread's pieces of the chunk's 288 weight quads
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + i * 256;    // quad q = (tap*8 + c4)*4 + n
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < 288) {
                const int grp = q >> 2, tap = grp >> 3, c4 = grp & 7;
                if (c4 < c4n) v = *reinterpret_cast<const f32x4*>(d.wp + ((size_t)(tap * cin4 + (cb >> 2) + c4) * CoutP + (q & 3)) * 4);
            }
            wst[i] = v;
        }
        __syncthreads();                    // previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < T4_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (soff[i] != -2) *reinterpret_cast<f32x4*>(halo + (f >> 3) * T4_PITCH + (f & 7) * 4) = st[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * 256 < 288) *reinterpret_cast<f32x4*>(wsh + (tid + i * 256) * 4) = wst[i];
        __syncthreads();
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const float* hp = hbase + (ky * T4_HW + kx) * T4_PITCH;
            const float* wt = wsh + tap * 128;
#pragma unroll 4
            for (int c4 = 0; c4 < c4n; ++c4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(hp + c4 * 4);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wt + c4 * 16 + n * 4);   // broadcast read
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[n] = fmaf(g[e], w[e], acc[n]);
                }
            }
        }
    }
    const int oy = ty * T4_TH + py, ox = tx * T4_TW + px;
    const bool inside = oy < d.Hout && ox < d.Wout;
    if (inside) {
        const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
        float* p = d.y + ((size_t)oy * pitch + ox) * d.Cy;
        if (d.bias != nullptr) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < ncols) acc[n] += d.bias[n];
        }
        if (ncols == 4) {
            if (d.accumulate) acc += *reinterpret_cast<const f32x4*>(p);
            *reinterpret_cast<f32x4*>(p) = acc;
        } else {
            for (int n = 0; n < ncols; ++n) {
                acc[n] = d.accumulate ? p[n] + acc[n] : acc[n];
                p[n] = acc[n];
            }
        }
    }
    // fused phase 1 of the BatchNorm backward of these columns (DipConvDesc.bnb_*, see conv_epilogue.h): one row of
    // {sum g*act'(z), sum g*act'(z)*xhat} per workgroup in bnb_partials_thin (channels 0..3)
    if (d.bnb_y != nullptr) {
        f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (inside) {
            const int pad = d.bnb_pad, Hi = d.Hout - 2 * pad, Wi = d.Wout - 2 * pad, Cs = d.bnb_Cs;
            const int iy = dip_reflect(oy - pad, Hi), ix = dip_reflect(ox - pad, Wi);
            const f32x4 yv = *reinterpret_cast<const f32x4*>(d.bnb_y + ((size_t)iy * Wi + ix) * d.bnb_Cy);
            const f32x4 mean = *reinterpret_cast<const f32x4*>(d.bnb_state), rstd = *reinterpret_cast<const f32x4*>(d.bnb_state + Cs),
                        sa = *reinterpret_cast<const f32x4*>(d.bnb_state + 2 * Cs), sb = *reinterpret_cast<const f32x4*>(d.bnb_state + 3 * Cs);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < ncols) {
                    const float z = fmaf(sa[n], yv[n], sb[n]);
                    const float gm = dip_mul_rn(acc[n], dip_act_grad(z, d.bnb_slope));
                    s1[n] = gm;
                    s2[n] = gm * ((yv[n] - mean[n]) * rstd[n]);
                }
            }
        }
        __syncthreads();                    // the halo is dead: reuse it for the tree (256 threads x 8 floats)
        dip_tree_sum8(halo, 1, 256, tid, 0, true, s1, s2);
        if (tid == 0) {
            float* o = d.bnb_partials_thin + (size_t)blockIdx.x * 2 * d.bnb_Cs;
            *reinterpret_cast<f32x4*>(o) = s1;
            *reinterpret_cast<f32x4*>(o + d.bnb_Cs) = s2;
        }
    }
}

}  // namespace

extern "C" int dip_conv_thin4_ntiles(int Hout, int Wout) { return dip_cdiv(Wout, T4_TW) * dip_cdiv(Hout, T4_TH); }

// columns [0, ncols) (ncols <= 4) of a 3x3 stride-1, undilated, transform-free convolution
extern "C" int dip_conv_thin4(const DipConvDesc* dp, int ncols, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.ks != 3 || d.stride != 1 || d.dil != 1 || d.tr.a != nullptr || ncols < 1 || ncols > 4 || (d.Cin & 3))
        DIP_FAIL("conv_thin4: unsupported configuration");
    if (d.bnb_y != nullptr && (d.bnb_partials_thin == nullptr || (d.bnb_Cy & 3) || (d.bnb_Cs & 3)))
        DIP_FAIL("conv_thin4: fused BatchNorm-backward partials need bnb_partials_thin and 4-aligned strides");
    const int ntx = dip_cdiv(d.Wout, T4_TW), nty = dip_cdiv(d.Hout, T4_TH);
    dip_launch_pair<DIP_FAM_THIN>(conv_thin4_kernel<false>, conv_thin4_kernel<true>, dim3(ntx * nty), dim3(256), 0, st, d, ntx,
                                  dip_round_up(d.Cout, 32), ncols);
    DIP_CHECK_LAUNCH();
    return 0;
}
