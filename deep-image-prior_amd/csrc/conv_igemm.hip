// Implicit-GEMM convolution on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32).
//
// One workgroup (4 waves) computes an 8x16-pixel x BN-channel output tile:
//   M = 128 output pixels, N = BN output channels, K = taps x input channels.
// K is walked in channel chunks; for each chunk the (transformed) input halo tile is staged ONCE
// in LDS ([pixel][channel], pixel pitch == 4 mod 8 dwords -> conflict-free ds_read_b128) and all
// KS*KS taps read shifted windows of it, so HBM/L2 sees each input pixel ~1.4x instead of 9x.
// The packed weights of one (chunk, tap) are a [cc/4][BN][4] slab, double-buffered in LDS.
// BatchNorm-apply + LeakyReLU of the PRODUCER layer, reflection/zero padding, the transposed
// (dilated) gather of the stride-2 data gradient, bias, and the BatchNorm partial statistics of
// the CONSUMER layer are all fused here, so activations cross HBM once per conv.
//
// K ordering trick: the 32x32x2 MFMA wants lane l to hold A[i=l&31][k=l>>5].  A lane reads FOUR
// consecutive channels with one ds_read_b128 (lanes 0-31: channels 8kk..8kk+3, lanes 32-63:
// 8kk+4..8kk+7) and feeds them to four MFMAs; B uses the same permutation ([c/4][n][c%4] slab),
// so the k-sum is merely re-ordered.
#include "dip_common.h"

namespace {

template <int KS, int S, int CCH, int BN>
struct Cfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int HTH = (TH - 1) * S + KS, HTW = (TW - 1) * S + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int CMAX = CCH + 8;
    static constexpr int LDP = CMAX + 4;  // == 4 (mod 8): 16 distinct 16-B slots per 16 pixels
    static constexpr int A_FLOATS = NPIX * LDP;
    static constexpr int B_FLOATS = CMAX * BN;
    static constexpr int WN = (BN >= 64) ? 2 : 1;
    static constexpr int WM = 4 / WN;
    static constexpr int MS = 4 / WM;
    static constexpr int NS = BN / 32 / WN;
    static constexpr int A_SLOTS = (NPIX * (CMAX / 4) + 255) / 256;
    static constexpr int B_SLOTS = ((CMAX / 4) * BN + 255) / 256;
    static constexpr int LDS_BYTES = (A_FLOATS + 2 * B_FLOATS + NPIX) * 4;
};

__device__ __forceinline__ int map_src(int v, int n_in, int dil, int reflect) {
    const int nv = (n_in - 1) * dil + 1;
    if (reflect) v = dip_reflect(v, nv);
    if (v < 0 || v >= nv) return -1;
    if (dil == 2) {
        if (v & 1) return -1;
        v >>= 1;
    }
    return v;
}

template <int KS, int S, int CCH, int BN>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const DipConvDesc d, const int ntx, const int ntiles,
                                                            const int CoutP, const int n_base) {
    using C = Cfg<KS, S, CCH, BN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;
    int* srcoff = reinterpret_cast<int*>(smem + C::A_FLOATS + 2 * C::B_FLOATS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int wn = wave % C::WN;
    const int wm = wave / C::WN;

    const int tile = dip_xcd_remap(blockIdx.x, ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int n0 = n_base + blockIdx.y * BN;

    // ---- per-tile source-pixel table (reflection / zero pad / dilation resolved once) ----
    for (int hp = tid; hp < C::NPIX; hp += 256) {
        const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
        const int sr = map_src(ty * C::TH * S + hr - d.off, d.Hin, d.dil, d.pad_mode);
        const int sc = map_src(tx * C::TW * S + hc - d.off, d.Win, d.dil, d.pad_mode);
        srcoff[hp] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
    }

    // ---- chunking of the input channels ----
    const int nfull = d.Cin / CCH, rem = d.Cin - nfull * CCH;
    int nchunks, last_cc;
    if (rem == 0) { nchunks = nfull; last_cc = CCH; }
    else if (rem <= 8 && nfull >= 1) { nchunks = nfull; last_cc = CCH + rem; }
    else { nchunks = nfull + 1; last_cc = rem; }
    const int cin4 = d.Cin >> 2;

    f32x16 acc[C::MS][C::NS];
#pragma unroll
    for (int i = 0; i < C::MS; ++i)
#pragma unroll
        for (int j = 0; j < C::NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS read bases (floats)
    int apix[C::MS];
#pragma unroll
    for (int ms = 0; ms < C::MS; ++ms) {
        const int sub = wm * C::MS + ms;          // 32-pixel sub-tile: rows 2*sub, 2*sub+1
        const int r = 2 * sub + (l31 >> 4), c = l31 & 15;
        apix[ms] = ((r * S) * C::HTW + c * S) * C::LDP + 4 * half;
    }
    int bcol[C::NS];
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) bcol[ns] = ((wn * C::NS + ns) * 32 + l31) * 4;

    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int cb = ch * CCH;
        const int cc = (ch == nchunks - 1) ? last_cc : CCH;
        const int c4n = cc >> 2;
        __syncthreads();  // previous chunk fully consumed (also orders srcoff writes on ch == 0)

        // ---- stage A: halo tile of this channel chunk, producer BN+LeakyReLU applied ----
        {
            const int nslots = C::NPIX * c4n;
            f32x4 v[C::A_SLOTS];
            int dst[C::A_SLOTS];
            int c4s[C::A_SLOTS];
#pragma unroll
            for (int i = 0; i < C::A_SLOTS; ++i) {
                const int f = tid + i * 256;
                dst[i] = -1;
                v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                c4s[i] = 0;
                if (f < nslots) {
                    const int hp = f / c4n, c4 = f - hp * c4n;
                    const int so = srcoff[hp];
                    dst[i] = hp * C::LDP + c4 * 4;
                    c4s[i] = so < 0 ? -1 : c4;
                    if (so >= 0)
                        v[i] = *reinterpret_cast<const f32x4*>(d.x + (size_t)so * d.Cx + cb + c4 * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < C::A_SLOTS; ++i) {
                if (dst[i] >= 0) {
                    f32x4 o = v[i];
                    if (has_tr && c4s[i] >= 0) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(d.tr.a + cb + c4s[i] * 4);
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(d.tr.b + cb + c4s[i] * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = dip_act(fmaf(a4[e], o[e], b4[e]), slope);
                    }
                    *reinterpret_cast<f32x4*>(As + dst[i]) = o;
                }
            }
        }
        // ---- stage B for tap 0 straight into buffer 0 ----
        {
            const int nb4 = c4n * BN;
#pragma unroll
            for (int i = 0; i < C::B_SLOTS; ++i) {
                const int f = tid + i * 256;
                if (f < nb4) {
                    const int c4 = f / BN, n = f - c4 * BN;
                    f32x4 w = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (n0 + n < CoutP)
                        w = *reinterpret_cast<const f32x4*>(
                            d.wp + ((size_t)(0 * cin4 + (cb >> 2) + c4) * CoutP + n0 + n) * 4);
                    *reinterpret_cast<f32x4*>(Bs + f * 4) = w;
                }
            }
        }
        __syncthreads();

#pragma unroll
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const float* Bcur = Bs + (tap & 1) * C::B_FLOATS;
            float* Bnxt = Bs + ((tap + 1) & 1) * C::B_FLOATS;
            const bool more = tap + 1 < KS * KS;
            // prefetch next tap's weights into registers
            f32x4 pre[C::B_SLOTS];
            const int nb4 = c4n * BN;
            if (more) {
#pragma unroll
                for (int i = 0; i < C::B_SLOTS; ++i) {
                    const int f = tid + i * 256;
                    if (f < nb4) {
                        const int c4 = f / BN, n = f - c4 * BN;
                        pre[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (n0 + n < CoutP)
                            pre[i] = *reinterpret_cast<const f32x4*>(
                                d.wp + ((size_t)((tap + 1) * cin4 + (cb >> 2) + c4) * CoutP + n0 + n) * 4);
                    }
                }
            }
            // ---- MFMA over this tap's cc channels ----
            const int tapoff = (ky * C::HTW + kx) * C::LDP;
            const int kk8 = cc >> 3;
            for (int kk = 0; kk < kk8; ++kk) {
                f32x4 a[C::MS], b[C::NS];
#pragma unroll
                for (int ms = 0; ms < C::MS; ++ms)
                    a[ms] = *reinterpret_cast<const f32x4*>(As + apix[ms] + tapoff + kk * 8);
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    b[ns] = *reinterpret_cast<const f32x4*>(Bcur + (kk * 2 + half) * (BN * 4) + bcol[ns]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                        for (int ns = 0; ns < C::NS; ++ns)
                            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms][j], b[ns][j], acc[ms][ns], 0, 0, 0);
            }
            if (cc & 4) {  // 4-channel tail: lanes 0-31 take channels cc-4, cc-3; lanes 32-63 cc-2, cc-1
                f32x2 a[C::MS], b[C::NS];
#pragma unroll
                for (int ms = 0; ms < C::MS; ++ms)
                    a[ms] = *reinterpret_cast<const f32x2*>(As + apix[ms] - 4 * half + tapoff + (cc - 4) + 2 * half);
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    b[ns] = *reinterpret_cast<const f32x2*>(Bcur + ((cc - 4) >> 2) * (BN * 4) + bcol[ns] + 2 * half);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                        for (int ns = 0; ns < C::NS; ++ns)
                            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms][j], b[ns][j], acc[ms][ns], 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < C::B_SLOTS; ++i) {
                    const int f = tid + i * 256;
                    if (f < nb4) *reinterpret_cast<f32x4*>(Bnxt + f * 4) = pre[i];
                }
                __syncthreads();
            }
        }
    }

    // ---- epilogue: bias, store, BatchNorm partial statistics ----
    const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
    float st_n[C::NS], st_k[C::NS], st_s1[C::NS], st_s2[C::NS];
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) {
        const int n = n0 + (wn * C::NS + ns) * 32 + l31;
        const float bias = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
        st_n[ns] = 0.f; st_k[ns] = 0.f; st_s1[ns] = 0.f; st_s2[ns] = 0.f;
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) {
            const int sub = wm * C::MS + ms;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int oy = ty * C::TH + 2 * sub + (m >> 4);
                const int ox = tx * C::TW + (m & 15);
                const bool valid = (oy < d.Hout) && (ox < d.Wout);
                float v = acc[ms][ns][r] + bias;
                if (valid && n < d.Cy) {
                    float* p = d.y + ((size_t)oy * pitch + ox) * d.Cy + n;
                    if (d.accumulate) v += *p;
                    *p = v;
                }
                if (valid) {  // shifted sums (shift = first value seen): cancellation-free variance
                    if (st_n[ns] == 0.f) st_k[ns] = v;
                    const float dv = v - st_k[ns];
                    st_n[ns] += 1.f;
                    st_s1[ns] += dv;
                    st_s2[ns] += dv * dv;
                }
            }
        }
    }
    if (d.stats != nullptr) {
        __syncthreads();  // LDS A/B no longer needed; reuse as reduction scratch
        float* red = smem;  // [WM][WN*NS*32][3]
#pragma unroll
        for (int ns = 0; ns < C::NS; ++ns) {
            float cn = st_n[ns];
            float mean = cn > 0.f ? st_k[ns] + st_s1[ns] / cn : 0.f;
            float M2 = cn > 0.f ? st_s2[ns] - st_s1[ns] * st_s1[ns] / cn : 0.f;
            const float on = __shfl_xor(cn, 32), om = __shfl_xor(mean, 32), oM = __shfl_xor(M2, 32);
            dip_chan(cn, mean, M2, on, om, oM);
            if (half == 0) {
                float* q = red + ((wm * (C::WN * C::NS * 32)) + (wn * C::NS + ns) * 32 + l31) * 3;
                q[0] = cn; q[1] = mean; q[2] = M2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            float cn = 0.f, mean = 0.f, M2 = 0.f;
#pragma unroll
            for (int w = 0; w < C::WM; ++w) {
                const float* q = red + (w * (C::WN * C::NS * 32) + tid) * 3;
                dip_chan(cn, mean, M2, q[0], q[1], q[2]);
            }
            const int n = n0 + tid;
            if (n < CoutP) {
                float* o = d.stats + (size_t)tile * 3 * CoutP + n;
                o[0] = cn; o[CoutP] = mean; o[2 * CoutP] = M2;
            }
        }
    }
}

template <int KS, int S, int CCH, int BN>
int launch(const DipConvDesc& d, hipStream_t st, int n_base, int grid_y) {
    using C = Cfg<KS, S, CCH, BN>;
    static bool attr_set = false;
    auto kern = conv_igemm_kernel<KS, S, CCH, BN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    const int CoutP = dip_round_up(d.Cout, 32);
    dim3 grid(ntiles, grid_y);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, CoutP, n_base);
    DIP_CHECK_LAUNCH();
    return 0;
}

// N is covered by full 128-wide blocks plus one narrower remainder launch (e.g. the 132-channel
// data gradient of the decoder convs = 128 + a 32-wide block instead of two 128-wide ones).
template <int KS, int S, int CCH>
int launch_bn(const DipConvDesc& d, hipStream_t st) {
    const int CoutP = dip_round_up(d.Cout, 32);
    const int nfull = CoutP / 128, rem = CoutP - nfull * 128;
    int rc = 0;
    if (nfull) rc = launch<KS, S, CCH, 128>(d, st, 0, nfull);
    if (rc || !rem) return rc;
    if (rem <= 32) return launch<KS, S, CCH, 32>(d, st, nfull * 128, 1);
    if (rem <= 64) return launch<KS, S, CCH, 64>(d, st, nfull * 128, 1);
    return launch<KS, S, CCH, 128>(d, st, nfull * 128, 1);
}

}  // namespace

extern "C" int dip_conv_ntiles(int Hout, int Wout) { return dip_cdiv(Wout, 16) * dip_cdiv(Hout, 8); }

extern "C" int dip_conv_igemm(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin <= 0 || d.Cin > d.Cx) DIP_FAIL("conv_igemm: channel strides must be multiples of 4");
    if (d.dil != 1 && d.dil != 2) DIP_FAIL("conv_igemm: dil must be 1 or 2");
    if (d.ks == 1 && d.stride == 1) return launch_bn<1, 1, 32>(d, st);
    if (d.ks == 3 && d.stride == 1) return launch_bn<3, 1, 32>(d, st);
    if (d.ks == 3 && d.stride == 2) return launch_bn<3, 2, 16>(d, st);
    if (d.ks == 5 && d.stride == 1) return launch_bn<5, 1, 16>(d, st);
    if (d.ks == 5 && d.stride == 2) return launch_bn<5, 2, 8>(d, st);
    DIP_FAIL("conv_igemm: unsupported kernel size / stride");
}
