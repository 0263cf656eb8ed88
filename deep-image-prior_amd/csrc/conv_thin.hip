// Thin-channel convolution (Cin, Cout <= 64) for the high-resolution layers of the narrow nets -- the 'library' inpainting
// net (inpainting.ipynb:222-232 of the reference: channels 16 / 32 / 64 at 448x704 .. 112x176, 5x5 down filters, 3x3 up
// filters) and the snail net (denoising.ipynb:143-150: 8 .. 64 channels); models/common.py:114-124 (ReflectionPad2d +
// Conv2d) and its stride-1 data gradient.
//
// Why a kernel of its own.  conv_igemm_kernel walks K in units of (16-channel chunk, tap): a barrier and a weight-slab DMA
// per unit, sized for 128 output columns (64 MFMAs per wave and unit).  At 32 columns a unit is 8 MFMAs, so a 5x5 layer is 25
// barrier + DMA round trips around 200 MFMAs: 37 us for the 1 GFLOP of 16 -> 16 channels at 224x352 (27 TF), half of whose
// 32x32 MFMA columns are padding.  Here
//   * ALL taps' weights of a channel chunk sit in LDS next to the chunk's halo: one barrier pair per CHUNK (one chunk for
//     <= 16 .. 64 input channels, by LDS budget; where even 16 channels x all taps do not fit -- 5x5 towards 64 columns --
//     the weights come in groups of filter rows), the K loop in between is straight-line MFMAs + LDS reads;
//   * the tile is v_mfma_f32_16x16x4_f32: 16 pixels x 16 output channels, so 16-channel layers waste nothing; a wave owns
//     two rows of an 8x16-pixel tile x all (<= 64) columns;
//   * K order: a lane reads FOUR consecutive channels with one ds_read_b128 (lane (pixel, q) channels 16 j + 4 q ..) and
//     feeds four MFMAs; the weights come in the same order from the packed layout [tap][c / 4][o][c % 4] (DipPackRec), so the
//     k-sum is merely re-ordered (as in conv_igemm.hip);
//   * 40 .. 60 VGPRs, 20 .. 60 KB of LDS: 2 .. 6 workgroups per CU hide each other's staging.
// Epilogue: bias, store, the consumer BatchNorm's {count, mean, M2} partials per tile (stats rows = tiles of 8x16 pixels:
// what dip_conv_plan reports for these shapes).  fp32 FMA chain on the fp32 MFMA, like every non-bf16 kernel here.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

typedef float f32x4t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int tmap_src(int v, int n_in, int pad_mode) {
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, n_in);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), n_in - 1);
    return (v < 0 || v >= n_in) ? -1 : v;
}

template <int KS, int S>
struct TCfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int HTH = (TH - 1) * S + KS, HTW = (TW - 1) * S + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int KK = KS * KS;
};

// NCB: 16-column blocks of the output (1 .. 4).  cc: input channels per chunk (16, 32 or 64).
template <int KS, int S, int NCB, bool GRP = false>
__global__ __launch_bounds__(256) void conv_thin_kernel(const DipConvDesc d_, const int ntx, const int ntiles, const int cc,
                                                        const int rpg, const int CoutP32, const int knock, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    using C = TCfg<KS, S>;
    constexpr int CP16 = NCB * 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldp = cc + 4;                               // pixel pitch: 16-byte slots of 8 consecutive pixels on distinct banks
    const int cc4 = cc >> 2;
    const int csh = cc == 64 ? 4 : (cc == 32 ? 3 : 2);    // log2(cc4): slot -> (pixel, channel group) by shifts
    float* Hs = smem;                                     // [NPIX][ldp]
    float* Ws = smem + C::NPIX * ldp;                     // [rpg * KS taps][cc / 4][CP16][4]: rpg filter rows at a time
    float* Rs = Ws + rpg * KS * cc * CP16;                // [4 waves][3][CP16] statistics hand-over

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = dip_xcd_remap(blockIdx.x, ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int cin4 = d.Cin >> 2;

    f32x4t acc[2][NCB];
#pragma unroll
    for (int pg = 0; pg < 2; ++pg)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[pg][cb] = f32x4t{0.f, 0.f, 0.f, 0.f};

    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;
    const int c4s = tid & (cc4 - 1);                      // this thread's 4-channel group in the halo staging (256 % cc4 == 0)
    const int nch = (d.Cin + cc - 1) / cc;

    for (int ch = 0; ch < nch; ++ch) {
        const int c0 = ch * cc;
        if (ch > 0) __syncthreads();                      // the previous chunk's reads are done
        // ---- halo of the chunk: producer BatchNorm + activation applied on the way, padding zeros after it --------------
        {
            const int c = c0 + c4s * 4;
            const bool cvalid = c < d.Cin;
            f32x4t ta = f32x4t{1.f, 1.f, 1.f, 1.f}, tb = f32x4t{0.f, 0.f, 0.f, 0.f};
            if (has_tr && cvalid) {
                ta = *reinterpret_cast<const f32x4t*>(d.tr.a + c);
                tb = *reinterpret_cast<const f32x4t*>(d.tr.b + c);
            }
            const int nslots = C::NPIX * cc4;
            // batches of 8 slots per thread: all eight loads are issued (clamped addresses, no control flow around them)
            // before the first one is used -- one memory round trip per batch instead of one per slot
            for (int f0 = tid; f0 < nslots && !(knock & 2); f0 += 256 * 8) {       // (knock: timing-only switches, DIP_THIN_KNOCK)
                f32x4t v[8];
                bool ok[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = f0 + i * 256;
                    const int hp = (f < nslots ? f : tid) >> csh;
                    const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
                    const int sr = tmap_src(ty * C::TH * S + hr - d.off, d.Hin, d.pad_mode);
                    const int sc = tmap_src(tx * C::TW * S + hc - d.off, d.Win, d.pad_mode);
                    ok[i] = sr >= 0 && sc >= 0 && cvalid;
                    v[i] = *reinterpret_cast<const f32x4t*>(d.x + (ok[i] ? ((size_t)sr * d.Win + sc) * d.Cx + c : 0));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = f0 + i * 256;
                    if (f < nslots) {
                        f32x4t u = v[i];
                        if (has_tr) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) u[e] = dip_act(fmaf(ta[e], u[e], tb[e]), slope);
                        }
                        if (!ok[i]) u = f32x4t{0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4t*>(Hs + (f >> csh) * ldp + c4s * 4) = u;
                    }
                }
            }
        }
        for (int r0 = 0; r0 < KS; r0 += rpg) {            // groups of rpg filter rows (all of them when the weights fit)
            const int nrows = min(rpg, KS - r0), tap0 = r0 * KS, ntap = nrows * KS;
            if (r0 > 0) __syncthreads();                  // the previous group's weight reads are done
            // ---- the group's weights of the chunk: [tap][c / 4][o][c % 4], CP16 columns of the CoutP32 packed ones -----
            {
                const int nslots = ntap * cc4 * CP16;
                for (int g0 = tid; g0 < nslots && !(knock & 4); g0 += 256 * 8) {          // batches of 8 loads in flight, as for the halo
                    f32x4t w[8];
                    bool ok[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int g = g0 + i * 256;
                        const int gg = g < nslots ? g : tid;
                        const int o = gg % CP16;
                        const int rest = gg / CP16;
                        const int c4 = rest & (cc4 - 1);
                        const int tap = tap0 + (rest >> csh);
                        const int gc4 = (c0 >> 2) + c4;
                        ok[i] = gc4 < cin4 && o < CoutP32;
                        w[i] = *reinterpret_cast<const f32x4t*>(d.wp + (ok[i] ? ((size_t)(tap * cin4 + gc4) * CoutP32 + o) * 4 : 0));
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int g = g0 + i * 256;
                        if (g < nslots) *reinterpret_cast<f32x4t*>(Ws + (size_t)g * 4) = ok[i] ? w[i] : f32x4t{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
            __syncthreads();
            // ---- K loop: (tap, 16 channels) steps, 8 * NCB MFMAs each; the next step's fragments are read under them ----
            const int nj = cc >> 4;
            const float* arow0 = Hs + ((2 * wave) * S * C::HTW + l15 * S) * ldp + 4 * q;
            const float* arow1 = arow0 + S * C::HTW * ldp;
            const float* brow = Ws + (q * CP16 + l15) * 4;
            f32x4t a0, a1, b[NCB];
            auto readA = [&](int tap, int j) {
                const int ky = tap / KS, kx = tap - ky * KS;
                const int o = (ky * C::HTW + kx) * ldp + 16 * j;
                a0 = *reinterpret_cast<const f32x4t*>(arow0 + o);
                a1 = *reinterpret_cast<const f32x4t*>(arow1 + o);
            };
            auto readB = [&](int tl, int j) {              // tl: tap index inside the group
                const float* p = brow + (size_t)(tl * cc4 + 4 * j) * CP16 * 4;
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) b[cb] = *reinterpret_cast<const f32x4t*>(p + cb * 64);
            };
            readA(tap0, 0);
            readB(0, 0);
            const int nsteps = (knock & 1) ? 0 : ntap * nj;
            int tl = 0, j = 0;
            for (int s = 0; s < nsteps; ++s) {
                const f32x4t ca0 = a0, ca1 = a1;
                f32x4t cb_[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) cb_[cb] = b[cb];
                if (++j == nj) { j = 0; ++tl; }
                if (s + 1 < nsteps) {
                    readA(tap0 + tl, j);
                    readB(tl, j);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) {
                        acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca0[e], cb_[cb][e], acc[0][cb], 0, 0, 0);
                        acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca1[e], cb_[cb][e], acc[1][cb], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue: accumulator register r of lane (l15, q) = pixel (row 2 wave + pg, column 4 q + r), channel cb * 16 + l15
    const int oy0 = ty * C::TH + 2 * wave;
    const int oxb = tx * C::TW + 4 * q;
    float cnt[NCB], mean[NCB], M2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int n = cb * 16 + l15;
        const float bias = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
        float k0 = 0.f, s1 = 0.f, s2 = 0.f, cn = 0.f;
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int oy = oy0 + pg;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ox = oxb + r;
                const float v = acc[pg][cb][r] + bias;
                if (oy < d.Hout && ox < d.Wout) {
                    if (n < d.Cy) d.y[((size_t)oy * d.Wout + ox) * d.Cy + n] = v;
                    if (cn == 0.f) k0 = v;
                    const float dv = v - k0;
                    cn += 1.f;
                    s1 += dv;
                    s2 = fmaf(dv, dv, s2);
                }
            }
        }
        cnt[cb] = cn;
        mean[cb] = cn > 0.f ? k0 + s1 / cn : 0.f;
        M2[cb] = cn > 0.f ? s2 - s1 * s1 / cn : 0.f;
    }
    if (d.stats == nullptr || (knock & 8)) return;
    // the four lanes (q) of a column, then the four waves, combined in a fixed order (Chan et al.)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int m = 16; m <= 32; m <<= 1) {
            const float on = __shfl_xor(cnt[cb], m), om = __shfl_xor(mean[cb], m), oM = __shfl_xor(M2[cb], m);
            // (both partners compute the same combination: order by lane so that the result is identical in both)
            float na = (lane & m) ? on : cnt[cb], ma = (lane & m) ? om : mean[cb], Ma = (lane & m) ? oM : M2[cb];
            const float nb = (lane & m) ? cnt[cb] : on, mb = (lane & m) ? mean[cb] : om, Mb = (lane & m) ? M2[cb] : oM;
            dip_chan(na, ma, Ma, nb, mb, Mb);
            cnt[cb] = na; mean[cb] = ma; M2[cb] = Ma;
        }
        if (q == 0) {
            float* o = Rs + (wave * 3) * CP16 + cb * 16 + l15;
            o[0] = cnt[cb]; o[CP16] = mean[cb]; o[2 * CP16] = M2[cb];
        }
    }
    __syncthreads();
    if (tid < CP16) {
        float na = Rs[tid], ma = Rs[CP16 + tid], Ma = Rs[2 * CP16 + tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) dip_chan(na, ma, Ma, Rs[(w * 3) * CP16 + tid], Rs[(w * 3 + 1) * CP16 + tid], Rs[(w * 3 + 2) * CP16 + tid]);
        float* o = d.stats + (size_t)tile * 3 * CoutP32 + tid;
        o[0] = na; o[CoutP32] = ma; o[2 * CoutP32] = Ma;
    }
}

// (channels per chunk, filter rows per weight group): the pair with the most MFMAs between two barriers whose halo + weights
// fit the LDS budget (64 KB: two workgroups per CU); failing that, the smallest pair that fits at all; cc == 0: none does
template <int KS, int S>
int thin_cc(int Cin, int ncb, int* rpg_out, int* lds_bytes) {
    using C = TCfg<KS, S>;
    const int cin16 = dip_round_up(Cin, 16);
    static const int budget = getenv("DIP_THIN_LDS_KB") ? atoi(getenv("DIP_THIN_LDS_KB")) * 1024 : 64 * 1024;
    int best = 0, best_work = 0, fb = 0, fb_bytes = 1 << 30, fb_rpg = 0;
    for (int cc = 16; cc <= 64 && cc <= cin16; cc *= 2)
        for (int rpg = 1; rpg <= KS; ++rpg) {
            if (rpg != 1 && rpg != KS && rpg != (KS + 1) / 2) continue;
            const int bytes = (C::NPIX * (cc + 4) + rpg * KS * cc * ncb * 16 + 4 * 3 * ncb * 16) * 4;
            if (bytes <= budget && cc * rpg > best_work) { best = cc; best_work = cc * rpg; *rpg_out = rpg; *lds_bytes = bytes; }
            if (bytes <= 150 * 1024 && bytes < fb_bytes) { fb = cc; fb_bytes = bytes; fb_rpg = rpg; }
        }
    if (best) return best;
    if (fb) { *rpg_out = fb_rpg; *lds_bytes = fb_bytes; }
    return fb;
}

bool thin_off() {
    static const bool off = getenv("DIP_CONV_NO_THIN") != nullptr;
    return off;
}
int thin_min_pixels() {
    static const int v = getenv("DIP_THIN_MIN_PIXELS") ? atoi(getenv("DIP_THIN_MIN_PIXELS")) : 4625;      // below: conv_small
    return v;
}

// shape part of the eligibility (what dip_conv_plan can know)
bool thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    if (thin_off()) return false;
    if (ks != 3 && ks != 5) return false;
    if (stride != 1 && stride != 2) return false;
    if (Cin < 1 || Cin > 64 || Cout < 1 || Cout > 64) return false;
    if (Hout * Wout < thin_min_pixels()) return false;
    // MEASURED (tools/thin_sweep.py on the 'library' net's shapes, profiles/r06_thin_sweep_conv_thin.txt; us, this kernel
    // against the register-staged / LDS-DMA kernels it replaces): it wins where ONE chunk holds all channels and all taps'
    // weights (<= 51 KB) --  16>16 5x5 @224x352 21.6 / 36.3 (data gradient 20.2 / 33.4), 16>32 5x5 s2 24.8 / 32.6, 32>16 3x3
    // @448x704 56.9 / 85.9 (data gradient 16>32: 52.2 / 68.6) -- is level at 32>32 5x5 (31.8 / 31.5) and loses with two
    // chunks or few tiles: 64>32 3x3 @224x352 61.8 / 52.1, 64>64 5x5 @56x88 97.6 / 31.8 (77 tiles, 4 column blocks: one heavy
    // workgroup per CU where split-K spreads the layer), and with 1..4 input channels (15 of 16 K lanes idle: 43 / 36).
    // DIP_THIN_ALL=1 lifts the two bounds (read at every call, unlike the other switches: tests/test_thin_gpu.py runs the
    // kernel on every shape it serves without changing what the rest of the suite exercises)
    if (getenv("DIP_THIN_ALL") != nullptr) return true;
    if (Cin < 8 || Cin * dip_round_up(Cout, 16) * ks * ks > 12800) return false;
    return true;
}

template <int KS, int S, int NCB>
int thin_launch(const DipConvDesc& d, hipStream_t st) {
    using C = TCfg<KS, S>;
    int lds = 0, rpg = KS;
    const int cc = thin_cc<KS, S>(d.Cin, NCB, &rpg, &lds);
    if (cc == 0) DIP_FAIL("conv_thin: the halo + weights of this shape do not fit into LDS");
    auto kern = conv_thin_kernel<KS, S, NCB>;
    auto kern_g = conv_thin_kernel<KS, S, NCB, true>;
    static bool attr_set[16] = {};
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, 150 * 1024);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    static const int knock = getenv("DIP_THIN_KNOCK") ? atoi(getenv("DIP_THIN_KNOCK")) : 0;      // timing-only knock-outs (wrong results)
    dip_launch_pair<DIP_FAM_CONV>(kern, kern_g, dim3(ntx * nty), dim3(256), (size_t)lds, st, d, ntx, ntx * nty, cc, rpg,
                                  dip_round_up(d.Cout, 32), knock);
    DIP_CHECK_LAUNCH();
    return 0;
}

template <int KS, int S>
int thin_launch_ncb(const DipConvDesc& d, hipStream_t st) {
    const int ncb = dip_cdiv(d.Cy > d.Cout ? d.Cy : d.Cout, 16);
    if (ncb == 1) return thin_launch<KS, S, 1>(d, st);
    if (ncb == 2) return thin_launch<KS, S, 2>(d, st);
    if (ncb == 3) return thin_launch<KS, S, 3>(d, st);
    return thin_launch<KS, S, 4>(d, st);
}

}  // namespace

// shape-only: used by dip_conv_plan (one pass, stats rows = 8x16-pixel tiles)
extern "C" int dip_conv_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    return thin_shape_ok(Hout, Wout, Cin, Cout, ks, stride) ? 1 : 0;
}

// 1 when dip_conv_igemm runs `d` on conv_thin_kernel: a thin shape, stride-1 / stride-2 forward or a stride-1 data gradient
// (dil == 1), one pass, plain store (no accumulate, no row pitch, no fused BatchNorm-backward partials)
extern "C" int dip_conv_thin_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    if (!thin_shape_ok(d.Hout, d.Wout, d.Cin, d.Cout, d.ks, d.stride)) return 0;
    if (d.dil != 1 || d.accumulate || d.y_pitch > 0 || d.bnb_y != nullptr || d.ksplit > 1) return 0;
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cy < d.Cout || d.Cy > 64 || d.Cin > d.Cx) return 0;
    return 1;
}

extern "C" int dip_conv_thin(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    if (!dip_conv_thin_eligible(dp)) DIP_FAIL("conv_thin: shape / mode not served (3x3 or 5x5, <= 64 channels, dil 1, one pass)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.ks == 3) return d.stride == 1 ? thin_launch_ncb<3, 1>(d, st) : thin_launch_ncb<3, 2>(d, st);
    return d.stride == 1 ? thin_launch_ncb<5, 1>(d, st) : thin_launch_ncb<5, 2>(d, st);
}
