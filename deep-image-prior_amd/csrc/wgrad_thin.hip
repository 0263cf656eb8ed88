// Weight gradient of the thin layers (8 .. 64 input channels, <= 64 output channels, 3x3 / 5x5, stride 1 / 2): the
// high-resolution convs of the narrow nets (inpainting.ipynb:222-232 'library': 16 / 32 / 64 channels with 5x5 down filters;
// denoising.ipynb:143-150 snail), autograd's ConvolutionBackward weight + bias part of models/common.py:120:
//   dW[tap][c][o] = sum_q u[src(q, tap)][c] * dy[q][o],  db[o] = sum_q dy[q][o],   u = transform(x).
//
// conv_wgrad_kernel keeps a 32-channel x 32-column MFMA accumulator per tap: a 16 -> 16 layer fills a quarter of it, and
// its stride-2 5x5 form stages an 11 x 35-pixel x 32-channel halo (49 KB) for 64 output pixels -- 76 .. 104 us for the
// 0.5 .. 1 GFLOP layers of the library net (5 .. 10 TF), 42 % of the kernel time of a grouped library iteration.  Here
//   * the tile is v_mfma_f32_16x16x4_f32 with M = 16 input channels of ONE tap, N = 16 output channels, K = 4 pixels: a
//     16 -> 16 layer wastes nothing;
//   * the taps are spread over the workgroup's four waves (tap = wave, wave + 4, ..: 7 / 6 / 6 / 6 of a 5x5 filter), which
//     all read the same staged tiles: 8x16 output pixels of dy and their halo of x, 16 or 32 channels each, [pixel][channel]
//     with pitches that put the four pixel groups of a ds_read_b32 on disjoint banks;
//   * a workgroup owns a (<= 32 input channels) x (<= 32 output channels) block of dW for a strided list of pixel tiles and
//     writes one slab -- the layout of conv_wgrad_kernel, summed by dip_wgrad_reduce in a fixed order (deterministic).
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

typedef float f32x4w __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wt_map_src(int v, int n_in, int pad_mode) {
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, n_in);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), n_in - 1);
    return (v < 0 || v >= n_in) ? -1 : v;
}

template <int KS, int S, int CI, int CO>
struct WTCfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int HTH = (TH - 1) * S + KS, HTW = (TW - 1) * S + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int KK = KS * KS;
    static constexpr int NTW = (KK + 3) / 4;                          // taps per wave
    // pitches (floats): S * LDC * 4 bytes == 64 (mod 128) so that pixel groups k and k + 1 of a ds_read_b32 hit disjoint banks
    static constexpr int LDC = S == 1 ? (CI == 1 ? 16 : 48) : (CI == 1 ? 24 : 40);
    static constexpr int LDO = CO == 1 ? 16 : 48;
    static constexpr int U_FLOATS = NPIX * LDC;
    static constexpr int D_FLOATS = TH * TW * LDO;
    static constexpr int LDS_BYTES = (U_FLOATS + D_FLOATS) * 4;
    static constexpr int U_SLOTS = (NPIX * CI * 4 + 255) / 256;        // float4 loads per thread and tile (halo)
    static constexpr int D_SLOTS = (TH * TW * CO * 4 + 255) / 256;
};

// grid: x = pixel-tile walkers (= slabs), y = (input-channel block, output-channel block)
template <int KS, int S, int CI, int CO, bool GRP = false>
__global__ __launch_bounds__(256) void wgrad_thin_kernel(const DipWgradDesc d_, const int ntx, const int ntiles, const int ncob,
                                                         const int CinP, const int CoutP, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipWgradDesc, d);
    using C = WTCfg<KS, S, CI, CO>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Us = smem;
    float* Ds = smem + C::U_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, k = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int walker = blockIdx.x, nwalk = gridDim.x;
    const int cib = blockIdx.y / ncob, cob = blockIdx.y - cib * ncob;
    const int c0 = cib * 16 * CI, o0 = cob * 16 * CO;

    f32x4w acc[C::NTW][CI][CO];
#pragma unroll
    for (int i = 0; i < C::NTW; ++i)
#pragma unroll
        for (int a = 0; a < CI; ++a)
#pragma unroll
            for (int b = 0; b < CO; ++b) acc[i][a][b] = f32x4w{0.f, 0.f, 0.f, 0.f};
    float bsum[CO];
#pragma unroll
    for (int b = 0; b < CO; ++b) bsum[b] = 0.f;
    const bool do_bias = d.bias_partial != nullptr && cib == 0 && wave == 0;

    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;
    // staging roles: halo slot f -> (pixel f / (4 CI), channel group f % (4 CI)); 256 % (4 CI) == 0: fixed group per thread
    const int ucg = tid & (4 * CI - 1);
    const int uc = c0 + ucg * 4;
    const bool ucv = uc < d.Cin;
    f32x4w ta = f32x4w{1.f, 1.f, 1.f, 1.f}, tb = f32x4w{0.f, 0.f, 0.f, 0.f};
    if (has_tr && ucv) {
        // (Cin is a multiple of 4 in memory: pad channels of x are zero and their coefficients are never applied: c < Cin)
        ta = *reinterpret_cast<const f32x4w*>(d.tr.a + uc);
        tb = *reinterpret_cast<const f32x4w*>(d.tr.b + uc);
    }
    const int dog = tid & (4 * CO - 1);
    const int dcol = o0 + dog * 4;
    const bool dov = dcol < d.Cdy;

    int toff[C::NTW];                                      // LDS offset of this wave's i-th tap (wave-uniform)
#pragma unroll
    for (int i = 0; i < C::NTW; ++i) {
        const int tap = min(wave + 4 * i, C::KK - 1);
        toff[i] = ((tap / KS) * C::HTW + (tap % KS)) * C::LDC;
    }

    for (int tile = walker; tile < ntiles; tile += nwalk) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
        if (tile != walker) __syncthreads();               // the previous tile's reads are done
        // ---- stage: all loads of the tile first (clamped addresses), then transform + LDS ---------------------------------
        f32x4w uv[C::U_SLOTS], dv[C::D_SLOTS];
        bool uok[C::U_SLOTS], dok[C::D_SLOTS];
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = min(f, C::NPIX * 4 * CI - 1) / (4 * CI);
            const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
            const int sr = wt_map_src(ty * C::TH * S + hr - d.off, d.Hin, d.pad_mode);
            const int sc = wt_map_src(tx * C::TW * S + hc - d.off, d.Win, d.pad_mode);
            uok[i] = sr >= 0 && sc >= 0 && ucv;
            uv[i] = *reinterpret_cast<const f32x4w*>(d.x + (uok[i] ? ((size_t)sr * d.Win + sc) * d.Cx + uc : 0));
        }
#pragma unroll
        for (int i = 0; i < C::D_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int px = min(f, C::TH * C::TW * 4 * CO - 1) / (4 * CO);
            const int oy = ty * C::TH + (px >> 4), ox = tx * C::TW + (px & 15);
            dok[i] = oy < d.Hout && ox < d.Wout && dov;
            dv[i] = *reinterpret_cast<const f32x4w*>(d.dy + (dok[i] ? ((size_t)oy * d.Wout + ox) * d.Cdy + dcol : 0));
        }
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < C::NPIX * 4 * CI) {
                f32x4w u = uv[i];
                if (has_tr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = dip_act(fmaf(ta[e], u[e], tb[e]), slope);
                }
                if (!uok[i]) u = f32x4w{0.f, 0.f, 0.f, 0.f};          // padding zeros come AFTER the activation
                *reinterpret_cast<f32x4w*>(Us + (f / (4 * CI)) * C::LDC + ucg * 4) = u;
            }
        }
#pragma unroll
        for (int i = 0; i < C::D_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < C::TH * C::TW * 4 * CO)
                *reinterpret_cast<f32x4w*>(Ds + (f / (4 * CO)) * C::LDO + dog * 4) = dok[i] ? dv[i] : f32x4w{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        // ---- K loop: steps of 4 pixels (row r, columns 4 s + k); this wave's taps x CI x CO MFMAs per step.  No control flow
        // inside: a wave whose last tap does not exist (tap >= KK: waves 1..3 of a 5x5 filter) multiplies the last valid tap
        // again into an accumulator that is never stored -- a branch around it would keep hipcc from issuing a step's LDS
        // reads ahead of its MFMAs (first version: one read -> wait -> MFMA at a time, ~210 cycles per MFMA) -------------------
        for (int r = 0; r < C::TH; ++r) {
#pragma unroll
            for (int s = 0; s < C::TW / 4; ++s) {
                float bv[CO], av[C::NTW][CI];
                const float* ub = Us + ((r * S) * C::HTW + (4 * s + k) * S) * C::LDC + l15;
#pragma unroll
                for (int b = 0; b < CO; ++b) bv[b] = Ds[(r * C::TW + 4 * s + k) * C::LDO + b * 16 + l15];
#pragma unroll
                for (int i = 0; i < C::NTW; ++i)
#pragma unroll
                    for (int a = 0; a < CI; ++a) av[i][a] = ub[toff[i] + a * 16];
                if (do_bias) {
#pragma unroll
                    for (int b = 0; b < CO; ++b) bsum[b] += bv[b];
                }
#pragma unroll
                for (int i = 0; i < C::NTW; ++i)
#pragma unroll
                    for (int a = 0; a < CI; ++a)
#pragma unroll
                        for (int b = 0; b < CO; ++b)
                            acc[i][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][a], bv[b], acc[i][a][b], 0, 0, 0);
            }
        }
    }

    // ---- slab: accumulator register q of lane (l15, k) = dW[tap][c0 + 16 a + 4 k + q][o0 + 16 b + l15] ----------------------
    float* slab = d.partial + (size_t)walker * KS * KS * CinP * CoutP;
#pragma unroll
    for (int i = 0; i < C::NTW; ++i) {
        const int tap = wave + 4 * i;
        if (tap < C::KK) {
#pragma unroll
            for (int a = 0; a < CI; ++a)
#pragma unroll
                for (int b = 0; b < CO; ++b) {
                    const int o = o0 + 16 * b + l15;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = c0 + 16 * a + 4 * k + q;
                        if (c < CinP && o < CoutP) slab[((size_t)tap * CinP + c) * CoutP + o] = acc[i][a][b][q];
                    }
                }
        }
    }
    if (do_bias) {
#pragma unroll
        for (int b = 0; b < CO; ++b) {
            float v = bsum[b];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int o = o0 + 16 * b + l15;
            if (k == 0 && o < CoutP) d.bias_partial[(size_t)walker * CoutP + o] = v;
        }
    }
}

bool wthin_off() {
    static const bool off = getenv("DIP_WGRAD_NO_THIN") != nullptr;
    return off;
}

// channel blocks per workgroup: 32 input channels (CI = 2) when the layer has more than 16 and the halo stays small
// (stride 1); 32 output channels (CO = 2) when it has more than 16
void wthin_blocks(int Cin, int Cout, int stride, int* ci, int* co) {
    *ci = (Cin > 16 && stride == 1) ? 2 : 1;
    *co = Cout > 16 ? 2 : 1;
}

bool wthin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    if (wthin_off()) return false;
    if (ks != 3 && ks != 5) return false;
    if (stride != 1 && stride != 2) return false;
    if (Cin < 8 || Cin > 64 || Cout < 1 || Cout > 64) return false;
    static const int minpix = getenv("DIP_WGRAD_THIN_MIN_PIXELS") ? atoi(getenv("DIP_WGRAD_THIN_MIN_PIXELS")) : 4096;
    return Hout * Wout >= minpix;
}

int wthin_nsplit(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    int ci, co;
    wthin_blocks(Cin, Cout, stride, &ci, &co);
    const int blocks = dip_cdiv(Cin, 16 * ci) * dip_cdiv(Cout, 16 * co);
    const int nt = dip_cdiv(Wout, 16) * dip_cdiv(Hout, 8);
    static const int target = getenv("DIP_WGRAD_THIN_WGS") ? atoi(getenv("DIP_WGRAD_THIN_WGS")) : 512;
    int n = target / blocks;
    if (n > nt) n = nt;
    if (n < 1) n = 1;
    const long long slab = (long long)ks * ks * dip_round_up(Cin, 32) * dip_round_up(Cout, 32);
    while (n > 1 && (long long)n * slab > (64ll << 20)) n /= 2;
    return n;
}

template <int KS, int S, int CI, int CO>
int wthin_launch(const DipWgradDesc& d, hipStream_t st) {
    using C = WTCfg<KS, S, CI, CO>;
    auto kern = wgrad_thin_kernel<KS, S, CI, CO>;
    auto kern_g = wgrad_thin_kernel<KS, S, CI, CO, true>;
    static bool attr_set[16] = {};
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    const int ncib = dip_cdiv(d.Cin, 16 * CI), ncob = dip_cdiv(d.Cout, 16 * CO);
    if (d.nsplit < 1 || d.nsplit > ntx * nty) DIP_FAIL("wgrad_thin: nsplit out of range (use dip_wgrad_plan2)");
    dip_launch_pair<DIP_FAM_WGRAD>(kern, kern_g, dim3(d.nsplit, ncib * ncob), dim3(256), C::LDS_BYTES, st, d, ntx, ntx * nty, ncob,
                                   CinP, CoutP);
    DIP_CHECK_LAUNCH();
    return 0;
}

template <int KS, int S>
int wthin_launch_blocks(const DipWgradDesc& d, hipStream_t st) {
    int ci, co;
    wthin_blocks(d.Cin, d.Cout, S, &ci, &co);
    if (ci == 1) return co == 1 ? wthin_launch<KS, S, 1, 1>(d, st) : wthin_launch<KS, S, 1, 2>(d, st);
    if constexpr (S == 1) return co == 1 ? wthin_launch<KS, S, 2, 1>(d, st) : wthin_launch<KS, S, 2, 2>(d, st);
    DIP_FAIL("wgrad_thin: 32-channel input blocks are a stride-1 form");
}

}  // namespace

extern "C" int dip_wgrad_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    return wthin_shape_ok(Hout, Wout, Cin, Cout, ks, stride) ? 1 : 0;
}
extern "C" int dip_wgrad_thin_nsplit(int Hout, int Wout, int Cin, int Cout, int ks, int stride) {
    return wthin_nsplit(Hout, Wout, Cin, Cout, ks, stride);
}

extern "C" int dip_wgrad_thin_eligible(const DipWgradDesc* dp) {
    const DipWgradDesc& d = *dp;
    if (!wthin_shape_ok(d.Hout, d.Wout, d.Cin, d.Cout, d.ks, d.stride)) return 0;
    // (Cin itself need not be a multiple of 4: the float4 of a ragged last group reads the zero pad channels of x, and the slab
    // rows c >= Cin it produces are never read by dip_wgrad_reduce)
    if ((d.Cx & 3) || (d.Cdy & 3) || d.Cin > d.Cx || d.Cdy < d.Cout) return 0;
    return 1;
}

extern "C" int dip_wgrad_thin(const DipWgradDesc* dp, void* stream) {
    const DipWgradDesc& d = *dp;
    if (!dip_wgrad_thin_eligible(dp)) DIP_FAIL("wgrad_thin: shape not served (3x3 / 5x5, 8..64 input channels, <= 64 output channels)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.ks == 3) return d.stride == 1 ? wthin_launch_blocks<3, 1>(d, st) : wthin_launch_blocks<3, 2>(d, st);
    return d.stride == 1 ? wthin_launch_blocks<5, 1>(d, st) : wthin_launch_blocks<5, 2>(d, st);
}
