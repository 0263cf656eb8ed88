// Latency-optimised implicit-GEMM convolution for the LOW-RESOLUTION scales of the hour-glass
// (<= 64x64 outputs: the three inner scales of the default net; models/skip.py:57-91 at depth >= 2).
//
// Why a second conv kernel.  conv_igemm_dma_kernel computes 128-pixel x 128-channel tiles; a 32x32
// layer has 8 of them, so it needs split-K over blockIdx.z, a workspace, a splitk_finish launch and a
// bn_finalize launch: three dependent launches (each a DMA pipeline prologue or a chain of dependent
// global round trips) for 0.3 GFLOP -- 30..45 us per layer where the matrix pipe needs 2 us, and the
// low-resolution walk of an iteration is ~35 such layers (profiles/r03_*timeline*).  Here a layer is
// ONE launch:
//   * work item = one wave: 32 output pixels x 32 output channels x a slice of K; the operands go
//     straight from L2 into the MFMA registers (global_load_dwordx4: lane (pixel, half) reads 4
//     consecutive channels -> four v_mfma_f32_32x32x2_f32, the K permutation of conv_igemm.hip), no
//     LDS staging, no barriers in the K loop, a ring of 3 x 4 K-steps of loads in flight per wave;
//   * split-K INSIDE the workgroup: 1, 2 or 4 waves share a tile and sum their accumulators through
//     LDS in a fixed order (deterministic), so there is no workspace and no finish launch;
//   * epilogue: bias, store, BatchNorm partial statistics {count, mean, M2} per workgroup row -- PARTIALS ONLY: the engine
//     emits a dip_bn_finalize launch after every dip_conv_small (the "last workgroup to arrive finalises" form of round 4,
//     bn_ticket.h, lives on in the opt-in *_fin entry points of other kernels; this kernel never had ticket code);
//   * data-gradient launches: phase 1 of the BatchNorm(+activation) backward of the conv's input
//     (DipConvDesc.bnb_*) rides in the epilogue as partial rows too (dip_bn_bwd_finalize2 follows); stride-2 transposed convs walk the four output-parity classes separately, so a
//     wave only visits the taps that hit non-zero positions of the dilated gradient (9 taps per 4
//     pixels instead of 36, as the phase mode of the LDS-DMA kernel).
// Any stride-1/2 1x1 / 3x3 convolution or data gradient with reflection / zero / replication padding,
// any activation, 1..160 output channels; the engine uses it below DIP_SMALL_MAX_PIXELS output pixels.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

constexpr int SM_TR_MAX = 512;      // input channels whose BatchNorm coefficients are cached in LDS
constexpr int SM_MAX_TAPS = 27;     // ring mode: up to 3 mirror sources x 9 taps per target pixel
constexpr int SM_BIG_TAPS = 49;     // KBIG instantiations: 5x5 / 7x7 filters (filter_size_down / _up = 5, 7: inpainting.ipynb:222-232)

struct SmallGeom;
template <class F> __host__ __device__ inline void dip_ptrs(SmallGeom&, F&) {}      // (dip_group.h: no pointers inside)
struct SmallGeom {
    int ncls;                 // 1, or 4 output-parity classes (dil == 2: transposed stride-2 conv)
    int Hc[4], Wc[4];         // pixels of class c = (py, px): rows py, py+2, ..; columns px, px+2, ..
    int gstart[5];            // first 32-pixel group of each class; gstart[ncls] = ngroups
    int ngroups;              // = grid.x = rows of the partial-statistics buffers
    int nblk;                 // 32-column blocks of the output = grid.y
    int nw;                   // waves per workgroup = K slices of its tile (4, 8, 16)
    int ring;                 // 1: dip_conv_dgrad_ring -- the tile's pixels are the frame rows 1, H-2 / columns 1, W-2 of
                              // an H x W image and every pixel sums the data gradient of the reflection-padded ring
                              // positions that mirror onto it (<= 3 sources x 9 taps), accumulated into y
};

// frame pixel i of an H x W image: rows 1 and H-2 (W pixels each), then columns 1 and W-2 without those rows
__device__ __forceinline__ void sm_ring_pixel(int i, int H, int W, int& r, int& c) {
    if (i < W) { r = 1; c = i; return; }
    i -= W;
    if (i < W) { r = H - 2; c = i; return; }
    i -= W;
    const int col = i < H - 2 ? 1 : W - 2;
    if (i >= H - 2) i -= H - 2;
    // rows 0 .. H-1 without 1 and H-2
    r = i == 0 ? 0 : (i < H - 3 ? i + 1 : H - 1);
    c = col;
}

// K steps (8 input channels of one tap = 4 MFMAs) a wave keeps in flight at once; 16 waves x 64 lanes x (8 * SMAX + ..)
// registers must fit the 512-register file of a SIMD four times
template <int NW> struct SmCfg { static constexpr int SMAX = NW == 16 ? 10 : 18; };

__device__ __forceinline__ int sm_map_src(int v, int n_in, int dil, int pad_mode) {
    const int nv = (n_in - 1) * dil + 1;
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, nv);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), nv - 1);
    if (v < 0 || v >= nv) return -1;
    if (dil == 2) {
        if (v & 1) return -1;
        v >>= 1;
    }
    return v;
}

// "the two loads of this K step have landed" (N younger loads may still be in flight); the registers are in/out operands
// so that no use of them can be scheduled above the wait
template <int N>
__device__ __forceinline__ void sm_wait(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// TR: 0 = no input transform, 1 = BatchNorm + LeakyReLU / identity (slope in (0, 1]), 2 = BatchNorm + Swish / ELU
// One workgroup = one 32-pixel x 32-channel output tile; its NW waves split K and are summed through LDS.
// KBIG: tap tables sized for 5x5 / 7x7 filters (6 KB of LDS more; the 1x1 / 3x3 instantiations keep their occupancy)
template <int TR, int NW, bool RING = false, bool GRP = false, bool KBIG = false>
__global__ __launch_bounds__(64 * NW) void conv_small_kernel(const DipConvDesc d_, const SmallGeom g, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    constexpr int SMAX = SmCfg<NW>::SMAX;
    constexpr int RPW = 16 / NW;                // accumulator registers a wave reduces across the K slices
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [NW][16][64] K-slice partial tiles
    __shared__ __attribute__((aligned(16))) float tra[TR ? SM_TR_MAX : 4];
    __shared__ __attribute__((aligned(16))) float trb[TR ? SM_TR_MAX : 4];
    constexpr int MAXT = RING ? SM_MAX_TAPS : (KBIG ? SM_BIG_TAPS : 9);
    __shared__ int srcoff[MAXT][32];            // source pixel of (tap, output pixel), -1: padding zero / outside
    __shared__ int yoff[32];                    // output pixel offset (oy * pitch + ox) of the 32 pixels, -1: none
    __shared__ int boff[32];                    // mirror pixel of the fused BatchNorm backward (bnb_y index), -1: none
    __shared__ int taps[NW][MAXT + 1];
    __shared__ float fin[16 * 64];              // the summed tile, handed to wave 0

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.x;
    const int n0 = blockIdx.y * 32;
    const int KK = d.ks * d.ks;
    const int CoutP = g.nblk * 32;

    // BatchNorm coefficients of the producer: issued first, they land while the tables below are built
    if (TR) {
        for (int c = tid; c < d.Cin; c += NT) { tra[c] = d.tr.a[c]; trb[c] = d.tr.b[c]; }
    }
    // ---- the tile's 32 output pixels and their source pixels, one table entry per thread -------------
    int Hc = g.Hc[0], Wc = g.Wc[0], gs = 0, py = 0, px = 0, pstep = 1;
    if (g.ncls > 1) {
        pstep = 2;
        if (group >= g.gstart[3]) { Hc = g.Hc[3]; Wc = g.Wc[3]; gs = g.gstart[3]; py = 1; px = 1; }
        else if (group >= g.gstart[2]) { Hc = g.Hc[2]; Wc = g.Wc[2]; gs = g.gstart[2]; py = 1; px = 0; }
        else if (group >= g.gstart[1]) { Hc = g.Hc[1]; Wc = g.Wc[1]; gs = g.gstart[1]; py = 0; px = 1; }
    }
    const int KV = RING ? 3 * KK : KK;        // (virtual) taps: ring mode walks up to 3 mirror sources per pixel
    for (int e = tid; e < (KV + 1) * 32; e += NT) {
        const int t = e >> 5, p = e & 31;
        const int pi = (group - gs) * 32 + p;
        if constexpr (RING) {
            // target pixel (r, c) of the frame; source s = t / KK: 0 = the ring position above / below it (row -1 mirrors
            // onto row 1, row H onto H-2), 1 = left / right of it, 2 = the corner; the ring position's data gradient
            // is the zero-padded 3x3 correlation of dy with the flipped filter at (rho, gam), rho / gam in -1 .. H / W
            const int H = d.Hout, W = d.Wout;
            const bool pvalid = pi < 2 * W + 2 * (H - 2);
            int r = 0, c = 0;
            if (pvalid) sm_ring_pixel(pi, H, W, r, c);
            if (t < KV) {
                const int src = t / KK, tt = t - src * KK;
                const int ky = tt / d.ks, kx = tt - ky * d.ks;
                const int mr = r == 1 ? -1 : (r == H - 2 ? H : -2), mc = c == 1 ? -1 : (c == W - 2 ? W : -2);   // -2: none
                const int rho = src == 1 ? r : mr, gam = src == 0 ? c : mc;
                const int sr = rho + ky - d.off, sc = gam + kx - d.off;
                const bool ok = pvalid && rho != -2 && gam != -2 && sr >= 0 && sr < d.Hin && sc >= 0 && sc < d.Win;
                srcoff[t][p] = ok ? sr * d.Win + sc : -1;
            } else {
                const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
                yoff[p] = pvalid ? r * pitch + c : -1;
                boff[p] = -1;
            }
        } else {
        const bool pvalid = pi < Hc * Wc;
        const int iy = pvalid ? pi / Wc : 0, ix = pvalid ? pi - iy * Wc : 0;
        const int oy = iy * pstep + py, ox = ix * pstep + px;
        if (t < KK) {
            const int ky = t / d.ks, kx = t - ky * d.ks;
            const int sr = sm_map_src(oy * d.stride + ky - d.off, d.Hin, d.dil, d.pad_mode);
            const int sc = sm_map_src(ox * d.stride + kx - d.off, d.Win, d.dil, d.pad_mode);
            srcoff[t][p] = (pvalid && sr >= 0 && sc >= 0) ? sr * d.Win + sc : -1;
        } else {
            const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
            yoff[p] = pvalid ? oy * pitch + ox : -1;
            int bo = -1;
            if (d.bnb_y != nullptr && pvalid) {
                const int Hi = d.Hout - 2 * d.bnb_pad, Wi = d.Wout - 2 * d.bnb_pad;
                bo = dip_reflect(oy - d.bnb_pad, Hi) * Wi + dip_reflect(ox - d.bnb_pad, Wi);
            }
            boff[p] = bo;
        }
        }
    }
    __syncthreads();
    // taps no pixel of the tile can see (zero padding; the other parities of a dilated gradient) are skipped
    int ntv = 0;
    for (int t = 0; t < KV; ++t) {
        if (__ballot(srcoff[t][l31] >= 0) != 0ull) {
            if (lane == 0) taps[w][ntv] = t;
            ++ntv;
        }
    }

    // ---- K loop: steps t in [t0, t1), t = (valid tap index, 8-channel group); ALL loads of a chunk of SMAX steps are
    // issued before the first MFMA (a low-resolution layer is latency-bound: memory-level parallelism, not reuse) -------
    const int nq = (d.Cin + 7) >> 3;
    const int T = ntv * nq;
    const int t0 = (int)(((long long)T * w) / NW), t1 = (int)(((long long)T * (w + 1)) / NW);
    const int cin4 = d.Cin >> 2;
    const float slope = d.tr.slope;
    const float* wcol = d.wp + (size_t)(n0 + l31) * 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int cs = t0; cs < t1; cs += SMAX) {                 // (one chunk unless Cin > 8 * SMAX * NW / taps)
        f32x4 A[SMAX], B[SMAX];
        unsigned amask = 0;
        int lti = cs / nq, lj = cs - lti * nq;
        // The loads are inline asm (volatile: issued in this order, before any wait below) and every step waits for
        // exactly its own pair (loads retire in order): left to hipcc's scheduler, the first steps' loads were waited for
        // before the rest was issued -- three memory round trips per slice instead of one.
#pragma unroll
        for (int i = 0; i < SMAX; ++i) {
            const bool on = cs + i < t1;                                 // wave-uniform
            const int vtap = taps[w][on ? lti : 0];
            const int so = srcoff[vtap][l31];
            const int tap = !RING ? vtap : (vtap >= 2 * KK ? vtap - 2 * KK : (vtap >= KK ? vtap - KK : vtap));   // its filter tap
            const int c = 8 * lj + 4 * half;
            const bool cv = on && c < d.Cin;                             // (the 4-channel tail of a 132-channel input)
            const bool av = cv && so >= 0;
            // unconditional loads from clamped addresses + selects: no control flow around a load
            const float* pa = d.x + (size_t)(av ? so : 0) * d.Cx + (cv ? c : 0);
            const float* pb = wcol + (size_t)((on ? tap : 0) * cin4 + (cv ? (c >> 2) : 0)) * CoutP * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(A[i]) : "v"(pa));
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(B[i]) : "v"(pb));
            amask |= av ? (1u << i) : 0u;
            if (++lj == nq) { lj = 0; ++lti; }
        }
        int cj = cs - (cs / nq) * nq;
        dip_static_for<0, SMAX>([&](auto I) {
            constexpr int i = decltype(I)::value;
            sm_wait<2 * (SMAX - 1 - i)>(A[i], B[i]);                     // (unconditional: the chunk's registers are
            if (cs + i < t1) {                                          //  quiet when its last step is through)
                f32x4 a = A[i];
                const bool av = (amask >> i) & 1u;
                if (TR) {
                    const int c = 8 * cj + 4 * half;
                    const int mc = c < d.Cin ? c : 0;
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + mc), b4 = *reinterpret_cast<const f32x4*>(trb + mc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tv = fmaf(a4[e], a[e], b4[e]);
                        a[e] = TR == 1 ? dip_act_leaky(tv, slope) : dip_act(tv, slope);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = av ? a[e] : 0.f;      // padding zeros come AFTER the activation
                const f32x4 b = B[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
            }
            if (++cj == nq) cj = 0;
        });
    }

    // ---- sum of the K slices: wave w adds accumulator registers [w * RPW, (w + 1) * RPW) over the slices in order,
    // wave 0 collects the finished tile ---------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int r = w * RPW + q;
        float v = red[r * 64 + lane];
#pragma unroll
        for (int s2 = 1; s2 < NW; ++s2) v += red[(s2 * 16 + r) * 64 + lane];
        fin[r * 64 + lane] = v;
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fin[r * 64 + lane];
    const int n = n0 + l31;

    // ---- bias, (accumulate), store ---------------------------------------------------------------------
    int yo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yo[r] = yoff[(r & 3) + 8 * (r >> 2) + 4 * half];
    {
        const float bias = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
        const bool ncol = n < d.Cy;
        if (d.accumulate || RING) {
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = (ncol && yo[r] >= 0) ? d.y[(size_t)yo[r] * d.Cy + n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += bias + old[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += bias;
        }
        if (ncol) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (yo[r] >= 0) d.y[(size_t)yo[r] * d.Cy + n] = acc[r];
        }
    }

    // ---- fused phase 1 of the BatchNorm(+activation) backward of the conv's input (data gradients) ------
    if (d.bnb_y != nullptr) {
        const int Cs = d.bnb_Cs;
        const bool nv = n < d.Cout;
        const int nn = nv ? n : 0;
        const float mean = d.bnb_state[nn], rstd = d.bnb_state[Cs + nn], sa = d.bnb_state[2 * Cs + nn],
                    sb = d.bnb_state[3 * Cs + nn];
        float yv[16];
        bool ok[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int bo = boff[(r & 3) + 8 * (r >> 2) + 4 * half];
            ok[r] = nv && bo >= 0;
            yv[r] = d.bnb_y[(size_t)(bo < 0 ? 0 : bo) * d.bnb_Cy + nn];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = fmaf(sa, yv[r], sb);
            const float gm = ok[r] ? dip_mul_rn(acc[r], dip_act_grad(z, d.bnb_slope)) : 0.f;
            const float xh = (yv[r] - mean) * rstd;
            s1 += gm;
            s2 = fmaf(gm, xh, s2);
        }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0 && n < Cs) {
            float* o = d.bnb_partials + (size_t)blockIdx.x * 2 * Cs + n;
            o[0] = s1;
            o[Cs] = s2;
        }
        return;
    }

    // ---- BatchNorm partial statistics of the consumer BatchNorm (forward launches) ----------------------
    if (d.stats == nullptr) return;
    float cn = 0.f, k = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (yo[r] >= 0) {
            if (cn == 0.f) k = acc[r];
            const float dv = acc[r] - k;
            cn += 1.f;
            s1 += dv;
            s2 = fmaf(dv, dv, s2);
        }
    }
    float mean = cn > 0.f ? k + s1 / cn : 0.f;
    float M2 = cn > 0.f ? s2 - s1 * s1 / cn : 0.f;
    const float on = __shfl_xor(cn, 32), om = __shfl_xor(mean, 32), oM = __shfl_xor(M2, 32);
    if (half == 0) {
        dip_chan(cn, mean, M2, on, om, oM);
        float* o = d.stats + (size_t)blockIdx.x * 3 * CoutP + n;
        o[0] = cn;
        o[CoutP] = mean;
        o[2 * CoutP] = M2;
    }
}

int small_max_pixels() {
    static const int v = [] {
        const char* e = getenv("DIP_SMALL_MAX_PIXELS");
        return e ? atoi(e) : 4624;             // 68 x 68: the 64 x 64 layers and their padded-domain data gradients
    }();
    return v;
}

bool small_geom(const DipConvDesc& d, SmallGeom* g) {
    if (d.ks != 1 && d.ks != 3 && d.ks != 5 && d.ks != 7) return false;
    if (d.stride != 1 && d.stride != 2) return false;
    if (d.dil != 1 && d.dil != 2) return false;
    if (d.dil == 2 && d.stride != 1) return false;
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin <= 0 || d.Cin > d.Cx) return false;
    if (d.tr.a != nullptr && d.Cin > SM_TR_MAX) return false;
    if (d.Cout < 1 || d.Cout > 160 || d.Cy < d.Cout) return false;
    if (d.Hout < 1 || d.Wout < 1) return false;
    g->ncls = d.dil == 2 ? 4 : 1;
    g->ring = 0;
    g->ngroups = 0;
    for (int c = 0; c < 4; ++c) { g->Hc[c] = g->Wc[c] = 0; g->gstart[c] = 0; }
    if (g->ncls == 1) {
        g->Hc[0] = d.Hout; g->Wc[0] = d.Wout;
        g->gstart[0] = 0;
        g->ngroups = dip_cdiv(d.Hout * d.Wout, 32);
        g->gstart[1] = g->ngroups;
    } else {
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            g->Hc[c] = (d.Hout - py + 1) / 2;
            g->Wc[c] = (d.Wout - px + 1) / 2;
            g->gstart[c] = g->ngroups;
            g->ngroups += dip_cdiv(g->Hc[c] * g->Wc[c], 32);
        }
        g->gstart[4] = g->ngroups;
    }
    g->nblk = dip_cdiv(d.Cout, 32);
    // waves per tile: every wave keeps its whole K slice in flight (<= SMAX steps), up to 16 waves
    const int kcls = d.dil == 2 ? (d.ks + 1) / 2 : d.ks;            // (a parity class of a dilated gradient sees <= ceil(ks / 2)^2 taps)
    const int tmax = kcls * kcls * ((d.Cin + 7) / 8);
    int nw = 4;
    while (nw < 16 && tmax > nw * SmCfg<8>::SMAX / 2) nw *= 2;
    static const char* force = getenv("DIP_SMALL_NW");
    if (force && (atoi(force) == 4 || atoi(force) == 8 || atoi(force) == 16)) nw = atoi(force);
    g->nw = nw;
    return true;
}

template <int TR, int NW, bool RING = false, bool KBIG = false>
int small_launch(const DipConvDesc& d, const SmallGeom& g, hipStream_t st) {
    auto kern = conv_small_kernel<TR, NW, RING, false, KBIG>;
    auto kern_g = conv_small_kernel<TR, NW, RING, true, KBIG>;   // grouped multi-instance form (dip_group.h)
    constexpr int lds = NW * 16 * 64 * 4;
    static bool attr_set[16] = {};
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, lds);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    dip_launch_pair<DIP_FAM_SMALL>(kern, kern_g, dim3(g.ngroups, g.nblk), dim3(64 * NW), lds, st, d, g);
    DIP_CHECK_LAUNCH();
    return 0;
}

template <int TR>
int small_launch_nw(const DipConvDesc& d, const SmallGeom& g, hipStream_t st) {
    if (d.ks > 3) {                              // 5x5 / 7x7: >= 50 K steps even at 16 channels -> 8 or 16 waves
        if (g.nw == 16) return small_launch<TR, 16, false, true>(d, g, st);
        return small_launch<TR, 8, false, true>(d, g, st);
    }
    if (g.nw == 16) return small_launch<TR, 16>(d, g, st);
    if (g.nw == 8) return small_launch<TR, 8>(d, g, st);
    return small_launch<TR, 4>(d, g, st);
}

}  // namespace

// Reflection-padded 3x3 stride-1 convs: the data gradient on the padded (H+2) x (W+2) domain costs the LDS-DMA kernel
// 561 tiles instead of 512 at 256^2 -- a lonely second round, 240 us instead of 160 (+14 for this launch).  The engine computes the INTERIOR
// H x W positions with the big kernel (a plain zero-padded correlation, off = 1) and this launch adds what the ring of
// the padded domain folds onto the frame rows 1, H-2 / columns 1, W-2 (adjoint of nn.ReflectionPad2d(1)): `d` is the
// interior descriptor (x = dy [H][W], y = the gradient [H][W][Cy], ks 3, stride 1, dil 1, off 1, zero padding).
extern "C" int dip_conv_dgrad_ring_ok(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    static const bool off = getenv("DIP_CONV_NO_RING") != nullptr;
    static const int maxpix = getenv("DIP_DGRAD_RING_MAX") ? atoi(getenv("DIP_DGRAD_RING_MAX")) : 300000;
    SmallGeom g;
    if (off || !small_geom(d, &g)) return 0;
    if (d.ks != 3 || d.stride != 1 || d.dil != 1 || d.off != 1 || d.pad_mode != DIP_PAD_ZERO) return 0;
    if (d.Hin != d.Hout || d.Win != d.Wout || d.Hout < 4 || d.Wout < 4 || d.tr.a != nullptr || d.bias != nullptr) return 0;
    return d.Hout * d.Wout <= maxpix ? 1 : 0;
}

extern "C" int dip_conv_dgrad_ring(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    SmallGeom g;
    if (!small_geom(d, &g) || d.ks != 3 || d.stride != 1 || d.dil != 1 || d.off != 1 || d.pad_mode != DIP_PAD_ZERO ||
        d.Hin != d.Hout || d.Win != d.Wout || d.Hout < 4 || d.Wout < 4 || d.tr.a != nullptr || d.bias != nullptr ||
        d.stats != nullptr || d.bnb_y != nullptr)
        DIP_FAIL("conv_dgrad_ring: needs the interior descriptor of a 3x3 stride-1 data gradient (off 1, zero padding)");
    g.ring = 1;
    g.ncls = 1;
    g.Hc[0] = 1; g.Wc[0] = 2 * d.Wout + 2 * (d.Hout - 2);
    g.gstart[0] = 0;
    g.ngroups = dip_cdiv(g.Wc[0], 32);
    g.gstart[1] = g.ngroups;
    g.nw = 16;
    return small_launch<0, 16, true>(d, g, reinterpret_cast<hipStream_t>(stream));
}

// 1 when the engine should run `d` through dip_conv_small: a shape the kernel serves and a small output
// 5x5 / 7x7 filters (round 6; tools/thin_sweep.py on the 'library' net's layers, profiles/r06_thin_sweep.txt): the tiled
// kernel's alternative is split-K + finish, so the bound depends on the shape --
//   * dilated (data gradient of a stride-2 conv): four parity classes of <= 9 taps each, the tiled kernel multiplies the
//     zeros of the dilation: conv_small wins up to ~21 k output pixels (116 x 180: 28 us against 50), loses at 228 x 356;
//   * stride 2 forward: up to ~5 k output pixels (56 x 88: 27 against 30 us);
//   * stride 1: the 3x3 bound; but K >= 3200 products on < 600 pixels (128 channels at 14 x 22 and below) is ONE 32 x 32
//     tile per CU doing 10 us of MFMAs on <= 40 CUs, and the tiled kernel's 24-way split-K + finish is 4 us faster.
extern "C" int dip_conv_small_eligible(const DipConvDesc* dp) {
    static const bool off = getenv("DIP_CONV_NO_SMALL") != nullptr;
    static const int k5_d2 = getenv("DIP_SMALL_K5_DIL2_MAX_PIXELS") ? atoi(getenv("DIP_SMALL_K5_DIL2_MAX_PIXELS")) : 24000;
    static const int k5_s2 = getenv("DIP_SMALL_K5_S2_MAX_PIXELS") ? atoi(getenv("DIP_SMALL_K5_S2_MAX_PIXELS")) : 5000;
    static const int k5_min = getenv("DIP_SMALL_K5_MIN_PIXELS") ? atoi(getenv("DIP_SMALL_K5_MIN_PIXELS")) : 600;
    SmallGeom g;
    if (off || !small_geom(*dp, &g)) return 0;
    const int px = dp->Hout * dp->Wout;
    if (dp->ks >= 5) {
        if (dp->dil == 2) return px <= (k5_d2 > small_max_pixels() ? k5_d2 : small_max_pixels()) ? 1 : 0;
        if (dp->stride == 2) return px <= (k5_s2 > small_max_pixels() ? k5_s2 : small_max_pixels()) ? 1 : 0;
        if (dp->ks * dp->ks * dp->Cin >= 3200 && px < k5_min) return 0;
    }
    return px <= small_max_pixels() ? 1 : 0;
}

// rows of the partial buffers (stats: [rows][3][CoutP32]; bnb_partials: [rows][2][bnb_Cs]) of dip_conv_small(d)
extern "C" int dip_conv_small_rows(const DipConvDesc* dp) {
    SmallGeom g;
    if (!small_geom(*dp, &g)) return 0;
    return g.ngroups;
}

extern "C" int dip_conv_small(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    SmallGeom g;
    if (!small_geom(d, &g)) DIP_FAIL("conv_small: unsupported shape (1x1 / 3x3 / 5x5 / 7x7, stride or dil <= 2, <= 160 output channels)");
    if (d.bnb_y != nullptr) {
        if (d.bnb_state == nullptr || d.bnb_partials == nullptr || d.bnb_Cs < d.Cout || d.bnb_pad < 0 ||
            d.Hout <= 2 * d.bnb_pad || d.Wout <= 2 * d.bnb_pad || d.stats != nullptr)
            DIP_FAIL("conv_small: inconsistent bnb_* fields");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.tr.a == nullptr) return small_launch_nw<0>(d, g, st);
    if (d.tr.slope > 0.f) return small_launch_nw<1>(d, g, st);
    return small_launch_nw<2>(d, g, st);
}
