// Implicit-GEMM convolution on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32):
// dispatcher (dip_conv_igemm / dip_conv_variant / dip_conv_plan), split-K finish kernel, and the
// register-staged kernel that serves what conv_igemm_dma.hip does not: stride-2 forwards, 5x5 and
// 7x7 filters, and the N = 160 one-pass variant for split-K data gradients towards 132 channels.
// Stride-1 1x1 / 3x3 convolutions -- the bulk of the net -- run conv_igemm_dma_kernel.
//
// One workgroup (4 waves) computes an 8x16-pixel x BN-channel output tile:
//   M = 128 output pixels, N = BN output channels, K = taps x input channels.
// K is walked in "units" = (channel chunk, tap).  For each chunk the (transformed) input halo
// tile is staged ONCE in LDS ([pixel][channel], pixel pitch == 4 mod 8 dwords -> conflict-free
// ds_read_b128) and all KS*KS taps read shifted windows of it, so HBM/L2 sees each input pixel
// ~1.4x instead of 9x.  The packed weights of one unit are a [cc/4][BN][4] slab that is copied
// global->LDS by the LDS-DMA path (global_load_lds_dwordx4, no VGPR round trip), double-buffered:
// the DMA for unit u+1 is in flight while the MFMAs of unit u run.  The next chunk's halo is
// prefetched into registers during the last tap of the current chunk.
// BatchNorm-apply + LeakyReLU of the PRODUCER layer, reflection/zero padding, the transposed
// (dilated) gather of the stride-2 data gradient, bias, and the BatchNorm partial statistics of
// the CONSUMER layer are all fused here, so activations cross HBM once per conv.
//
// K ordering trick: the 32x32x2 MFMA wants lane l to hold A[i=l&31][k=l>>5].  A lane reads FOUR
// consecutive channels with one ds_read_b128 (lanes 0-31: channels 8kk..8kk+3, lanes 32-63:
// 8kk+4..8kk+7) and feeds them to four MFMAs; B uses the same permutation ([c/4][n][c%4] slab),
// so the k-sum is merely re-ordered.
//
// Small images: a grid of <= 256 tiles cannot fill 256 CUs and each workgroup would walk all of K
// serially; blockIdx.z then splits the unit range (split-K) into a workspace and
// splitk_finish_kernel sums the slices in a fixed order, adds the bias and emits the statistics.
#include "dip_common.h"
#include "conv_epilogue.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

constexpr int TR_MAX = 512;   // max input channels whose BN coefficients are cached in LDS

// EXTRA: a 5th 32-wide column block (N = 160 = 128 + 32) is spread over the four waves, one extra
// 32x32 sub-tile each (wave (wm,wn) takes pixel sub-tile 2*wm+wn, whose A fragment it already
// holds).  This is the data gradient of the 132-channel decoder convs: +25 % MFMAs instead of a
// second launch that re-stages the whole input for 4 useful columns.  Needs Cin % CCH == 0 (no
// merged tail chunk: the LDS budget is spent on the wider weight slab) and no statistics.
template <int KS, int S, int CCH, int BN, bool EXTRA = false>
struct Cfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int HTH = (TH - 1) * S + KS, HTW = (TW - 1) * S + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int CMAX = EXTRA ? CCH : CCH + 8;
    static constexpr int BNB = EXTRA ? BN + 32 : BN;   // columns in the LDS weight slab
    static constexpr int C4MAX = CMAX / 4;
    static constexpr int LDP = CMAX + 4;  // == 4 (mod 8): 16 distinct 16-B slots per 16 pixels
    static constexpr int A_FLOATS = NPIX * LDP;
    static constexpr int B_FLOATS = CMAX * BNB;
    static constexpr int WN = (BN >= 64) ? 2 : 1;
    static constexpr int WM = 4 / WN;
    static constexpr int MS = 4 / WM;
    static constexpr int NS = BN / 32 / WN;
    static constexpr int A_SLOTS = (NPIX * C4MAX + 255) / 256;
    static constexpr int B_SLOTS = (C4MAX * BNB + 255) / 256;
    static constexpr int LDS_BYTES = (A_FLOATS + 2 * B_FLOATS + NPIX + 2 * TR_MAX) * 4;
    // Holding the next halo in VGPRs across the MFMAs spills for the 3x3 tiles at BN=128 (and a
    // scratch reload drains vmcnt, i.e. the weight DMA); the 1x1 tile needs only 5 slots and every
    // unit of a 1x1 conv is a new chunk, so there the prefetch pays.
    static constexpr bool PREFETCH_A = (KS == 1);
};

__device__ __forceinline__ int map_src(int v, int n_in, int dil, int pad_mode) {
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

typedef __attribute__((address_space(3))) void* lptr_t;

// One LDS-DMA piece: every active lane copies 16 B from its own global address to
// LDS[m0_base + lane*16].  Issued through inline asm so that hipcc neither counts it nor fences
// the following ds_reads of the OTHER weight buffer behind it (it cannot prove the two LDS
// buffers disjoint and would drain vmcnt(0) before every MFMA block); the kernel drains the DMA
// itself with dma_wait() in front of the barrier that publishes the buffer.  m0 is saved/restored
// inside the statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_dst_wave_uniform) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)lds_dst_wave_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(base)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int KS, int S, int CCH, int BN, bool EXTRA, bool GRP = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const DipConvDesc d_, const int ntx, const int ntiles,
                                                            const int CoutP, const int n_base, const int ksplit,
                                                            float* __restrict__ ws_, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    DIP_GRP_PTR(float*, ws);
    using C = Cfg<KS, S, CCH, BN, EXTRA>;
    constexpr int BNB = C::BNB;
    constexpr int KK = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + C::A_FLOATS;
    int* srcoff = reinterpret_cast<int*>(smem + C::A_FLOATS + 2 * C::B_FLOATS);
    float* tra = smem + C::A_FLOATS + 2 * C::B_FLOATS + C::NPIX;
    float* trb = tra + TR_MAX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;
    if (has_tr) {
        for (int c = tid; c < d.Cin; c += 256) {
            tra[c] = d.tr.a[c];
            trb[c] = d.tr.b[c];
        }
    }

    // ---- chunking of the input channels; K "units" = (chunk, tap) ----
    const int nfull = d.Cin / CCH, rem = d.Cin - nfull * CCH;
    int nchunks, last_cc;
    if (rem == 0) { nchunks = nfull; last_cc = CCH; }
    else if (rem <= 8 && nfull >= 1) { nchunks = nfull; last_cc = CCH + rem; }
    else { nchunks = nfull + 1; last_cc = rem; }
    const int cin4 = d.Cin >> 2;
    const int nunits = nchunks * KK;
    const int z = dip_grp_z<GRP>(grp);
    const int u0 = (int)(((long long)z * nunits) / ksplit);
    const int u1 = (int)(((long long)(z + 1) * nunits) / ksplit);

    f32x16 acc[C::MS][C::NS];
#pragma unroll
    for (int i = 0; i < C::MS; ++i)
#pragma unroll
        for (int j = 0; j < C::NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x16 accx;                                   // EXTRA: sub-tile (2*wm + wn) x columns [BN, BN+32)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[r] = 0.f;

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

    // ---- staging helpers -------------------------------------------------------------------
    f32x4 av[C::A_SLOTS];
    auto chunk_cc = [&](int ch) { return (ch == nchunks - 1) ? last_cc : CCH; };
    auto loadA = [&](int ch) {       // issue the global loads of chunk ch's halo tile into registers
        const int cb = ch * CCH, c4n = chunk_cc(ch) >> 2;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f / C::C4MAX, c4 = f - hp * C::C4MAX;
            const int so = ((hp < C::NPIX) && (c4 < c4n)) ? srcoff[hp] : -1;
            av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (so >= 0) av[i] = *reinterpret_cast<const f32x4*>(d.x + (size_t)so * d.Cx + cb + c4 * 4);
        }
    };
    auto storeA = [&](int ch) {      // producer BN + LeakyReLU, then LDS (slot geometry recomputed, not kept live)
        const int cb = ch * CCH, c4n = chunk_cc(ch) >> 2;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f / C::C4MAX, c4 = f - hp * C::C4MAX;
            if ((hp < C::NPIX) && (c4 < c4n)) {
                f32x4 o = av[i];
                if (has_tr && srcoff[hp] >= 0) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + cb + c4 * 4);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(trb + cb + c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = dip_act(fmaf(a4[e], o[e], b4[e]), slope);
                }
                *reinterpret_cast<f32x4*>(As + hp * C::LDP + c4 * 4) = o;
            }
        }
    };
    auto dmaB = [&](int u, float* Bdst) {   // LDS-DMA of unit u's weight slab (linear image)
        const int ch = u / KK, tap = u - ch * KK;
        const int cb = ch * CCH, nb4 = (chunk_cc(ch) >> 2) * BNB;
#pragma unroll
        for (int i = 0; i < C::B_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < nb4) {
                const int c4 = f / BNB, n = f - c4 * BNB;
                const int nn = min(n0 + n, CoutP - 1);   // columns past CoutP are never stored
                const float* src = d.wp + ((size_t)(tap * cin4 + (cb >> 2) + c4) * CoutP + nn) * 4;
                float* dst = Bdst + (i * 256 + wave * 64) * 4;   // wave-uniform base, lane*16 B added by HW
                lds_dma16(src, dst);
            }
        }
    };
    auto mma8 = [&](const float* Ab, const float* Bb) {   // 8 channels: 2x b128 A, 2x b128 B, 16 MFMA
        f32x4 a[C::MS], b[C::NS];
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) a[ms] = *reinterpret_cast<const f32x4*>(Ab + apix[ms]);
#pragma unroll
        for (int ns = 0; ns < C::NS; ++ns) b[ns] = *reinterpret_cast<const f32x4*>(Bb + half * (BNB * 4) + bcol[ns]);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms][j], b[ns][j], acc[ms][ns], 0, 0, 0);
        if constexpr (EXTRA) {
            const f32x4 bx = *reinterpret_cast<const f32x4*>(Bb + half * (BNB * 4) + (BN + l31) * 4);
            const f32x4 ax = wn ? a[C::MS - 1] : a[0];
#pragma unroll
            for (int j = 0; j < 4; ++j) accx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[j], bx[j], accx, 0, 0, 0);
        }
    };
    auto mma4 = [&](const float* Ab, const float* Bb) {   // 4-channel tail: lanes 0-31 ch 0,1; 32-63 ch 2,3
        f32x2 a[C::MS], b[C::NS];
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) a[ms] = *reinterpret_cast<const f32x2*>(Ab + apix[ms] - 2 * half);
#pragma unroll
        for (int ns = 0; ns < C::NS; ++ns) b[ns] = *reinterpret_cast<const f32x2*>(Bb + bcol[ns] + 2 * half);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ms][j], b[ns][j], acc[ms][ns], 0, 0, 0);
    };

    // ---- prologue: first chunk's halo + first unit's weights ----
    __syncthreads();                       // srcoff / tr tables visible
    loadA(u0 / KK);
    dmaB(u0, Bs);
    storeA(u0 / KK);
    dma_wait();
    __syncthreads();

    for (int u = u0; u < u1; ++u) {
        const int ch = u / KK, tap = u - ch * KK;
        const int ky = tap / KS, kx = tap - ky * KS;
        const int cc = chunk_cc(ch);
        const float* Bcur = Bs + ((u - u0) & 1) * C::B_FLOATS;
        float* Bnxt = Bs + ((u - u0 + 1) & 1) * C::B_FLOATS;
        const bool more = (u + 1) < u1;
        const bool newchunk = more && (tap == KK - 1);
        if (more) dmaB(u + 1, Bnxt);
        if constexpr (C::PREFETCH_A) {
            if (newchunk) loadA(ch + 1);   // global loads fly under this unit's MFMAs
        }
        const float* Ab = As + (ky * C::HTW + kx) * C::LDP;
        if (cc == CCH) {
#pragma unroll
            for (int kk = 0; kk < CCH / 8; ++kk) mma8(Ab + kk * 8, Bcur + kk * 2 * (BNB * 4));
        } else {
            const int kk8 = cc >> 3;
            for (int kk = 0; kk < kk8; ++kk) mma8(Ab + kk * 8, Bcur + kk * 2 * (BNB * 4));
            if (cc & 4) mma4(Ab + (cc - 4), Bcur + ((cc - 4) >> 2) * (BNB * 4));
        }
        if (newchunk) {
            __syncthreads();               // every wave is done reading the old halo
            if constexpr (!C::PREFETCH_A) loadA(ch + 1);
            storeA(ch + 1);
        }
        if (more) {
            dma_wait();                    // this wave's DMA pieces have landed ...
            __syncthreads();               // ... and so have everyone else's; halo stores visible
        }
    }

    // ---- epilogue ----
    if (ksplit > 1) {                      // split-K slice: raw partial sums to the workspace
        float* wz = ws + (size_t)z * d.Hout * d.Wout * d.Cy;
#pragma unroll
        for (int ns = 0; ns < C::NS; ++ns) {
            const int n = n0 + (wn * C::NS + ns) * 32 + l31;
#pragma unroll
            for (int ms = 0; ms < C::MS; ++ms) {
                const int sub = wm * C::MS + ms;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int oy = ty * C::TH + 2 * sub + (m >> 4);
                    const int ox = tx * C::TW + (m & 15);
                    if (oy < d.Hout && ox < d.Wout && n < d.Cy)
                        wz[((size_t)oy * d.Wout + ox) * d.Cy + n] = acc[ms][ns][r];
                }
            }
        }
        if constexpr (EXTRA) {
            const int n = n0 + BN + l31, sub = wm * C::MS + wn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int oy = ty * C::TH + 2 * sub + (m >> 4), ox = tx * C::TW + (m & 15);
                if (oy < d.Hout && ox < d.Wout && n < d.Cy) wz[((size_t)oy * d.Wout + ox) * d.Cy + n] = accx[r];
            }
        }
        return;
    }
    const DipEpi epi = dip_epi_make(d, ty, tx, C::TH, C::TW);
    if constexpr (EXTRA) {
        const int n = n0 + BN + l31;
        const float bias = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
        dip_epi_store16(epi, accx, wm * C::MS + wn, n, bias, half);
    }
    dip_conv_epilogue<C, BN>(d, acc, epi, n0, wn, wm, l31, half, tid, tile, CoutP, smem);
}

// Sum the split-K slices (fixed order), add bias, store (honouring y_pitch / accumulate) and emit
// {count, mean, M2} partials per block: thread (prow, cg) owns 4 channels of pixels prow, prow+rpi...
template <bool GRP = false>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ ws_, int ksplit, DipConvDesc d_,
                                                            int CoutP, int ppb, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, ws);
    DIP_GRP_DESC(DipConvDesc, d);
    __shared__ __attribute__((aligned(16))) float sh[256 * 12];
    const int nc4 = d.Cy >> 2;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    const int prow = threadIdx.x / nc4, cg = threadIdx.x - prow * nc4;
    const bool active = (int)threadIdx.x < rpi * nc4;
    const int npix = d.Hout * d.Wout;
    const size_t zs = (size_t)npix * d.Cy;
    const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
    f32x4 K = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = K, s2 = K;
    float n = 0.f;
    if (active) {
        const int ch = cg * 4;
        f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.bias != nullptr) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ch + e < d.Cout) bias[e] = d.bias[ch + e];
        }
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        for (int p = p0 + prow; p < p1; p += rpi) {
            f32x4 v = bias;
            const float* wp0 = ws + (size_t)p * d.Cy + ch;
            int k = 0;
            for (; k + 8 <= ksplit; k += 8) {           // 8 independent loads in flight, fixed add order
                f32x4 t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const f32x4*>(wp0 + (size_t)(k + q) * zs);
#pragma unroll
                for (int q = 0; q < 8; ++q) v += t[q];
            }
            for (; k + 4 <= ksplit; k += 4) {
                const f32x4 t0 = *reinterpret_cast<const f32x4*>(wp0 + (size_t)(k + 0) * zs);
                const f32x4 t1 = *reinterpret_cast<const f32x4*>(wp0 + (size_t)(k + 1) * zs);
                const f32x4 t2 = *reinterpret_cast<const f32x4*>(wp0 + (size_t)(k + 2) * zs);
                const f32x4 t3 = *reinterpret_cast<const f32x4*>(wp0 + (size_t)(k + 3) * zs);
                v += t0; v += t1; v += t2; v += t3;
            }
            for (; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(wp0 + (size_t)k * zs);
            const int oy = p / d.Wout, ox = p - oy * d.Wout;
            float* o = d.y + ((size_t)oy * pitch + ox) * d.Cy + ch;
            if (d.accumulate) v += *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = v;
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
    if (d.stats == nullptr) return;
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, M2 = mean;
    if (n > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = K[e] + s1[e] / n;
            M2[e] = s2[e] - s1[e] * s1[e] / n;
        }
    }
    dip_tree_chan4(sh, nc4, rpi, prow, cg, active, n, mean, M2);
    if (active && prow == 0) {
        float* o = d.stats + (size_t)blockIdx.x * 3 * CoutP + cg * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (cg * 4 + e < CoutP) {
                o[e] = n; o[CoutP + e] = mean[e]; o[2 * CoutP + e] = M2[e];
            }
        }
    }
}

int finish_ppb(int npix, int Cy, int* nblk) {
    const int nc4 = Cy / 4;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    // <= 512 blocks; small images get one pixel row per thread (a split-K layer is latency-bound: 8 blocks
    // of 4 sequential pixels per thread took 14 us for a 16x16 layer, mostly dependent round trips)
    int ppb = dip_cdiv(npix, 512);
    if (ppb < rpi) ppb = rpi;
    *nblk = dip_cdiv(npix, ppb);
    return ppb;
}

template <int KS, int S, int CCH, int BN, bool EXTRA = false>
int launch(const DipConvDesc& d, hipStream_t st, int n_base, int grid_y, int ksplit, float* ws) {
    using C = Cfg<KS, S, CCH, BN, EXTRA>;
    static bool attr_set[16] = {};
    auto kern = conv_igemm_kernel<KS, S, CCH, BN, EXTRA>;
    auto kern_g = conv_igemm_kernel<KS, S, CCH, BN, EXTRA, true>;        // grouped multi-instance form (dip_group.h)
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    const int CoutP = dip_round_up(d.Cout, 32);
    dim3 grid(ntiles, grid_y, ksplit);
    dip_launch_pair<DIP_FAM_CONV>(kern, kern_g, grid, dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, CoutP, n_base, ksplit, ws);
    DIP_CHECK_LAUNCH();
    return 0;
}

// N is covered by full 128-wide blocks plus one narrower remainder launch (e.g. the 132-channel
// data gradient of the decoder convs = 128 + a 32-wide block instead of two 128-wide ones).
template <int KS, int S, int CCH>
int launch_bn(const DipConvDesc& d, hipStream_t st, int ksplit, float* ws) {
    const int CoutP = dip_round_up(d.Cout, 32);
    const int nfull = CoutP / 128, rem = CoutP - nfull * 128;
    int rc = 0;
    if constexpr (KS == 3 && S == 1) {
        if (nfull == 1 && rem == 32 && d.stats == nullptr && (d.Cin % CCH) == 0)
            return launch<KS, S, CCH, 128, true>(d, st, 0, 1, ksplit, ws);      // N = 160 in one pass
    }
    if (nfull) rc = launch<KS, S, CCH, 128>(d, st, 0, nfull, ksplit, ws);
    if (rc || !rem) return rc;
    if (rem <= 32) return launch<KS, S, CCH, 32>(d, st, nfull * 128, 1, ksplit, ws);
    if (rem <= 64) return launch<KS, S, CCH, 64>(d, st, nfull * 128, 1, ksplit, ws);
    return launch<KS, S, CCH, 128>(d, st, nfull * 128, 1, ksplit, ws);
}

int cch_of(int ks, int stride) {
    if (ks == 1 && stride == 1) return 32;
    if (ks == 3 && stride == 1) return 32;
    if (ks == 3 && stride == 2) return 16;
    if (ks == 5 && stride == 1) return 16;
    if (ks == 5 && stride == 2) return 8;
    if (ks == 7 && stride == 1) return 16;     // feature_inversion.ipynb: filter_size_down/up = [7, 7, 5, 5, 3, 3]
    if (ks == 7 && stride == 2) return 8;
    if (ks == 8 || ks == 12) return 8;         // the dense Lanczos Downsampler conv inside conv() (stride 2) and its data gradient
    return 0;
}

int units_of(int Cin, int ks, int stride) {
    const int cch = cch_of(ks, stride);
    if (!cch) return 0;
    const int nfull = Cin / cch, rem = Cin - nfull * cch;
    const int nchunks = (rem == 0) ? nfull : ((rem <= 8 && nfull >= 1) ? nfull : nfull + 1);
    return nchunks * ks * ks;
}

}  // namespace

extern "C" int dip_conv_ntiles(int Hout, int Wout) { return dip_cdiv(Wout, 16) * dip_cdiv(Hout, 8); }
extern "C" int dip_conv_bf3_n64_plan(int ntiles, int Cin, int Cout);       // conv_bf3.hip

// Launch plan of one convolution: split-K factor, rows of the BatchNorm partial buffer, and the
// split-K workspace size in floats (0 when ksplit == 1).
extern "C" int dip_conv_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride);      // conv_thin.hip
static int conv_plan_impl(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit, int* stats_rows,
                          int64_t* ws_floats, bool allow_thin, bool allow_n64 = true);
extern "C" int dip_conv_plan(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit,
                             int* stats_rows, int64_t* ws_floats) {
    return conv_plan_impl(Hout, Wout, Cin, Cout, ks, stride, ksplit, stats_rows, ws_floats, true);
}
// the plan of a layer that turned out NOT to be eligible for the bf16-pipe kernel although its shape is (no split weights in
// the descriptor, a transform over > 512 input channels, fused BatchNorm-backward partials): dip_conv_plan gives a 3x3
// stride-1 layer with 96..255 tiles ONE pass because the 64-column bf16 form fills the chip there; on the fp32 kernels that
// pass is 96..255 workgroups, so this plan splits K as it did before round 5 (the engine re-plans with it)
extern "C" int dip_conv_plan_fp32(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit,
                                  int* stats_rows, int64_t* ws_floats) {
    return conv_plan_impl(Hout, Wout, Cin, Cout, ks, stride, ksplit, stats_rows, ws_floats, true, false);
}
static int conv_plan_impl(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* ksplit, int* stats_rows,
                          int64_t* ws_floats, bool allow_thin, bool allow_n64) {
    const int ntiles = dip_conv_ntiles(Hout, Wout);
    // thin layers (<= 64 channels in and out, 3x3 / 5x5, above conv_small's range): conv_thin_kernel, always one pass
    if (allow_thin && dip_conv_thin_shape_ok(Hout, Wout, Cin, Cout, ks, stride)) {
        *ksplit = 1;
        *stats_rows = ntiles;
        *ws_floats = 0;
        return 0;
    }
    const int gy = dip_cdiv(dip_round_up(Cout, 32), 128);
    int units = units_of(dip_round_up(Cin, 4), ks, stride);
    if (!units) DIP_FAIL("conv_plan: unsupported kernel size / stride");
    // 3x3 stride 2 normally runs on the LDS-DMA kernel (strided forward mode): 32-channel units
    static const bool s2dma = getenv("DIP_CONV_NO_S2DMA") == nullptr && getenv("DIP_CONV_NO_DMA") == nullptr;
    if (s2dma && ks == 3 && stride == 2 && Cin <= 288) units = dip_cdiv(dip_round_up(Cin, 4), 32) * 9;
    int k = 1;
    const int wgs = ntiles * gy;
    if (wgs <= 256) {          // one full round of 2 workgroups per CU (768 was 1.3 % slower end to end: a half-empty
                               // second round + more split-K slices to reduce), >= 2 K-units per slice, <= 24 slices
        static const char* tw = getenv("DIP_CONV_PLAN_WGS");
        k = (tw ? atoi(tw) : 512) / wgs;
        if (k > units / 2) k = units / 2;
        if (k > 24) k = 24;
        if (k < 1) k = 1;
    }
    // accuracy rule: one fp32 accumulator never sums more than ~2300 products (2x the 3x3 x 128-channel layers the
    // per-op tolerance is calibrated on).  Only the dense 8x8 / 12x12 Lanczos convs of conv(..., 'lanczos*') at >= 72
    // channels get here (K = 8192 at 128 channels: 4 slices, summed in fixed order by splitk_finish_kernel)
    const int Kred = ks * ks * dip_round_up(Cin, 4);
    if (Kred > 4608 && ks >= 8) {          // (7x7 x >= 96 channels etc. keep their one-pass plans: round-3 advisor finding)
        int kmin = dip_cdiv(Kred, 2304);
        if (kmin > units / 2) kmin = units / 2;
        if (k < kmin) k = kmin;
    }
    // 96..255 tiles: the 64-column bf16-pipe kernel, one pass
    if (allow_n64 && ks == 3 && stride == 1 && dip_conv_bf3_n64_plan(ntiles, dip_round_up(Cin, 4), Cout)) k = 1;
    const int Cy = dip_round_up(Cout, 4);
    *ksplit = k;
    if (k > 1) {
        int nblk;
        finish_ppb(Hout * Wout, Cy, &nblk);
        *stats_rows = nblk;
        *ws_floats = (int64_t)k * Hout * Wout * Cy;
    } else {
        *stats_rows = ntiles;
        *ws_floats = 0;
    }
    return 0;
}

// Launch plan of the data gradient of a stride-2 3x3 convolution (dil == 2 descriptor, Hout x Wout = the
// gradient's domain): the phase mode of the LDS-DMA kernel runs 4 workgroups per 8x16 tile of the
// half-resolution grid with 4/2/2/1 taps; split-K is bounded by the channel chunks of the 1-tap phase.
// Falls back to dip_conv_plan's answer (ksplit for the dilated evaluation) when the phase mode cannot run it.
extern "C" int dip_conv_plan_dil2(int Hout, int Wout, int Cin, int Cout, int ks, int* ksplit, int* stats_rows,
                                  int64_t* ws_floats) {
    static const bool off = getenv("DIP_CONV_NO_PHASE") != nullptr || getenv("DIP_CONV_NO_DMA") != nullptr;
    const int CoutP = dip_round_up(Cout, 32);
    if (off || ks != 3 || (CoutP % 128) != 0) return conv_plan_impl(Hout, Wout, Cin, Cout, ks, 1, ksplit, stats_rows, ws_floats, false);
    const int ntiles = dip_conv_ntiles((Hout + 1) / 2, (Wout + 1) / 2);
    const int wgs = 4 * ntiles * (CoutP / 128);
    const int nchunks = dip_cdiv(dip_round_up(Cin, 4), 32);
    int k = 1;
    if (wgs < 512) {
        k = dip_cdiv(720, wgs);
        if (k > 4) k = 4;
        if (k > nchunks) k = nchunks;
    }
    static const char* force = getenv("DIP_CONV_PHASE_KSPLIT");
    if (force && atoi(force) >= 1 && atoi(force) <= nchunks) k = atoi(force);
    const int Cy = dip_round_up(Cout, 4);
    *ksplit = k;
    if (k > 1) {
        int nblk;
        finish_ppb(Hout * Wout, Cy, &nblk);
        *stats_rows = nblk;
        *ws_floats = (int64_t)k * Hout * Wout * Cy;
    } else {
        *stats_rows = ntiles;
        *ws_floats = 0;
    }
    return 0;
}

extern "C" int dip_conv_dma_eligible(const DipConvDesc* dp);
extern "C" int dip_conv_phase_eligible(const DipConvDesc* dp);
extern "C" int dip_conv_igemm_dma(const DipConvDesc* dp, int ksplit, void* stream);
extern "C" int dip_conv_igemm_dma_cols(const DipConvDesc* dp, int n_base, void* stream);
extern "C" int dip_conv_thin4(const DipConvDesc* dp, int ncols, void* stream);
extern "C" int dip_conv1x1_res_eligible(const DipConvDesc* dp);
extern "C" int dip_conv1x1_res(const DipConvDesc* dp, void* stream);
extern "C" int dip_conv_bf3_eligible(const DipConvDesc* dp);
extern "C" int dip_conv_bf3_cols(const DipConvDesc* dp, int n_base, int ncols, void* stream);

extern "C" int dip_conv_thin_eligible(const DipConvDesc* dp);
extern "C" int dip_conv_thin(const DipConvDesc* dp, void* stream);

extern "C" int dip_conv_variant(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    // 1x1 layers with >= 256 tiles and split weights: the bf16 matrix pipe (conv_bf3_k1_kernel) before the fp32 weights-resident kernel
    if (d.ks == 1 && dip_conv_bf3_eligible(dp)) return 7;
    if (dip_conv1x1_res_eligible(dp)) return 6;
    if (dip_conv_thin_eligible(dp)) return 8;
    static const bool no_dma = getenv("DIP_CONV_NO_DMA") != nullptr;      // A/B switches for profiling
    static const bool no_extra = getenv("DIP_CONV_NO_EXTRA") != nullptr;
    const int CoutP = dip_round_up(d.Cout, 32);
    if (!no_dma && dip_conv_phase_eligible(dp)) return 4;
    static const bool no_thin4 = getenv("DIP_CONV_NO_THIN4") != nullptr;
    // 129..132 output channels (the data gradient towards [4 skip | 128 up-sampled] channels): the
    // 1..4 leading columns on the vector ALU (conv_thin4.hip), the other 128 on the DMA kernel
    if (!no_thin4 && !no_dma && d.ks == 3 && d.stride == 1 && d.dil == 1 && d.Cout > 128 && d.Cout <= 132 &&
        d.stats == nullptr && d.tr.a == nullptr && d.ksplit <= 1 && dip_conv_dma_eligible(dp))
        return 3;
    // 3x3 stride-1 layers with >= 256 tiles on the bf16 matrix pipe (DIP_CONV_BF3, conv_bf3.hip)
    if (dip_conv_bf3_eligible(dp)) return 7;
    // otherwise N = 160 in one pass: a fifth 32-column block spread over the four waves
    if (!no_extra && d.ks == 3 && d.stride == 1 && CoutP == 160 && d.stats == nullptr && (d.Cin % 32) == 0) return 2;
    if (!no_dma && dip_conv_dma_eligible(dp)) return d.stride == 2 ? 5 : 1;
    return 0;
}

// fused BatchNorm-backward partials (bnb_* fields) ride in the shared one-pass epilogue (conv_epilogue.h):
// every one-pass variant except the phase mode (4 workgroups per tile, interleaved pixels) and the N = 160 variant
extern "C" int dip_conv_bnb_fusable(const DipConvDesc* dp) {
    if (dp->ksplit > 1) return 0;
    const int v = dip_conv_variant(dp);
    return (v == 0 || v == 1 || v == 3 || v == 6) ? 1 : 0;
}

// second half of a split-K dispatch: sums the d.ksplit workspace slices in a fixed order, adds the
// bias, stores and emits the BatchNorm partials (exported so that a profiler can time it on its own)
extern "C" int dip_conv_splitk_finish(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    if (d.ksplit <= 1 || d.ws == nullptr) DIP_FAIL("conv_splitk_finish: descriptor is not split-K");
    int nblk;
    const int ppb = finish_ppb(d.Hout * d.Wout, d.Cy, &nblk);
    dip_launch_pair<DIP_FAM_CONV>(splitk_finish_kernel<false>, splitk_finish_kernel<true>, dim3(nblk), dim3(256), 0,
                                  reinterpret_cast<hipStream_t>(stream), (const float*)d.ws, d.ksplit, d, dip_round_up(d.Cout, 32), ppb);
    DIP_CHECK_LAUNCH();
    return 0;
}

extern "C" int dip_conv_igemm(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin <= 0 || d.Cin > d.Cx) DIP_FAIL("conv_igemm: channel strides must be multiples of 4");
    if (d.dil != 1 && d.dil != 2) DIP_FAIL("conv_igemm: dil must be 1 or 2");
    if (d.tr.a != nullptr && d.Cin > TR_MAX) DIP_FAIL("conv_igemm: more than 512 input channels with a fused transform");
    int ksplit = d.ksplit > 1 ? d.ksplit : 1;
    if (d.bnb_y != nullptr) {
        if (!dip_conv_bnb_fusable(dp)) DIP_FAIL("conv_igemm: fused BatchNorm-backward partials need a one-pass launch (dip_conv_bnb_fusable)");
        if (d.bnb_state == nullptr || d.bnb_partials == nullptr || d.bnb_Cs < d.Cout || d.bnb_pad < 0 ||
            d.Hout <= 2 * d.bnb_pad || d.Wout <= 2 * d.bnb_pad)
            DIP_FAIL("conv_igemm: inconsistent bnb_* fields");
    }
    if (ksplit > 1) {
        if (d.ws == nullptr) DIP_FAIL("conv_igemm: ksplit > 1 needs a workspace");
        int units = units_of(d.Cin, d.ks, d.stride);
        if (dip_conv_variant(dp) == 5) units = dip_cdiv(d.Cin, 32) * 9;     // 32-channel units of the LDS-DMA kernel
        if (ksplit > units) DIP_FAIL("conv_igemm: ksplit exceeds the number of K units");
    }
    int rc;
    const int variant = dip_conv_variant(dp);
    if (variant == 3) {
        const int ncols = d.Cout - 128;
        rc = dip_conv_thin4(dp, ncols, stream);
        if (rc) return rc;
        return dip_conv_igemm_dma_cols(dp, ncols, stream);
    }
    if (variant == 6) return dip_conv1x1_res(dp, stream);
    if (variant == 8) return dip_conv_thin(dp, stream);
    if (variant == 7) return dip_conv_bf3_cols(dp, 0, dip_round_up(d.Cout, 128), stream);
    if (variant == 1 || variant == 4 || variant == 5) rc = dip_conv_igemm_dma(dp, ksplit, stream);
    else if (d.ks == 1 && d.stride == 1) rc = launch_bn<1, 1, 32>(d, st, ksplit, d.ws);
    else if (d.ks == 3 && d.stride == 1) rc = launch_bn<3, 1, 32>(d, st, ksplit, d.ws);
    else if (d.ks == 3 && d.stride == 2) rc = launch_bn<3, 2, 16>(d, st, ksplit, d.ws);
    else if (d.ks == 5 && d.stride == 1) rc = launch_bn<5, 1, 16>(d, st, ksplit, d.ws);
    else if (d.ks == 5 && d.stride == 2) rc = launch_bn<5, 2, 8>(d, st, ksplit, d.ws);
    else if (d.ks == 7 && d.stride == 1) rc = launch_bn<7, 1, 16>(d, st, ksplit, d.ws);
    else if (d.ks == 7 && d.stride == 2) rc = launch_bn<7, 2, 8>(d, st, ksplit, d.ws);
    else if (d.ks == 8 && d.stride == 2) rc = launch_bn<8, 2, 8>(d, st, ksplit, d.ws);
    else if (d.ks == 8 && d.stride == 1) rc = launch_bn<8, 1, 8>(d, st, ksplit, d.ws);
    else if (d.ks == 12 && d.stride == 2) rc = launch_bn<12, 2, 8>(d, st, ksplit, d.ws);
    else if (d.ks == 12 && d.stride == 1) rc = launch_bn<12, 1, 8>(d, st, ksplit, d.ws);
    else DIP_FAIL("conv_igemm: unsupported kernel size / stride");
    if (rc || ksplit == 1) return rc;
    return dip_conv_splitk_finish(dp, stream);
}
