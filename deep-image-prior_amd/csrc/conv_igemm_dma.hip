// Implicit-GEMM convolution, fully asynchronous staging variant: 1x1 and 3x3 at stride 1 (MODE 0), the data
// gradient of a stride-2 3x3 as four dense sub-filter convolutions (MODE 1) and the stride-2 3x3 forward with the
// input split by pixel parity (MODE 2) -- see the comment in front of the kernel.
//
// Same tile / wave / MFMA layout as conv_igemm.hip (8x16 pixels x BN channels per 4-wave
// workgroup, two workgroups per CU), but NOTHING is staged through registers any more:
//   * the input halo tile of the NEXT channel chunk is copied global->LDS by LDS-DMA
//     (global_load_lds_dwordx4) into a second halo buffer while the current chunk's 9 taps run;
//   * the producer's BatchNorm-apply + LeakyReLU cannot ride on a DMA, so it is applied to the
//     landed halo in LDS (see below);
//   * zero padding (data gradients) is a DMA from a 16-byte zero page.
// LDS-DMA writes are lane-linear (dest = wave base + lane*16 B), so the halo image cannot be padded
// against bank conflicts; instead the 16-byte slot index is XOR-swizzled with the pixel index on the
// SOURCE side (slot (hp, s) holds channel group s ^ (hp & 7)) and un-swizzled on the read.
// LDS: 2 x 23.0 KB halo + 2 x 16 KB weights + tables = 79.95 KB -> still two workgroups per CU.
// 3x3: the producer transform is applied IN PLACE in LDS, once per chunk, by the thread that DMA'd
// the slot (after its own vmcnt(0)); 1x1: at the fragment read.
// Restrictions (checked by the dispatcher, dip_conv_dma_eligible / dip_conv_phase_eligible): ks in {1,3} (stride 2:
// 3x3 only), Cin <= 288 and a LeakyReLU-type activation when a transform is fused; everything else takes the
// register-staged kernel in conv_igemm.hip.
#include "dip_common.h"
#include "conv_epilogue.h"
#include "lds_dma.h"
#include "dip_group.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int TRN = 288;

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N)
template <int I, int N>
struct SFor {
    template <class F>
    static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, I>{});
        SFor<I + 1, N>::run(f);
    }
};
template <int N>
struct SFor<N, N> {
    template <class F>
    static __device__ __forceinline__ void run(F&&) {}
};

__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

template <int KS, int BN, int NTAB = 1>
struct DCfg {
    static constexpr int TH = 8, TW = 16;
    static constexpr int HTH = TH - 1 + KS, HTW = TW - 1 + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int CCH = 32;
    static constexpr int A_BUF = NPIX * 32;                 // floats per halo buffer (8 slots of 16 B per pixel)
    static constexpr int B_BUF = CCH * BN;                  // floats per weight slab
    static constexpr int WN = (BN >= 64) ? 2 : 1;
    static constexpr int WM = 4 / WN;
    static constexpr int MS = 4 / WM;
    static constexpr int NS = BN / 32 / WN;
    static constexpr int A_SLOTS = (NPIX * 8 + 255) / 256;
    static constexpr int B_SLOTS = (8 * BN + 255) / 256;
    static constexpr int LDS_BYTES = (2 * A_BUF + 2 * B_BUF + NTAB * NPIX + 2 * TRN) * 4;
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

#ifdef DIP_CLK_PROFILE
// debug build only: per-workgroup shader-clock totals of the K-loop segments (wave 0)
__device__ unsigned long long g_prof[8192 * 16];
__device__ unsigned g_trace[128 * 8];
#define PROBE(i) do { } while (0)
#else
#define PROBE(i) do { } while (0)
#endif

// MODE 1 (phase mode, the data gradient of a stride-2 convolution): the descriptor describes the dilated
// problem (dil == 2, filter d.ks = 2*KS-1, zero padding), the kernel runs it as 4 dense convolutions,
// one per parity (py, px) of the output pixel: output (2a+py, 2b+px) only sees the filter taps
// k = k0 + 2j with k0 = (off - p) & 1, which read dy[a + j - (off - p - k0)/2] -- a KS x KS (or smaller)
// dense filter over the UNdilated input.  9 taps per 4 pixels instead of 36; the terms that remain are
// summed in the same order as in the dilated evaluation, so the result is bit-identical to it.
// Workgroup id -> (tile, phase): the 4 phases of a tile are neighbours (same dy halo in L2).
// MODE 2 (strided forward: stride 2, filter d.ks = 2*KS-1): the mirror image -- the INPUT is split by the parity
// (py, px) of its pixel; tap k = k0 + 2j of parity p reads x[2(q + m_min + j) + p], a dense KS x KS (or smaller)
// filter over the sub-grid of that parity.  One workgroup walks the 4 sub-grids one after the other: K units =
// (parity, channel chunk, sub-filter tap), 9 per chunk in total as for a stride-1 3x3, and each
// (parity, chunk) pair has its own halo (gathered by the LDS-DMA with a pixel stride of 2).
// GRP: grouped multi-instance launch (dip_group.h) -- gridDim.z = ksplit x instances, the descriptor shifted to this
// workgroup's instance; GRP = false is the solo kernel, instruction for instruction what it was before the parameter existed.
template <int KS, int BN, bool TR, int MODE = 0, bool GRP = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_kernel(const DipConvDesc d_, const int ntx, const int ntiles,
                                                                const int CoutP, const int n_base, const int ksplit,
                                                                float* __restrict__ ws_, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    DIP_GRP_PTR(float*, ws);
    constexpr bool PH = (MODE == 1);
    constexpr bool SF = (MODE == 2);
    using C = DCfg<KS, BN, SF ? 4 : 1>;
#ifdef DIP_CLK_PROFILE
    const unsigned long long wall0 = wall_clock64(), clk0 = clock64();
#endif
    constexpr int KK = KS * KS;
    constexpr int CCH = C::CCH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                   // 2 halo buffers
    float* Bs = smem + 2 * C::A_BUF;                    // 2 weight slabs
    int* srcoff = reinterpret_cast<int*>(Bs + 2 * C::B_BUF);
    float* tra = reinterpret_cast<float*>(srcoff + (SF ? 4 : 1) * C::NPIX);
    float* trb = tra + TRN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int wn = wave % C::WN;
    const int wm = wave / C::WN;

    int tile, py = 0, px = 0, k0y = 0, k0x = 0, nky = KS, nkx = KS, offy = d.off, offx = d.off;
    if constexpr (PH) {
        // this XCD's contiguous range [a, b) of (tile, phase) pairs, walked phase-major: the 4-tap phase of
        // all its tiles first, the 1-tap phase last (longest-first keeps the tail of the launch short)
        const int nwg = 4 * ntiles, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        const int a = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int b = a + q + (xcd < r ? 1 : 0);
        int lin = a, left = idx;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int first = a + ((p - a) & 3);                     // first id >= a with id & 3 == p
            const int cnt = first < b ? ((b - 1 - first) >> 2) + 1 : 0;
            if (left >= 0 && left < cnt) lin = first + 4 * left;
            left -= cnt;
            if (left < 0) left = -0x40000000;
        }
        tile = lin >> 2;
        const int p2 = d.off & 1;                       // the parity that sees the even taps
        py = p2 ^ ((lin >> 1) & 1);
        px = p2 ^ (lin & 1);
        k0y = (d.off - py) & 1;
        k0x = (d.off - px) & 1;
        nky = (d.ks - k0y + 1) >> 1;
        nkx = (d.ks - k0x + 1) >> 1;
        offy = (d.off - py - k0y) >> 1;
        offx = (d.off - px - k0x) >> 1;
    } else {
        tile = dip_xcd_remap(blockIdx.x, ntiles);
    }
    int kkr = PH ? nky * nkx : KS * KS;                 // taps of the current filter (SF: of the current parity)
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int n0 = n_base + blockIdx.y * BN;

    // SF: parity p of an input coordinate v = 2a + p -> first tap k0, taps, and a - q of the first tap
    auto sf_dim = [&](int p_, int& k0_, int& nk_, int& mmin_) {
        k0_ = (p_ + d.off) & 1;
        nk_ = (d.ks - k0_ + 1) >> 1;
        mmin_ = (k0_ - d.off - p_) >> 1;                // (even numerator)
    };
    if constexpr (SF) {
        for (int i = tid; i < 4 * C::NPIX; i += 256) {
            const int ph_ = i / C::NPIX, hp = i - ph_ * C::NPIX;
            const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
            int a0, a1, my, mx;
            sf_dim(ph_ >> 1, a0, a1, my);
            sf_dim(ph_ & 1, a0, a1, mx);
            const int sr = map_src(2 * (ty * C::TH + hr + my) + (ph_ >> 1), d.Hin, 1, d.pad_mode);
            const int sc = map_src(2 * (tx * C::TW + hc + mx) + (ph_ & 1), d.Win, 1, d.pad_mode);
            srcoff[i] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
        }
    } else {
        for (int hp = tid; hp < C::NPIX; hp += 256) {
            const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
            const int sr = map_src(ty * C::TH + hr - offy, d.Hin, PH ? 1 : d.dil, d.pad_mode);
            const int sc = map_src(tx * C::TW + hc - offx, d.Win, PH ? 1 : d.dil, d.pad_mode);
            srcoff[hp] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
        }
    }
    // 3x3: the thread that DMA'd a slot transforms it IN PLACE once its own DMA has landed (after its
    // own vmcnt(0), before the unit's barrier) -- 1/18 of the VALU work of transforming at every
    // fragment read, no extra barrier, and zero-padded slots simply stay zero.
    // 1x1: every unit is a new chunk, so the transform stays at the fragment read (2x redundancy only).
    constexpr bool has_tr = TR && (KS == 1);      // transform at fragment read
    const bool fix_inplace = (KS != 1) && (TR || d.pad_mode != DIP_PAD_REFLECT || d.dil != 1);
    const size_t wtap = (size_t)(d.Cin >> 2) * CoutP * 4;            // floats between taps in the packed weights
    // packed-weight offset of tap (ky, kx) of this workgroup's filter
    auto wofs = [&](int ky_, int kx_) -> size_t {
        if constexpr (PH || SF) return (size_t)((k0y + 2 * ky_) * d.ks + (k0x + 2 * kx_)) * wtap;
        else return (size_t)(ky_ * KS + kx_) * wtap;
    };
    const float slope = d.tr.slope;
    if constexpr (TR) {
        for (int c = tid; c < d.Cin; c += 256) {
            tra[c] = d.tr.a[c];
            trb[c] = d.tr.b[c];
        }
    }

    const int nchunks = (d.Cin + CCH - 1) / CCH;
    const int last_cc = d.Cin - (nchunks - 1) * CCH;
    const int nunits = SF ? nchunks * d.ks * d.ks : nchunks * kkr;
    const int z = dip_grp_z<GRP>(grp);
    const int u0 = (int)(((long long)z * nunits) / ksplit);
    const int u1 = (int)(((long long)(z + 1) * nunits) / ksplit);
    // SF: units run parity-major: parity ph owns nchunks * taps(ph) consecutive units
    int ph = 0, mmy = 0, mmx = 0, tap0 = 0;
    auto sf_set_phase = [&](int ph_) {
        sf_dim(ph_ >> 1, k0y, nky, mmy);
        sf_dim(ph_ & 1, k0x, nkx, mmx);
        kkr = nky * nkx;
    };
    int ch0;
    if constexpr (SF) {
        int rem = u0;
        for (ph = 0; ph < 4; ++ph) {
            sf_set_phase(ph);
            if (rem < nchunks * kkr) break;
            rem -= nchunks * kkr;
        }
        ch0 = rem / kkr;
        tap0 = rem - ch0 * kkr;
    } else {
        ch0 = u0 / kkr;
        tap0 = u0 - ch0 * kkr;
    }

    f32x16 acc[C::MS][C::NS];
#pragma unroll
    for (int i = 0; i < C::MS; ++i)
#pragma unroll
        for (int j = 0; j < C::NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int hp0[C::MS];
#pragma unroll
    for (int ms = 0; ms < C::MS; ++ms) {
        const int sub = wm * C::MS + ms;
        const int r = 2 * sub + (l31 >> 4), c = l31 & 15;
        hp0[ms] = r * C::HTW + c;
    }
    int bcol[C::NS];
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) bcol[ns] = ((wn * C::NS + ns) * 32 + l31) * 4;

    auto chunk_cc = [&](int ch) { return (ch == nchunks - 1) ? last_cc : CCH; };
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;            // LDS byte address of smem
    const unsigned lds_piece = (unsigned)wave * 1024u;                    // this wave's 1 KiB inside a 4 KiB piece row

    __syncthreads();                       // srcoff / tr tables visible
    // Per-thread DMA geometry, computed once: byte offsets relative to a per-unit / per-chunk
    // wave-uniform base, so the issue inside the K loop is two SALU moves + one VMEM instruction.
    constexpr unsigned NONE = 0xffffffffu;
    unsigned aoff[C::A_SLOTS];             // halo piece i: (src_pixel*Cx + c4*4)*4 bytes, NONE = zero pad
    int ac4[C::A_SLOTS];                   // its channel group (un-swizzled), -1 = no such slot
    // SF: the geometry registers describe the halo that is fetched / fixed up NEXT; gph = its parity
    int gph = 0;
    auto load_geom = [&](int ph_) {
        gph = ph_;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f >> 3;
            const int c4 = (f & 7) ^ (hp & 7);
            ac4[i] = hp < C::NPIX ? c4 : -1;
            const int so = hp < C::NPIX ? srcoff[ph_ * C::NPIX + hp] : -1;
            aoff[i] = so >= 0 ? (unsigned)(so * d.Cx + c4 * 4) * 4u : NONE;
        }
    };
    load_geom(SF ? ph : 0);
    unsigned boff[C::B_SLOTS];             // weight piece i: ((c4*CoutP + column)*4)*4 bytes
#pragma unroll
    for (int i = 0; i < C::B_SLOTS; ++i) {
        const int f = tid + i * 256;
        const int c4 = f / BN, n = f - c4 * BN;
        boff[i] = (unsigned)((c4 * CoutP + min(n0 + n, CoutP - 1)) * 4) * 4u;
    }
    auto dmaA = [&](int ch, int abuf) {    // halo tile of chunk ch: slot (hp, s) <- channel group s ^ (hp & 7)
        const int c4n = chunk_cc(ch) >> 2;
        const float* sb = d.x + ch * CCH;
        const unsigned m0b = lds_base + (unsigned)(abuf * C::A_BUF) * 4u + lds_piece;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            if (ac4[i] >= 0 && ac4[i] < c4n) {
                if constexpr (KS == 1) {
                    if (aoff[i] != NONE) lds_dma16_s(sb, aoff[i], m0b + i * 4096u);
                    else lds_dma16_s(g_zero_page, 0u, m0b + i * 4096u);
                } else {
                    // exactly ONE instruction per slot and wave (the unit-end wait counts them): a padded
                    // lane copies 16 valid bytes from offset 0 and fixA() zeroes the slot afterwards
                    lds_dma16_s(sb, aoff[i] != NONE ? aoff[i] : 0u, m0b + i * 4096u);
                }
            }
        }
    };
    auto fixA = [&](int ch, float* Abuf) {  // in-place pass over this thread's own landed slots:
        const int cb = ch * CCH, c4n = chunk_cc(ch) >> 2;   // BatchNorm+LeakyReLU, or zero for padding
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            if (ac4[i] >= 0 && ac4[i] < c4n) {
                float* p = Abuf + (tid + i * 256) * 4;
                if (aoff[i] == NONE) {
                    *reinterpret_cast<f32x4*>(p) = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if constexpr (TR) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(p);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(tra + cb + ac4[i] * 4);
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(trb + cb + ac4[i] * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dip_act_leaky(fmaf(ta[e], v[e], tb[e]), slope);
                    *reinterpret_cast<f32x4*>(p) = v;
                }
            }
        }
    };
    auto dmaB = [&](const float* sb, int ch, int bbuf) {      // weight slab of one unit: sb = its packed weights
        const int nb4 = (chunk_cc(ch) >> 2) * BN;
        const unsigned m0b = lds_base + (unsigned)(2 * C::A_BUF + bbuf * C::B_BUF) * 4u + lds_piece;
#pragma unroll
        for (int i = 0; i < C::B_SLOTS; ++i) {
            if (tid + i * 256 < nb4) lds_dma16_s(sb, boff[i], m0b + i * 4096u);
        }
    };

    // One DMA piece each, so the K loop can drop them between MFMAs (a block of DMA issue code
    // between two MFMA clusters leaves the matrix pipe idle: the wave issues in order).
    auto dmaA_slot = [&](auto I, const float* sb, unsigned m0b, int c4n) {
        constexpr int i = decltype(I)::value;
        if constexpr (i < C::A_SLOTS) {
            if constexpr (KS == 1) {
                if (ac4[i] >= 0 && ac4[i] < c4n) {
                    if (aoff[i] != NONE) lds_dma16_s(sb, aoff[i], m0b + i * 4096u);
                    else lds_dma16_s(g_zero_page, 0u, m0b + i * 4096u);
                }
            } else {
                // full chunks only (c4n == 8): every lane of a full slot copies; a padded lane copies 16
                // valid bytes from offset 0 and fixA zeroes the slot.  ONE instruction per slot and wave.
                const unsigned vo = aoff[i] != NONE ? aoff[i] : 0u;
                if constexpr ((i + 1) * 256 <= C::NPIX * 8) {
                    lds_dma16_s(sb, vo, m0b + i * 4096u);
                } else {
                    if (ac4[i] >= 0) lds_dma16_s(sb, vo, m0b + i * 4096u);
                }
            }
        }
    };
    auto dmaB_slot = [&](auto I, const float* sb, unsigned m0b) {     // full chunks: 8*BN pieces == B_SLOTS*256
        constexpr int i = decltype(I)::value;
        if constexpr (i < C::B_SLOTS) lds_dma16_s(sb, boff[i], m0b + i * 4096u);
    };
    auto fixA_slot = [&](auto I, int cb, float* Abuf) {               // full chunks
        constexpr int i = decltype(I)::value;
        if constexpr (i < C::A_SLOTS) {
            if (ac4[i] >= 0) {
                float* p = Abuf + (tid + i * 256) * 4;
                if (aoff[i] == NONE) {
                    *reinterpret_cast<f32x4*>(p) = f32x4{0.f, 0.f, 0.f, 0.f};
                } else if constexpr (TR) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(p);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(tra + cb + ac4[i] * 4);
                    const f32x4 tb = *reinterpret_cast<const f32x4*>(trb + cb + ac4[i] * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dip_act_leaky(fmaf(ta[e], v[e], tb[e]), slope);
                    *reinterpret_cast<f32x4*>(p) = v;
                }
            }
        }
    };

    // incremental bookkeeping (no divisions in the loop)
    int ch = ch0, tap = tap0;
    int ky = (PH || SF) ? tap / nkx : tap / KS, kx = (PH || SF) ? tap - ky * nkx : tap - ky * KS;
    int abuf = 0, bbuf = 0;
    const float* wcur = d.wp + wofs(ky, kx) + (size_t)(ch * (CCH / 4)) * CoutP * 4;
    // SF: the (parity, chunk) pair after (ph_, ch_); parity 4 = none
    auto sf_next = [&](int ph_, int ch_, int& phn_, int& chn_) {
        chn_ = ch_ + 1;
        phn_ = ph_;
        if (chn_ == nchunks) { chn_ = 0; phn_ = ph_ + 1; }
    };

    // ---- prologue ----
    dmaA(ch0, 0);
    dmaB(wcur, ch0, 0);
    dma_wait();
    if (fix_inplace) fixA(ch0, As);
    if constexpr (SF) {                    // geometry of the halo after the first one
        int phn, chn;
        sf_next(ph, ch, phn, chn);
        if (phn < 4 && phn != gph) load_geom(phn);
    }
    __syncthreads();
    // Halo DMAs of this wave per chunk (wave-uniform).  The unit that issues the next chunk's halo
    // waits only for its weights: loads retire in order and the halo is issued AFTER the weights, so
    // vmcnt(nA) leaves exactly the halo in flight.
    int nA = 0;
#pragma unroll
    for (int i = 0; i < C::A_SLOTS; ++i) nA += (wave * 64 + i * 256 < C::NPIX * 8) ? 1 : 0;
    int a_state = 0;                       // next halo: 0 none/ready, 1 in flight, 2 landed but not fixed up yet
#ifdef DIP_CLK_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tstart = clock64();
#endif

    // One K step of 8 channels: the fragments of both operands (lane = 4 consecutive channels of its
    // pixel / column -> 4 MFMAs per ds_read_b128 pair).
    struct Frag {
        f32x4 a[C::MS], b[C::NS], ta, tb;
    };
    int abase[C::MS], sw[C::MS];
    auto set_tap = [&](int ky, int kx) {
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) {
            const int hp = hp0[ms] + ky * C::HTW + kx;
            abase[ms] = hp * 32;
            sw[ms] = hp & 7;
        }
    };
    auto read_frag = [&](Frag& f, const float* Acur, const float* Bcur, int cb, int kk) {
        const int c4 = 2 * kk + half;
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms)
            f.a[ms] = *reinterpret_cast<const f32x4*>(Acur + abase[ms] + ((c4 ^ sw[ms]) << 2));
#pragma unroll
        for (int ns = 0; ns < C::NS; ++ns)
            f.b[ns] = *reinterpret_cast<const f32x4*>(Bcur + c4 * (BN * 4) + bcol[ns]);
        if constexpr (has_tr) {
            f.ta = *reinterpret_cast<const f32x4*>(tra + cb + c4 * 4);
            f.tb = *reinterpret_cast<const f32x4*>(trb + cb + c4 * 4);
        }
    };
    // 16 MFMAs of one fragment set; step(J) runs after the 4 MFMAs of channel J
    auto mma_frag = [&](Frag& f, auto&& step) {
        if constexpr (has_tr) {
#pragma unroll
            for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[ms][e] = dip_act_leaky(fmaf(f.ta[e], f.a[ms][e], f.tb[e]), slope);
        }
        SFor<0, 4>::run([&](auto J) {
            constexpr int j = decltype(J)::value;
#pragma unroll
            for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[ms][j], f.b[ns][j], acc[ms][ns], 0, 0, 0);
            step(J);
        });
    };

    bool have_f0 = false;
    Frag F0;
    set_tap(ky, kx);

    for (int u = u0; u < u1; ++u) {
        const int cc = chunk_cc(ch), cb = ch * CCH;
        const float* Acur = As + abuf * C::A_BUF;
        float* Anxt = As + (abuf ^ 1) * C::A_BUF;
        const float* Bcur = Bs + bbuf * C::B_BUF;
        const bool more = (u + 1) < u1;
        // the halo after the current one: chunk chx (SF: of parity phx)
        int chx = ch + 1, phx = ph;
        if constexpr (SF) sf_next(ph, ch, phx, chx);
        const bool fetch_next = (u == u0 || tap == 0) && (SF ? (u - tap + kkr) < u1 : (ch + 1) * kkr < u1);
        const bool last_tap = (tap == kkr - 1);
        // next unit
        const int ch_n = last_tap ? chx : ch;
        const float* wnext;
        if constexpr (SF) {
            const bool row_end = (kx + 1 == nkx);
            if (!last_tap) {
                wnext = d.wp + wofs(row_end ? ky + 1 : ky, row_end ? 0 : kx + 1) + (size_t)(ch * (CCH / 4)) * CoutP * 4;
            } else {                        // first tap of the next (parity, chunk) pair
                int a0, a1, b0, b1;
                sf_dim(phx >> 1, a0, a1, b1);
                sf_dim(phx & 1, b0, a1, b1);
                wnext = d.wp + (size_t)(a0 * d.ks + b0) * wtap + (size_t)(chx * (CCH / 4)) * CoutP * 4;
            }
        } else if constexpr (PH) {
            const bool row_end = (kx + 1 == nkx);
            wnext = last_tap ? d.wp + wofs(0, 0) + (size_t)((ch + 1) * (CCH / 4)) * CoutP * 4
                             : d.wp + wofs(row_end ? ky + 1 : ky, row_end ? 0 : kx + 1) + (size_t)(ch * (CCH / 4)) * CoutP * 4;
        } else {
            wnext = last_tap ? d.wp + (size_t)((ch + 1) * (CCH / 4)) * CoutP * 4 : wcur + wtap;
        }
        const unsigned m0B = lds_base + (unsigned)(2 * C::A_BUF + (bbuf ^ 1) * C::B_BUF) * 4u + lds_piece;
        const unsigned m0A = lds_base + (unsigned)((abuf ^ 1) * C::A_BUF) * 4u + lds_piece;
        const float* asrc = d.x + chx * CCH;
        const bool nextB_full = more && chunk_cc(ch_n) == CCH;
        const bool nextA_full = fetch_next && chunk_cc(chx) == CCH;

        auto unit_end = [&]() {             // publish the next unit's operands
            if (!more) return;
            const bool need_now = last_tap;                  // the next unit reads the next halo
            if (fetch_next && !need_now && KS != 1) {
                dma_wait_keep(nA);
                a_state = 1;
            } else {
                dma_wait();
                if (fetch_next || a_state == 1) a_state = 2;
            }
            if (need_now && a_state == 2) {
                if (fix_inplace) fixA(chx, Anxt);
                a_state = 0;
            }
            __syncthreads();
        };
        auto advance = [&]() {              // bookkeeping of unit u+1 (valid only if more)
            wcur = wnext;
            bbuf ^= 1;
            if (last_tap) {
                if constexpr (SF) {
                    // into (phx, chx); its halo is landed and fixed up, so the geometry registers move on to the pair
                    // after it
                    if (phx != ph) sf_set_phase(phx);
                    ph = phx;
                    ch = chx;
                    int ph2, ch2;
                    sf_next(ph, ch, ph2, ch2);
                    if (ph2 < 4 && ph2 != gph) load_geom(ph2);
                } else {
                    ch += 1;
                }
                tap = 0; ky = 0; kx = 0; abuf ^= 1;
            } else {
                tap += 1; kx += 1;
                if (kx == ((PH || SF) ? nkx : KS)) { kx = 0; ky += 1; }
            }
            set_tap(ky, kx);
        };

        if (cc == CCH) {
            Frag F1, F2, F3;
            if (!have_f0) read_frag(F0, Acur, Bcur, cb, 0);
            read_frag(F1, Acur, Bcur, cb, 1);
            mma_frag(F0, [&](auto J) {                       // weights of unit u+1
                if (nextB_full) dmaB_slot(J, wnext, m0B);
            });
            if (more && !nextB_full) dmaB(wnext, ch_n, bbuf ^ 1);  // ragged last chunk: generic issue
            read_frag(F2, Acur, Bcur, cb, 2);
            mma_frag(F1, [&](auto J) {                       // halo of chunk ch+1, pieces 0..3
                if (nextA_full) dmaA_slot(J, asrc, m0A, 8);
            });
            read_frag(F3, Acur, Bcur, cb, 3);
            mma_frag(F2, [&](auto J) {                       // pieces 4..7
                constexpr int j = decltype(J)::value;
                if (nextA_full) dmaA_slot(std::integral_constant<int, 4 + j>{}, asrc, m0A, 8);
            });
            if (fetch_next && !nextA_full) dmaA(chx, abuf ^ 1);
            unit_end();
            // The last 16 MFMAs of this unit run AFTER the barrier (their operands are in registers):
            // they cover the next unit's first LDS reads and, one unit after its halo has landed,
            // the in-place fix-up of that halo.
            const bool do_fix = more && a_state == 2;
            const int cb_n = chx * CCH;
            const bool fix_full = chunk_cc(chx) == CCH;
            const bool pre = more && chunk_cc(ch_n) == CCH;
            if (more) advance();
            if (pre) read_frag(F0, As + abuf * C::A_BUF, Bs + bbuf * C::B_BUF, ch * CCH, 0);
            have_f0 = pre;
            mma_frag(F3, [&](auto J) {
                constexpr int j = decltype(J)::value;
                if (do_fix && fix_inplace && fix_full) {
                    fixA_slot(J, cb_n, Anxt);
                    fixA_slot(std::integral_constant<int, 4 + j>{}, cb_n, Anxt);
                }
            });
            if (do_fix) {
                if (fix_inplace && !fix_full) fixA(cb_n / CCH, Anxt);
                a_state = 0;
            }
        } else {
            // ragged last chunk (cc < 32): issue first, then a rolled loop
            if (more) dmaB(wnext, ch_n, bbuf ^ 1);
            if (fetch_next) dmaA(chx, abuf ^ 1);
            if (a_state == 2) {
                if (fix_inplace) fixA(chx, Anxt);
                a_state = 0;
            }
            const int kk8 = cc >> 3;
            for (int kk = 0; kk < kk8; ++kk) {
                Frag f;
                read_frag(f, Acur, Bcur, cb, kk);
                mma_frag(f, [&](auto) {});
            }
            if (cc & 4) {       // tail group c4t: lanes 0-31 take its channels 0,1; lanes 32-63 channels 2,3
                const int c4t = (cc - 4) >> 2;
                f32x2 a2[C::MS], b2[C::NS];
#pragma unroll
                for (int ms = 0; ms < C::MS; ++ms)
                    a2[ms] = *reinterpret_cast<const f32x2*>(Acur + abase[ms] + ((c4t ^ sw[ms]) << 2) + 2 * half);
#pragma unroll
                for (int ns = 0; ns < C::NS; ++ns)
                    b2[ns] = *reinterpret_cast<const f32x2*>(Bcur + c4t * (BN * 4) + bcol[ns] + 2 * half);
                if constexpr (has_tr) {
                    const f32x2 ta = *reinterpret_cast<const f32x2*>(tra + cb + c4t * 4 + 2 * half);
                    const f32x2 tb = *reinterpret_cast<const f32x2*>(trb + cb + c4t * 4 + 2 * half);
#pragma unroll
                    for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                        for (int e = 0; e < 2; ++e) a2[ms][e] = dip_act_leaky(fmaf(ta[e], a2[ms][e], tb[e]), slope);
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ms = 0; ms < C::MS; ++ms)
#pragma unroll
                        for (int ns = 0; ns < C::NS; ++ns)
                            acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ms][j], b2[ns][j], acc[ms][ns], 0, 0, 0);
            }
            unit_end();
            if (more) advance();
            have_f0 = false;
        }
    }
#ifdef DIP_CLK_PROFILE
    const unsigned long long clk_kend = clock64();
    prof[7] = clk_kend - tstart;
#endif

    // ---- epilogue (conv_epilogue.h) ----
    if (ksplit > 1) {
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
                    int oy = ty * C::TH + 2 * sub + (m >> 4);
                    int ox = tx * C::TW + (m & 15);
                    if constexpr (PH) { oy = 2 * oy + py; ox = 2 * ox + px; }
                    if (oy < d.Hout && ox < d.Wout && n < d.Cy)
                        wz[((size_t)oy * d.Wout + ox) * d.Cy + n] = acc[ms][ns][r];
                }
            }
        }
        return;
    }
    DipEpi epi = dip_epi_make(d, ty, tx, C::TH, C::TW);
    if constexpr (PH) {                    // tile (ty, tx) of the phase sub-grid: pixels (2a+py, 2b+px)
        const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
        epi.yt = d.y + ((size_t)(2 * ty * C::TH + py) * pitch + (size_t)(2 * tx * C::TW + px)) * d.Cy;
        epi.row_stride = 2 * pitch * d.Cy;
        epi.Cy = 2 * d.Cy;
        epi.rows_left = ((d.Hout - py + 1) >> 1) - ty * C::TH;
        epi.cols_left = ((d.Wout - px + 1) >> 1) - tx * C::TW;
        epi.full = (epi.rows_left >= C::TH) && (epi.cols_left >= C::TW);
    }
    dip_conv_epilogue<C, BN>(d, acc, epi, n0, wn, wm, l31, half, tid, tile, CoutP, smem);
#ifdef DIP_CLK_PROFILE
    __syncthreads();
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned long long* o = g_prof + (size_t)blockIdx.x * 16;
        for (int i = 0; i < 8; ++i) o[i] = prof[i];
        o[8] = wall0;
        o[9] = wall_clock64();
        o[10] = __builtin_amdgcn_s_getreg(63492);      // HW_ID
        o[11] = __builtin_amdgcn_s_getreg(63508);      // XCC_ID
        o[12] = tstart - clk0;                          // prologue cycles
        o[13] = clock64() - clk_kend;                   // epilogue cycles
    }
#endif
}

template <int KS, int BN, bool TR, int MODE = 0>
int launch_tr(const DipConvDesc& d, hipStream_t st, int n_base, int grid_y, int ksplit, float* ws) {
    using C = DCfg<KS, BN, MODE == 2 ? 4 : 1>;
    constexpr bool PH = (MODE == 1);
    static bool attr_set[16] = {};
    auto kern = conv_igemm_dma_kernel<KS, BN, TR, MODE>;
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, conv_igemm_dma_kernel<KS, BN, TR, MODE, true>, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    // phase mode tiles the (Hout+1)/2 x (Wout+1)/2 sub-grid of one parity, 4 workgroups per tile
    const int ntx = dip_cdiv(PH ? (d.Wout + 1) / 2 : d.Wout, C::TW), nty = dip_cdiv(PH ? (d.Hout + 1) / 2 : d.Hout, C::TH);
    const int ntiles = ntx * nty;
    const int CoutP = dip_round_up(d.Cout, 32);
    dim3 grid(PH ? 4 * ntiles : ntiles, grid_y, ksplit);
    dip_launch_pair<DIP_FAM_DMA>(kern, conv_igemm_dma_kernel<KS, BN, TR, MODE, true>, grid, dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, CoutP, n_base,
                                 ksplit, ws);
    DIP_CHECK_LAUNCH();
    return 0;
}

template <int KS, int BN, int MODE = 0>
int launch(const DipConvDesc& d, hipStream_t st, int n_base, int grid_y, int ksplit, float* ws) {
    return d.tr.a != nullptr ? launch_tr<KS, BN, true, MODE>(d, st, n_base, grid_y, ksplit, ws)
                             : launch_tr<KS, BN, false, MODE>(d, st, n_base, grid_y, ksplit, ws);
}

template <int KS, int MODE = 0>
int launch_bn(const DipConvDesc& d, hipStream_t st, int ksplit, float* ws) {
    const int CoutP = dip_round_up(d.Cout, 32);
    const int nfull = CoutP / 128, rem = CoutP - nfull * 128;
    int rc = 0;
    if (nfull) rc = launch<KS, 128, MODE>(d, st, 0, nfull, ksplit, ws);
    if (rc || !rem) return rc;
    if (rem <= 32) return launch<KS, 32, MODE>(d, st, nfull * 128, 1, ksplit, ws);
    if (rem <= 64) return launch<KS, 64, MODE>(d, st, nfull * 128, 1, ksplit, ws);
    return launch<KS, 128, MODE>(d, st, nfull * 128, 1, ksplit, ws);
}

}  // namespace

// true when this variant can run the descriptor (see the restrictions in the file header)
extern "C" int dip_conv_dma_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    static const bool no_s2 = getenv("DIP_CONV_NO_S2DMA") != nullptr;
    if (d.stride == 2) {                             // strided forward mode: 3x3, undilated input
        if (no_s2 || d.ks != 3 || d.dil != 1 || d.off != 1) return 0;
    } else if (d.stride != 1 || (d.ks != 1 && d.ks != 3)) {
        return 0;
    }
    if (d.pad_mode == DIP_PAD_REPLICATE) return 0;      // (replication padding: the register-staged kernel)
    const bool has_tr = d.tr.a != nullptr;
    if (has_tr && d.Cin > TRN) return 0;
    if (has_tr && d.tr.slope <= 0.f) return 0;      // Swish / ELU: the register-staged kernel applies them at staging
    // 1x1 transforms at the fragment read: a zero-padded or dilated gather would turn pad zeros into
    // act(b) there (never happens for the 1x1 convs of this net: off == 0, dil == 1)
    if (has_tr && d.ks == 1 && (d.off != 0 || d.dil != 1)) return 0;
    return 1;
}

#ifdef DIP_CLK_PROFILE
extern "C" int dip_debug_trace_read(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_trace), sizeof(unsigned) * 128 * 8);
}
extern "C" int dip_debug_prof_read(void* dst, int nwg) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_prof), (size_t)nwg * 16 * sizeof(unsigned long long));
}
#endif

// one 128-column block starting at column n_base (3x3 only): the 132-column data gradients run as
// conv_thin4 (columns 0..3) + this (columns 4..131)
extern "C" int dip_conv_bf3_eligible(const DipConvDesc* dp);
extern "C" int dip_conv_bf3_cols(const DipConvDesc* dp, int n_base, int ncols, void* stream);

extern "C" int dip_conv_igemm_dma_cols(const DipConvDesc* dp, int n_base, void* stream) {
    if (dp->ks != 3) DIP_FAIL("conv_igemm_dma_cols: 3x3 only");
    if (dip_conv_bf3_eligible(dp)) return dip_conv_bf3_cols(dp, n_base, 128, stream);      // (DIP_CONV_BF3: bf16 matrix pipe)
    return launch<3, 128>(*dp, reinterpret_cast<hipStream_t>(stream), n_base, 1, 1, nullptr);
}

// true when the descriptor is the data gradient of a stride-2 3x3 convolution that the phase mode runs:
// dilated input, zero padding, no fused transform / statistics, whole 128-column blocks
extern "C" int dip_conv_phase_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    static const bool off = getenv("DIP_CONV_NO_PHASE") != nullptr;
    if (off) return 0;
    return d.dil == 2 && d.stride == 1 && d.ks == 3 && d.pad_mode == DIP_PAD_ZERO && d.tr.a == nullptr &&
           d.stats == nullptr && d.off >= 1 && d.off <= 2 && (dip_round_up(d.Cout, 32) % 128) == 0 &&
           d.ksplit <= (d.Cin + 31) / 32;
}

extern "C" int dip_conv_igemm_dma(const DipConvDesc* dp, int ksplit, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.dil == 2 && dip_conv_phase_eligible(dp))
        return launch_tr<2, 128, false, 1>(d, st, 0, dip_round_up(d.Cout, 32) / 128, ksplit, d.ws);
    if (d.stride == 2) return launch_bn<2, 2>(d, st, ksplit, d.ws);        // (ks == 3: dip_conv_dma_eligible)
    if (d.ks == 1) return launch_bn<1>(d, st, ksplit, d.ws);
    return launch_bn<3>(d, st, ksplit, d.ws);
}
