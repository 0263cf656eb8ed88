// fp32 convolution on the bf16 matrix pipe ("bf16x9"): 3x3 stride-1 layers at >= 128 x 128.
//
// gfx950 has no TF32-like mode; its fp32 MFMA runs at the fp32 vector rate (157 TFLOP/s), 1/16 of the bf16 MFMA rate.
// But an fp32 number is EXACTLY the sum of three bf16 numbers (a = a1 + a2 + a3: 8 + 8 + 8 significand bits, same
// exponent range), the product of two bf16 numbers is exact in fp32 (8 x 8 bits), and v_mfma_f32_32x32x16_bf16
// accumulates in fp32.  So
//     a * b = sum_{i,j} a_i * b_j        nine exact partial products, summed in fp32
// costs 9 bf16 MFMAs of 32 cycles (K = 16) where the fp32 path needs 8 MFMAs of 64 cycles (K = 2): 0.5625 of the
// matrix-pipe time, with NO loss of precision -- the only roundings are those of the fp32 accumulation, as in the
// fp32 MFMA's own fmaf chain.  Measured (tools/ubench/bf16x9.hip, K = 1152): rel-L2 error against fp64 5.3e-7 for this
// scheme, 6.1e-7 for v_mfma_f32_32x32x2_f32; the inner loop below sustains 235-247 TFLOP/s fp32-equivalent (the fp32 MFMA
// peak is 157.3, the LDS-DMA fp32 kernel reaches 119).
//
// Kernel: implicit GEMM, one workgroup (4 waves) = 8x16 output pixels x 128 output channels, a wave owns 64 x 64 =
// 2 x 2 accumulators.  K is walked in units = (16-channel chunk, tap) = ONE K = 16 MFMA step: 12 ds_read_b128 (2 pixel
// blocks + 2 channel blocks, 3 planes each) feed 36 MFMAs.
//   * A (activations): the halo tile of a chunk (10 x 18 pixels x 16 channels) is loaded fp32 into registers one chunk
//     ahead, put through the producer's BatchNorm + LeakyReLU, split into three bf16 planes and written to the OTHER
//     of two LDS buffers ([plane][pixel][16 k] bf16 = 32 B per pixel; the two 16-B slots of a pixel are swapped
//     with bit 3 of the pixel index, so the 16 lanes of a ds_read_b128 phase hit 16 distinct 4-bank groups);
//   * B (weights): split once per iteration by dip_pack_weights_bf3 ([tap][chunk][plane][n][16 k] bf16) and copied
//     global -> LDS by LDS-DMA two units ahead into one of three 12 KB buffers, the same swizzle applied on the source side;
//   * one barrier per unit publishes the next B buffer (and, every ninth, the next A buffer); two workgroups per CU
//     (76 KB LDS, <= 256 VGPRs) overlap each other's barriers and staging;
//   * epilogue: conv_epilogue.h (bias, store, BatchNorm partial statistics), shared with the fp32 kernels.
// Everything else (stride 2, 1x1, 5x5 / 7x7, the low-resolution layers) stays on the fp32-MFMA kernels.
#include "dip_common.h"
#include "conv_epilogue.h"
#include "lds_dma.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int B3_TR_MAX = 512;
constexpr int B3_CCH = 16;

struct B3Cfg {
    static constexpr int TH = 8, TW = 16, KS = 3;
    static constexpr int HTH = TH - 1 + KS, HTW = TW - 1 + KS, NPIX = HTH * HTW;      // 10 x 18 = 180
    static constexpr int WN = 2, WM = 2, MS = 2, NS = 2;
    static constexpr int A_PLANE = NPIX * 32;            // bytes: [pixel][16 bf16]
    static constexpr int A_BYTES = 3 * A_PLANE;
    static constexpr int B_PLANE = 128 * 32;             // [n][16 bf16]
    static constexpr int B_BYTES = 3 * B_PLANE;
    static constexpr int A_SLOTS = (NPIX * 4 + 255) / 256;      // float4 slots per thread and chunk (3)
    static constexpr int NPIX_PAD = (NPIX + 3) & ~3;
    static constexpr int NBUF_B = 3;                      // weight buffers: the DMA of unit u + 2 is issued in unit u
    static constexpr int LDS_BYTES = 2 * A_BYTES + NBUF_B * B_BYTES + NPIX_PAD * 4 + 2 * B3_TR_MAX * 4;
};

__device__ __forceinline__ int b3_map_src(int v, int n_in, int pad_mode) {
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, n_in);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), n_in - 1);
    return (v < 0 || v >= n_in) ? -1 : v;
}

// exact three-way split by truncation: a == hi + mid + lo, each with <= 8 significand bits (a bf16 number)
__device__ __forceinline__ void b3_split(float a, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned uh = __float_as_uint(a) & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = uh;
    m = um;
    l = __float_as_uint(r2) & 0xFFFF0000u;        // (<= 8 significand bits are left: the mask only drops zeros)
}

// NT = 9: all cross products (exact); NT = 6: without lo*lo, lo*mid, mid*lo (each < 2^-24 of the product)
template <int NT, int TR>
__global__ __launch_bounds__(256, 2) void conv_bf3_kernel(const DipConvDesc d, const int ntx, const int ntiles,
                                                          const int CoutP, const int n_base, const int dbg) {
    using C = B3Cfg;
    constexpr int KK = 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Abuf = smem;
    unsigned char* Bbuf = smem + 2 * C::A_BYTES;
    int* srcoff = reinterpret_cast<int*>(Bbuf + C::NBUF_B * C::B_BYTES);
    float* tra = reinterpret_cast<float*>(srcoff + C::NPIX_PAD);
    float* trb = tra + B3_TR_MAX;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tile = dip_xcd_remap(blockIdx.x, ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int n0 = n_base + blockIdx.y * 128;

    for (int hp = tid; hp < C::NPIX; hp += 256) {
        const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
        const int sr = b3_map_src(ty * C::TH + hr - d.off, d.Hin, d.pad_mode);
        const int sc = b3_map_src(tx * C::TW + hc - d.off, d.Win, d.pad_mode);
        srcoff[hp] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
    }
    const float slope = d.tr.slope;
    if (TR) {
        for (int c = tid; c < d.Cin; c += 256) { tra[c] = d.tr.a[c]; trb[c] = d.tr.b[c]; }
    }

    const int nch = (d.Cin + B3_CCH - 1) / B3_CCH;
    const int nunits = nch * KK;
    const unsigned short* w3 = reinterpret_cast<const unsigned short*>(d.wp3);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS byte offsets of this lane's fragments (without plane / buffer / tap terms)
    int a_pix[2];                       // halo pixel of (ms, lane) at tap (0, 0)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        const int sub = wm * 2 + ms;
        a_pix[ms] = (2 * sub + (l31 >> 4)) * C::HTW + (l31 & 15);
    }
    int b_off[2];
#pragma unroll
    for (int ns = 0; ns < 2; ++ns) {
        const int nl = (wn * 2 + ns) * 32 + l31;
        b_off[ns] = nl * 32 + ((half ^ ((nl >> 3) & 1)) << 4);
    }

    f32x4 av[C::A_SLOTS];
    auto loadA = [&](int ch) {          // fp32 halo of chunk ch -> registers (global loads stay in flight)
        const int cb = ch * B3_CCH;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f >> 2, c = cb + (f & 3) * 4;
            const int so = (hp < C::NPIX && c < d.Cin) ? srcoff[hp < C::NPIX ? hp : 0] : -1;
            // unconditional load from a clamped address (storeA zeroes what is padding): no control flow around the load,
            // and nothing here consumes the loaded registers, so the loads stay in flight under the MFMAs
            av[i] = *reinterpret_cast<const f32x4*>(d.x + (size_t)(so >= 0 ? so : 0) * d.Cx + (so >= 0 ? c : 0));
        }
    };
    auto storeA = [&](int ch, int buf) {          // BatchNorm + activation, exact 3-way split, three bf16 planes
        const int cb = ch * B3_CCH;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f >> 2, c4 = f & 3, c = cb + c4 * 4;
            if (hp < C::NPIX) {
                const bool valid = c < d.Cin && srcoff[hp] >= 0;          // else: zero padding / channels past Cin
                f32x4 o = av[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = valid ? o[e] : 0.f;
                if (TR && valid) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + c), b4 = *reinterpret_cast<const f32x4*>(trb + c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tv = fmaf(a4[e], o[e], b4[e]);
                        o[e] = TR == 1 ? dip_act_leaky(tv, slope) : dip_act(tv, slope);
                    }
                }
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) b3_split(o[e], h[e], m[e], l[e]);
                unsigned char* base = Abuf + buf * C::A_BYTES + hp * 32 + ((((c4 >> 1) ^ (hp >> 3)) & 1) << 4) + ((c4 & 1) << 3);
                // two bf16 per dword: element e in the low half, e + 1 in the high half
                *reinterpret_cast<u32x2*>(base) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
                *reinterpret_cast<u32x2*>(base + C::A_PLANE) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
                *reinterpret_cast<u32x2*>(base + 2 * C::A_PLANE) = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
            }
        }
    };
    // weight DMA of a unit: 3 planes x 128 n x 32 B = 12 pieces of 1 KB, three per wave; the per-lane part of the source
    // offset never changes (piece -> plane, column, swizzled 16-B slot), the unit only moves a wave-uniform base
    unsigned b_src[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int piece = wave + 4 * q, p = piece >> 2, nb = piece & 3;
        const int nl = nb * 32 + (lane >> 1);
        const int slot = (lane & 1) ^ ((nl >> 3) & 1);
        const int nn = min(n0 + nl, CoutP - 1);                  // columns past CoutP are never stored
        b_src[q] = (unsigned)((((p * CoutP + nn) << 4) + (slot << 3)) * 2);       // bytes
    }
    const unsigned b_lds0 = (unsigned)(size_t)(lptr_t)Bbuf;
    auto dmaB = [&](int u, int buf) {
        const int ch = u / KK, tap = u - ch * KK;
        const unsigned char* ubase = reinterpret_cast<const unsigned char*>(w3) + (size_t)(tap * nch + ch) * 3 * CoutP * 32;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int piece = wave + 4 * q;
            lds_dma16_s(ubase, b_src[q], b_lds0 + buf * C::B_BYTES + (piece >> 2) * C::B_PLANE + (piece & 3) * 1024);
        }
    };
    auto compute = [&](int ky, int kx, int abuf, int bbuf) {
        bf16x8 a[2][3], b[2][3];
        const unsigned char* Ab = Abuf + abuf * C::A_BYTES;
        const unsigned char* Bb = Bbuf + bbuf * C::B_BYTES;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int hp = a_pix[ms] + ky * C::HTW + kx;
            const unsigned char* pa = Ab + hp * 32 + (((half ^ (hp >> 3)) & 1) << 4);
#pragma unroll
            for (int p = 0; p < 3; ++p) a[ms][p] = *reinterpret_cast<const bf16x8*>(pa + p * C::A_PLANE);
        }
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[ns][p] = *reinterpret_cast<const bf16x8*>(Bb + p * C::B_PLANE + b_off[ns]);
        // smallest partial products first
#pragma unroll
        for (int s = 4; s >= 0; --s) {
            if (NT == 6 && s > 2) continue;
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                const int pb = s - pa;
                if (pb < 0 || pb > 2) continue;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int ns = 0; ns < 2; ++ns)
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ms][pa], b[ns][pb], acc[ms][ns], 0, 0, 0);
            }
        }
    };

    // ---- prologue ----
    __syncthreads();                    // srcoff / tr tables
    loadA(0);
    dmaB(0, 0);
    if (nunits > 1) dmaB(1, 1);
    storeA(0, 0);
    dma_wait();
    __syncthreads();

    int ch = 0, tap = 0, bcur = 0;
    for (int u = 0; u < nunits; ++u) {
        const bool next_chunk = ch + 1 < nch;
        // (the halo loads go first: hipcc guards the reuse of their registers with s_waitcnt vmcnt(0), which would also
        // wait for the weight DMA if that were already in flight)
        const bool ld = tap == 0 && next_chunk;
        if (ld && !(dbg & 2)) loadA(ch + 1);                              // in flight under this chunk's first MFMAs
        // the next chunk's halo goes to the other A buffer (last read a chunk ago) BEFORE this unit's DMA is issued: hipcc
        // waits for the halo registers with vmcnt counts that do not know about the DMA pieces
        if (tap == 4 && next_chunk && !(dbg & 2)) storeA(ch + 1, (ch + 1) & 1);
        int bnext2 = bcur + 2;
        if (bnext2 >= C::NBUF_B) bnext2 -= C::NBUF_B;
        const bool dma2 = u + 2 < nunits && !(dbg & 1);
        if (dma2) dmaB(u + 2, bnext2);                      // two units ahead: ~1 us to land
        const int ky = tap / 3, kx = tap - 3 * ky;
        if (!(dbg & 8)) compute(ky, kx, ch & 1, bcur);
        if (u + 1 < nunits) {
            // unit u + 1's weights have landed (loads retire in order: at most the 3 DMA pieces of unit u + 2 -- and the
            // halo loads issued before them in this iteration -- may still be in flight)
            if (!dma2) dma_wait();
            else if (ld && !(dbg & 2)) dma_wait_keep(3 + C::A_SLOTS);
            else dma_wait_keep(3);
            if (!(dbg & 4)) __syncthreads();            // ... and everyone else's; A stores visible
        }
        if (++tap == KK) { tap = 0; ++ch; }
        if (++bcur == C::NBUF_B) bcur = 0;
    }

    // ---- epilogue (conv_epilogue.h) ----
    __syncthreads();                    // every wave is done with the staging buffers (the epilogue reuses them)
    const DipEpi epi = dip_epi_make(d, ty, tx, C::TH, C::TW);
    dip_conv_epilogue<C, 128>(d, acc, epi, n0, wn, wm, l31, half, tid, tile, CoutP, reinterpret_cast<float*>(smem));
}

// ------------------------------------------------------------------------------------------------------------------
// weights -> three bf16 planes, [tap][chunk][plane][n][16 k]; forward: k = input channel, n = output channel;
// data gradient: k = output channel, n = input channel, flipped taps
__global__ __launch_bounds__(256) void pack_weights_bf3_kernel(const float* __restrict__ params, unsigned short* __restrict__ out,
                                                               const DipPackRec3* __restrict__ recs) {
    const DipPackRec3 r = recs[blockIdx.y];
    const int KK = r.KS * r.KS;
    const long long nf = r.fwd_off >= 0 ? (long long)KK * r.nchF * r.CoutP32 * 16 : 0;
    const long long nd = r.dgrad_off >= 0 ? (long long)KK * r.nchD * r.CinP32 * 16 : 0;
    const float* w = params + r.w_off;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long long)gridDim.x * 256) {
        float v = 0.f;
        unsigned short* o;
        long long plane;
        if (i < nf) {
            const int kk = (int)(i & 15);
            const int n = (int)((i >> 4) % r.CoutP32);
            const long long rest = (i >> 4) / r.CoutP32;
            const int ch = (int)(rest % r.nchF), tap = (int)(rest / r.nchF);
            const int c = ch * 16 + kk;
            if (n < r.Cout && c < r.Cin) v = w[((size_t)n * r.Cin + c) * KK + tap];
            plane = (long long)r.CoutP32 * 16;
            o = out + r.fwd_off + ((long long)(tap * r.nchF + ch) * 3) * plane + (long long)n * 16 + kk;
        } else {
            const long long j = i - nf;
            const int kk = (int)(j & 15);
            const int n = (int)((j >> 4) % r.CinP32);
            const long long rest = (j >> 4) / r.CinP32;
            const int ch = (int)(rest % r.nchD), tap = (int)(rest / r.nchD);
            const int oc = ch * 16 + kk;
            if (oc < r.Cout && n < r.Cin) v = w[((size_t)oc * r.Cin + n) * KK + (KK - 1 - tap)];
            plane = (long long)r.CinP32 * 16;
            o = out + r.dgrad_off + ((long long)(tap * r.nchD + ch) * 3) * plane + (long long)n * 16 + kk;
        }
        const unsigned uh = __float_as_uint(v) & 0xFFFF0000u;
        const float r1 = v - __uint_as_float(uh);
        const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(um);
        o[0] = (unsigned short)(uh >> 16);
        o[plane] = (unsigned short)(um >> 16);
        o[2 * plane] = (unsigned short)(__float_as_uint(r2) >> 16);
    }
}

int g_bf3_override = -1;             // dip_conv_bf3_set_terms

int bf3_terms() {
    // default: all nine cross products (exact); DIP_CONV_BF3=0: the fp32-MFMA kernels; =6: the six largest products
    static const int v = [] {
        const char* e = getenv("DIP_CONV_BF3");
        if (e == nullptr) return 9;
        const int t = atoi(e);
        return t == 6 ? 6 : (t == 0 ? 0 : 9);
    }();
    return g_bf3_override >= 0 ? g_bf3_override : v;
}

template <int NT, int TR>
int bf3_launch(const DipConvDesc& d, int n_base, int ncols, hipStream_t st) {
    using C = B3Cfg;
    auto kern = conv_bf3_kernel<NT, TR>;
    static bool attr_set[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set[dev] = true;
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    static const int dbg = getenv("DIP_BF3_DEBUG") ? atoi(getenv("DIP_BF3_DEBUG")) : 0;      // timing experiments only
    hipLaunchKernelGGL(kern, dim3(ntiles, dip_cdiv(ncols, 128)), dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, dip_round_up(d.Cout, 32),
                       n_base, dbg);
    DIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// 1 when dip_conv_igemm runs `d` on the bf16 matrix pipe (d->wp3 set; DIP_CONV_BF3=0 switches it off, =6 drops three terms): 3x3, stride 1, dil 1, one pass,
// >= 256 tiles (the layers that are MFMA-bound), at least one full 128-column block
extern "C" int dip_conv_bf3_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    if (bf3_terms() == 0 || d.wp3 == nullptr) return 0;
    if (d.ks != 3 || d.stride != 1 || d.dil != 1 || d.ksplit > 1 || d.accumulate || d.y_pitch > 0) return 0;
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin > d.Cx || d.Cin < 16) return 0;
    if (d.tr.a != nullptr && d.Cin > B3_TR_MAX) return 0;
    if (d.Cout < 128 || d.bnb_y != nullptr) return 0;
    return dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 8) >= 256 ? 1 : 0;
}

// columns [n_base, n_base + ncols) of `d` (ncols a multiple of 128, or the rest of the row of blocks)
extern "C" int dip_conv_bf3_cols(const DipConvDesc* dp, int n_base, int ncols, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nt = bf3_terms();
    if (nt == 0 || d.wp3 == nullptr) DIP_FAIL("conv_bf3: DIP_CONV_BF3 is off or the descriptor has no split weights (wp3)");
    const int tr = d.tr.a == nullptr ? 0 : (d.tr.slope > 0.f ? 1 : 2);
    if (nt == 6) {
        if (tr == 0) return bf3_launch<6, 0>(d, n_base, ncols, st);
        if (tr == 1) return bf3_launch<6, 1>(d, n_base, ncols, st);
        return bf3_launch<6, 2>(d, n_base, ncols, st);
    }
    if (tr == 0) return bf3_launch<9, 0>(d, n_base, ncols, st);
    if (tr == 1) return bf3_launch<9, 1>(d, n_base, ncols, st);
    return bf3_launch<9, 2>(d, n_base, ncols, st);
}

extern "C" int dip_conv_bf3_terms(void) { return bf3_terms(); }
// 0 / 6 / 9: overrides DIP_CONV_BF3 for this process (tests, A/B runs inside one process); -1: back to the environment
extern "C" int dip_conv_bf3_set_terms(int terms) {
    if (terms != -1 && terms != 0 && terms != 6 && terms != 9) DIP_FAIL("conv_bf3_set_terms: 0, 6, 9 or -1");
    g_bf3_override = terms;
    return 0;
}

extern "C" int dip_pack_weights_bf3(const float* params, void* packed3, const DipPackRec3* recs_dev, int nrec, long long max_elems,
                                    void* stream) {
    if (nrec <= 0) return 0;
    long long gx = (max_elems + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(pack_weights_bf3_kernel, dim3((unsigned)gx, nrec), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<unsigned short*>(packed3), recs_dev);
    DIP_CHECK_LAUNCH();
    return 0;
}
