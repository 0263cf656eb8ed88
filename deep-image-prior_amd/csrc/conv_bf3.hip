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
// peak is 157.3, the LDS-DMA fp32 kernel reaches 119).  The default (NT = 8) leaves a3 * b3 out: below 2^-32 of a * b, 2^-8 of
// the rounding error of one accumulation step (see bf3_terms() below).
//
// Kernel: implicit GEMM, one workgroup (4 waves) = 8x16 output pixels x 128 output channels, a wave owns 64 x 64 =
// 2 x 2 accumulators.  K is walked in units = (16-channel chunk, tap) = ONE K = 16 MFMA step: 6 ds_read_b128 (A: 2 pixel
// blocks x 3 planes) + 6 global_load_dwordx4 (B: 2 channel blocks x 3 planes) feed 36 MFMAs.
//   * A (activations): the halo tile of a chunk (10 x 18 pixels x 16 channels) is loaded fp32 into registers one chunk
//     ahead, put through the producer's BatchNorm + LeakyReLU, split into three bf16 planes and written to the OTHER
//     of two LDS buffers ([plane][pixel][16 k] bf16 = 32 B per pixel; the two 16-B slots of a pixel are swapped
//     with bit 3 of the pixel index, so the 16 lanes of a ds_read_b128 phase hit 16 distinct 4-bank groups);
//   * B (weights): split once per iteration by dip_pack_weights_bf3 ([tap][chunk][plane][n][16 k] bf16) and loaded from
//     L2 straight into the MFMA operand registers, one unit ahead (see the comment on the kernel);
//   * one barrier per CHUNK hands the A buffers over; two workgroups per CU (39 KB LDS, ~200 VGPRs);
//   * epilogue: conv_epilogue.h (bias, store, BatchNorm partial statistics), shared with the fp32 kernels.
// Everything else (stride 2, 1x1, 5x5 / 7x7, the low-resolution layers) stays on the fp32-MFMA kernels.
#include "dip_common.h"
#include "dip_group.h"
#include "conv_epilogue.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef DIP_W3_PROFILE
// clock probe of the profile build (tools/w3_profile.py): per workgroup {s_memtime cycles, s_memrealtime ticks (100 MHz)} of wave 0
__device__ unsigned long long g_b3_prof[8192 * 2];
#endif

constexpr int B3_TR_MAX = 512;
constexpr int B3_CCH = 16;

// BN_ = 128: the form of the big layers (a wave owns 64 pixels x 64 columns).  BN_ = 64: the same tile with 32 columns per wave,
// i.e. two workgroups per pixel tile -- for layers with < 256 tiles, which 128-column workgroups cannot spread over 256 CUs
// (round 5: 128^2 layers 66-85 us on the fp32 split-K kernels -> this form, +1.9 % per iteration; profiles/r05_ab_n64.txt)
template <int BN_ = 128, int KS_ = 3>
struct B3Cfg {
    static constexpr int TH = 8, TW = 16, KS = KS_;
    static constexpr int HTH = TH - 1 + KS, HTW = TW - 1 + KS, NPIX = HTH * HTW;      // 10 x 18 = 180
    static constexpr int BN = BN_;
    static constexpr int WN = 2, WM = 2, MS = 2, NS = BN_ / 64;
    static constexpr int A_PLANE = NPIX * 32;            // bytes: [pixel][16 bf16]
    static constexpr int A_BYTES = 3 * A_PLANE;
    static constexpr int A_SLOTS = (NPIX * 4 + 255) / 256;      // float4 slots per thread and chunk (3)
    static constexpr int NPIX_PAD = (NPIX + 3) & ~3;
    static constexpr int LDS_BYTES = 2 * A_BYTES + NPIX_PAD * 4 + 2 * B3_TR_MAX * 4;      // 39.4 KB
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

// NT = 9: all cross products (exact); NT = 8: without lo*lo (< 2^-32 of the product: 2^-8 of the rounding error of ONE fp32
// accumulation step, so the sum is as accurate as with it); NT = 6: also without lo*mid, mid*lo (each < 2^-24 of the product).
//
// The B operand (weights) goes straight from L2 into registers, one unit ahead: no weight buffers in LDS, no LDS-DMA, and
// ONE workgroup barrier per 16-channel chunk (9 units) -- the waves of a workgroup only meet where the A buffers change hands.
// Round 4 started with the LDS-DMA form of the fp32 kernel (weights of a unit DMA'd two units ahead into one of three 12 KB
// LDS buffers, one barrier per unit); per-workgroup wall-clock stamps and per-wave s_memtime sums showed a wave spending about
// as long outside its MFMA burst (DMA issue, vmcnt wait, barrier, LDS-read latency: ~1000 cycles per unit) as in it (1152),
// and the two waves that share a SIMD falling into step -- both in their bursts at half rate, then both outside with the
// matrix pipe idle (counter utilisation 0.62; s_setprio by wave slot made one workgroup fast and the other slow, the sum
// stayed).  This form: 540 -> 495 us on 128 -> 128 @ 512^2, 156 -> 132 us @ 256^2, +3.2 % per iteration.
// A lane's B fragment is 16 contiguous bytes of the packed plane row of its column ([tap][chunk][plane][n][16 k]: lanes
// 0..31 the first 8 k of 32 rows, lanes 32..63 the second 8 -- one contiguous KB per load instruction), 6 loads per unit;
// the co-resident workgroup and the wave with the same column block read the same lines (vL1D / L2 hits).
// A fragments are read from LDS one unit ahead as well (the A buffer of a chunk does not change while it is walked).
template <int NT, int TR, int BN = 128>
__global__ __launch_bounds__(256, 2) void conv_bf3_kernel(const DipConvDesc d, const int ntx, const int ntiles,
                                                           const int CoutP, const int n_base, const int tailk) {
    using C = B3Cfg<BN>;
    constexpr int KK = 9;
    constexpr int NS = C::NS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Abuf = smem;
    int* srcoff = reinterpret_cast<int*>(smem + 2 * C::A_BYTES);
    float* tra = reinterpret_cast<float*>(srcoff + C::NPIX_PAD);
    float* trb = tra + B3_TR_MAX;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
#ifdef DIP_W3_PROFILE
    const unsigned long long pk0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int tile = dip_xcd_remap(blockIdx.x, ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int n0 = n_base + blockIdx.y * BN;

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
    // A 4-channel last chunk (132 = 8 x 16 + 4 input channels: the decoder convs on [4 skip | 128] channels) would spend nine
    // units on 4 real k of 16.  Its K is PACKED instead: k = (tap, channel), three units of 4 taps x 4 channels (the 36 real
    // products in 48 slots instead of 144); a lane's eight k are two taps of its pixel = two 8-byte reads at two halo offsets;
    // dip_pack_weights_bf3 stores the matching B rows in the unit slots (tap 0..2, last chunk) of the ordinary layout, so the
    // weight loads do not change.  Six units of 81 less: 482 -> ~450 us on the 512^2 layer.
    const bool tail = tailk && nch > 1 && (d.Cin & 15) == 4;
    const int nunits = tail ? (nch - 1) * KK + 3 : nch * KK;

    f32x16 acc[2][NS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int a_pix[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) a_pix[ms] = (2 * (wm * 2 + ms) + (l31 >> 4)) * C::HTW + (l31 & 15);

    // ---- A: as in conv_bf3_kernel (halo fp32 -> registers one chunk ahead -> transform, exact split, three bf16 planes) ----
    f32x4 av[C::A_SLOTS];
#pragma unroll
    for (int i = 0; i < C::A_SLOTS; ++i) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // halo loads of chunk `ch`, issued only when `on` (wave-uniform): the branch is INSIDE the asm statement, so that for the
    // compiler the statement is unconditional straight-line code -- a compiler-visible branch around an asm load makes it
    // merge the "loaded" and "not loaded" values of av[] behind the branch with register copies, i.e. it reads registers
    // whose load is still in flight (seen in the ISA of an earlier form of this kernel)
    auto loadA = [&](int ch, bool on) {
        const int cb = ch * B3_CCH;
        const float* src[C::A_SLOTS];
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) src[i] = d.x;
        if (on) {
#pragma unroll
            for (int i = 0; i < C::A_SLOTS; ++i) {
                const int f = tid + i * 256;
                const int hp = f >> 2, c = cb + (f & 3) * 4;
                const int so = (hp < C::NPIX && c < d.Cin) ? srcoff[hp < C::NPIX ? hp : 0] : -1;
                src[i] = d.x + (size_t)(so >= 0 ? so : 0) * d.Cx + (so >= 0 ? c : 0);     // clamped: storeA zeroes the padding
            }
        }
        const int on_s = __builtin_amdgcn_readfirstlane(on ? 1 : 0);
        asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lb3r_noA_%=\n\t"
                     "global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %4, off\n\tglobal_load_dwordx4 %2, %5, off\n"
                     ".Lb3r_noA_%=:"
                     : "+v"(av[0]), "+v"(av[1]), "+v"(av[2])
                     : "v"(src[0]), "v"(src[1]), "v"(src[2]), "s"(on_s)
                     : "scc");
    };
    auto storeA = [&](int ch, int buf) {
        const int cb = ch * B3_CCH;
#pragma unroll
        for (int i = 0; i < C::A_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f >> 2, c4 = f & 3, c = cb + c4 * 4;
            if (hp < C::NPIX) {
                const bool valid = c < d.Cin && srcoff[hp] >= 0;
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
                *reinterpret_cast<u32x2*>(base) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
                *reinterpret_cast<u32x2*>(base + C::A_PLANE) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
                *reinterpret_cast<u32x2*>(base + 2 * C::A_PLANE) = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
            }
        }
    };

    // ---- B: two register sets (current unit, next unit in flight: a third set / two units ahead measured 2 % slower) of
    // 2 column blocks x 3 planes; per-lane byte offset inside a plane, wave-uniform unit base ----
    bf16x8 bq[2][NS][3];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[sidx][ns][p] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    unsigned b_voff[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nn = min(n0 + (wn * NS + ns) * 32 + l31, CoutP - 1);          // columns past CoutP are never stored
        b_voff[ns] = (unsigned)(nn * 32 + half * 16);
    }
    const unsigned char* w3 = reinterpret_cast<const unsigned char*>(d.wp3);
    const size_t plane_bytes = (size_t)CoutP * 32;
    typedef bf16x8 (&BSet)[NS][3];
    auto loadB = [&](BSet bs, int tapn, int chn) {
        const unsigned char* ub = w3 + (size_t)(tapn * nch + chn) * 3 * plane_bytes;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            // wave-uniform base (SGPR pair) + 32-bit per-lane offset
            const unsigned long long b64 = (unsigned long long)(ub + p * plane_bytes);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b64);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
            const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(bs[ns][p]) : "v"(b_voff[ns]), "s"(sb));
        }
    };
    // all B loads of the set have landed; keep3 (wave-uniform): the 3 halo loads issued after them may stay in flight.
    // One unconditional statement with the choice inside (see loadA).
    auto waitB = [&](BSet bs, bool keep3) {
        const int k_s = __builtin_amdgcn_readfirstlane(keep3 ? 1 : 0);
        if constexpr (NS == 2) {
            asm volatile("s_cmp_eq_u32 %6, 0\n\ts_cbranch_scc1 .Lb3r_w0_%=\n\ts_waitcnt vmcnt(3)\n\ts_branch .Lb3r_we_%=\n"
                         ".Lb3r_w0_%=:\n\ts_waitcnt vmcnt(0)\n.Lb3r_we_%=:"
                         : "+v"(bs[0][0]), "+v"(bs[0][1]), "+v"(bs[0][2]), "+v"(bs[1][0]), "+v"(bs[1][1]), "+v"(bs[1][2])
                         : "s"(k_s)
                         : "scc");
        } else {
            asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lb3r_w0_%=\n\ts_waitcnt vmcnt(3)\n\ts_branch .Lb3r_we_%=\n"
                         ".Lb3r_w0_%=:\n\ts_waitcnt vmcnt(0)\n.Lb3r_we_%=:"
                         : "+v"(bs[0][0]), "+v"(bs[0][1]), "+v"(bs[0][2])
                         : "s"(k_s)
                         : "scc");
        }
    };
    // A fragments of a unit: 2 pixel blocks x 3 planes, read one unit ahead into the other register set (the A buffer of a
    // chunk does not change while the chunk is walked, so this needs no synchronisation beyond the chunk boundary's)
    typedef bf16x8 (&ASet)[2][3];
    bf16x8 aq[2][2][3];
    auto readA = [&](ASet as, int tapn, int abuf, bool packed) {
        const unsigned char* Ab = Abuf + abuf * C::A_BYTES;
        if (packed) {                   // (wave-uniform) packed unit tapn of the 4-channel chunk: this lane's k = taps tA, tA + 1
            const int tA = 4 * tapn + 2 * half, tB = tA + 1;
            const int oA = tA < KK ? (tA / 3) * C::HTW + tA % 3 : 0, oB = tB < KK ? (tB / 3) * C::HTW + tB % 3 : 0;   // (taps >= 9: zero weights)
#pragma unroll
            for (int ms = 0; ms < 2; ++ms) {
                const int hA = a_pix[ms] + oA, hB = a_pix[ms] + oB;
                const unsigned char* pA = Ab + hA * 32 + (((hA >> 3) & 1) << 4);      // channels 0..3 of the chunk: k 0..3 of the pixel's slot
                const unsigned char* pB = Ab + hB * 32 + (((hB >> 3) & 1) << 4);
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const u32x2 vA = *reinterpret_cast<const u32x2*>(pA + p * C::A_PLANE), vB = *reinterpret_cast<const u32x2*>(pB + p * C::A_PLANE);
                    as[ms][p] = __builtin_bit_cast(bf16x8, u32x4{vA[0], vA[1], vB[0], vB[1]});
                }
            }
            return;
        }
        const int ky = tapn / 3, kx = tapn - 3 * ky;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int hp = a_pix[ms] + ky * C::HTW + kx;
            const unsigned char* pa = Ab + hp * 32 + (((half ^ (hp >> 3)) & 1) << 4);
#pragma unroll
            for (int p = 0; p < 3; ++p) as[ms][p] = *reinterpret_cast<const bf16x8*>(pa + p * C::A_PLANE);
        }
    };
    auto compute = [&](BSet bs, ASet as) {
#pragma unroll
        for (int sm = 4; sm >= 0; --sm) {          // smallest partial products first
            if ((NT == 6 && sm > 2) || (NT == 8 && sm > 3)) continue;
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                const int pb = sm - pa;
                if (pb < 0 || pb > 2) continue;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[ms][pa], bs[ns][pb], acc[ms][ns], 0, 0, 0);
            }
        }
    };

    // ---- prologue ----
    __syncthreads();                    // srcoff / tr tables
    loadA(0, true);
    loadB(bq[0], 0, 0);
    if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(6)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]));       // the halo (the 6 B loads are younger)
    else asm volatile("s_waitcnt vmcnt(3)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]));                          // (3 B loads)
    storeA(0, 0);
    waitB(bq[0], false);
    __syncthreads();
    readA(aq[0], 0, 0, false);

    int u = 0, ch = 0, tap = 0;
    auto unit = [&](BSet cur, BSet nxt, ASet acur, ASet anxt) {
        const bool next_chunk = ch + 1 < nch;
        // the next chunk's halo (loaded at tap 0, landed by the end of tap 1: the B loads of tap 1 are younger and were waited
        // for) goes to the other A buffer, last read a chunk ago (barrier at the end of tap 7)
        if (tap == 4 && next_chunk) storeA(ch + 1, (ch + 1) & 1);
        // next unit's weights (the last unit re-loads unit 0: never used, keeps the statement unconditional)
        const bool more = u + 1 < nunits;
        const bool wrap = tap == KK - 1;
        const int tapn = more ? (wrap ? 0 : tap + 1) : 0, chn = more ? (wrap ? ch + 1 : ch) : 0;
        // (issuing the six loads one by one between the groups of four MFMAs instead: no faster at 512^2, 5 % slower at 256^2;
        // a third register set, i.e. two units ahead: 2 % slower)
        loadB(nxt, tapn, chn);
        const bool ld = tap == 0 && next_chunk;
        loadA(ch + 1, ld);                                       // AFTER this unit's B loads: they can be waited for alone
        readA(anxt, tapn, chn & 1, tail && chn == nch - 1);      // next unit's A fragments, under this unit's MFMAs
        compute(cur, acur);
        __builtin_amdgcn_sched_barrier(0);          // (hipcc hoisted the wait to the 4th MFMA: a full L2 latency exposed per unit)
        waitB(nxt, ld);
        // A buffers change hands: the stores of tap 4 become visible before tap 8 reads the next chunk's first fragments, and
        // every wave has issued (and, __syncthreads waits lgkmcnt(0), received) its last reads of the buffer that tap 4 of the
        // next chunk overwrites
        if (tap == KK - 2 && next_chunk) __syncthreads();
        ++u;
        if (++tap == KK) { tap = 0; ++ch; }
    };
    while (u < nunits) {
        unit(bq[0], bq[1], aq[0], aq[1]);
        if (u < nunits) unit(bq[1], bq[0], aq[1], aq[0]);
    }
    // the last unit's (unused) weight loads have been waited for (keep3 is false in the last chunk); nothing is in flight

    // ---- epilogue (conv_epilogue.h) ----
    __syncthreads();
#ifdef DIP_W3_PROFILE
    if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 8192) {
        g_b3_prof[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - pk0;
        g_b3_prof[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - pr0;
    }
#endif
    const DipEpi epi = dip_epi_make(d, ty, tx, C::TH, C::TW);
    dip_conv_epilogue<C, BN>(d, acc, epi, n0, wn, wm, l31, half, tid, tile, CoutP, reinterpret_cast<float*>(smem));
}

// ------------------------------------------------------------------------------------------------------------------
// The 1x1 form (round 6): the need1x1_up convs of models/skip.py:88-91 at the high resolutions, forward and data gradient, on
// the same arithmetic.  A 1x1 layer has ONE unit per 16-channel chunk, so the kernel above -- which hides a chunk's staging
// under nine units of MFMAs and meets at a barrier once per chunk -- has nothing to hide under; the pipeline here is per
// chunk: in unit ch a wave issues the weights of chunk ch + 1 and the fp32 pixels of chunk ch + 2 (two register sets take
// turns), splits chunk ch + 1 out of the other register set into the other A buffer, runs the MFMAs of chunk ch out of
// fragments it read at the end of the previous unit, meets the workgroup, and reads the fragments of chunk ch + 1.  Every
// load is unconditional (a chunk past the end re-reads the last one), so the vmcnt values below are exact: 8 = the 6 weight
// loads + 2 pixel loads issued behind the pixel loads being waited for (5 in the 64-column form), 2 = the pixel loads behind
// the weights.  Tile, accumulator layout and epilogue are the 3x3 kernel's (8 x 16 pixels, no halo).
// MEASURED, NOT THE DEFAULT (DIP_CONV_BF3_1X1=1 switches it on).  Against the fp32 form (conv1x1_res.hip, weights-resident,
// persistent, LDS-DMA: 102 us forward / 85 us data gradient at 512^2 under rocprofv3) this kernel takes 98 / 85 us -- eight
// units per tile do not amortise a tile's prologue and epilogue, which the persistent fp32 kernel hides under the next tile --
// and the ITERATION loses 3.3 % with it (186.5 -> 180.3 it/s, interleaved A/B): every other kernel runs 1 - 4 % slower behind
// two more launches on the bf16 pipe (the chip is power-managed: DESIGN 3.6).  profiles/r06_conv_bf3_1x1_dead_end.txt.
template <int NT, int TR, int BN = 128>
__global__ __launch_bounds__(256, 2) void conv_bf3_k1_kernel(const DipConvDesc d, const int ntx, const int ntiles,
                                                              const int CoutP, const int n_base, const int /*tailk*/) {
    using C = B3Cfg<BN, 1>;
    constexpr int NS = C::NS;
    static_assert(C::A_SLOTS == 2 && C::NPIX == 128, "one float4 pair per thread and chunk");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Abuf = smem;
    int* srcoff = reinterpret_cast<int*>(smem + 2 * C::A_BYTES);
    float* tra = reinterpret_cast<float*>(srcoff + C::NPIX_PAD);
    float* trb = tra + B3_TR_MAX;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tile = dip_xcd_remap(blockIdx.x, ntiles);
    const int ty = tile / ntx, tx = tile - ty * ntx;
    const int n0 = n_base + blockIdx.y * BN;

    if (tid < C::NPIX) {
        const int hr = tid / C::HTW, hc = tid - hr * C::HTW;
        const int sr = b3_map_src(ty * C::TH + hr - d.off, d.Hin, d.pad_mode);
        const int sc = b3_map_src(tx * C::TW + hc - d.off, d.Win, d.pad_mode);
        srcoff[tid] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
    }
    const float slope = d.tr.slope;
    if (TR) {
        for (int c = tid; c < d.Cin; c += 256) { tra[c] = d.tr.a[c]; trb[c] = d.tr.b[c]; }
    }
    const int nch = (d.Cin + B3_CCH - 1) / B3_CCH;

    f32x16 acc[2][NS];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_pix[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) a_pix[ms] = (2 * (wm * 2 + ms) + (l31 >> 4)) * C::HTW + (l31 & 15);

    __syncthreads();                    // srcoff / tr tables
    // this thread's two staging slots: float4 f = tid + 256 i -> pixel f / 4, channels 4 (f % 4) .. + 3 of the chunk
    int so[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) so[i] = srcoff[(tid + i * 256) >> 2];
    const int c4 = tid & 3;
    typedef f32x4 (&AReg)[2];
    f32x4 av[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) av[0][i] = av[1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto loadA = [&](AReg a, int ch) {
        const int c = min(ch, nch - 1) * B3_CCH + c4 * 4;
        const bool cv = c < d.Cin;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float* src = d.x + (size_t)(so[i] >= 0 ? so[i] : 0) * d.Cx + (cv ? c : 0);      // clamped: storeA zeroes the padding
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a[i]) : "v"(src));
        }
    };
    auto storeA = [&](AReg a, int ch, int buf) {
        const int c = ch * B3_CCH + c4 * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hp = (tid + i * 256) >> 2;
            const bool valid = c < d.Cin && so[i] >= 0;
            f32x4 o = a[i];
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
            *reinterpret_cast<u32x2*>(base) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
            *reinterpret_cast<u32x2*>(base + C::A_PLANE) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
            *reinterpret_cast<u32x2*>(base + 2 * C::A_PLANE) = u32x2{(l[0] >> 16) | l[1], (l[2] >> 16) | l[3]};
        }
    };
    bf16x8 bq[2][NS][3];
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
        for (int ns = 0; ns < NS; ++ns)
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[sidx][ns][p] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    unsigned b_voff[NS];
#pragma unroll
    for (int ns = 0; ns < NS; ++ns) {
        const int nn = min(n0 + (wn * NS + ns) * 32 + l31, CoutP - 1);
        b_voff[ns] = (unsigned)(nn * 32 + half * 16);
    }
    const unsigned char* w3 = reinterpret_cast<const unsigned char*>(d.wp3);
    const size_t plane_bytes = (size_t)CoutP * 32;
    typedef bf16x8 (&BSet)[NS][3];
    auto loadB = [&](BSet bs, int ch) {
        const unsigned char* ub = w3 + (size_t)min(ch, nch - 1) * 3 * plane_bytes;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const unsigned long long b64 = (unsigned long long)(ub + p * plane_bytes);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)b64);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(b64 >> 32));
            const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
#pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(bs[ns][p]) : "v"(b_voff[ns]), "s"(sb));
        }
    };
    auto waitA = [&](AReg a) {          // the pixel loads of `a` have landed; the 3 NS weight + 2 pixel loads behind them stay in flight
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(3 * NS + 2));
    };
    auto waitB = [&](BSet bs) {         // the weight loads of `bs` have landed; the 2 pixel loads behind them stay in flight
        if constexpr (NS == 2)
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(bs[0][0]), "+v"(bs[0][1]), "+v"(bs[0][2]), "+v"(bs[1][0]), "+v"(bs[1][1]), "+v"(bs[1][2]));
        else
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(bs[0][0]), "+v"(bs[0][1]), "+v"(bs[0][2]));
    };
    typedef bf16x8 (&ASet)[2][3];
    bf16x8 aq[2][2][3];
    auto readA = [&](ASet as, int abuf) {
        const unsigned char* Ab = Abuf + abuf * C::A_BYTES;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            const int hp = a_pix[ms];
            const unsigned char* pa = Ab + hp * 32 + (((half ^ (hp >> 3)) & 1) << 4);
#pragma unroll
            for (int p = 0; p < 3; ++p) as[ms][p] = *reinterpret_cast<const bf16x8*>(pa + p * C::A_PLANE);
        }
    };
    auto compute = [&](BSet bs, ASet as) {
#pragma unroll
        for (int sm = 4; sm >= 0; --sm) {          // smallest partial products first
            if ((NT == 6 && sm > 2) || (NT == 8 && sm > 3)) continue;
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                const int pb = sm - pa;
                if (pb < 0 || pb > 2) continue;
#pragma unroll
                for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        acc[ms][ns] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as[ms][pa], bs[ns][pb], acc[ms][ns], 0, 0, 0);
            }
        }
    };

    // ---- prologue: chunk 0 staged and its fragments read, chunk 1's pixels in flight ----
    loadA(av[0], 0);
    loadB(bq[0], 0);
    loadA(av[1], 1);
    waitA(av[0]);
    storeA(av[0], 0, 0);
    waitB(bq[0]);
    __syncthreads();
    readA(aq[0], 0);

    // unit ch: register sets / buffers of chunk ch are `cur`, of chunk ch + 1 `nxt`
    auto unit = [&](int ch, BSet bcur, BSet bnxt, ASet acur, ASet anxt, AReg vcur, AReg vnxt, int nbuf) {
        loadB(bnxt, ch + 1);
        loadA(vcur, ch + 2);                       // (chunk ch left this set in unit ch - 1)
        waitA(vnxt);                               // chunk ch + 1, issued a unit ago
        storeA(vnxt, min(ch + 1, nch - 1), nbuf);  // buffer nbuf was last read before the previous barrier
        compute(bcur, acur);
        __builtin_amdgcn_sched_barrier(0);
        waitB(bnxt);
        __syncthreads();
        readA(anxt, nbuf);
    };
    for (int ch = 0; ch < nch; ch += 2) {
        unit(ch, bq[0], bq[1], aq[0], aq[1], av[0], av[1], 1);
        if (ch + 1 < nch) unit(ch + 1, bq[1], bq[0], aq[1], aq[0], av[1], av[0], 0);
    }
    asm volatile("s_waitcnt vmcnt(0)");             // the loads issued past the last chunk

    // ---- epilogue (conv_epilogue.h) ----
    __syncthreads();
    const DipEpi epi = dip_epi_make(d, ty, tx, C::TH, C::TW);
    dip_conv_epilogue<C, BN>(d, acc, epi, n0, wn, wm, l31, half, tid, tile, CoutP, reinterpret_cast<float*>(smem));
}

// ------------------------------------------------------------------------------------------------------------------
// weights -> three bf16 planes, [tap][chunk][plane][n][16 k]; forward: k = input channel, n = output channel;
// data gradient: k = output channel, n = input channel, flipped taps
template <bool GRP = false>
__global__ __launch_bounds__(256) void pack_weights_bf3_kernel(const float* __restrict__ params_, unsigned short* __restrict__ out_,
                                                               const DipPackRec3* __restrict__ recs_, const int tailk,
                                                               const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, params);
    DIP_GRP_PTR(unsigned short*, out);
    DIP_GRP_PTR(const DipPackRec3*, recs);
    const DipPackRec3 r = recs[blockIdx.y];
    const int KK = r.KS * r.KS;
    const long long nf = r.fwd_off >= 0 ? (long long)KK * r.nchF * r.CoutP32 * 16 : 0;
    const long long nd = r.dgrad_off >= 0 ? (long long)KK * r.nchD * r.CinP32 * 16 : 0;
    const float* w = params + r.w_off;
    // a last chunk of 4 channels is stored with its K packed as (tap, channel) -- conv_bf3_kernel's `tail` (same rule: the
    // contraction length rounded up to 4 is 16 m + 4, m >= 1; 3x3 only)
    const bool tailF = tailk && r.KS == 3 && r.nchF > 1 && (((r.Cin + 3) & ~3) & 15) == 4;
    const bool tailD = tailk && r.KS == 3 && r.nchD > 1 && (((r.Cout + 3) & ~3) & 15) == 4;
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
            if (tailF && ch == r.nchF - 1) {        // packed K of the 4-channel chunk: unit slot `tap` < 3 holds taps 4 tap .. 4 tap + 3 x 4 channels
                const int t = 4 * tap + (kk >> 2), cc = ch * 16 + (kk & 3);
                if (tap < 3 && t < KK && n < r.Cout && cc < r.Cin) v = w[((size_t)n * r.Cin + cc) * KK + t];
            } else if (n < r.Cout && c < r.Cin) v = w[((size_t)n * r.Cin + c) * KK + tap];
            plane = (long long)r.CoutP32 * 16;
            o = out + r.fwd_off + ((long long)(tap * r.nchF + ch) * 3) * plane + (long long)n * 16 + kk;
        } else {
            const long long j = i - nf;
            const int kk = (int)(j & 15);
            const int n = (int)((j >> 4) % r.CinP32);
            const long long rest = (j >> 4) / r.CinP32;
            const int ch = (int)(rest % r.nchD), tap = (int)(rest / r.nchD);
            const int oc = ch * 16 + kk;
            if (tailD && ch == r.nchD - 1) {
                const int t = 4 * tap + (kk >> 2), occ = ch * 16 + (kk & 3);
                if (tap < 3 && t < KK && occ < r.Cout && n < r.Cin) v = w[((size_t)occ * r.Cin + n) * KK + (KK - 1 - t)];
            } else if (oc < r.Cout && n < r.Cin) v = w[((size_t)oc * r.Cin + n) * KK + (KK - 1 - tap)];
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

// DIP_CONV_BF3_NO_TAILK=1: a 4-channel last chunk walks nine zero-padded units as before round 6 (A/B switch; the weight pack
// and the kernel read the same answer)
bool bf3_tailk() {
    static const bool on = getenv("DIP_CONV_BF3_NO_TAILK") == nullptr;
    return on;
}

int bf3_terms() {
    // default: eight cross products -- all but lo x lo, which is < 2^-32 of a product (2^-8 of the rounding error of one fp32
    // accumulation step: measured error vs fp64 identical to the nine-product form, tests/test_bf3_gpu.py; +2.6 % per iteration);
    // DIP_CONV_BF3=9: all nine (every product exact); =0: the fp32-MFMA kernels; =6: the six largest products
    static const int v = [] {
        const char* e = getenv("DIP_CONV_BF3");
        if (e == nullptr) return 8;
        const int t = atoi(e);
        return t == 6 ? 6 : (t == 9 ? 9 : (t == 0 ? 0 : 8));
    }();
    return g_bf3_override >= 0 ? g_bf3_override : v;
}

// layers with 96..255 tiles (the 128^2 layers of the default net) run the 64-column form of the kernel -- two workgroups per
// pixel tile -- instead of the fp32 split-K kernels
constexpr int B3_N64_MIN_TILES = 96;
// 1x1 layers: from 256 tiles (the 128-column form; below, conv_small / the split-K kernels keep them)
constexpr int B3_K1_MIN_TILES = 256;

template <int NT, int TR, int BN, int KS = 3>
int bf3_launch_bn(const DipConvDesc& d, int n_base, int ncols, hipStream_t st) {
    using C = B3Cfg<BN, KS>;
    auto kern = [] {
        if constexpr (KS == 1) return conv_bf3_k1_kernel<NT, TR, BN>;
        else return conv_bf3_kernel<NT, TR, BN>;
    }();
    static bool attr_set[16] = {};
    if (dip_once_per_device(attr_set)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    dip_launch(kern, dim3(ntiles, dip_cdiv(ncols, BN)), dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, dip_round_up(d.Cout, 32), n_base,
               bf3_tailk() ? 1 : 0);
    DIP_CHECK_LAUNCH();
    return 0;
}

template <int NT, int TR>
int bf3_launch(const DipConvDesc& d, int n_base, int ncols, hipStream_t st) {
    if (d.ks == 1) {
        if (dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 8) < 256) return bf3_launch_bn<NT, TR, 64, 1>(d, n_base, ncols, st);
        return bf3_launch_bn<NT, TR, 128, 1>(d, n_base, ncols, st);
    }
    if (dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 8) < 256) return bf3_launch_bn<NT, TR, 64>(d, n_base, ncols, st);
    return bf3_launch_bn<NT, TR, 128>(d, n_base, ncols, st);
}

}  // namespace

#ifdef DIP_W3_PROFILE
extern "C" int dip_b3_prof_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_b3_prof), (size_t)n * sizeof(unsigned long long));
}
#endif

// 1 when dip_conv_igemm runs `d` on the bf16 matrix pipe (d->wp3 set; DIP_CONV_BF3=0 switches it off, =6 drops three terms): 3x3, stride 1, dil 1, one pass,
// >= 96 tiles (128 columns per workgroup from 256 tiles, 64 below), at least one full 128-column block
extern "C" int dip_conv_bf3_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    if (bf3_terms() == 0 || d.wp3 == nullptr) return 0;
    // the 1x1 form is opt-in (DIP_CONV_BF3_1X1=1, read at every call so that a test can switch it): measured no faster than the
    // fp32 weights-resident kernel per launch and 3.3 % SLOWER per iteration (profiles/r06_conv_bf3_1x1_dead_end.txt)
    const char* k1e = getenv("DIP_CONV_BF3_1X1");
    const bool k1 = k1e != nullptr && k1e[0] == '1';
    if ((d.ks != 3 && !(d.ks == 1 && k1)) || d.stride != 1 || d.dil != 1 || d.ksplit > 1 || d.accumulate || d.y_pitch > 0) return 0;
    if (d.ks == 1 && ((d.Cin & 15) || dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 8) < B3_K1_MIN_TILES)) return 0;
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin > d.Cx || d.Cin < 16) return 0;
    if (d.tr.a != nullptr && d.Cin > B3_TR_MAX) return 0;
    if (d.Cout < 128 || d.bnb_y != nullptr) return 0;
    return dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 8) >= B3_N64_MIN_TILES ? 1 : 0;
}

// 1 when the planner (dip_conv_plan) must give a 3x3 stride-1 layer with `ntiles` tiles a one-pass plan because it will run
// on the 64-column form of the bf16-pipe kernel
extern "C" int dip_conv_bf3_n64_plan(int ntiles, int Cin, int Cout) {
    return bf3_terms() != 0 && ntiles >= B3_N64_MIN_TILES && ntiles < 256 && Cout >= 128 && Cin >= 16 ? 1 : 0;
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
    if (nt == 8) {
        if (tr == 0) return bf3_launch<8, 0>(d, n_base, ncols, st);
        if (tr == 1) return bf3_launch<8, 1>(d, n_base, ncols, st);
        return bf3_launch<8, 2>(d, n_base, ncols, st);
    }
    if (tr == 0) return bf3_launch<9, 0>(d, n_base, ncols, st);
    if (tr == 1) return bf3_launch<9, 1>(d, n_base, ncols, st);
    return bf3_launch<9, 2>(d, n_base, ncols, st);
}

extern "C" int dip_conv_bf3_terms(void) { return bf3_terms(); }
// 0 / 6 / 8 / 9: overrides DIP_CONV_BF3 for this process (tests, A/B runs inside one process); -1: back to the environment
extern "C" int dip_conv_bf3_set_terms(int terms) {
    if (terms != -1 && terms != 0 && terms != 6 && terms != 8 && terms != 9) DIP_FAIL("conv_bf3_set_terms: 0, 6, 8, 9 or -1");
    g_bf3_override = terms;
    return 0;
}

extern "C" int dip_pack_weights_bf3(const float* params, void* packed3, const DipPackRec3* recs_dev, int nrec, long long max_elems,
                                    void* stream) {
    if (nrec <= 0) return 0;
    long long gx = (max_elems + 256 * 8 - 1) / (256 * 8);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    dip_launch_pair<DIP_FAM_MISC>(pack_weights_bf3_kernel<false>, pack_weights_bf3_kernel<true>, dim3((unsigned)gx, nrec), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<unsigned short*>(packed3), recs_dev, bf3_tailk() ? 1 : 0);
    DIP_CHECK_LAUNCH();
    return 0;
}
