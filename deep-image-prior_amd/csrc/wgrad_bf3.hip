// Weight gradient of the 3x3 stride-1 convolutions on the bf16 matrix pipe (the arithmetic of conv_bf3.hip: every fp32
// operand split exactly into three bf16 terms, the cross products -- each exact in fp32 -- accumulated in fp32: eight of the
// nine by default, lo x lo (< 2^-32 of a product) left out; NT = 9 sums them all).
//
//   dW[tap][c][o] = sum_p  u[p + tap][c] * dy[p][o]            (u = transform(x), as in the forward)
//
// Two forms of the kernel live in this file: the ping-pong of two wave groups that runs every eligible layer since round 5
// (wgrad_bf3_kernel, described where it is defined) and the round-4 form described next (wgrad_bf3_v1_kernel: the reference
// of the bit-identity test, DIP_WGRAD_BF3_V1=1).  Both share the LDS layouts, the split and the per-accumulator MFMA order.
//
// GEMM per tap: M = 32 input channels per workgroup, N = 128 output channels (32 per wave), K = output pixels.  The
// contraction runs over PIXELS, so v_mfma_f32_32x32x16_bf16 wants 8 consecutive pixels of one channel per lane: both
// operands are staged TRANSPOSED in LDS ([plane][channel][pixel] bf16; the fp32 kernel keeps them pixel-major and feeds
// one pixel per MFMA).  A workgroup walks a strided list of 2 x 16-pixel tiles (two K = 16 steps) and keeps the nine
// 32 x 32 accumulators of its taps per wave (conv_wgrad.hip's scheme), one partial slab per workgroup, summed in a fixed
// order by dip_wgrad_reduce.
//   * staging: a thread loads TWO horizontally adjacent pixels x 4 channels (2 x 16 B), applies the producer's
//     BatchNorm + LeakyReLU (u only), splits, and writes 4 channels x 3 planes x one dword (the pixel pair) --
//     the transposition costs 12 ds_write_b32 per pair; the global loads of tile t + 1 are issued under the MFMAs of tile t;
//   * the three horizontal taps read the SAME 16-byte-aligned window of a halo row (8 pixels + 2): kx = 1 is a 2-byte
//     funnel shift (v_alignbyte_b32), kx = 2 a register rename -- one ds_read_b128 + one ds_read_b32 per (row, plane)
//     serve 3 taps; row pitches (208 B per u channel, 80 B per dy channel) put the 16 lanes of a
//     ds_read_b128 phase on 16 distinct 4-bank groups;
//   * per K step and wave: 81 MFMAs (9 taps x 9 products), 21 LDS reads; 144 accumulator + ~100 other registers.
// The <= 4-channel tail of a 132-channel layer stays on the fp32 kernel's (tap, channel)-packed phase 2
// (dip_conv_wgrad_tail): 2 MFMAs per K step instead of 9 with 28 of 32 rows idle.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef DIP_W3_PROFILE
// per-wave cycle sums of the ping-pong kernel's phases (s_memtime ticks = shader cycles): [workgroup][wave][8]
//   0 MFMA phases, 1 barrier waits after an MFMA phase, 2 commit (transform + split + LDS writes), 3 fetch issue,
//   4 barrier waits after a staging phase, 5 half-periods, 6 whole kernel, 7 whole kernel in s_memrealtime ticks (100 MHz)
__device__ unsigned long long g_w3_prof[1024 * 8 * 8];
__device__ int g_w3_mode;          // knock-outs (results are garbage): 1 = no MFMAs, 2 = no staging after the prologue
#define W3_T() __builtin_amdgcn_s_memtime()
#define W3_ADD(slot, t0, t1) prof[slot] += (t1) - (t0)
#else
#define W3_T() 0ull
#define W3_ADD(slot, t0, t1) ((void)0)
#endif

struct W3Cfg {
    static constexpr int TH = 2, TW = 16;                   // output pixels per tile: two K = 16 steps
    static constexpr int HTH = TH + 2, HTW = TW + 2;        // 4 x 18 halo
    static constexpr int U_ROW = 48;                        // bytes per halo row of a channel (24 pixels: 16-B aligned rows)
    static constexpr int U_CH = HTH * U_ROW + 16;           // 208 B per channel (52 dwords: conflict-free b128 phases)
    static constexpr int U_PLANE = 32 * U_CH;               // 6656
    static constexpr int D_ROW = 32;                        // bytes per tile row of a channel (16 pixels)
    static constexpr int D_CH = TH * D_ROW + 16;            // 80 B per channel (20 dwords)
    static constexpr int D_PLANE = 128 * D_CH;              // 10240
    static constexpr int U_PAIRS = HTH * 9 * 8;             // (row, pixel pair, 4-channel group) = 288 slots per tile
    static constexpr int D_PAIRS = TH * 8 * 32;             // 512
    static constexpr int U_SLOTS = 2, D_SLOTS = 2;          // per thread
    static constexpr int LDS_BYTES = 3 * U_PLANE + 3 * D_PLANE + 2 * 512 * 4;      // + the transform tables
};

__device__ __forceinline__ int w3_map_src(int v, int n_in, int pad_mode) {
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, n_in);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), n_in - 1);
    return (v < 0 || v >= n_in) ? -1 : v;
}

// a == h + m + l exactly (three bf16 numbers, by truncation); returned in the HIGH halves
__device__ __forceinline__ void w3_split(float a, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned uh = __float_as_uint(a) & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = uh; m = um; l = __float_as_uint(r2) & 0xFFFF0000u;
}

// The three horizontal taps of one (tile row, tap row) out of the 16-byte-aligned windows w0 (8 pixels) + w1 (2 more) of a halo
// row, all partial products, smallest first, the three taps' accumulators taking turns MFMA by MFMA.  Per accumulator the
// order of the products is that of the 4-wave form, so the result is bit-identical.  (The rotation was written on the
// hypothesis that a run of v_mfma_f32_32x32x16_bf16 on ONE accumulator issues at half rate; measured in round 5, it does not --
// same-accumulator chains issue back to back, the kernel's time did not move.  It is kept because it costs nothing and keeps
// the operand registers of the three taps live together, one window load per row.)
template <int NT>
__device__ __forceinline__ void w3_taps_of_row(const u32x4 (&w0)[3], const unsigned (&w1)[3], const bf16x8 (&b)[3], f32x16& acc0,
                                               f32x16& acc1, f32x16& acc2) {
    bf16x8 a[3][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        a[0][p] = __builtin_bit_cast(bf16x8, w0[p]);
        a[1][p] = __builtin_bit_cast(bf16x8, u32x4{__builtin_amdgcn_alignbyte(w0[p][1], w0[p][0], 2), __builtin_amdgcn_alignbyte(w0[p][2], w0[p][1], 2),
                                                     __builtin_amdgcn_alignbyte(w0[p][3], w0[p][2], 2), __builtin_amdgcn_alignbyte(w1[p], w0[p][3], 2)});
        a[2][p] = __builtin_bit_cast(bf16x8, u32x4{w0[p][1], w0[p][2], w0[p][3], w1[p]});
    }
#pragma unroll
    for (int sm = 4; sm >= 0; --sm) {               // smallest partial products first
        if ((NT == 6 && sm > 2) || (NT == 8 && sm > 3)) continue;
#pragma unroll
        for (int pa = 0; pa < 3; ++pa) {
            const int pb = sm - pa;
            if (pb < 0 || pb > 2) continue;
            // (sched_barrier 0x7F6: everything but MFMAs may cross, so the MFMAs stay in this order)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][pa], b[pb], acc0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][pa], b[pb], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][pa], b[pb], acc2, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x7F6);
        }
    }
}

template <int NT, int TR>
__global__ __launch_bounds__(256, 2) void wgrad_bf3_v1_kernel(const DipWgradDesc d, const int ntx, const int ntiles, const int CinP,
                                                              const int CoutP) {
    using C = W3Cfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Us = smem;                               // [3][32 c][208 B]
    unsigned char* Ds = smem + 3 * C::U_PLANE;              // [3][128 o][80 B]
    float* tra = reinterpret_cast<float*>(Ds + 3 * C::D_PLANE);
    float* trb = tra + 512;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int walker = blockIdx.x, nwalk = gridDim.x;
    const int c0 = blockIdx.y * 32;
    const int o0 = blockIdx.z * 128;
    const bool wave_active = (o0 + wave * 32) < CoutP;
    const bool do_bias = d.bias_partial != nullptr && blockIdx.y == 0;
    const float slope = d.tr.slope;

    if (TR) {
        for (int c = tid; c < 32; c += 256) {
            const bool ok = c0 + c < d.Cin;
            tra[c] = ok ? d.tr.a[c0 + c] : 1.f;
            trb[c] = ok ? d.tr.b[c0 + c] : 0.f;
        }
    }
    // zero the pad bytes / the pixels 18..23 of the halo rows once (never written by the staging, read by nobody that matters,
    // but NaN bit patterns must not sit there: 0 * NaN)
    for (int i = tid; i < (3 * C::U_PLANE + 3 * C::D_PLANE) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bs[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};          // bias gradient of this thread's dy channels

    // ---- staging slots of this thread ------------------------------------------------------------------------------
    // u: slot -> (4-channel group cg = s / 36, halo row = (s % 36) / 9, pixel pair pp = s % 9)
    // dy: slot -> (4-channel group cg = s / 16 (0..31), tile row = (s % 16) / 8, pixel pair pp = s % 8)
    f32x4 ur[C::U_SLOTS][2], dr[C::D_SLOTS][2];
    auto fetch = [&](int tile) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int s = tid + i * 256;
            const int cg = s / 36, rem = s - cg * 36, hr = rem / 9, pp = rem - hr * 9;
            const int c = c0 + cg * 4;
            const int sr = w3_map_src(ty * C::TH + hr - d.off, d.Hin, d.pad_mode);
            const bool okc = s < C::U_PAIRS && c < d.Cin && sr >= 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int sc = w3_map_src(tx * C::TW + 2 * pp + q - d.off, d.Win, d.pad_mode);
                const bool ok = okc && sc >= 0;
                // unconditional load from a clamped address; commit() zeroes what is padding (same predicate)
                ur[i][q] = *reinterpret_cast<const f32x4*>(d.x + ((size_t)(ok ? sr : 0) * d.Win + (ok ? sc : 0)) * d.Cx + (ok ? c : 0));
            }
        }
#pragma unroll
        for (int i = 0; i < C::D_SLOTS; ++i) {
            const int s = tid + i * 256;
            const int cg = s >> 4, rem = s & 15, r = rem >> 3, pp = rem & 7;
            const int o = o0 + cg * 4;
            const int oy = ty * C::TH + r;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ox = tx * C::TW + 2 * pp + q;
                const bool ok = oy < d.Hout && ox < d.Wout && o < d.Cdy;
                dr[i][q] = *reinterpret_cast<const f32x4*>(d.dy + ((size_t)(ok ? oy : 0) * d.Wout + (ok ? ox : 0)) * d.Cdy + (ok ? o : 0));
            }
        }
    };
    auto commit = [&](int tile) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int s = tid + i * 256;
            if (s < C::U_PAIRS) {
                const int cg = s / 36, rem = s - cg * 36, hr = rem / 9, pp = rem - hr * 9;
                const int c = c0 + cg * 4;
                const int sr = w3_map_src(ty * C::TH + hr - d.off, d.Hin, d.pad_mode);
                unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int sc = w3_map_src(tx * C::TW + 2 * pp + q - d.off, d.Win, d.pad_mode);
                    const bool ok = c < d.Cin && sr >= 0 && sc >= 0;
                    f32x4 v = ur[i][q];
                    if (TR) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + cg * 4), b4 = *reinterpret_cast<const f32x4*>(trb + cg * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float tv = fmaf(a4[e], v[e], b4[e]);
                            v[e] = TR == 1 ? dip_act_leaky(tv, slope) : dip_act(tv, slope);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) w3_split(ok ? v[e] : 0.f, h[q][e], m[q][e], l[q][e]);
                }
                unsigned char* base = Us + (cg * 4) * C::U_CH + hr * C::U_ROW + pp * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH) = (h[0][e] >> 16) | h[1][e];
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH + C::U_PLANE) = (m[0][e] >> 16) | m[1][e];
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH + 2 * C::U_PLANE) = (l[0][e] >> 16) | l[1][e];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::D_SLOTS; ++i) {
            const int s = tid + i * 256;
            const int cg = s >> 4, rem = s & 15, r = rem >> 3, pp = rem & 7;
            const int o = o0 + cg * 4;
            const int oy = ty * C::TH + r;
            unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ox = tx * C::TW + 2 * pp + q;
                const bool ok = oy < d.Hout && ox < d.Wout && o < d.Cdy;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ok ? dr[i][q][e] : 0.f;
                    bs[i][e] += v;
                    w3_split(v, h[q][e], m[q][e], l[q][e]);
                }
            }
            unsigned char* base = Ds + (cg * 4) * C::D_CH + r * C::D_ROW + pp * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<unsigned*>(base + e * C::D_CH) = (h[0][e] >> 16) | h[1][e];
                *reinterpret_cast<unsigned*>(base + e * C::D_CH + C::D_PLANE) = (m[0][e] >> 16) | m[1][e];
                *reinterpret_cast<unsigned*>(base + e * C::D_CH + 2 * C::D_PLANE) = (l[0][e] >> 16) | l[1][e];
            }
        }
    };

    const unsigned char* ua = Us + l31 * C::U_CH + 16 * half;                   // channel l31, pixels 8 * half ..
    const unsigned char* da = Ds + (wave * 32 + l31) * C::D_CH + 16 * half;     // output channel of this lane

    __syncthreads();                       // tables + zeroed LDS
    if (walker < ntiles) fetch(walker);
    for (int tile = walker; tile < ntiles; tile += nwalk) {
        __syncthreads();                   // every wave is done with the previous tile
        commit(tile);
        __syncthreads();
        if (tile + nwalk < ntiles) fetch(tile + nwalk);          // in flight under this tile's MFMAs
        if (wave_active) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {               // tile row = one K = 16 step
                bf16x8 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(da + p * C::D_PLANE + s * C::D_ROW);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    // the 16-byte-aligned window of halo row s + ky: pixels 8 * half .. + 9
                    const int hr = s + ky;
                    u32x4 w0[3];
                    unsigned w1[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        w0[p] = *reinterpret_cast<const u32x4*>(ua + p * C::U_PLANE + hr * C::U_ROW);
                        w1[p] = *reinterpret_cast<const unsigned*>(ua + p * C::U_PLANE + hr * C::U_ROW + 16);
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        bf16x8 a[3];
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            u32x4 v;
                            if (kx == 0) v = w0[p];
                            else if (kx == 2) v = u32x4{w0[p][1], w0[p][2], w0[p][3], w1[p]};
                            else v = u32x4{__builtin_amdgcn_alignbyte(w0[p][1], w0[p][0], 2), __builtin_amdgcn_alignbyte(w0[p][2], w0[p][1], 2),
                                           __builtin_amdgcn_alignbyte(w0[p][3], w0[p][2], 2), __builtin_amdgcn_alignbyte(w1[p], w0[p][3], 2)};
                            a[p] = __builtin_bit_cast(bf16x8, v);
                        }
                        const int t = ky * 3 + kx;
#pragma unroll
                        for (int sm = 4; sm >= 0; --sm) {               // smallest partial products first
                            if ((NT == 6 && sm > 2) || (NT == 8 && sm > 3)) continue;
#pragma unroll
                            for (int pa = 0; pa < 3; ++pa) {
                                const int pb = sm - pa;
                                if (pb < 0 || pb > 2) continue;
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[pa], b[pb], acc[t], 0, 0, 0);
                            }
                        }
                    }
                    // one halo row at a time: hoisting the windows of later rows above these MFMAs spills the accumulators
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- this workgroup's partial slab: rows c0 .. c0 + 31 of slab `walker` ----
    if (wave_active) {
        const int o = o0 + wave * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < CinP) d.partial[(((size_t)walker * 9 + t) * CinP + c) * CoutP + o] = acc[t][r];
            }
    }
    if (do_bias) {
        // thread (cg, pixel-pair lane): sum over the 16 threads that share a channel group, through LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int i = 0; i < C::D_SLOTS; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(tid + i * 256) * 4 + e] = bs[i][e];
        __syncthreads();
        if (tid < 128 && o0 + tid < CoutP) {
            const int cg = tid >> 2, e = tid & 3;
            float sum = 0.f;
            for (int k = 0; k < 16; ++k) sum += red[(cg * 16 + k) * 4 + e];
            d.bias_partial[(size_t)walker * CoutP + o0 + tid] = sum;
        }
    }
}

template <int NT, int TR>
int w3_launch_v1(const DipWgradDesc& d, hipStream_t st) {
    using C = W3Cfg;
    auto kern = wgrad_bf3_v1_kernel<NT, TR>;
    static bool attr_set[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set[dev] = true;
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    const int nfull = ((d.Cin & 31) >= 1 && (d.Cin & 31) <= 4 && d.Cin > 32) ? (d.Cin >> 5) : dip_cdiv(d.Cin, 32);
    dip_launch(kern, dim3(d.nsplit, nfull, dip_cdiv(CoutP, 128)), dim3(256), C::LDS_BYTES, st, d, ntx, ntx * nty, CinP, CoutP);
    DIP_CHECK_LAUNCH();
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the same arithmetic, bit for bit (same walkers, same tiles per walker, same order of MFMAs per accumulator), as a
// PING-PONG of two wave groups.  ONE workgroup of 8 waves owns a CU: group g = wave / 4 owns the 32 input channels c0 + 32 g
// (its own nine accumulators per wave, its own partial-slab rows: what a round-4 workgroup owned), both groups share the
// staged dy tile, and the groups alternate by construction: in half-period h group (h & 1) runs the MFMAs of tile h / 2 out of
// LDS buffer (h / 2) & 1 while the other group stages ITS share of the next tile (its 32 channels of u, half of dy) into the
// other buffer; one barrier per half-period.  A SIMD holds one wave of each group.
// What it was built for and what the counters say (tools/w3_profile.py): the round-4 kernel keeps the matrix
// pipes busy 0.55 of the time and the hypothesis was lock-step of its two co-resident workgroups (both in their MFMA bursts at
// half rate, then both staging with the pipes idle).  The ping-pong removes that by construction -- and times the same in
// isolation: an MFMA phase is 5 300 cycles per tile (4 608 ideal), the staging 3 000 cycles alone but 6 200 beside the other
// group's MFMA wave on the same SIMD, and the clock inside the kernel is 1.66 GHz (1.89 with the staging knocked out, 2.34
// with the MFMAs knocked out): the kernel is bound by what a SIMD can issue for two waves and by the power budget, not by
// idle pipes.  In the iteration the form is worth +1.1 %.
//   * dy is transformed / split once per 64 input channels instead of once per 32;
//   * the A windows are walked by HALO ROW (row hr serves tap row ky = hr of tile row 0 and ky = hr - 1 of tile row 1: 4 row
//     loads per tile instead of 6), the next row's windows are read under the current row's MFMAs;
//   * LDS: 2 x (3 x 64 x 208 + 3 x 128 x 80) B = 138 KB + tables.
template <int NT, int TR>
__global__ __launch_bounds__(512) void wgrad_bf3_kernel(const DipWgradDesc d, const int ntx, const int ntiles, const int CinP,
                                                        const int CoutP, const int c_end) {
    using C = W3Cfg;
    constexpr int U_PLANE2 = 64 * C::U_CH;                       // 13312: 64 channels per workgroup
    constexpr int BUF = 3 * U_PLANE2 + 3 * C::D_PLANE;           // 70656
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tra = reinterpret_cast<float*>(smem + 2 * BUF);
    float* trb = tra + 64;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wq = wave & 3, tg = tid & 255;
    const int walker = blockIdx.x, nwalk = gridDim.x;
    const int c0 = blockIdx.y * 64 + grp * 32;                   // this group's 32 input channels
    const int o0 = blockIdx.z * 128;
    const bool wave_active = (o0 + wq * 32) < CoutP;
    const bool do_bias = d.bias_partial != nullptr && blockIdx.y == 0;
    const float slope = d.tr.slope;

    if (TR) {
        if (tid < 64) {
            const int c = blockIdx.y * 64 + tid;
            const bool ok = c < d.Cin;
            tra[tid] = ok ? d.tr.a[c] : 1.f;
            trb[tid] = ok ? d.tr.b[c] : 0.f;
        }
    }
    for (int i = tid; i < 2 * BUF / 4; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;     // pad bytes / pixels 18..23: no NaN patterns

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};                          // bias gradient of this thread's 4 dy channels

    // ---- staging slots ----------------------------------------------------------------------------------------------
    // A slot = two horizontally adjacent pixels x 4 channels (2 x 16-byte loads).  The CHANNEL GROUP is the fastest lane index:
    // 8 lanes read the 128 contiguous bytes of one pixel's 32 channels, a wave-wide load touches 8-16 cache lines.  (Round 4's
    // map -- channel group slowest, so that the transposing LDS writes were conflict-free -- made every lane of a load hit a
    // line of its own, 64 per instruction: tools/w3_profile.py measured 4000 cycles per tile for ISSUING the six loads, more than
    // the 2400 of the transform + split + LDS writes and, with them, more than the 5200-cycle MFMA phase they should hide
    // under.  The LDS writes now pay a 2-way bank conflict instead.)
    // u (per group, its 32 channels): slot s = tg + 256 i < 288 -> (4-channel group cg = s % 8, halo row (s / 8) / 9, pixel pair (s / 8) % 9)
    // dy (whole workgroup): slot = tid -> (pixel pair tid % 8, cg = 8 (tid / 128) + (tid / 8) % 8, tile row (tid / 64) % 2)
    // Per-thread constants of the slots (the staging is instruction-issue bound -- it shares its SIMD with the MFMA wave of the
    // other group -- so nothing that does not depend on the tile is recomputed per tile): source offsets relative to the tile's
    // first halo pixel / first output pixel, LDS write addresses, channel validity.  INTERIOR tiles (whole halo inside the
    // image, whole tile inside the output: 93 % of the tiles at 512^2) take these as they are; border tiles map every row /
    // column through the padding rule as before.
    int u_off[2], u_lds[2], u_hr[2], u_pp[2];
    bool u_valid[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tg + i * 256;
        const int cg = s & 7, rem = s >> 3, hr = rem / 9, pp = rem - hr * 9;
        u_hr[i] = hr;
        u_pp[i] = pp;
        u_valid[i] = s < C::U_PAIRS && (c0 + cg * 4) < d.Cin;
        u_off[i] = u_valid[i] ? (hr * d.Win + 2 * pp) * d.Cx + c0 + cg * 4 : 0;
        u_lds[i] = (grp * 32 + cg * 4) * C::U_CH + hr * C::U_ROW + pp * 4;
    }
    const int u_c = c0 + (tg & 7) * 4;                              // (both slots of a thread have the same channel group)
    const int d_pp = tid & 7, d_r = (tid >> 6) & 1, d_cg = ((tid >> 7) << 3) | ((tid >> 3) & 7);
    const int d_o = o0 + d_cg * 4;
    const bool d_valid = d_o < d.Cdy;
    const int d_off = d_valid ? (d_r * d.Wout + 2 * d_pp) * d.Cdy + d_o : 0;
    const int d_lds = 3 * U_PLANE2 + (d_cg * 4) * C::D_CH + d_r * C::D_ROW + d_pp * 4;
    auto interior = [&](int ty, int tx) {
        return ty * C::TH - d.off >= 0 && ty * C::TH + C::HTH - 1 - d.off < d.Hin && tx * C::TW - d.off >= 0 &&
               tx * C::TW + C::HTW - 1 - d.off < d.Win && ty * C::TH + C::TH - 1 < d.Hout && tx * C::TW + C::TW - 1 < d.Wout;
    };

    f32x4 ur[2][2], dr[2];
    auto fetch = [&](int tile) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
        const float* pu[2][2];
        const float* pd[2];
        if (interior(ty, tx)) {
            const float* bx = d.x + ((size_t)(ty * C::TH - d.off) * d.Win + (tx * C::TW - d.off)) * d.Cx;
            const float* by = d.dy + ((size_t)(ty * C::TH) * d.Wout + tx * C::TW) * d.Cdy;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                pu[i][0] = bx + u_off[i];
                pu[i][1] = pu[i][0] + d.Cx;
            }
            pd[0] = by + d_off;
            pd[1] = pd[0] + d.Cdy;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sr = w3_map_src(ty * C::TH + u_hr[i] - d.off, d.Hin, d.pad_mode);
                const bool okc = u_valid[i] && sr >= 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int sc = w3_map_src(tx * C::TW + 2 * u_pp[i] + q - d.off, d.Win, d.pad_mode);
                    const bool ok = okc && sc >= 0;
                    // unconditional load from a clamped address; commit() zeroes what is padding (same predicate)
                    pu[i][q] = d.x + ((size_t)(ok ? sr : 0) * d.Win + (ok ? sc : 0)) * d.Cx + (ok ? u_c : 0);
                }
            }
            const int oy = ty * C::TH + d_r;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int ox = tx * C::TW + 2 * d_pp + q;
                const bool ok = oy < d.Hout && ox < d.Wout && d_valid;
                pd[q] = d.dy + ((size_t)(ok ? oy : 0) * d.Wout + (ok ? ox : 0)) * d.Cdy + (ok ? d_o : 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 2; ++q) ur[i][q] = *reinterpret_cast<const f32x4*>(pu[i][q]);
#pragma unroll
        for (int q = 0; q < 2; ++q) dr[q] = *reinterpret_cast<const f32x4*>(pd[q]);
    };
    auto commit = [&](int tile, int buf) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
        unsigned char* Bs = smem + buf * BUF;
        bool oku[2][2], okd[2];
        if (interior(ty, tx)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) oku[i][0] = oku[i][1] = u_valid[i];
            okd[0] = okd[1] = d_valid;
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int sr = w3_map_src(ty * C::TH + u_hr[i] - d.off, d.Hin, d.pad_mode);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int sc = w3_map_src(tx * C::TW + 2 * u_pp[i] + q - d.off, d.Win, d.pad_mode);
                    oku[i][q] = u_valid[i] && sr >= 0 && sc >= 0;
                }
            }
            const int oy = ty * C::TH + d_r;
#pragma unroll
            for (int q = 0; q < 2; ++q) okd[q] = oy < d.Hout && (tx * C::TW + 2 * d_pp + q) < d.Wout && d_valid;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (tg + i * 256 < C::U_PAIRS) {
                unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x4 v = ur[i][q];
                    if (TR) {
                        const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + grp * 32 + (tg & 7) * 4);
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(trb + grp * 32 + (tg & 7) * 4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float tv = fmaf(a4[e], v[e], b4[e]);
                            v[e] = TR == 1 ? dip_act_leaky(tv, slope) : dip_act(tv, slope);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) w3_split(oku[i][q] ? v[e] : 0.f, h[q][e], m[q][e], l[q][e]);
                }
                unsigned char* base = Bs + u_lds[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH) = (h[0][e] >> 16) | h[1][e];
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH + U_PLANE2) = (m[0][e] >> 16) | m[1][e];
                    *reinterpret_cast<unsigned*>(base + e * C::U_CH + 2 * U_PLANE2) = (l[0][e] >> 16) | l[1][e];
                }
            }
        }
        {
            unsigned h[2][4], m[2][4], l[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = okd[q] ? dr[q][e] : 0.f;
                    bs[e] += v;
                    w3_split(v, h[q][e], m[q][e], l[q][e]);
                }
            }
            unsigned char* base = Bs + d_lds;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *reinterpret_cast<unsigned*>(base + e * C::D_CH) = (h[0][e] >> 16) | h[1][e];
                *reinterpret_cast<unsigned*>(base + e * C::D_CH + C::D_PLANE) = (m[0][e] >> 16) | m[1][e];
                *reinterpret_cast<unsigned*>(base + e * C::D_CH + 2 * C::D_PLANE) = (l[0][e] >> 16) | l[1][e];
            }
        }
    };

    // ---- the MFMAs of one tile out of buffer `buf`, by halo row ----
    const int ua_off = (grp * 32 + l31) * C::U_CH + 16 * half;                   // channel l31 of the group, pixels 8 * half ..
    const int da_off = 3 * U_PLANE2 + (wq * 32 + l31) * C::D_CH + 16 * half;     // output channel of this lane
    auto mfma_tile = [&](int buf) {
        const unsigned char* ua = smem + buf * BUF + ua_off;
        const unsigned char* da = smem + buf * BUF + da_off;
        bf16x8 b[2][3];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[s][p] = *reinterpret_cast<const bf16x8*>(da + p * C::D_PLANE + s * C::D_ROW);
        u32x4 w0[2][3];
        unsigned w1[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            w0[0][p] = *reinterpret_cast<const u32x4*>(ua + p * U_PLANE2);
            w1[0][p] = *reinterpret_cast<const unsigned*>(ua + p * U_PLANE2 + 16);
        }
#pragma unroll
        for (int hr = 0; hr < 4; ++hr) {
            const int cur = hr & 1, nxt = cur ^ 1;
            if (hr < 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    w0[nxt][p] = *reinterpret_cast<const u32x4*>(ua + p * U_PLANE2 + (hr + 1) * C::U_ROW);
                    w1[nxt][p] = *reinterpret_cast<const unsigned*>(ua + p * U_PLANE2 + (hr + 1) * C::U_ROW + 16);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {               // tile row s reads halo row hr as its tap row ky = hr - s
                const int ky = hr - s;
                if (ky < 0 || ky > 2) continue;
                w3_taps_of_row<NT>(w0[cur], w1[cur], b[s], acc[ky * 3], acc[ky * 3 + 1], acc[ky * 3 + 2]);
            }
            __builtin_amdgcn_sched_barrier(0);          // one halo row at a time (hoisting more windows spills the accumulators)
        }
    };

    // ---- prologue: both groups stage their share of the first tile; the second tile's loads go into flight ----
#ifdef DIP_W3_PROFILE
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const unsigned long long tk0 = W3_T();
#ifdef DIP_W3_PROFILE
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    const int w3_mode = __builtin_amdgcn_readfirstlane(g_w3_mode);
#else
    constexpr int w3_mode = 0;
#endif
    __syncthreads();                       // tables + zeroed LDS
    const int ntl = walker < ntiles ? (ntiles - 1 - walker) / nwalk + 1 : 0;        // tiles of this walker
    if (ntl > 0) {
        fetch(walker);
        commit(walker, 0);
        if (ntl > 1) fetch(walker + nwalk);
    }
    __syncthreads();
    // half-period h: group (h & 1) multiplies tile h / 2, the other group stages its share of tile h / 2 + 1
    for (int h = 0; h < 2 * ntl; ++h) {
        const int kt = h >> 1, buf = kt & 1;
        const unsigned long long t0 = W3_T();
        if ((h & 1) == grp) {
            if (wave_active && w3_mode != 1) {
                __builtin_amdgcn_s_setprio(2);
                mfma_tile(buf);
                __builtin_amdgcn_s_setprio(0);
            }
            const unsigned long long t1 = W3_T();
            __syncthreads();
            W3_ADD(0, t0, t1);
            W3_ADD(1, t1, W3_T());
        } else {
            unsigned long long t1 = t0, t2 = t0;
            if (kt + 1 < ntl && w3_mode != 2) {
                commit(walker + (kt + 1) * nwalk, buf ^ 1);
                t1 = W3_T();
                if (kt + 2 < ntl) fetch(walker + (kt + 2) * nwalk);          // in flight under this group's next MFMA phase
                t2 = W3_T();
            }
            __syncthreads();
            W3_ADD(2, t0, t1);
            W3_ADD(3, t1, t2);
            W3_ADD(4, t2, W3_T());
        }
#ifdef DIP_W3_PROFILE
        prof[5] += 1;
#endif
    }
#ifdef DIP_W3_PROFILE
    prof[6] = W3_T() - tk0;
    prof[7] = __builtin_amdgcn_s_memrealtime() - tr0;
    if (lane == 0) {
        const int wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (wg < 1024)
            for (int i = 0; i < 8; ++i) g_w3_prof[(wg * 8 + wave) * 8 + i] = prof[i];
    }
#endif

    // ---- this group's rows c0 .. c0 + 31 of partial slab `walker` ----
    // c_end = channels of the full 32-channel chunks: with an odd number of them in front of a <= 4-channel tail (Cin = 100,
    // 164) group 1 of the last workgroup has no chunk; it still stages and multiplies (the barriers are workgroup-wide) but
    // must not write rows that belong to the tail launch (they were right only because dip_conv_wgrad_tail,
    // issued later on the same stream, overwrote them)
    if (wave_active && c0 < c_end) {
        const int o = o0 + wq * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < CinP) d.partial[(((size_t)walker * 9 + t) * CinP + c) * CoutP + o] = acc[t][r];
            }
    }
    if (do_bias) {
        // thread (cg, tile row, pixel pair): sum over the 16 threads that share a channel group, through LDS
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int e = 0; e < 4; ++e) red[tid * 4 + e] = bs[e];
        __syncthreads();
        if (tid < 128 && o0 + tid < CoutP) {
            const int cg = tid >> 2, e = tid & 3;
            float sum = 0.f;
            // (the 4-wave form's order: tile row 0, pixel pairs 0..7, then tile row 1 -- the bias gradient stays bit-identical)
            for (int k = 0; k < 16; ++k) sum += red[((k & 7) + 8 * (cg & 7) + 64 * (k >> 3) + 128 * (cg >> 3)) * 4 + e];
            d.bias_partial[(size_t)walker * CoutP + o0 + tid] = sum;
        }
    }
}

template <int NT, int TR>
int w3_launch(const DipWgradDesc& d, hipStream_t st) {
    using C = W3Cfg;
    constexpr int LDS2 = 2 * (3 * 64 * C::U_CH + 3 * C::D_PLANE) + 2 * 64 * 4;
    auto kern = wgrad_bf3_kernel<NT, TR>;
    static bool attr_set[16] = {};
    int dev = 0;
    hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set[dev] = true;
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    const int nfull = ((d.Cin & 31) >= 1 && (d.Cin & 31) <= 4 && d.Cin > 32) ? (d.Cin >> 5) : dip_cdiv(d.Cin, 32);
    const bool has_tail = (d.Cin & 31) >= 1 && (d.Cin & 31) <= 4 && d.Cin > 32;
    dip_launch(kern, dim3(d.nsplit, dip_cdiv(nfull, 2), dip_cdiv(CoutP, 128)), dim3(512), LDS2, st, d, ntx, ntx * nty, CinP, CoutP,
               has_tail ? nfull * 32 : CinP);
    DIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

#ifdef DIP_W3_PROFILE
extern "C" int dip_w3_prof_mode(int mode) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_w3_mode), &mode, sizeof(int)); }
extern "C" int dip_w3_prof_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w3_prof), (size_t)n * sizeof(unsigned long long));
}
#endif

extern "C" int dip_conv_bf3_terms(void);
extern "C" int dip_conv_wgrad_tail(const DipWgradDesc* dp, void* stream);

// 1 when dip_conv_wgrad runs `d` on the bf16 matrix pipe: 3x3, stride 1, >= 32 input channels, one tap group, one slab per
// walker, and >= 512 tiles of 2 x 16 pixels (128 x 128; round 5: the 128^2 layers' weight gradients 59-78 us on the fp32
// kernel -> here, +1.3 % per iteration, profiles/r05_ab_n64.txt)
extern "C" int dip_wgrad_bf3_eligible(const DipWgradDesc* dp) {
    const DipWgradDesc& d = *dp;
    static const bool off = getenv("DIP_WGRAD_NO_BF3") != nullptr;
    if (off || dip_conv_bf3_terms() == 0) return 0;
    if (d.ks != 3 || d.stride != 1 || d.Cin < 32 || d.Cout < 97 || d.tap_groups > 1) return 0;
    if ((d.Cx & 3) || (d.Cdy & 3) || (d.tr.a != nullptr && d.Cin > 512)) return 0;
    const int tail = d.Cin & 31;
    if (tail >= 1 && tail <= 4 && (d.Cin >> 5) > 8) return 0;            // (dip_conv_wgrad_tail shares the tail among <= 8 chunks)
    const int ntiles = dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 2);
    if (d.nsplit < 1 || d.nsplit > dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 4)) return 0;
    return ntiles >= 512 ? 1 : 0;
}

extern "C" int dip_wgrad_bf3(const DipWgradDesc* dp, void* stream) {
    const DipWgradDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int nt = dip_conv_bf3_terms();
    if (nt == 0) DIP_FAIL("wgrad_bf3: the bf16-pipe arithmetic is switched off (DIP_CONV_BF3=0)");
    const int tr = d.tr.a == nullptr ? 0 : (d.tr.slope > 0.f ? 1 : 2);
    int rc;
    // DIP_WGRAD_BF3_V1=1: the round-4 form of the kernel (4 waves, two workgroups per CU, one accumulator at a time) for every
    // layer: the REFERENCE of the bit-identity test (tests/test_bf3_gpu.py) and of A/B runs.  =2: only below 2048 tiles.
    static const int v1_mode = [] { const char* e = getenv("DIP_WGRAD_BF3_V1"); return e ? atoi(e) : 0; }();
    const bool v1 = v1_mode == 1 || (v1_mode == 2 && dip_cdiv(d.Wout, 16) * dip_cdiv(d.Hout, 2) < 2048);
    if (v1) {
        if (nt == 6) rc = tr == 0 ? w3_launch_v1<6, 0>(d, st) : (tr == 1 ? w3_launch_v1<6, 1>(d, st) : w3_launch_v1<6, 2>(d, st));
        else if (nt == 8) rc = tr == 0 ? w3_launch_v1<8, 0>(d, st) : (tr == 1 ? w3_launch_v1<8, 1>(d, st) : w3_launch_v1<8, 2>(d, st));
        else rc = tr == 0 ? w3_launch_v1<9, 0>(d, st) : (tr == 1 ? w3_launch_v1<9, 1>(d, st) : w3_launch_v1<9, 2>(d, st));
    } else
    if (nt == 6) rc = tr == 0 ? w3_launch<6, 0>(d, st) : (tr == 1 ? w3_launch<6, 1>(d, st) : w3_launch<6, 2>(d, st));
    else if (nt == 8) rc = tr == 0 ? w3_launch<8, 0>(d, st) : (tr == 1 ? w3_launch<8, 1>(d, st) : w3_launch<8, 2>(d, st));
    else rc = tr == 0 ? w3_launch<9, 0>(d, st) : (tr == 1 ? w3_launch<9, 1>(d, st) : w3_launch<9, 2>(d, st));
    if (rc) return rc;
    // the <= 4-channel tail of a 132-channel layer: the fp32 kernel's (tap, channel)-packed phase 2 on its own
    if ((d.Cin & 31) >= 1 && (d.Cin & 31) <= 4 && d.Cin > 32) return dip_conv_wgrad_tail(dp, stream);
    return 0;
}
