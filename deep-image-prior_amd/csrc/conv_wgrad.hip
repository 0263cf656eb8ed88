// Convolution weight gradient on the gfx950 fp32 matrix core.
//
//   dW[tap][c][o] = sum_q  u[src(q,tap)][c] * dy[q][o]          (u = transform(x), as in the forward)
//
// GEMM view per tap: M = input channels (32 per workgroup), N = output channels (128 per
// workgroup, 32 per wave), K = output pixels.  A workgroup owns a [NT taps][32 c][128 o]
// accumulator block (NT 32x32 MFMA accumulators per wave) and walks a strided list of 4x16-pixel
// tiles ("split-K" over pixels): per tile the input halo (32 channels, producer BN+LeakyReLU
// applied on load) and the dy tile are staged in LDS once and serve all taps.  Both MFMA operands
// are pixel-major in LDS ([pixel][channel]), so each ds_read_b32 has the 32 lanes of a half-wave
// on consecutive banks: conflict-free without padding.
// Each workgroup finally writes ONE partial slab; dip_wgrad_reduce sums the slabs in a fixed
// order (deterministic, no float atomics) straight into the OIHW gradient arena.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

// marks halo entries outside the image while they wait in registers: they must become 0, not
// act(b), when the producer transform is applied on the way to LDS
constexpr float PADV = -3.0e38f;

template <int KS, int S, int NT, int CB>
struct WCfg {
    static constexpr int TH = 4, TW = 16, NPX = TH * TW;   // 64 output pixels per step
    static constexpr int HTH = (TH - 1) * S + KS, HTW = (TW - 1) * S + KS;
    static constexpr int NPIX = HTH * HTW;
    static constexpr int CW = 32 * CB;                      // input channels per workgroup
    static constexpr int U_FLOATS = NPIX * CW;
    static constexpr int DY_FLOATS = NPX * 128;
    static constexpr int U_SLOTS = (NPIX * (CW / 4) + 255) / 256;
    static constexpr int LDS_BYTES = (U_FLOATS + DY_FLOATS) * 4;
    static constexpr int NGROUPS = (KS * KS + NT - 1) / NT;
};

__device__ __forceinline__ int wmap_src(int v, int n_in, int pad_mode) {
    if (pad_mode == DIP_PAD_REFLECT) v = dip_reflect(v, n_in);
    else if (pad_mode == DIP_PAD_REPLICATE) v = min(max(v, 0), n_in - 1);
    return (v < 0 || v >= n_in) ? -1 : v;
}

template <int KS, int S, int NT, int CB, bool SLIDE = false, bool GRP = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const DipWgradDesc d_, const int ntx, const int ntiles,
                                                            const int CinP, const int CoutP, const int ragged_parts,
                                                            const int kw, const int phase2_only, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipWgradDesc, d);
    using C = WCfg<KS, S, NT, CB>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Us = smem;
    float* Ds = smem + C::U_FLOATS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int half = lane >> 5;

    // kw = 1: the four waves own four 32-column blocks.  Narrow layers (<= 64 / <= 32 output columns) would leave
    // two / three waves idle: there kw = 2 / 4 waves share a column block and split the K steps (pixel pairs) of
    // every tile between them, each writing a slab of its own (slab index blockIdx.x * kw + wk).
    const int wcol = wave % (4 / kw), wk = wave / (4 / kw);
    const int walker = blockIdx.x, nwalk = gridDim.x;      // pixel-tile walkers
    const int split = walker * kw + wk;                    // slab
    // 32*CB input channels.  Workgroups are dispatched in blockIdx order; the ragged tail chunk of a
    // 132-channel layer is light (2 of 9 MFMAs per K step), so it goes FIRST: its workgroups retire early
    // and the slots go to the full chunks, instead of forming a lonely last round behind them.
    // (With ragged_parts > 0 there are no tail workgroups at all: see phase 2 below.)
    const int cchunk = ragged_parts > 0 ? (int)blockIdx.y : (int)((blockIdx.y + gridDim.y - 1) % gridDim.y);
    const int group = dip_grp_z<GRP>(grp) % C::NGROUPS;     // tap group
    const int nblk = dip_grp_z<GRP>(grp) / C::NGROUPS;      // 128 output channels
    const int tap0 = group * NT;
    int c0 = cchunk * C::CW;
    const int o0 = nblk * 128;
    const bool wave_active = (o0 + wcol * 32) < CoutP;
    const bool do_bias = (d.bias_partial != nullptr) && cchunk == 0 && group == 0;

    f32x16 acc[NT * CB];
#pragma unroll
    for (int t = 0; t < NT * CB; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    const bool has_tr = d.tr.a != nullptr;
    const float slope = d.tr.slope;
    const int c4 = tid & (C::CW / 4 - 1);          // this thread's 4-channel group in the staging
    bool cvalid = (c0 + c4 * 4) < d.Cin;
    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_tr && cvalid) {
        ta = *reinterpret_cast<const f32x4*>(d.tr.a + c0 + c4 * 4);
        tb = *reinterpret_cast<const f32x4*>(d.tr.b + c0 + c4 * 4);
    }

    // Per-lane A-operand offsets (floats, relative to the K step's pixel) of the active accumulators.
    // Normal chunk: accumulator t = tap tap0+t, row = input channel l31.
    // Ragged 4-channel tail chunk of a 3x3 conv (the skip branch of a 132-channel concat): rows are
    // (tap, channel) pairs -- taps 0..7 in accumulator 0, tap 8 in rows 0..3 of accumulator 1 -- so the
    // chunk costs 2 MFMAs per K step instead of 9 with 28 of 32 rows idle.
    constexpr bool CAN_PACK = (KS == 3) && (NT == 9) && (CB == 1);
    bool pack = CAN_PACK && (d.Cin - c0 <= 4);
    // accumulators 0 and 1 read through per-lane offsets so that the packed chunk shares the loop
    int aoff0 = l31, aoff1 = C::CW + l31;                        // taps 0 and 1, row = channel l31
    if (pack) {
        const int ptap = l31 >> 2, pch = l31 & 3;
        aoff0 = ((ptap / 3) * C::HTW + (ptap % 3)) * C::CW + pch;
        aoff1 = (2 * C::HTW + 2) * C::CW + pch;
    }

    // Software pipeline over the pixel tiles: the global loads of tile t+1 are issued before the
    // MFMAs of tile t and parked in registers (12 x 16 B per thread); they are written to LDS after
    // the barrier that ends tile t's reads.  Without it a workgroup alternated between a load phase
    // and an MFMA phase and only the co-resident workgroup could fill the gaps.
    f32x4 ureg[C::U_SLOTS], dreg[8];
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < C::NPIX * (C::CW / 4)) {
                const int hp = f / (C::CW / 4);
                const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
                const int sr = wmap_src(ty * C::TH * S + hr - d.off, d.Hin, d.pad_mode);
                const int sc = wmap_src(tx * C::TW * S + hc - d.off, d.Win, d.pad_mode);
                if (sr >= 0 && sc >= 0 && cvalid)
                    v = *reinterpret_cast<const f32x4*>(d.x + ((size_t)sr * d.Win + sc) * d.Cx + c0 + c4 * 4);
                else
                    v = f32x4{PADV, PADV, PADV, PADV};
            }
            ureg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + i * 256;           // float4 index: pixel = f >> 5, o4 = f & 31
            const int px = f >> 5, o4 = f & 31;
            const int oy = ty * C::TH + (px >> 4), ox = tx * C::TW + (px & 15);
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int o = o0 + o4 * 4;
            if (oy < d.Hout && ox < d.Wout && o < d.Cdy)
                v = *reinterpret_cast<const f32x4*>(d.dy + ((size_t)oy * d.Wout + ox) * d.Cdy + o);
            dreg[i] = v;
        }
    };
    // Compact halo of the packed tail phase of the SLIDE kernel: [pixel][4 channels] with a row pitch of PKP = 19
    // pixels.  In the 32-channel layout the packed A operand (lane = (tap, channel)) reads addresses
    // (ky*HTW + kx)*32 + ch: all 32 lanes of a half-wave in banks 0..3 (8-way conflict); with 4-float pixels and pitch
    // 19 the eight taps land on banks 0, 4, ..., 28.  (Measured: the phase is bound by its per-tile staging latency, not
    // by LDS -- 62 of the 780 us of the 132 -> 128 layer at 512^2 with either layout; another 70 us of that layer's
    // excess over 128 -> 128 is the 528-byte pixel pitch of the 132-channel tensor, whose 128-byte channel slices
    // straddle cache lines.)
    constexpr int PKP = C::HTW + 1;
    auto commit = [&](auto compactc) __attribute__((always_inline)) {     // registers -> LDS, producer BatchNorm+LeakyReLU on the way
        constexpr bool COMPACT = decltype(compactc)::value;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < C::NPIX * (C::CW / 4) && (!COMPACT || c4 == 0)) {
                f32x4 v = ureg[i];
                if (has_tr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] == PADV) ? 0.f : dip_act(fmaf(ta[e], v[e], tb[e]), slope);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (v[e] == PADV) ? 0.f : v[e];
                }
                if constexpr (COMPACT) {
                    const int hp = f / (C::CW / 4);
                    const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
                    *reinterpret_cast<f32x4*>(Us + (hr * PKP + hc) * 4) = v;
                } else {
                    *reinterpret_cast<f32x4*>(Us + f * 4) = v;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(Ds + (tid + i * 256) * 4) = dreg[i];
    };

    // (the stride-2 and 5x5 halos are too large to park next to the accumulators without spilling:
    // those variants stage global -> LDS directly, slot by slot)
    constexpr bool PF = (KS <= 3) && (S == 1);
    auto stage_direct = [&](int tile) __attribute__((always_inline)) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < C::NPIX * (C::CW / 4)) {
                const int hp = f / (C::CW / 4);
                const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
                const int sr = wmap_src(ty * C::TH * S + hr - d.off, d.Hin, d.pad_mode);
                const int sc = wmap_src(tx * C::TW * S + hc - d.off, d.Win, d.pad_mode);
                f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (sr >= 0 && sc >= 0 && cvalid) {
                    v = *reinterpret_cast<const f32x4*>(d.x + ((size_t)sr * d.Win + sc) * d.Cx + c0 + c4 * 4);
                    if (has_tr) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dip_act(fmaf(ta[e], v[e], tb[e]), slope);
                    }
                }
                *reinterpret_cast<f32x4*>(Us + f * 4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + i * 256;
            const int px = f >> 5, o4 = f & 31;
            const int oy = ty * C::TH + (px >> 4), ox = tx * C::TW + (px & 15);
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            const int o = o0 + o4 * 4;
            if (oy < d.Hout && ox < d.Wout && o < d.Cdy)
                v = *reinterpret_cast<const f32x4*>(d.dy + ((size_t)oy * d.Wout + ox) * d.Cdy + o);
            *reinterpret_cast<f32x4*>(Ds + f * 4) = v;
        }
    };
    // the pixel tiles first, first + step, ...; packc = integral_constant<bool>: the SLIDE kernel's compile-time
    // choice between the (tap, channel)-packed loop of a ragged tail (phase 2) and the sliding-window loop (phase 1)
    auto walk = [&](auto packc, const int first, const int step) __attribute__((always_inline)) {
    constexpr bool PACKED = decltype(packc)::value;
    if (PF && first < ntiles) fetch(first);
    for (int tile = first; tile < ntiles; tile += step) {
        __syncthreads();                   // every wave is done with the previous tile
        if constexpr (PF) commit(std::integral_constant<bool, SLIDE && PACKED>{}); else stage_direct(tile);
        __syncthreads();
        if (PF && tile + step < ntiles) fetch(tile + step);
        if (wave_active) {
            if constexpr (SLIDE) {
                // Sliding A-operand window (3x3, stride 1).  A K step pairs the pixels (row r, col j) [lanes 0..31] and
                // (row r, col j + 8) [lanes 32..63]; tap kx of step j reads halo column j + kx (+ 8), which is the
                // value tap kx + 1 read one step earlier: the window a[ky][.] is kept in registers, so a step costs
                // NKY new ds_read_b32 (+ 1 for dy) for its 3*NKY MFMAs instead of one read per MFMA.  The column of
                // step j + 1 is read one step ahead (ring of 4), the MFMA that needs the newest value goes last.
                constexpr int NKY = NT / 3;
                const int ky0 = tap0 / 3;
                if constexpr (PACKED) {
                    // (tap, channel)-packed rows on the compact halo (see commit): lane l31 = tap * 4 + channel
                    const int ptap = l31 >> 2, pch = l31 & 3;
                    const int po0 = ((ptap / 3) * PKP + (ptap % 3)) * 4 + pch, po1 = (2 * PKP + 2) * 4 + pch;
#pragma unroll 2
                    for (int s = wk; s < C::NPX / 2; s += kw) {
                        const int px = 2 * s + half;
                        const int r = px >> 4, c = px & 15;
                        const float b = Ds[px * 128 + wcol * 32 + l31];
                        const float* ub = Us + (r * PKP + c) * 4;
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[po0], b, acc[0], 0, 0, 0);
                        acc[NT > 1 ? 1 : 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[po1], b, acc[NT > 1 ? 1 : 0], 0, 0, 0);
                    }
                } else {
#pragma unroll 1
                    for (int r = wk; r < C::TH; r += kw) {
                        const float* urow = Us + ((r + ky0) * C::HTW + 8 * half) * C::CW + l31;
                        const float* drow = Ds + (r * C::TW + 8 * half) * 128 + wcol * 32 + l31;
                        float a[NKY][4];
#pragma unroll
                        for (int ky = 0; ky < NKY; ++ky)
#pragma unroll
                            for (int q = 0; q < 3; ++q) a[ky][q] = urow[(ky * C::HTW + q) * C::CW];
                        float b = drow[0];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            float bn = 0.f;
                            if (j < 7) {            // operands of step j + 1, in flight under this step's MFMAs
                                bn = drow[(j + 1) * 128];
#pragma unroll
                                for (int ky = 0; ky < NKY; ++ky) a[ky][(j + 3) & 3] = urow[(ky * C::HTW + j + 3) * C::CW];
                            }
                            bsum += b;
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                                for (int ky = 0; ky < NKY; ++ky)
                                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ky][(j + kx) & 3], b,
                                                                                            acc[ky * 3 + kx], 0, 0, 0);
                            b = bn;
                            __builtin_amdgcn_sched_barrier(0);      // keep the loads of later steps out of this one
                        }
                    }
                }
            } else if constexpr (CAN_PACK) {
#pragma unroll 2
                for (int s = wk; s < C::NPX / 2; s += kw) {
                    const int px = 2 * s + half;
                    const int r = px >> 4, c = px & 15;
                    const float b = Ds[px * 128 + wcol * 32 + l31];
                    bsum += b;
                    const float* ub = Us + ((r * S) * C::HTW + c * S) * C::CW;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[aoff0], b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[aoff1], b, acc[1], 0, 0, 0);
                    if (!pack) {
#pragma unroll
                        for (int t = 2; t < 9; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ub[((t / 3) * C::HTW + (t % 3)) * C::CW + l31], b,
                                                                          acc[t], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll 2
                for (int s = wk; s < C::NPX / 2; s += kw) {
                    const int px = 2 * s + half;
                    const int r = px >> 4, c = px & 15;
                    const float b = Ds[px * 128 + wcol * 32 + l31];
                    bsum += b;
                    const float* ub = Us + ((r * S) * C::HTW + c * S) * C::CW + l31;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int tap = tap0 + t;
                        if (tap < KS * KS) {
                            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
                            for (int cb = 0; cb < CB; ++cb) {
                                const float a = ub[(ky * C::HTW + kx) * C::CW + cb * 32];
                                acc[t * CB + cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t * CB + cb], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
    }
    };
    // phase2_only (dip_conv_wgrad_tail): the full 32-channel chunks were done by another kernel (wgrad_bf3.hip); this
    // launch only adds the shared-out tail below
    if (!phase2_only) walk(std::false_type{}, walker, nwalk);

    // ---- write this workgroup's partial slab ----
    if (phase2_only) {
    } else if (pack) {
      if (wave_active) {
        const int o = o0 + wcol * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;       // row = tap * 4 + channel
            const int c = c0 + (m & 3);
            if (c < CinP) {
                d.partial[(((size_t)split * 9 + (m >> 2)) * CinP + c) * CoutP + o] = acc[0][r];
                if (m < 4) d.partial[(((size_t)split * 9 + 8) * CinP + c) * CoutP + o] = acc[(NT * CB > 1) ? 1 : 0][r];
            }
        }
        if (c0 > 0) {          // the slab reduction sums the 8 four-row parts of a tail chunk: parts 1..7 are zero here
#pragma unroll 1
            for (int t = 0; t < 9; ++t)
                for (int rr = 4 + half; rr < 32; rr += 2)
                    if (c0 + rr < CinP) d.partial[(((size_t)split * 9 + t) * CinP + c0 + rr) * CoutP + o] = 0.f;
        }
        if (do_bias) {
            const float tot = bsum + __shfl_xor(bsum, 32);
            if (half == 0) d.bias_partial[(size_t)split * CoutP + o] = tot;
        }
      }
    } else if (wave_active) {
        const int o = o0 + wcol * 32 + l31;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tap = tap0 + t;
            if (tap < KS * KS) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = c0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (c < CinP)
                            d.partial[(((size_t)split * (KS * KS) + tap) * CinP + c) * CoutP + o] = acc[t * CB + cb][r];
                    }
            }
        }
        if (do_bias) {
            const float tot = bsum + __shfl_xor(bsum, 32);
            if (half == 0) d.bias_partial[(size_t)split * CoutP + o] = tot;
        }
    }

    // ---- phase 2: the ragged <= 4-channel tail of a 132-channel layer, shared out ------------------
    // A tail chunk of its own was 128 extra light workgroups behind 512 heavy ones = a lonely second
    // round (+25 % time for 3 % of the FLOPs).  Instead every full-chunk workgroup also does its share
    // (every ragged_parts-th tile of its split) of the tail in the (tap, channel)-packed form -- 2 MFMAs per
    // K step -- and writes it as part `cchunk` (rows CinMain + 4*cchunk ..+3) of the tail's 32 slab rows;
    // the slab reduction adds the parts.
    if constexpr (CAN_PACK) {
        if (ragged_parts > 0) {
            __syncthreads();
            const int CinMain = d.Cin & ~31;
            c0 = CinMain;
            cvalid = (c0 + c4 * 4) < d.Cin;
            ta = f32x4{1.f, 1.f, 1.f, 1.f};
            tb = f32x4{0.f, 0.f, 0.f, 0.f};
            if (has_tr && cvalid) {
                ta = *reinterpret_cast<const f32x4*>(d.tr.a + c0 + c4 * 4);
                tb = *reinterpret_cast<const f32x4*>(d.tr.b + c0 + c4 * 4);
            }
            pack = true;
            {
                const int ptap = l31 >> 2, pch = l31 & 3;
                aoff0 = ((ptap / 3) * C::HTW + (ptap % 3)) * C::CW + pch;
                aoff1 = (2 * C::HTW + 2) * C::CW + pch;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
            walk(std::true_type{}, walker + cchunk * nwalk, nwalk * ragged_parts);       // (kw == 1 here)
            if (wave_active) {
                const int o = o0 + wcol * 32 + l31;
                const int cbase = CinMain + 4 * cchunk;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;       // row = tap * 4 + channel
                    const int c = cbase + (m & 3);
                    d.partial[(((size_t)split * 9 + (m >> 2)) * CinP + c) * CoutP + o] = acc[0][r];
                    if (m < 4) d.partial[(((size_t)split * 9 + 8) * CinP + c) * CoutP + o] = acc[1][r];
                }
                if (cchunk == 0) {         // parts ragged_parts..7 do not exist: zeros
#pragma unroll 1
                    for (int t = 0; t < 9; ++t)
                        for (int rr = 4 * ragged_parts + half; rr < 32; rr += 2)
                            d.partial[(((size_t)split * 9 + t) * CinP + CinMain + rr) * CoutP + o] = 0.f;
                }
            }
        }
    }
}

// waves that share a 32-column block (and split the K steps) in layers with <= 64 / <= 32 output columns
int wgrad_kw(int CoutP) {
    static const bool off = getenv("DIP_WGRAD_NO_KW") != nullptr;
    if (off) return 1;
    return CoutP <= 32 ? 4 : (CoutP <= 64 ? 2 : 1);
}

template <int KS, int S, int NT, int CB, bool SLIDE = false>
int launch(const DipWgradDesc& d, hipStream_t st, int CinP_slab = 0) {
    using C = WCfg<KS, S, NT, CB>;
    static bool attr_set[16] = {};
    auto kern = conv_wgrad_kernel<KS, S, NT, CB, SLIDE>;
    auto kern_g = conv_wgrad_kernel<KS, S, NT, CB, SLIDE, true>;         // grouped multi-instance form (dip_group.h)
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    if (d.nsplit < 1) DIP_FAIL("conv_wgrad: nsplit out of range");
    // a <= 4-channel tail behind 1..8 full 32-channel chunks is shared out among the full-chunk workgroups
    // (phase 2 of the kernel) instead of getting workgroups of its own
    int ragged_parts = 0;
    if (KS == 3 && NT == 9 && CB == 1) {
        static const bool no_parts = getenv("DIP_WGRAD_NO_RAGGED_PARTS") != nullptr;
        const int tail = d.Cin & 31, nfull = d.Cin >> 5;
        if (!no_parts && tail >= 1 && tail <= 4 && nfull >= 1 && nfull <= 8) ragged_parts = nfull;
    }
    // narrow layers: kw waves per 32-column block, each with a slab of its own (d.nsplit counts slabs)
    int kw = wgrad_kw(CoutP);
    if (ragged_parts > 0 || (d.nsplit % kw) != 0) kw = 1;
    if (d.nsplit / kw > ntiles) DIP_FAIL("conv_wgrad: nsplit out of range (more walkers than pixel tiles)");
    if (SLIDE && ragged_parts == 0 && KS == 3 && NT == 9 && (d.Cin & 31) >= 1 && (d.Cin & 31) <= 4)
        return launch<KS, S, NT, CB, false>(d, st, CinP_slab);      // a packed tail chunk in phase 1: the one-loop kernel
    dim3 grid(d.nsplit / kw, ragged_parts > 0 ? ragged_parts : dip_cdiv(CinP, C::CW), C::NGROUPS * dip_cdiv(CoutP, 128));
    // CinP_slab: row count of the slabs when this launch covers only the leading channels of the layer
    dip_launch_pair<DIP_FAM_WGRAD>(kern, kern_g, grid, dim3(256), C::LDS_BYTES, st, d, ntx, ntiles, CinP_slab > 0 ? CinP_slab : CinP, CoutP,
                                   ragged_parts, kw, 0);
    DIP_CHECK_LAUNCH();
    return 0;
}

// only phase 2 of conv_wgrad_kernel<3, 1, 9, 1, SLIDE>: the (tap, channel)-packed <= 4-channel tail of a 132-channel layer,
// shared out among `nfull` workgroups per walker (rows CinMain .. CinMain + 31 of every slab)
int launch_tail(const DipWgradDesc& d, hipStream_t st) {
    using C = WCfg<3, 1, 9, 1>;
    static bool attr_set[16] = {};
    auto kern = conv_wgrad_kernel<3, 1, 9, 1, true>;
    auto kern_g = conv_wgrad_kernel<3, 1, 9, 1, true, true>;
    if (dip_once_per_device(attr_set)) {
        hipError_t e = dip_pair_lds_attr(kern, kern_g, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    const int tail = d.Cin & 31, nfull = d.Cin >> 5;
    if (!(tail >= 1 && tail <= 4 && nfull >= 1 && nfull <= 8)) DIP_FAIL("conv_wgrad_tail: needs 1..8 full 32-channel chunks + a 1..4-channel tail");
    if (d.nsplit < 1 || d.nsplit > ntx * nty) DIP_FAIL("conv_wgrad_tail: nsplit out of range");
    dip_launch_pair<DIP_FAM_WGRAD>(kern, kern_g, dim3(d.nsplit, nfull, dip_cdiv(CoutP, 128)), dim3(256), C::LDS_BYTES, st, d, ntx, ntx * nty, CinP,
                                   CoutP, nfull, 1, 1);
    DIP_CHECK_LAUNCH();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Thin 1x1 weight gradient (Cout <= 8: the 4-channel skip convs and the RGB output conv).
// HBM-bound streaming kernel on the vector ALU: thread (prow, cg) owns 4 input channels and NO
// output channels, walks its pixels, and the block tree-reduces over prow.  One slab per block,
// same layout as the MFMA kernel's slabs, so dip_wgrad_reduce finishes it.
// ------------------------------------------------------------------------------------------
template <int NO, bool GRP = false>
__global__ __launch_bounds__(256) void thin1x1_wgrad_kernel(const DipWgradDesc d_, const int CinP, const int CoutP,
                                                            const int ppb, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipWgradDesc, d);
    constexpr int W = NO * 4 + NO;                  // per-thread accumulators: dW[o][4] + dbias[o]
    extern __shared__ __attribute__((aligned(16))) float sh[];
    const int nc4 = (d.Cin + 3) >> 2;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    const int prow = threadIdx.x / nc4, cg = threadIdx.x - prow * nc4;
    const bool active = (int)threadIdx.x < rpi * nc4;
    float acc[W];
#pragma unroll
    for (int i = 0; i < W; ++i) acc[i] = 0.f;
    if (active) {
        const int ch = cg * 4;
        const bool has_tr = d.tr.a != nullptr;
        f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_tr) {
            ta = *reinterpret_cast<const f32x4*>(d.tr.a + ch);
            tb = *reinterpret_cast<const f32x4*>(d.tr.b + ch);
        }
        const int npix = d.Hout * d.Wout;
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        // 4 pixels per trip: all loads of the trip are issued before the first use (the one-pixel walk was a chain
        // of dependent memory round trips: 80 us for the 512^2 output conv, 1.7 TB/s); a pixel past the end reads
        // pixel p0 and is weighted 0
        const float slope = d.tr.slope;
        for (int pq = p0 + prow; pq < p1; pq += 4 * rpi) {
            f32x4 u[4], t[4][NO / 4];
            float wgt[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = pq + j * rpi;
                wgt[j] = p < p1 ? 1.f : 0.f;
                const size_t pp = (size_t)(p < p1 ? p : p0);
                u[j] = *reinterpret_cast<const f32x4*>(d.x + pp * d.Cx + ch);
#pragma unroll
                for (int q = 0; q < NO / 4; ++q) t[j][q] = *reinterpret_cast<const f32x4*>(d.dy + pp * d.Cdy + q * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 uu = u[j];
                if (has_tr) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) uu[e] = dip_act(fmaf(ta[e], uu[e], tb[e]), slope);
                }
#pragma unroll
                for (int o = 0; o < NO; ++o) {
                    const float g = t[j][o / 4][o % 4] * wgt[j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[o * 4 + e] = fmaf(g, uu[e], acc[o * 4 + e]);
                    acc[NO * 4 + o] += g;
                }
            }
        }
    }
    float* mine = sh + (size_t)threadIdx.x * W;
#pragma unroll
    for (int i = 0; i < W; ++i) mine[i] = acc[i];
    for (int s = dip_pow2_ceil(rpi) >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (active && prow < s && prow + s < rpi) {
            const float* q = sh + (size_t)((prow + s) * nc4 + cg) * W;
#pragma unroll
            for (int i = 0; i < W; ++i) { acc[i] += q[i]; mine[i] = acc[i]; }
        }
    }
    if (active && prow == 0) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
            if (o < d.Cout) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = cg * 4 + e;
                    if (c < CinP) d.partial[((size_t)blockIdx.x * CinP + c) * CoutP + o] = acc[o * 4 + e];
                }
                if (cg == 0 && d.bias_partial != nullptr) d.bias_partial[(size_t)blockIdx.x * CoutP + o] = acc[NO * 4 + o];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a conv with <= 4 INPUT channels (the first conv of a net whose input is an image / a
// 1..4-plane noise tensor: inpainting 'library' has 1, the snail net 3, the meshgrid input 2): M = taps x 4
// channels is far too thin for the MFMA kernel, which stages a 32-channel halo per tile (5x5 stride 2, 1 -> 16
// channels at 448x704: 327 us for 0.06 GFLOP).  Vector-ALU streaming kernel instead: thread (prow, og) owns 4
// output channels and the KS taps of ONE filter row (blockIdx.y) x 4 channels = KS*16 accumulators, walks its
// pixels (dy once, KS neighbouring x pixels of 16 bytes each), then the block tree-reduces over prow, one filter
// column at a time, into one slab per block -- same slab layout as the MFMA kernel, dip_wgrad_reduce finishes it.
// ------------------------------------------------------------------------------------------
template <int KS, bool GRP = false>
__global__ __launch_bounds__(256) void thin_cin_wgrad_kernel(const DipWgradDesc d_, const int CinP, const int CoutP,
                                                             const int ppb, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipWgradDesc, d);
    __shared__ __attribute__((aligned(16))) float sh[256 * 16];
    const int nog = (d.Cout + 3) >> 2;                   // groups of 4 output channels (<= 64)
    const int rpi = 256 / nog;
    const int prow = threadIdx.x / nog, og = threadIdx.x - prow * nog;
    const bool active = (int)threadIdx.x < rpi * nog;
    const int ky = blockIdx.y;
    float acc[KS][16];                                   // [kx][c * 4 + o]
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        const bool has_tr = d.tr.a != nullptr;
        const float slope = has_tr ? d.tr.slope : 1.f;
        f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_tr) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < d.Cin) { ta[e] = d.tr.a[e]; tb[e] = d.tr.b[e]; }
        }
        const int npix = d.Hout * d.Wout;
        const int p0 = blockIdx.x * ppb, p1 = min(p0 + ppb, npix);
        const bool reflect = d.pad_mode == DIP_PAD_REFLECT;
        for (int p = p0 + prow; p < p1; p += rpi) {
            const int oy = p / d.Wout, ox = p - oy * d.Wout;
            const f32x4 g = *reinterpret_cast<const f32x4*>(d.dy + (size_t)p * d.Cdy + og * 4);
            int sy = oy * d.stride + ky - d.off;
            if (reflect) sy = dip_reflect(sy, d.Hin);
            const bool yok = (unsigned)sy < (unsigned)d.Hin;
            f32x4 xs[KS];
            float ok[KS];
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {            // all loads first (clamped address, validity as a weight)
                int sx = ox * d.stride + kx - d.off;
                if (reflect) sx = dip_reflect(sx, d.Win);
                const bool v = yok & ((unsigned)sx < (unsigned)d.Win);
                ok[kx] = v ? 1.f : 0.f;
                xs[kx] = *reinterpret_cast<const f32x4*>(d.x + (v ? ((size_t)sy * d.Win + sx) * d.Cx : 0));
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                f32x4 u = xs[kx];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    // zero padding contributes 0 (not act(b)); channels >= Cin are zero pad channels of x
                    const float t = has_tr ? dip_act(fmaf(ta[c], u[c], tb[c]), slope) : u[c];
                    const float uc = (c < d.Cin) ? t * ok[kx] : 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[kx][c * 4 + o] = fmaf(uc, g[o], acc[kx][c * 4 + o]);
                }
            }
            bsum += g;
        }
    }
    // block reduction over prow, one filter column (16 values per thread) at a time
    const int tap0 = ky * KS;
    for (int kx = 0; kx <= KS; ++kx) {                   // kx == KS: the bias sums (filter row 0 only)
        if (kx == KS && (ky != 0 || d.bias_partial == nullptr)) break;
        float* mine = sh + (size_t)threadIdx.x * 16;
        __syncthreads();                                 // the previous column's reads are done
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = 0.f;
            if (kx < KS) {
#pragma unroll
                for (int k = 0; k < KS; ++k) v = (k == kx) ? acc[k][i] : v;
            } else {
                v = i < 4 ? bsum[i] : 0.f;
            }
            mine[i] = v;
        }
        for (int st = dip_pow2_ceil(rpi) >> 1; st >= 1; st >>= 1) {
            __syncthreads();
            if (active && prow < st && prow + st < rpi) {
                const float* q = sh + (size_t)((prow + st) * nog + og) * 16;
#pragma unroll
                for (int i = 0; i < 16; ++i) mine[i] += q[i];
            }
        }
        if (active && prow == 0) {
            if (kx < KS) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        const int oo = og * 4 + o;
                        if (c < d.Cin && oo < CoutP)
                            d.partial[(((size_t)blockIdx.x * (KS * KS) + tap0 + kx) * CinP + c) * CoutP + oo] = mine[c * 4 + o];
                    }
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o)
                    if (og * 4 + o < CoutP) d.bias_partial[(size_t)blockIdx.x * CoutP + og * 4 + o] = mine[o];
            }
        }
    }
}

// pixels per block of the thin-input kernel: <= 512 slabs and <= 64 MB of slabs
int thin_cin_ppb(int npix, int Cout, int ks, int* nblk) {
    const int nog = (Cout + 3) / 4;
    const int rpi = 256 / nog;
    int ppb = dip_cdiv(npix, 512);
    if (ppb < rpi * 4) ppb = rpi * 4;
    const long long slab = (long long)ks * ks * 32 * dip_round_up(Cout, 32);
    while ((long long)dip_cdiv(npix, ppb) * slab > (16ll << 20)) ppb *= 2;
    *nblk = dip_cdiv(npix, ppb);
    return ppb;
}

bool is_thin_cin(int ks, int Cin, int Cout) {
    static const bool off = getenv("DIP_WGRAD_NO_THIN_CIN") != nullptr;
    return !off && Cin <= 4 && Cout <= 256 && (ks == 3 || ks == 5 || ks == 7);
}

int thin_ppb(int npix, int Cin, int* nblk) {
    const int nc4 = (Cin + 3) / 4;
    int rpi = 256 / nc4;
    if (rpi < 1) rpi = 1;
    int ppb = dip_cdiv(npix, 512);      // <= 512 slabs: the slab reduction reads nblk x CinP x CoutP floats
    if (ppb < rpi * 8) ppb = rpi * 8;
    *nblk = dip_cdiv(npix, ppb);
    return ppb;
}

bool is_thin(int ks, int Cin, int Cout) { return ks == 1 && Cout <= 8 && Cin <= 1024; }


// block = 8 split-lanes x 32 consecutive outputs (o fastest -> 128-B coalesced slab reads); each
// split-lane sums slabs sl, sl+8, ... then the 8 partials are added in a fixed order.
template <bool GRP = false>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial_,
                                                           const float* __restrict__ bias_partial_, int nsplit, int KK,
                                                           int Cin, int Cout, int CinP, int CoutP, float* dw_,
                                                           float* dbias_, const DipGrpArg<GRP> grp) {
    DIP_GRP_PTR(const float*, partial);
    DIP_GRP_PTR(const float*, bias_partial);
    DIP_GRP_PTR(float*, dw);
    DIP_GRP_PTR(float*, dbias);
    __shared__ float sh[8][32];
    const int oi = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int id = blockIdx.x * 32 + oi;
    const int total = KK * Cin * Cout;
    float s = 0.f;
    int o = 0, c = 0, tap = 0;
    if (id < total) {
        o = id % Cout;
        c = (id / Cout) % Cin;
        tap = id / (Cout * Cin);
        const size_t slab = (size_t)KK * CinP * CoutP;
        const float* p = partial + ((size_t)tap * CinP + c) * CoutP + o;
        // the <= 4-channel tail chunk of a 3x3 layer arrives in 8 four-row parts (conv_wgrad_kernel phase 2;
        // kernels that do not share it out leave parts 1..7 zero)
        const int tail = Cin & 31;
        if (KK == 9 && Cin > 32 && tail >= 1 && tail <= 4 && c >= Cin - tail) {
#pragma unroll 2
            for (int k = sl; k < nsplit; k += 8)
#pragma unroll
                for (int q = 0; q < 8; ++q) s += p[k * slab + (size_t)q * 4 * CoutP];
        } else {
#pragma unroll 4
            for (int k = sl; k < nsplit; k += 8) s += p[k * slab];      // 4 loads in flight, same summation order
        }
    } else if (dbias != nullptr && id < total + Cout) {
        o = id - total;
#pragma unroll 4
        for (int k = sl; k < nsplit; k += 8) s += bias_partial[(size_t)k * CoutP + o];
    }
    sh[sl][oi] = s;
    __syncthreads();
    if (sl == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) s += sh[r][oi];
        if (id < total) dw[((size_t)o * Cin + c) * KK + tap] = s;
        else if (dbias != nullptr && id < total + Cout) dbias[o] = s;
    }
}

}  // namespace

extern "C" int dip_conv_wgrad_ntiles(int Hout, int Wout) { return dip_cdiv(Wout, 16) * dip_cdiv(Hout, 4); }

// Number of partial slabs (= nsplit of DipWgradDesc) the weight-gradient kernels should be run
// with: ~512 workgroups in flight for the MFMA kernels, one slab per block for the thin kernel,
// and never more than 256 MB of slabs.
extern "C" int dip_wgrad_thin_shape_ok(int Hout, int Wout, int Cin, int Cout, int ks, int stride);      // wgrad_thin.hip
extern "C" int dip_wgrad_thin_nsplit(int Hout, int Wout, int Cin, int Cout, int ks, int stride);
extern "C" int dip_wgrad_thin_eligible(const DipWgradDesc* dp);
extern "C" int dip_wgrad_thin(const DipWgradDesc* dp, void* stream);

extern "C" int dip_wgrad_plan(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* nsplit) {
    // thin layers (8..64 input channels, <= 64 output channels, 3x3 / 5x5): wgrad_thin_kernel, one slab per pixel-tile walker
    if (dip_wgrad_thin_shape_ok(Hout, Wout, Cin, Cout, ks, stride)) {
        *nsplit = dip_wgrad_thin_nsplit(Hout, Wout, Cin, Cout, ks, stride);
        return 0;
    }
    if (is_thin(ks, Cin, Cout)) {
        int nblk;
        thin_ppb(Hout * Wout, Cin, &nblk);
        *nsplit = nblk;
        return 0;
    }
    if (is_thin_cin(ks, Cin, Cout)) {
        int nblk;
        thin_cin_ppb(Hout * Wout, Cout, ks, &nblk);
        *nsplit = nblk;
        return 0;
    }
    const int CinP = dip_round_up(Cin, 32), CoutP = dip_round_up(Cout, 32);
    const int nt = dip_conv_wgrad_ntiles(Hout, Wout);
    const int cw = (ks == 1) ? 128 : 32;
    const int groups = (ks == 5) ? 5 : (ks == 7 ? 7 : (ks == 8 ? 8 : (ks == 12 ? 24 : 1)));
    int chunks = dip_cdiv(CinP, cw);
    // a ragged <= 4-channel tail of a 3x3 conv runs the packed variant (2 of 9 MFMAs): its
    // workgroups are light, so fill the chip with the full chunks' workgroups
    if (ks == 3 && chunks > 1 && Cin - (chunks - 1) * 32 <= 4) chunks -= 1;
    const int per_split = chunks * groups * dip_cdiv(CoutP, 128);
    int n = 512 / per_split;
    // >= 4 pixel tiles per workgroup: amortise its slab write + the reduce.  3x3 stride 2: >= 2 -- a tile stages a 9 x 33-pixel
    // halo, and 512 workgroups with 2 tiles each beat 256 with 4 (tools/s2_time.py on MI355X, kernel + reduce:
    // 32 > 128 @ 256 x 256 out: 97 + 8 -> 74 + 13 us, 128 > 128 @ 128 x 128 out: 92 + 8 -> 72 + 14 us; twice as many again: slower)
    const int min_tiles = (ks == 3 && stride == 2) ? 2 : 4;
    if (n > nt / min_tiles) n = nt / min_tiles;
    if (n < 1) n = 1;
    // the layers wgrad_bf3_kernel takes (3x3 stride 1, >= 512 tiles of 2 x 16 pixels): ONE 8-wave workgroup per CU that fills its
    // register file, resident for the whole launch -- 256 of them leave no CU to the dependent chain of the main stream, which then
    // waits for the launch to end (profiles/r06_timeline_three_streams.txt: a 6 us bn_bwd_finalize "runs" 450 us beside the 512^2
    // weight gradient).  DIP_WGRAD_BF3_WGS caps the grid so that 256 - cap CUs stay free for the chain.
    static const int bf3_wgs = [] { const char* e = getenv("DIP_WGRAD_BF3_WGS"); return e ? atoi(e) : 0; }();
    if (bf3_wgs > 0 && ks == 3 && stride == 1 && Cin >= 32 && Cout >= 97 && dip_cdiv(Wout, 16) * dip_cdiv(Hout, 2) >= 512) {
        const int cap = bf3_wgs / (dip_cdiv(chunks, 2) * dip_cdiv(CoutP, 128));
        if (cap >= 1 && n > cap) n = cap;
    }
    const long long slab = (long long)ks * ks * CinP * CoutP;
    while (n > 1 && (long long)n * slab > (64ll << 20)) n /= 2;
    *nsplit = n * wgrad_kw(CoutP);         // narrow layers: kw slabs per workgroup
    return 0;
}

// Cost model of one MFMA weight-gradient launch (microseconds): `n` pixel splits x `chunks` channel
// chunks x `groups` tap groups x `oblk` 128-column blocks of workgroups, two resident per CU; every
// workgroup walks ceil(nt / n) 64-pixel tiles, each costing its MFMAs (taps-per-workgroup x 32 K steps
// x 64 cycles) plus the staging of the tile; then the n slabs are written once and read once by the
// reduction.  Only used for layers with <= 256 tiles, where launch shape, not MFMA rate, sets the time.
static double wgrad_cost(int nt, int n, int chunks, int groups, int oblk, int taps_per_wg, double slab_bytes) {
    const int wgs = n * chunks * groups * oblk;
    const int rounds = dip_cdiv(wgs, 512);
    const int tiles = dip_cdiv(nt, n);
    const double t_tile = (taps_per_wg * 32.0 * 64.0 + 3000.0) / 2400.0;
    const double t_main = rounds * tiles * t_tile + 4.0;
    const double t_slab = 2.0 * n * slab_bytes / 3.0e6 + 3.0;
    return t_main + t_slab;
}

extern "C" int dip_wgrad_plan2(int Hout, int Wout, int Cin, int Cout, int ks, int stride, int* nsplit, int* tap_groups,
                               int* chan_block) {
    *tap_groups = 1;
    *chan_block = (ks == 1) ? 4 : 1;
    int rc = dip_wgrad_plan(Hout, Wout, Cin, Cout, ks, stride, nsplit);
    if (rc) return rc;
    if (is_thin(ks, Cin, Cout) || is_thin_cin(ks, Cin, Cout) || dip_wgrad_thin_shape_ok(Hout, Wout, Cin, Cout, ks, stride)) return 0;
    const int nt = dip_conv_wgrad_ntiles(Hout, Wout);
    static const bool off = getenv("DIP_WGRAD_NO_SMALL_PLAN") != nullptr;
    if (ks == 5 && !off) {
        // 5x5 layers (round 6; tools/wgrad_sweep.py k5 on the 'library' net's shapes, kernel + slab reduction in us,
        // profiles/r06_wgrad_sweep_k5.txt): dip_wgrad_plan's ">= 4 pixel tiles per workgroup, ~512 workgroups" gave the
        // low-resolution 128-channel layers 20..100 heavy workgroups; one pixel tile per walker (<= 128 walkers) and, at
        // <= 2 tiles, one TAP per workgroup instead of one filter row:
        //   128>128 @28x44 (nt 21): (5 rows, n 5) 55+11 -> (n 21) 27+15      64>128 s2: 68+7 -> 22+8
        //   128>128 @14x22 (nt 8):  (n 2) 44+11 -> (n 8) 18+12               s2: 55+11 -> 20+12
        //   128>128 @7x11  (nt 2):  (n 1) 24+11 -> (25 taps, n 2) 10+11      s2: 31+11 -> 13+11
        //   16>32 s2 @112x176 (nt 308): (n 77) 95+5 -> (n 128) 76+8          16>16 @224x352: 101+5 -> 91+5
        const int CoutP5 = dip_round_up(Cout, 32);
        int n = nt < 128 ? nt : 128;
        const long long slab5 = 25ll * dip_round_up(Cin, 32) * CoutP5;
        while (n > 1 && (long long)n * slab5 > (64ll << 20)) n /= 2;
        *nsplit = n * wgrad_kw(CoutP5);
        *tap_groups = nt <= 2 ? 25 : 5;
        return 0;
    }
    if (ks != 3 && ks != 1) return 0;
    if (nt > 256 || off) return 0;
    if (ks == 3 && nt > 128) return 0;
    const int CinP = dip_round_up(Cin, 32), CoutP = dip_round_up(Cout, 32);
    const int oblk = dip_cdiv(CoutP, 128);
    const double slab = 4.0 * ks * ks * CinP * CoutP;
    double best = 1e30;
    int bn = *nsplit, bg = 1, bcb = *chan_block;
    if (ks == 3) {
        // Measured on MI355X (tools/wgrad_sweep.py small, kernel + slab reduction, 128 -> 128 channels):
        //   64x64 out (nt 64): (g3, n32) 32 us | (g1, n64) 34 | old plan (g1, n16) 80
        //   32x32 out (nt 16): (g3, n16) 18 us | (g9, n16) 18 | old plan (g1, n4) 79
        //   16x16 out (nt  4): (g9, n4) 13 us | (g3, n4) 16  | old plan (g1, n1) 118
        //   128x128 out (nt 256): (g3, n64) 70 ~ (g1, n64) 73: the large-layer plan stays.
        if (nt <= 8) { bn = nt; bg = 9; }
        else if (nt <= 32) { bn = nt; bg = 3; }
        else if (nt <= 128) { bn = nt / 2; bg = 3; }
        (void)best;
    } else {
        const int cbs[2] = {4, 1};
        for (int ci = 0; ci < 2; ++ci) {
            const int chunks = dip_cdiv(CinP, 32 * cbs[ci]);
            for (int n = 1; n <= nt; n *= 2) {
                const double c = wgrad_cost(nt, n, chunks, 1, oblk, cbs[ci], slab);
                if (c < best) { best = c; bn = n; bcb = cbs[ci]; }
            }
        }
    }
    *nsplit = bn * ((ks == 3) ? wgrad_kw(CoutP) : 1);
    *tap_groups = bg;
    *chan_block = bcb;
    return 0;
}

extern "C" int dip_wgrad_bf3_eligible(const DipWgradDesc* dp);
extern "C" int dip_wgrad_bf3(const DipWgradDesc* dp, void* stream);

extern "C" int dip_wgrad_tail_stream_ok(const DipWgradDesc* dp);            // wgrad_tail.hip
extern "C" int dip_wgrad_tail_stream(const DipWgradDesc* dp, void* stream);
extern "C" int dip_conv_wgrad_tail(const DipWgradDesc* dp, void* stream) {
    if (dip_wgrad_tail_stream_ok(dp)) return dip_wgrad_tail_stream(dp, stream);      // round 6: the streaming form
    return launch_tail(*dp, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int dip_conv_wgrad(const DipWgradDesc* dp, void* stream) {
    const DipWgradDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((d.Cx & 3) || (d.Cdy & 3)) DIP_FAIL("conv_wgrad: channel strides must be multiples of 4");
    if (dip_wgrad_bf3_eligible(dp)) return dip_wgrad_bf3(dp, stream);        // big 3x3 stride-1 layers: bf16 matrix pipe
    if (dip_wgrad_thin_eligible(dp)) return dip_wgrad_thin(dp, stream);      // thin layers: 16x16x4 MFMA tiles (wgrad_thin.hip)
    if (is_thin(d.ks, d.Cin, d.Cout)) {
        int nblk;
        const int ppb = thin_ppb(d.Hout * d.Wout, d.Cin, &nblk);
        if (nblk != d.nsplit) DIP_FAIL("conv_wgrad: thin 1x1 path needs nsplit from dip_wgrad_plan");
        const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
        if (d.Cout <= 4) {
            dip_launch_pair<DIP_FAM_THIN>(thin1x1_wgrad_kernel<4>, thin1x1_wgrad_kernel<4, true>, dim3(nblk), dim3(256), 256 * 20 * 4, st, d, CinP,
                                          CoutP, ppb);
        } else {
            dip_launch_pair<DIP_FAM_THIN>(thin1x1_wgrad_kernel<8>, thin1x1_wgrad_kernel<8, true>, dim3(nblk), dim3(256), 256 * 40 * 4, st, d, CinP,
                                          CoutP, ppb);
        }
        DIP_CHECK_LAUNCH();
        return 0;
    }
    if (is_thin_cin(d.ks, d.Cin, d.Cout)) {
        int nblk;
        const int ppb = thin_cin_ppb(d.Hout * d.Wout, d.Cout, d.ks, &nblk);
        if (nblk != d.nsplit) DIP_FAIL("conv_wgrad: the <= 4 input channel path needs nsplit from dip_wgrad_plan");
        if (d.Cx < 4) DIP_FAIL("conv_wgrad: channel stride of x must be >= 4");
        const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
        const dim3 grid(nblk, d.ks);
        if (d.ks == 3) dip_launch_pair<DIP_FAM_THIN>(thin_cin_wgrad_kernel<3>, thin_cin_wgrad_kernel<3, true>, grid, dim3(256), 0, st, d, CinP, CoutP, ppb);
        else if (d.ks == 5) dip_launch_pair<DIP_FAM_THIN>(thin_cin_wgrad_kernel<5>, thin_cin_wgrad_kernel<5, true>, grid, dim3(256), 0, st, d, CinP, CoutP, ppb);
        else dip_launch_pair<DIP_FAM_THIN>(thin_cin_wgrad_kernel<7>, thin_cin_wgrad_kernel<7, true>, grid, dim3(256), 0, st, d, CinP, CoutP, ppb);
        DIP_CHECK_LAUNCH();
        return 0;
    }
    const int g = d.tap_groups > 1 ? d.tap_groups : 1;
    if (d.ks == 1 && d.stride == 1) return d.chan_block == 1 ? launch<1, 1, 1, 1>(d, st) : launch<1, 1, 1, 4>(d, st);
    if (d.ks == 3 && (g != 1 && g != 3 && g != 9)) DIP_FAIL("conv_wgrad: tap_groups must be 1, 3 or 9");
    if (d.ks == 3 && d.stride == 1) {
        // sliding A-operand window (see the kernel); DIP_WGRAD_NO_SLIDE=1 keeps the one-read-per-MFMA loop (A/B)
        static const bool slide = getenv("DIP_WGRAD_NO_SLIDE") == nullptr;
        if (slide && g == 1) return launch<3, 1, 9, 1, true>(d, st);
        if (slide && g == 3) return launch<3, 1, 3, 1, true>(d, st);
        return g == 1 ? launch<3, 1, 9, 1>(d, st) : (g == 3 ? launch<3, 1, 3, 1>(d, st) : launch<3, 1, 1, 1>(d, st));
    }
    if (d.ks == 3 && d.stride == 2)
        return g == 1 ? launch<3, 2, 9, 1>(d, st) : (g == 3 ? launch<3, 2, 3, 1>(d, st) : launch<3, 2, 1, 1>(d, st));
    // 5x5: one filter row per workgroup (5 tap groups) or -- low-resolution layers, tap_groups == 25 from dip_wgrad_plan2 --
    // one tap per workgroup
    if (d.ks == 5 && g != 1 && g != 5 && g != 25) DIP_FAIL("conv_wgrad: tap_groups of a 5x5 layer must be 5 (0, 1) or 25");
    if (d.ks == 5 && d.stride == 1) return g == 25 ? launch<5, 1, 1, 1>(d, st) : launch<5, 1, 5, 1>(d, st);
    if (d.ks == 5 && d.stride == 2) return g == 25 ? launch<5, 2, 1, 1>(d, st) : launch<5, 2, 5, 1>(d, st);
    if (d.ks == 7 && d.stride == 1) return launch<7, 1, 7, 1>(d, st);
    if (d.ks == 7 && d.stride == 2) return launch<7, 2, 7, 1>(d, st);
    if (d.ks == 8 && d.stride == 2) return launch<8, 2, 8, 1>(d, st);        // Lanczos2 Downsampler conv: one filter row per workgroup
    if (d.ks == 12 && d.stride == 2) return launch<12, 2, 6, 1>(d, st);      // Lanczos3: half a filter row
    DIP_FAIL("conv_wgrad: unsupported kernel size / stride");
}

extern "C" int dip_wgrad_reduce(const float* partial, const float* bias_partial, int nsplit, int ks, int Cin,
                                int Cout, float* dw, float* dbias, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int KK = ks * ks;
    const int CinP = dip_round_up(Cin, 32), CoutP = dip_round_up(Cout, 32);
    const int total = KK * Cin * Cout + (dbias ? Cout : 0);
    dip_launch_pair<DIP_FAM_WGRAD>(wgrad_reduce_kernel<false>, wgrad_reduce_kernel<true>, dim3(dip_cdiv(total, 32)), dim3(256), 0, st, partial,
                                   dbias ? bias_partial : (const float*)nullptr, nsplit, KK, Cin, Cout, CinP, CoutP, dw, dbias);
    DIP_CHECK_LAUNCH();
    return 0;
}
