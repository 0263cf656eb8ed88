// 3x3 stride-1 weight gradient, large layers: 64 input channels x 128 output channels per workgroup.
//
// Same GEMM view and slab format as conv_wgrad.hip (M = input channels, N = output channels, K = output
// pixels in 4x16 tiles, one partial slab per pixel split), but a wave owns 32 channels x 64 columns x
// 9 taps = 18 MFMA accumulators, so every A operand (one ds_read_b32 per tap) feeds TWO MFMAs and a K
// step needs 11 LDS reads for 18 MFMAs instead of 10 for 9.  The instruction mix alone
// (tools/ubench/wgrad_mix64.hip) reaches 146 TFLOP/s against 127-133 for the 9-accumulator mix.
// 288 accumulator registers do not fit next to anything at two waves per SIMD: ONE workgroup per CU
// (4 waves, one per SIMD), 16 accumulators pinned in AGPRs and 2 in arch VGPRs by inline asm -- hipcc left
// to itself moves ~100 registers per K step between the two files.  With a single wave per SIMD nothing else
// hides the staging, so the tiles are double-buffered in LDS (2 x 59 KB): the global loads of tile t+1 are
// issued before the MFMAs of tile t, parked in registers, and written to the other buffer afterwards -- one
// barrier per tile.
// A <= 4-channel tail behind the 64-channel chunks (the 132-channel concat layers) is shared out as in
// conv_wgrad.hip phase 2: every workgroup does its share of the pixel tiles in the (tap, channel)-packed
// form and writes it as one or two of the 8 four-row parts that dip_wgrad_reduce adds up.
#include "dip_common.h"
#include <stdlib.h>

namespace {

struct W64 {
    static constexpr int TH = 4, TW = 16, NPX = 64;
    static constexpr int HTH = 6, HTW = 18, NPIX = HTH * HTW;
    static constexpr int CW = 64;
    static constexpr int U_FLOATS = NPIX * CW;
    static constexpr int D_FLOATS = NPX * 128;
    static constexpr int BUF = U_FLOATS + D_FLOATS;
    static constexpr int U_SLOTS = (NPIX * (CW / 4) + 255) / 256;
    static constexpr int LDS_BYTES = 2 * BUF * 4;
};

#define DIP_MFMA_A(acc, x, y) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))
#define DIP_MFMA_V(acc, x, y) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

__global__ __launch_bounds__(256, 1) void conv_wgrad64_kernel(const DipWgradDesc d, const int ntx, const int ntiles,
                                                              const int CinP, const int CoutP, const int tail_parts) {
    using C = W64;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;       // 32-channel half, 64-column half

    const int split = blockIdx.x;
    const int cchunk = blockIdx.y;
    const int o0 = blockIdx.z * 128;
    int c0 = cchunk * C::CW;
    const bool do_bias = (d.bias_partial != nullptr) && cchunk == 0 && wm == 0;

    f32x16 ca[16], cv[2];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ca[t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[t][r] = 0.f;
    float bsum0 = 0.f, bsum1 = 0.f;

    const bool has_tr = d.tr.a != nullptr;
    const float slope = has_tr ? d.tr.slope : 1.0f;
    const bool leaky = slope > 0.f;
    const int c4 = tid & 15;                       // this thread's 4-channel group in the staging
    bool cvalid = (c0 + c4 * 4) < d.Cin;
    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_tr && cvalid) {
        ta = *reinterpret_cast<const f32x4*>(d.tr.a + c0 + c4 * 4);
        tb = *reinterpret_cast<const f32x4*>(d.tr.b + c0 + c4 * 4);
    }

    // Staging registers of the NEXT tile.  Every load is unconditional (a lane outside the image / channel range
    // reads a valid dummy address) and the validity goes into a bit mask that commit() applies: a branch between
    // a load and the next load would serialise the memory pipeline of the wave.
    f32x4 ureg[C::U_SLOTS], dreg[8];
    unsigned umask = 0, dmask = 0;
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const int ty = tile / ntx, tx = tile - ty * ntx;
        umask = 0;
        dmask = 0;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            const int hp = f >> 4;
            const int hr = hp / C::HTW, hc = hp - hr * C::HTW;
            int sr = ty * C::TH + hr - d.off, sc = tx * C::TW + hc - d.off;
            if (d.pad_mode == DIP_PAD_REFLECT) {
                sr = dip_reflect(sr, d.Hin);
                sc = dip_reflect(sc, d.Win);
            }
            const bool ok = (f < C::NPIX * (C::CW / 4)) & ((unsigned)sr < (unsigned)d.Hin) & ((unsigned)sc < (unsigned)d.Win) & cvalid;
            const int off = ok ? (sr * d.Win + sc) * d.Cx + c0 + c4 * 4 : 0;      // (< 2^31 floats: checked by the launcher)
            ureg[i] = *reinterpret_cast<const f32x4*>(d.x + off);
            umask |= (ok ? 1u : 0u) << i;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + i * 256;           // float4 index: pixel = f >> 5, o4 = f & 31
            const int px = f >> 5, o4 = f & 31;
            const int oy = ty * C::TH + (px >> 4), ox = tx * C::TW + (px & 15);
            const int o = o0 + o4 * 4;
            const bool ok = (oy < d.Hout) & (ox < d.Wout) & (o < d.Cdy);
            const int off = ok ? (oy * d.Wout + ox) * d.Cdy + o : 0;
            dreg[i] = *reinterpret_cast<const f32x4*>(d.dy + off);
            dmask |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit = [&](float* buf) __attribute__((always_inline)) {      // registers -> LDS, producer BatchNorm+act on the way
        float* Us = buf;
        float* Ds = buf + C::U_FLOATS;
#pragma unroll
        for (int i = 0; i < C::U_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (f < C::NPIX * (C::CW / 4)) {
                f32x4 v = ureg[i];
                const bool ok = (umask >> i) & 1u;
                if (leaky) {                       // (workgroup-uniform; without a transform ta = 1, tb = 0, slope = 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? dip_act_leaky(fmaf(ta[e], v[e], tb[e]), slope) : 0.f;
                } else {                           // Swish / ELU
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? dip_act(fmaf(ta[e], v[e], tb[e]), slope) : 0.f;
                }
                *reinterpret_cast<f32x4*>(Us + f * 4) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x4 v = dreg[i];
            const bool ok = (dmask >> i) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            *reinterpret_cast<f32x4*>(Ds + (tid + i * 256) * 4) = v;
        }
    };

    // K steps of one tile: step s = output pixels 2s, 2s+1 (lanes 0-31 / 32-63); all LDS offsets are immediates
    auto compute_main = [&](const float* buf) __attribute__((always_inline)) {
        const float* ul = buf + half * C::CW + wm * 32 + l31;
        const float* dl = buf + C::U_FLOATS + half * 128 + wn * 64 + l31;
        auto rd = [&](auto S, float (&a)[9], float (&b)[2]) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            constexpr int base = ((s >> 3) * C::HTW + ((2 * s) & 15)) * C::CW;
            b[0] = dl[2 * s * 128];
            b[1] = dl[2 * s * 128 + 32];
#pragma unroll
            for (int t = 0; t < 9; ++t) a[t] = ul[base + ((t / 3) * C::HTW + (t % 3)) * C::CW];
        };
        auto mm = [&](float (&a)[9], float (&b)[2]) __attribute__((always_inline)) {
            bsum0 += b[0];
            bsum1 += b[1];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                DIP_MFMA_A(ca[2 * t], a[t], b[0]);
                DIP_MFMA_A(ca[2 * t + 1], a[t], b[1]);
            }
            DIP_MFMA_V(cv[0], a[8], b[0]);
            DIP_MFMA_V(cv[1], a[8], b[1]);
        };
        float a0[9], b0[2], a1[9], b1[2];
        rd(std::integral_constant<int, 0>{}, a0, b0);
        dip_static_for<0, 16>([&](auto I) {
            constexpr int s = 2 * decltype(I)::value;
            rd(std::integral_constant<int, s + 1>{}, a1, b1);
            mm(a0, b0);
            if constexpr (s + 2 < 32) rd(std::integral_constant<int, s + 2>{}, a0, b0);
            mm(a1, b1);
        });
    };
    // (tap, channel)-packed rows of the <= 4-channel tail: accumulator rows 4*tap + ch for taps 0..7 (ca[0], ca[1]
    // for the two column blocks), tap 8 in rows 0..3 of ca[2], ca[3].  The two channel-half waves of a column
    // half split the tile's K steps.
    auto compute_tail = [&](const float* buf) __attribute__((always_inline)) {
        const int ptap = l31 >> 2, pch = l31 & 3;
        const float* ul = buf + (wm * 2 * C::HTW) * C::CW + half * C::CW;
        const float* u0 = ul + ((ptap / 3) * C::HTW + (ptap % 3)) * C::CW + pch;
        const float* u1 = ul + (2 * C::HTW + 2) * C::CW + pch;
        const float* dl = buf + C::U_FLOATS + (wm * 32 + half) * 128 + wn * 64 + l31;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int base = ((s >> 3) * C::HTW + ((2 * s) & 15)) * C::CW;
            const float b0 = dl[2 * s * 128], b1 = dl[2 * s * 128 + 32];
            const float x0 = u0[base], x1 = u1[base];
            DIP_MFMA_A(ca[0], x0, b0);
            DIP_MFMA_A(ca[1], x0, b1);
            DIP_MFMA_A(ca[2], x1, b0);
            DIP_MFMA_A(ca[3], x1, b1);
        }
    };

    auto walk = [&](const int first, const int step, const bool tail) __attribute__((always_inline)) {
        int cur = 0;
        if (first < ntiles) {
            fetch(first);
            commit(smem);
        }
        __syncthreads();
        for (int tile = first; tile < ntiles; tile += step) {
            const bool more = tile + step < ntiles;
            if (more) fetch(tile + step);
            if (tail) compute_tail(smem + cur * C::BUF);
            else compute_main(smem + cur * C::BUF);
            if (more) commit(smem + (cur ^ 1) * C::BUF);
            __syncthreads();
            cur ^= 1;
        }
    };
    walk(split, d.nsplit, false);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");        // the asm MFMAs' results are read by ordinary code below

    // ---- this workgroup's partial slab: rows c0 + wm*32 .. +31, columns o0 + wn*64 .. +63 ----
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = o0 + wn * 64 + j * 32 + l31;
            if (o < CoutP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = c0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = (t < 8) ? ca[(2 * t + j) & 15][r] : cv[j][r];
                    if (c < CinP) d.partial[(((size_t)split * 9 + t) * CinP + c) * CoutP + o] = v;
                }
            }
        }
    }
    if (do_bias) {
        const float t0 = bsum0 + __shfl_xor(bsum0, 32), t1 = bsum1 + __shfl_xor(bsum1, 32);
        const int o = o0 + wn * 64 + l31;
        if (half == 0) {
            if (o < CoutP) d.bias_partial[(size_t)split * CoutP + o] = t0;
            if (o + 32 < CoutP) d.bias_partial[(size_t)split * CoutP + o + 32] = t1;
        }
    }

    // ---- the ragged <= 4-channel tail, shared out (see the file header) ----
    if (tail_parts > 0) {
        const int CinMain = d.Cin & ~31;
        c0 = CinMain;
        cvalid = (c0 + c4 * 4) < d.Cin;
        ta = f32x4{1.f, 1.f, 1.f, 1.f};
        tb = f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_tr && cvalid) {
            ta = *reinterpret_cast<const f32x4*>(d.tr.a + c0 + c4 * 4);
            tb = *reinterpret_cast<const f32x4*>(d.tr.b + c0 + c4 * 4);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) ca[t][r] = 0.f;
        const int nch = gridDim.y;
        walk(split + cchunk * d.nsplit, d.nsplit * nch, true);
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        const int part = cchunk * 2 + wm;
        const int cbase = CinMain + 4 * part;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = o0 + wn * 64 + j * 32 + l31;
            if (o < CoutP) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;       // row = tap * 4 + channel
                    d.partial[(((size_t)split * 9 + (m >> 2)) * CinP + cbase + (m & 3)) * CoutP + o] = ca[j][r];
                    if (m < 4) d.partial[(((size_t)split * 9 + 8) * CinP + cbase + m) * CoutP + o] = ca[2 + j][r];
                }
                if (cchunk == 0 && wm == 0) {          // parts tail_parts..7 do not exist: zeros
#pragma unroll 1
                    for (int t = 0; t < 9; ++t)
                        for (int rr = 4 * tail_parts + half; rr < 32; rr += 2)
                            d.partial[(((size_t)split * 9 + t) * CinP + CinMain + rr) * CoutP + o] = 0.f;
                }
            }
        }
    }
}

}  // namespace

// domain of the 64-channel kernel: 3x3, stride 1, whole 64-channel chunks (at most 4) + an optional <= 4-channel tail
extern "C" int dip_wgrad64_eligible(int Cin, int ks, int stride) {
    const int main = Cin & ~31, tail = Cin & 31;
    return ks == 3 && stride == 1 && main >= 64 && (main % 64) == 0 && main <= 256 && tail <= 4;
}

extern "C" int dip_conv_wgrad64(const DipWgradDesc* dp, void* stream) {
    const DipWgradDesc& d = *dp;
    using C = W64;
    if (!dip_wgrad64_eligible(d.Cin, d.ks, d.stride)) DIP_FAIL("conv_wgrad64: layer outside the 64-channel kernel's domain");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad64_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
        attr_set = true;
    }
    const int ntx = dip_cdiv(d.Wout, C::TW), nty = dip_cdiv(d.Hout, C::TH);
    const int ntiles = ntx * nty;
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    if (d.nsplit < 1 || d.nsplit > ntiles) DIP_FAIL("conv_wgrad64: nsplit out of range");
    if ((long long)d.Hin * d.Win * d.Cx >= (1ll << 31) || (long long)d.Hout * d.Wout * d.Cdy >= (1ll << 31))
        DIP_FAIL("conv_wgrad64: tensor larger than 2^31 floats");
    const int nch = (d.Cin & ~31) / 64;
    const int tail_parts = (d.Cin & 31) ? 2 * nch : 0;
    dim3 grid(d.nsplit, nch, dip_cdiv(CoutP, 128));
    hipLaunchKernelGGL(conv_wgrad64_kernel, grid, dim3(256), C::LDS_BYTES, reinterpret_cast<hipStream_t>(stream), d, ntx,
                       ntiles, CinP, CoutP, tail_parts);
    DIP_CHECK_LAUNCH();
    return 0;
}
