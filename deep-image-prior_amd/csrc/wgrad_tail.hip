// Weight gradient of the <= 4-channel tail of a 132-channel 3x3 stride-1 layer (the decoder convs read a concat of
// 4 skip + 128 up-sampled channels, reference models/skip.py:50-53; autograd ConvolutionBackward, weight part, of
// models/common.py:120) -- the rows c_base .. c_base + 3 of
//
//   dW[tap][c][o] = sum_p  u[p + tap][c] * dy[p][o]            (u = transform(x), as in the forward)
//
// that wgrad_bf3_kernel leaves out (it takes whole 32-channel chunks).  Until round 6 these rows ran phase 2 of the fp32
// weight-gradient kernel on its own (conv_wgrad.hip, launch_tail): 4 x 16-pixel tiles with a 32-channel halo staged in LDS
// for 4 useful channels -- 74 us at 512 x 512 for 2.4 GFLOP and one pass over dy.  This kernel streams instead:
//   * (tap, channel) packed into the rows of the fp32 MFMA: taps 0..7 x 4 channels = its 32 rows; tap 8 (4 rows: a second,
//     empty MFMA per K step and column block) on the vector ALU;
//     v_mfma_f32_32x32x2_f32, K = a pair of output pixels.  B comes straight from global memory: a lane's B values are
//     NCB consecutive floats of its pixel's dy row, so a half-wave reads 32 NCB consecutive columns of one pixel and
//     accumulator block e holds columns o0 + NCB * lane + e.  A: the 3 x 18-pixel window of a RUN of 16 output pixels is
//     loaded once (54 lanes, one float4 = the four tail channels of a pixel), transformed and put into a wave-private
//     LDS window; a lane reads its (tap, channel) value of a K step from there.  (Reading A per K step from global
//     memory -- 16 scattered 4-byte loads per run, 16 cache lines each -- made the launch bound by the L1's line rate:
//     87 us at 512 x 512, against 74 for the kernel this one replaces.)
//   * an output row is cut into runs of 16 pixels; slab `walker` (DipWgradDesc.nsplit of them, the bf16-pipe kernel's) =
//     a contiguous range of runs; the 128 / (32 NCB) workgroups of a slab split the columns, a workgroup's 8 waves take
//     its runs in turn; the loads of WT_DEPTH - 1 runs are in flight under a run's MFMAs (dy is read once: 134 MB at 512 x 512);
//   * a wave writes its sums as one of the EIGHT four-row parts of the tail rows that dip_wgrad_reduce adds up (the layout of
//     phase 2 of the fp32 kernel, which shared the tail out among <= 8 workgroups per walker): no LDS, no barrier.
#include "dip_common.h"
#include "dip_group.h"
#include <stdlib.h>

namespace {

constexpr int WT_UNROLL = 8;
constexpr int WT_WAVES = 8;
constexpr int WT_DEPTH = 3;

template <int NCB> struct WtVec;
template <> struct WtVec<1> { typedef float T; };
template <> struct WtVec<2> { typedef f32x2 T; };
template <> struct WtVec<4> { typedef f32x4 T; };
template <int NCB> __device__ __forceinline__ float wt_get(const typename WtVec<NCB>::T& v, int e) { return v[e]; }
template <> __device__ __forceinline__ float wt_get<1>(const float& v, int) { return v; }

// source index of v under the padding rule, -1 outside (zero padding); selects, not a wave-uniform switch per call
__device__ __forceinline__ int wt_map(int v, int n, int pad_mode) {
    const int a = v < 0 ? -v : v;
    const int refl = a >= n ? 2 * (n - 1) - a : a;
    const int repl = min(max(v, 0), n - 1);
    const int m = pad_mode == DIP_PAD_REFLECT ? refl : (pad_mode == DIP_PAD_REPLICATE ? repl : v);
    return (m < 0 || m >= n) ? -1 : m;
}

constexpr int WT_RUN = 2 * WT_UNROLL;                 // output pixels of a run: WT_UNROLL K steps of the MFMA
constexpr int WT_WW = WT_RUN + 2;                      // its input window: 3 rows x (WT_RUN + 2) pixels x 4 channels
constexpr int WT_PITCH = WT_WW * 4 + 4;                // floats per window row in LDS (76: the three rows' reads on disjoint banks)

template <int NCB, int TR, bool GRP = false>
__global__ __launch_bounds__(64 * WT_WAVES) void wgrad_tail_kernel(const DipWgradDesc d_, const int CinP, const int CoutP, const int c_base,
                                                                  const int rpr, const int rps, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipWgradDesc, d);
    typedef typename WtVec<NCB>::T BV;
    __shared__ __attribute__((aligned(16))) float win[WT_WAVES][3 * WT_PITCH];      // one window per wave (wave-private: no barrier)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int walker = blockIdx.x;
    const int o0 = blockIdx.y * 32 * NCB;
    const int tail = d.Cin - c_base;
    // runs: an output row is cut into rpr runs of WT_RUN pixels (the last one ragged); slab `walker` owns rps consecutive runs
    const int nrun_all = d.Hout * rpr;
    const int R0 = walker * rps, R1 = min(R0 + rps, nrun_all);

    // MFMA row of this lane: (tap l31 / 4, channel l31 % 4)
    const int ch = l31 & 3, tap0 = l31 >> 2;
    const int ky0 = tap0 / 3, kx0 = tap0 - 3 * ky0;
    const float slope = d.tr.slope;
    // staging role of this lane: window pixel (hr, hc) for lane < 3 * WT_WW, all four channels
    const bool stager = lane < 3 * WT_WW;
    const int hr = lane / WT_WW, hc = lane - hr * WT_WW;
    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (TR != 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < tail) { ta[e] = d.tr.a[c_base + e]; tb[e] = d.tr.b[c_base + e]; }
    }
    const float* xc = d.x + c_base;
    const int ocol = o0 + NCB * l31;
    const bool colok = ocol < d.Cdy;                  // (Cdy is a multiple of 4: a whole vector is inside or outside)
    const float* dyc = d.dy + (colok ? ocol : 0);
    float* wl = &win[wave][0];
    const int rd0 = (ky0 * WT_PITCH) + (kx0 + half) * 4 + ch;          // + 8 s: this lane's block-0 value of K step s
    const int rd8 = (2 * WT_PITCH) + (2 + half) * 4;                   // + 8 s: tap 8 (ky 2, kx 2) of this lane's pixel

    f32x16 acc[NCB];                                  // taps 0..7 x 4 channels: the MFMA's 32 rows
    float acc8[4][NCB];                               // tap 8: 4 rows would cost a second MFMA per K step and column block; vector ALU
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc8[c][cb] = 0.f;
    }

    struct Run {                                     // what a run loads: its window pixel (stagers), its dy values, their validity
        f32x4 xw;
        BV b[WT_UNROLL];
        unsigned mb;                                 // bit s: K step s of this lane is inside the row and the column range
        bool xok;
    };
    // Everything a run needs is loaded by straight-line, unconditional code (clamped addresses, validity applied later): the
    // compiler's vmcnt bookkeeping then lets the loads of WT_DEPTH - 1 runs stay in flight under the MFMAs of a run.
    auto load = [&](Run& R, const int rg, const bool live) {
        const int oy = rg / rpr, oxf = (rg - oy * rpr) * WT_RUN;               // (wave-uniform)
        const int sy = wt_map(oy + hr - d.off, d.Hin, d.pad_mode), sx = wt_map(oxf + hc - d.off, d.Win, d.pad_mode);
        R.xok = live && stager && sy >= 0 && sx >= 0;
        R.xw = *reinterpret_cast<const f32x4*>(xc + (R.xok ? (sy * d.Win + sx) * d.Cx : 0));
        unsigned mb = 0u;
        const int qrow = oy * d.Wout;
#pragma unroll
        for (int s = 0; s < WT_UNROLL; ++s) {
            const int ox = oxf + 2 * s + half;
            const bool ok = ox < d.Wout;
            mb |= ok ? (1u << s) : 0u;
            R.b[s] = *reinterpret_cast<const BV*>(dyc + (qrow + (ok ? ox : 0)) * d.Cdy);
        }
        R.mb = (live && colok) ? mb : 0u;
    };
    auto mfmas = [&](const Run& R) {
        // the window: transform (producer BatchNorm + activation), zero for padding / channels past the tail, into LDS
        // (the window is wave-private and a wave's LDS operations complete in order, so no workgroup barrier -- but the compiler
        // must not move the float reads below across the float4 write, or the next write across them: wavefront-scope fences)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (stager) {
            f32x4 v = R.xw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                if (TR == 1) t = dip_act_leaky(fmaf(ta[e], t, tb[e]), slope);
                else if (TR == 2) t = dip_act(fmaf(ta[e], t, tb[e]), slope);
                v[e] = (R.xok && e < tail) ? t : 0.f;
            }
            *reinterpret_cast<f32x4*>(wl + hr * WT_PITCH + hc * 4) = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float u0[WT_UNROLL];
        f32x4 u8[WT_UNROLL];
#pragma unroll
        for (int s = 0; s < WT_UNROLL; ++s) {
            u0[s] = wl[rd0 + 8 * s];
            u8[s] = *reinterpret_cast<const f32x4*>(wl + rd8 + 8 * s);       // tap 8 of this lane's pixel, all four channels
        }
#pragma unroll
        for (int s = 0; s < WT_UNROLL; ++s) {
            const bool bok = (R.mb >> s) & 1u;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float bv = bok ? wt_get<NCB>(R.b[s], cb) : 0.f;
                acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(u0[s], bv, acc[cb], 0, 0, 0);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc8[c][cb] = fmaf(u8[s][c], bv, acc8[c][cb]);
            }
        }
    };
    // wave w takes the runs R0 + w, R0 + w + WT_WAVES, ...; runs past its last one re-load the slab's first run (cache hits)
    // with nothing valid, so that the loads of the steady state are unconditional
    const int first = R0 + wave;
    const int nruns = first < R1 ? (R1 - first - 1) / WT_WAVES + 1 : 0;
    auto run_id = [&](int r) { return r < nruns ? first + r * WT_WAVES : min(R0, nrun_all - 1); };
    Run buf[WT_DEPTH];
#pragma unroll
    for (int k = 0; k < WT_DEPTH - 1; ++k) load(buf[k], run_id(k), k < nruns);
    for (int r = 0; r < nruns; r += WT_DEPTH) {
#pragma unroll
        for (int k = 0; k < WT_DEPTH; ++k) {
            const int rn = r + k + WT_DEPTH - 1;
            load(buf[(k + WT_DEPTH - 1) % WT_DEPTH], run_id(rn), rn < nruns);
            if (r + k < nruns) mfmas(buf[k]);
        }
    }

    // ---- this wave's sums = part `wave` of the tail rows of slab `walker` (rows c_base + 4 * wave + c: dip_wgrad_reduce sums the
    // eight four-row parts of a tail, as it does for phase 2 of the fp32 kernel), columns o0 .. o0 + 32 NCB - 1 ----
    float* slab = d.partial + (size_t)walker * 9 * CinP * CoutP;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int col = ocol + cb;
        // tap 8: this lane summed the pixels of its half of every pair; the other half's sums sit 32 lanes away
        float t8[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t8[c] = acc8[c][cb] + __shfl_xor(acc8[c][cb], 32, 64);
        if (col >= CoutP) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;            // (tap row / 4, channel row % 4)
            slab[((size_t)(row >> 2) * CinP + c_base + 4 * wave + (r & 3)) * CoutP + col] = acc[cb][r];
        }
        if (half == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c) slab[((size_t)8 * CinP + c_base + 4 * wave + c) * CoutP + col] = t8[c];
        }
    }
}

template <int NCB>
int wt_launch(const DipWgradDesc& d, hipStream_t st, int c_base) {
    const int CinP = dip_round_up(d.Cin, 32), CoutP = dip_round_up(d.Cout, 32);
    const int rpr = dip_cdiv(d.Wout, WT_RUN);                       // runs per output row
    const int rps = dip_cdiv(d.Hout * rpr, d.nsplit);               // runs per slab
    const dim3 grid(d.nsplit, dip_cdiv(CoutP, 32 * NCB), 1), block(64 * WT_WAVES);
    const int tr = d.tr.a == nullptr ? 0 : (d.tr.slope > 0.f ? 1 : 2);
    if (tr == 0) dip_launch_pair<DIP_FAM_WGRAD>(wgrad_tail_kernel<NCB, 0, false>, wgrad_tail_kernel<NCB, 0, true>, grid, block, 0, st, d, CinP, CoutP, c_base, rpr, rps);
    else if (tr == 1) dip_launch_pair<DIP_FAM_WGRAD>(wgrad_tail_kernel<NCB, 1, false>, wgrad_tail_kernel<NCB, 1, true>, grid, block, 0, st, d, CinP, CoutP, c_base, rpr, rps);
    else dip_launch_pair<DIP_FAM_WGRAD>(wgrad_tail_kernel<NCB, 2, false>, wgrad_tail_kernel<NCB, 2, true>, grid, block, 0, st, d, CinP, CoutP, c_base, rpr, rps);
    DIP_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// 1 when dip_wgrad_tail_stream serves the tail of `d`: 3x3, stride 1, 1..4 channels behind >= 1 whole 32-channel chunk
// (DIP_WGRAD_TAIL_OLD=1: phase 2 of the fp32 kernel as before, for A/B runs)
extern "C" int dip_wgrad_tail_stream_ok(const DipWgradDesc* dp) {
    const DipWgradDesc& d = *dp;
    static const bool off = getenv("DIP_WGRAD_TAIL_OLD") != nullptr;
    const int tail = d.Cin & 31;
    return (!off && d.ks == 3 && d.stride == 1 && tail >= 1 && tail <= 4 && d.Cin > 32 && d.nsplit >= 1 && !(d.Cx & 3) && !(d.Cdy & 3) &&
            (long long)d.Hin * d.Win * d.Cx < (1ll << 31) && (long long)d.Hout * d.Wout * d.Cdy < (1ll << 31)) ? 1 : 0;
}

extern "C" int dip_wgrad_tail_stream(const DipWgradDesc* dp, void* stream) {
    const DipWgradDesc& d = *dp;
    if (!dip_wgrad_tail_stream_ok(dp)) DIP_FAIL("wgrad_tail_stream: needs a 3x3 stride-1 layer with a 1..4-channel tail behind whole 32-channel chunks");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int c_base = (d.Cin >> 5) << 5;
    // columns per workgroup: as many as still give >= 256 workgroups (DIP_WGRAD_TAIL_NCB overrides)
    static const int forced = [] { const char* e = getenv("DIP_WGRAD_TAIL_NCB"); return e ? atoi(e) : 0; }();
    const int CoutP = dip_round_up(d.Cout, 32);
    int ncb = 2;                                     // (4 columns per lane: the pipelined loads do not fit the register file)
    while (ncb > 1 && d.nsplit * dip_cdiv(CoutP, 32 * ncb) < 256) ncb >>= 1;
    if (forced == 1 || forced == 2) ncb = forced;
    return ncb == 2 ? wt_launch<2>(d, st, c_base) : wt_launch<1>(d, st, c_base);
}
