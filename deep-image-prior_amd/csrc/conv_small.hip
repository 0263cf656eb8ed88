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
//   * epilogue: bias, store, BatchNorm partial statistics {count, mean, M2} per workgroup row, and the
//     LAST workgroup of a 32-channel block to arrive (fence-free ticket: write-through sc1 stores,
//     s_waitcnt, relaxed agent-scope atomic add; tools/ubench/ticket_sc1.hip) reduces the <= 128 rows
//     in fp64 and writes the BatchNorm state + running statistics -- what dip_bn_finalize does in a
//     launch of its own;
//   * data-gradient launches: phase 1 of the BatchNorm(+activation) backward of the conv's input
//     (DipConvDesc.bnb_*) rides in the epilogue the same way, finalised (dgamma, dbeta, k1, k2) by the
//     last arriver; stride-2 transposed convs walk the four output-parity classes separately, so a
//     wave only visits the taps that hit non-zero positions of the dilated gradient (9 taps per 4
//     pixels instead of 36, as the phase mode of the LDS-DMA kernel).
// Any stride-1/2 1x1 / 3x3 convolution or data gradient with reflection / zero / replication padding,
// any activation, 1..160 output channels; the engine uses it below DIP_SMALL_MAX_PIXELS output pixels.
#include "dip_common.h"
#include "bn_ticket.h"
#include <stdlib.h>

namespace {

constexpr int SM_NB = 4;            // K steps (8 input channels of one tap = 4 MFMAs) per register batch
constexpr int SM_TR_MAX = 512;      // input channels whose BatchNorm coefficients are cached in LDS
constexpr int SM_MAX_TAPS = 9;

struct SmallGeom {
    int ncls;                 // 1, or 4 output-parity classes (dil == 2: transposed stride-2 conv)
    int Hc[4], Wc[4];         // pixels of class c = (py, px): rows py, py+2, ..; columns px, px+2, ..
    int gstart[5];            // first 32-pixel group of each class; gstart[ncls] = ngroups
    int ngroups;
    int nblk;                 // 32-column blocks of the output
    int ksl;                  // K slices per tile = waves of a workgroup that share a tile (1, 2, 4)
    int gpw;                  // pixel groups per workgroup = 4 / ksl
    int nwg_x;                // ceil(ngroups / gpw) = rows of the partial-statistics buffers
};

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

// TR: 0 = no input transform, 1 = BatchNorm + LeakyReLU / identity (slope in (0, 1]), 2 = BatchNorm + Swish / ELU
template <int TR>
__global__ __launch_bounds__(256) void conv_small_kernel(const DipConvDesc d, const SmallGeom g) {
    __shared__ __attribute__((aligned(16))) float tra[TR ? SM_TR_MAX : 4];
    __shared__ __attribute__((aligned(16))) float trb[TR ? SM_TR_MAX : 4];
    __shared__ int srcoff[4][SM_MAX_TAPS][32];
    __shared__ int yoff[4][32];                 // output pixel offset (oy * pitch + ox) of the wave's 32 pixels, -1: none
    __shared__ int boff[4][32];                 // mirror pixel of the fused BatchNorm backward (bnb_y index), -1: none
    __shared__ int taps[4][12];
    __shared__ unsigned flag;
    // K-slice reduction [4 waves][16 regs][64 lanes] floats (16 KB); later the fp64 trees of the finalisations
    __shared__ __attribute__((aligned(16))) double red_d[DIP_TICKET_SH_DOUBLES];
    float* red = reinterpret_cast<float*>(red_d);

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = w % g.ksl, gsub = w / g.ksl;
    const int group = blockIdx.x * g.gpw + gsub;
    const bool gvalid = group < g.ngroups;
    const int n0 = blockIdx.y * 32;
    const int KK = d.ks * d.ks;
    const int CoutP = g.nblk * 32;

    // ---- the wave's 32 output pixels ------------------------------------------------------------
    int Hc = g.Hc[0], Wc = g.Wc[0], gs = 0, py = 0, px = 0, pstep = 1;
    if (g.ncls > 1) {
        pstep = 2;
        if (group >= g.gstart[3]) { Hc = g.Hc[3]; Wc = g.Wc[3]; gs = g.gstart[3]; py = 1; px = 1; }
        else if (group >= g.gstart[2]) { Hc = g.Hc[2]; Wc = g.Wc[2]; gs = g.gstart[2]; py = 1; px = 0; }
        else if (group >= g.gstart[1]) { Hc = g.Hc[1]; Wc = g.Wc[1]; gs = g.gstart[1]; py = 0; px = 1; }
    }
    const int pi = (group - gs) * 32 + l31;
    const bool pvalid = gvalid && pi < Hc * Wc;
    const int iy = pvalid ? pi / Wc : 0, ix = pvalid ? pi - iy * Wc : 0;
    const int oy = iy * pstep + py, ox = ix * pstep + px;
    int ntv = 0;
    for (int t = 0; t < KK; ++t) {
        const int ky = t / d.ks, kx = t - ky * d.ks;
        const int sr = sm_map_src(oy * d.stride + ky - d.off, d.Hin, d.dil, d.pad_mode);
        const int sc = sm_map_src(ox * d.stride + kx - d.off, d.Win, d.dil, d.pad_mode);
        const int so = (pvalid && sr >= 0 && sc >= 0) ? sr * d.Win + sc : -1;
        if (half == 0) srcoff[w][t][l31] = so;
        if (__ballot(so >= 0) != 0ull) {        // wave-uniform: a tap no pixel of the wave can see is skipped
            if (lane == 0) taps[w][ntv] = t;
            ++ntv;
        }
    }
    if (half == 0) {
        const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
        yoff[w][l31] = pvalid ? oy * pitch + ox : -1;
        int bo = -1;
        if (d.bnb_y != nullptr && pvalid) {
            const int Hi = d.Hout - 2 * d.bnb_pad, Wi = d.Wout - 2 * d.bnb_pad;
            bo = dip_reflect(oy - d.bnb_pad, Hi) * Wi + dip_reflect(ox - d.bnb_pad, Wi);
        }
        boff[w][l31] = bo;
    }
    if (TR) {
        for (int c = tid; c < d.Cin; c += 256) { tra[c] = d.tr.a[c]; trb[c] = d.tr.b[c]; }
    }
    __syncthreads();

    // ---- K loop: steps t in [t0, t1), t = (valid tap index, 8-channel group) ----------------------
    const int nq = (d.Cin + 7) >> 3;
    const int T = ntv * nq;
    const int t0 = (int)(((long long)T * slice) / g.ksl), t1 = (int)(((long long)T * (slice + 1)) / g.ksl);
    const int cin4 = d.Cin >> 2;
    const float slope = d.tr.slope;
    const float* wcol = d.wp + (size_t)(n0 + l31) * 4;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // Ring of three register batches: while batch b is multiplied, the loads of b+1 have (mostly) landed and those of
    // b+2 are being issued.  The loads are inline asm and the waits explicit (s_waitcnt vmcnt(16) = "all but the two
    // younger batches"; loads retire in order): left to hipcc, the s_waitcnt pass merges to vmcnt(0) at the loop head
    // (3x unrolled body) or waits for the loads it has just issued (rotating registers), and the ring collapses.
    // The wait statements name the batch's registers as in/out operands, so no use can be scheduled above them.
    struct Batch { f32x4 A[SM_NB], B[SM_NB]; int M[SM_NB]; };
    Batch P0 = {}, P1 = {}, P2 = {};
    int lt = t0, lti = t0 / nq, lj = t0 - lti * nq;      // next step to load
    auto load_batch = [&](Batch& P) {
#pragma unroll
        for (int k = 0; k < SM_NB; ++k) {
            const bool on = lt < t1;                                   // wave-uniform
            const int tap = taps[w][on ? lti : 0];
            const int so = srcoff[w][tap][l31];
            const int c = 8 * lj + 4 * half;
            const bool cv = on && c < d.Cin;                           // (the 4-channel tail of a 132-channel input)
            const bool av = cv && so >= 0;
            // unconditional loads from clamped addresses + selects: no control flow around a load; steps past t1
            // (the ring runs two batches ahead) contribute zeros
            const float* pa = d.x + (size_t)(av ? so : 0) * d.Cx + (cv ? c : 0);
            const float* pb = wcol + (size_t)((on ? tap : 0) * cin4 + (cv ? (c >> 2) : 0)) * CoutP * 4;
            // ("+v": the destination is tied to the batch's previous value, so the register allocator keeps ONE physical
            // register per slot around the loop -- with a plain output it renamed the slot and copied the registers of
            // loads still in flight at the back-edge)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(P.A[k]) : "v"(pa));
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(P.B[k]) : "v"(pb));
            P.M[k] = av ? c : -1;
            ++lt;
            if (++lj == nq) { lj = 0; ++lti; }
        }
    };
    static_assert(SM_NB == 4, "the wait statements list 8 registers per batch");
#define SM_WAIT(P, CNT)                                                                                           \
    asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                      \
                 : "+v"(P.A[0]), "+v"(P.A[1]), "+v"(P.A[2]), "+v"(P.A[3]), "+v"(P.B[0]), "+v"(P.B[1]), "+v"(P.B[2]), \
                   "+v"(P.B[3]))
    auto comp_batch = [&](const Batch& P) {
#pragma unroll
        for (int k = 0; k < SM_NB; ++k) {
            f32x4 a = P.A[k];
            const int m = P.M[k];
            if (TR) {
                const int mc = m < 0 ? 0 : m;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(tra + mc), b4 = *reinterpret_cast<const f32x4*>(trb + mc);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float tv = fmaf(a4[e], a[e], b4[e]);
                    a[e] = TR == 1 ? dip_act_leaky(tv, slope) : dip_act(tv, slope);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = m < 0 ? 0.f : a[e];      // padding zeros come AFTER the activation
            const f32x4 b = P.B[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
    };
    const int nb = (t1 - t0 + SM_NB - 1) / SM_NB;
    if (nb > 0) {
        load_batch(P0);
        load_batch(P1);
        for (int b = 0; b < nb; b += 3) {
            load_batch(P2);
            SM_WAIT(P0, 16);
            comp_batch(P0);
            load_batch(P0);
            if (b + 1 < nb) {
                SM_WAIT(P1, 16);
                comp_batch(P1);
            }
            load_batch(P1);
            if (b + 2 < nb) {
                SM_WAIT(P2, 16);
                comp_batch(P2);
            }
        }
        // drain the loads that ran ahead: their destination registers are dead to the compiler from here on
        SM_WAIT(P0, 0);
        SM_WAIT(P1, 0);
        SM_WAIT(P2, 0);
    }
#undef SM_WAIT

    // ---- split-K inside the workgroup: slice 0 adds slices 1.. in order -----------------------------
    if (g.ksl > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(w * 16 + r) * 64 + lane] = acc[r];
        __syncthreads();
        if (slice == 0) {
            for (int s = 1; s < g.ksl; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += red[((w + s) * 16 + r) * 64 + lane];
        }
    }
    const bool fin_wave = slice == 0 && gvalid;       // holds a finished 32 x 32 tile
    const int n = n0 + l31;

    // ---- bias, (accumulate), store ---------------------------------------------------------------------
    int yo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) yo[r] = yoff[w][(r & 3) + 8 * (r >> 2) + 4 * half];
    if (fin_wave) {
        const float bias = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
        const bool ncol = n < d.Cy;
        if (d.accumulate) {
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
        float s1 = 0.f, s2 = 0.f;
        if (fin_wave) {
            const float mean = d.bnb_state[nn], rstd = d.bnb_state[Cs + nn], sa = d.bnb_state[2 * Cs + nn],
                        sb = d.bnb_state[3 * Cs + nn];
            float yv[16];
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int bo = boff[w][(r & 3) + 8 * (r >> 2) + 4 * half];
                ok[r] = nv && bo >= 0;
                yv[r] = d.bnb_y[(size_t)(bo < 0 ? 0 : bo) * d.bnb_Cy + nn];
            }
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
        }
        __syncthreads();                       // (`red` is reused)
        if (half == 0) { red[(w * 32 + l31) * 2] = s1; red[(w * 32 + l31) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < 32) {
            float t1s = 0.f, t2s = 0.f;
            for (int q = 0; q < g.gpw; ++q) {           // waves q * ksl hold the tiles of this workgroup
                t1s += red[((q * g.ksl) * 32 + tid) * 2];
                t2s += red[((q * g.ksl) * 32 + tid) * 2 + 1];
            }
            if (n0 + tid < Cs) {
                float* o = d.bnb_partials + (size_t)blockIdx.x * 2 * Cs + n0 + tid;
                dip_st_sc1(o, t1s);
                dip_st_sc1(o + Cs, t2s);
            }
        }
        if (d.bnb_fin.coef != nullptr) {
            if (dip_ticket_last(d.bnb_fin.ticket + blockIdx.y, g.nwg_x, &flag)) {
                dip_bnb_fin_rows(d.bnb_partials, g.nwg_x, Cs, n0, 32, d.bnb_fin, red_d);
                dip_ticket_reset(d.bnb_fin.ticket + blockIdx.y);
            }
        }
        return;
    }

    // ---- BatchNorm partial statistics of the consumer BatchNorm (forward launches) ----------------------
    if (d.stats == nullptr) return;
    float cn = 0.f, mean = 0.f, M2 = 0.f;
    if (fin_wave) {
        float k = 0.f, s1 = 0.f, s2 = 0.f;
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
        mean = cn > 0.f ? k + s1 / cn : 0.f;
        M2 = cn > 0.f ? s2 - s1 * s1 / cn : 0.f;
        const float on = __shfl_xor(cn, 32), om = __shfl_xor(mean, 32), oM = __shfl_xor(M2, 32);
        if (half == 0) dip_chan(cn, mean, M2, on, om, oM);      // (lower half holds rows 0..3, 8..11, ..: fixed order)
    }
    __syncthreads();                           // (`red` is reused)
    if (half == 0) { float* q = red + (w * 32 + l31) * 3; q[0] = cn; q[1] = mean; q[2] = M2; }
    __syncthreads();
    if (tid < 32) {
        float c2 = 0.f, m2 = 0.f, q2 = 0.f;
        for (int q = 0; q < g.gpw; ++q) {
            const float* p = red + ((q * g.ksl) * 32 + tid) * 3;
            dip_chan(c2, m2, q2, p[0], p[1], p[2]);
        }
        float* o = d.stats + (size_t)blockIdx.x * 3 * CoutP + n0 + tid;
        dip_st_sc1(o, c2);
        dip_st_sc1(o + CoutP, m2);
        dip_st_sc1(o + 2 * CoutP, q2);
    }
    if (d.fin.state == nullptr) return;
    if (dip_ticket_last(d.fin.ticket + blockIdx.y, g.nwg_x, &flag)) {
        dip_bn_fin_rows(d.stats, g.nwg_x, CoutP, n0, 32, d.fin, red_d);
        dip_ticket_reset(d.fin.ticket + blockIdx.y);
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
    if (d.ks != 1 && d.ks != 3) return false;
    if (d.stride != 1 && d.stride != 2) return false;
    if (d.dil != 1 && d.dil != 2) return false;
    if (d.dil == 2 && d.stride != 1) return false;
    if ((d.Cin & 3) || (d.Cx & 3) || (d.Cy & 3) || d.Cin <= 0 || d.Cin > d.Cx) return false;
    if (d.tr.a != nullptr && d.Cin > SM_TR_MAX) return false;
    if (d.Cout < 1 || d.Cout > 160 || d.Cy < d.Cout) return false;
    if (d.Hout < 1 || d.Wout < 1) return false;
    g->ncls = d.dil == 2 ? 4 : 1;
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
    // K slices: enough waves for ~2 per SIMD (2048), at least 8 K-steps per slice
    const int ksteps = d.ks * d.ks * ((d.Cin + 7) / 8) / (g->ncls == 4 ? 2 : 1);      // (a parity class sees ~9/4 taps)
    int ksl = 1;
    while (ksl < 4 && g->ngroups * g->nblk * ksl < 2048 && ksteps / (2 * ksl) >= 8) ksl *= 2;
    static const char* force = getenv("DIP_SMALL_KSL");
    if (force && (atoi(force) == 1 || atoi(force) == 2 || atoi(force) == 4)) ksl = atoi(force);
    g->ksl = ksl;
    g->gpw = 4 / ksl;
    g->nwg_x = dip_cdiv(g->ngroups, g->gpw);
    return true;
}

}  // namespace

// 1 when the engine should run `d` through dip_conv_small: a shape the kernel serves and a small output
extern "C" int dip_conv_small_eligible(const DipConvDesc* dp) {
    static const bool off = getenv("DIP_CONV_NO_SMALL") != nullptr;
    SmallGeom g;
    if (off || !small_geom(*dp, &g)) return 0;
    return dp->Hout * dp->Wout <= small_max_pixels() ? 1 : 0;
}

// rows of the partial buffers (stats: [rows][3][CoutP32]; bnb_partials: [rows][2][bnb_Cs]) of dip_conv_small(d)
extern "C" int dip_conv_small_rows(const DipConvDesc* dp) {
    SmallGeom g;
    if (!small_geom(*dp, &g)) return 0;
    return g.nwg_x;
}

extern "C" int dip_conv_small(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    SmallGeom g;
    if (!small_geom(d, &g)) DIP_FAIL("conv_small: unsupported shape (1x1 / 3x3, stride or dil <= 2, <= 160 output channels)");
    if (d.bnb_y != nullptr) {
        if (d.bnb_state == nullptr || d.bnb_partials == nullptr || d.bnb_Cs < d.Cout || d.bnb_pad < 0 ||
            d.Hout <= 2 * d.bnb_pad || d.Wout <= 2 * d.bnb_pad || d.stats != nullptr)
            DIP_FAIL("conv_small: inconsistent bnb_* fields");
        if (d.bnb_fin.coef != nullptr && d.bnb_fin.ticket == nullptr) DIP_FAIL("conv_small: bnb_fin needs a ticket buffer");
    }
    if (d.fin.state != nullptr && (d.stats == nullptr || d.fin.ticket == nullptr || d.fin.gamma == nullptr ||
                                   d.fin.beta == nullptr || d.fin.C > d.Cout))
        DIP_FAIL("conv_small: fin needs stats, ticket, gamma, beta");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(g.nwg_x, g.nblk);
    if (d.tr.a == nullptr) hipLaunchKernelGGL(conv_small_kernel<0>, grid, dim3(256), 0, st, d, g);
    else if (d.tr.slope > 0.f) hipLaunchKernelGGL(conv_small_kernel<1>, grid, dim3(256), 0, st, d, g);
    else hipLaunchKernelGGL(conv_small_kernel<2>, grid, dim3(256), 0, st, d, g);
    DIP_CHECK_LAUNCH();
    return 0;
}
