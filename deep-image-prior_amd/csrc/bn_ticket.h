// "The last workgroup to arrive finalises" -- WITHOUT an agent-scope fence.
// OPT-IN (DIP_TICKET_FIN=1), gfx950-ONLY BEHAVIOUR: the protocol below has no release/acquire pair -- it relies on a store's vmcnt
// retirement implying agent-scope visibility across the XCD L2s, which the HIP / LLVM memory model does not guarantee.
// Measured slower than the separate finalisation launch; kept for the record and covered by tests/test_small_gpu.py.
//
// A BatchNorm needs a reduction over the whole image between two element-wise passes; the producers of
// the partial rows (conv epilogues, up-sample + concat, the statistics passes of the backward) used to
// hand them to a finalisation launch of their own (dip_bn_finalize / dip_bn_bwd_finalize): 60 launches
// per iteration, each a ~3 us launch floor + two dependent global round trips in the serial chain.  For
// launches with few partial rows (the low-resolution scales) the finalisation rides in the producer:
//   * partial rows are written with write-through stores (relaxed agent-scope atomic stores = `sc1`),
//   * every thread waits for its own stores (s_waitcnt vmcnt(0): stores count in vmcnt on gfx9),
//   * one thread draws a ticket with a relaxed agent-scope atomic add,
//   * the workgroup that draws the last ticket reads all rows with L2-bypassing (sc1) loads, reduces
//     them in fp64 in a fixed order (deterministic) and writes the result with plain stores (the next
//     launch sees them), then puts the counter back to zero.
// __threadfence() instead would write back / invalidate the XCD's whole L2 per workgroup (+200 us on
// a 2048-workgroup conv); the protocol was validated on its own in tools/ubench/ticket_sc1.hip.
// The arithmetic is that of bn_finalize_kernel / bn_bwd_finalize_kernel (bn_kernels.hip).
#pragma once
#include "dip_common.h"

constexpr int DIP_TICKET_SH_DOUBLES = 256 * 12;      // LDS scratch of the finalisation trees

__device__ __forceinline__ void dip_st_sc1(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float dip_ld_sc1(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Call from EVERY thread of the workgroup after the partial-row stores were issued.
// True in every thread of the workgroup that drew the last of `nwg` tickets.
__device__ __forceinline__ bool dip_ticket_last(unsigned* ticket, int nwg, unsigned* flag_lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *flag_lds = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *flag_lds == (unsigned)(nwg - 1);
}
__device__ __forceinline__ void dip_ticket_reset(unsigned* ticket) {
    if (threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// thread layout of a finalisation: `ncg` 4-channel groups x `nrl` row lanes (256 threads)
struct DipFinLayout {
    int ncg, nrl, cg, rl;
    bool active;
};
__device__ __forceinline__ DipFinLayout dip_fin_layout(int nch) {
    DipFinLayout L;
    L.ncg = (nch + 3) >> 2;
    L.nrl = 256 / L.ncg;
    if (L.nrl < 1) L.nrl = 1;
    L.rl = threadIdx.x / L.ncg;
    L.cg = threadIdx.x - L.rl * L.ncg;
    L.active = L.rl < L.nrl;
    return L;
}

// fixed-order tree over the row lanes; NV doubles per thread at sh[tid * NV]; result at rl == 0
template <int NV>
__device__ __forceinline__ void dip_fin_tree(const DipFinLayout& L, double* sh) {
    for (int s = dip_pow2_ceil(L.nrl) >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (L.active && L.rl < s && L.rl + s < L.nrl) {
            double* mine = sh + (size_t)threadIdx.x * NV;
            const double* q = sh + (size_t)((L.rl + s) * L.ncg + L.cg) * NV;
#pragma unroll
            for (int k = 0; k < NV; ++k) mine[k] += q[k];
        }
    }
    __syncthreads();
}

// Forward: rows [nrows][3][Cstride] of {count, mean, M2} -> state block + running statistics of channels
// [c_first, c_first + nch) (nch <= 256: one thread per channel at the end; columns >= f.C are skipped).  Whole
// workgroup (256 threads).
__device__ __forceinline__ void dip_bn_fin_rows(const float* stats, int nrows, int Cstride, int c_first, int nch,
                                                const DipBnFin& f, double* sh) {
    const DipFinLayout L = dip_fin_layout(nch);
    const int cb = c_first + L.cg * 4;
    double K[4], acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    if (L.active) {
        // means are accumulated relative to the first row's mean K: M2 = S2 - N d^2 then cancels only d = mean - K
#pragma unroll
        for (int e = 0; e < 4; ++e) K[e] = (cb + e < Cstride) ? (double)dip_ld_sc1(stats + Cstride + cb + e) : 0.0;
#pragma unroll 4
        for (int row = L.rl; row < nrows; row += L.nrl) {
            const float* p = stats + (size_t)row * 3 * Cstride + cb;
            float n[4], m[4], q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = cb + e < Cstride;
                n[e] = ok ? dip_ld_sc1(p + e) : 0.f;
                m[e] = ok ? dip_ld_sc1(p + Cstride + e) : 0.f;
                q[e] = ok ? dip_ld_sc1(p + 2 * Cstride + e) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double ni = (double)n[e], mi = (double)m[e] - K[e];
                acc[e] += ni;
                acc[4 + e] += ni * mi;
                acc[8 + e] += (double)q[e] + ni * mi * mi;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) sh[(size_t)threadIdx.x * 12 + k] = acc[k];
    dip_fin_tree<12>(L, sh);
    if ((int)threadIdx.x < nch && c_first + (int)threadIdx.x < f.C) {
        const int c = c_first + threadIdx.x, e = threadIdx.x & 3;
        const double* r0 = sh + (size_t)(threadIdx.x >> 2) * 12;        // thread (rl 0, cg = tid / 4)
        const double N = r0[e], S1 = r0[4 + e], S2 = r0[8 + e];
        const double dmean = N > 0.0 ? S1 / N : 0.0;
        const double mean = (double)dip_ld_sc1(stats + Cstride + c) + dmean;
        double M2 = S2 - N * dmean * dmean;
        if (M2 < 0.0) M2 = 0.0;
        const double var = N > 0.0 ? M2 / N : 0.0;          // biased (normalisation)
        const float rstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float a = f.gamma[c] * rstd;
        const float fm = (float)mean;
        f.state[c] = fm;
        f.state[f.Cs + c] = rstd;
        f.state[2 * f.Cs + c] = a;
        f.state[3 * f.Cs + c] = f.beta[c] - fm * a;
        if (f.running_mean != nullptr) {
            const double unb = N > 1.0 ? M2 / (N - 1.0) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * fm;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
        }
    }
}

// Backward: rows [nrows][2][Cs] of {sum dz, sum dz * xhat} -> dgamma, dbeta, coef = {S1 / npix, S2 / npix} of channels
// [c_first, c_first + nch).
__device__ __forceinline__ void dip_bnb_fin_rows(const float* partials, int nrows, int Cs, int c_first, int nch,
                                                 const DipBnbFin& f, double* sh) {
    const DipFinLayout L = dip_fin_layout(nch);
    const int cb = c_first + L.cg * 4;
    double acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.0;
    if (L.active) {
#pragma unroll 4
        for (int row = L.rl; row < nrows; row += L.nrl) {
            const float* p = partials + (size_t)row * 2 * Cs + cb;
            float v1[4], v2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = cb + e < Cs;
                v1[e] = ok ? dip_ld_sc1(p + e) : 0.f;
                v2[e] = ok ? dip_ld_sc1(p + Cs + e) : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[e] += (double)v1[e]; acc[4 + e] += (double)v2[e]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[(size_t)threadIdx.x * 8 + k] = acc[k];
    dip_fin_tree<8>(L, sh);
    if ((int)threadIdx.x < nch && c_first + (int)threadIdx.x < f.C) {
        const int c = c_first + threadIdx.x, e = threadIdx.x & 3;
        const double* r0 = sh + (size_t)(threadIdx.x >> 2) * 8;
        const double s1 = r0[e], s2 = r0[4 + e];
        if (f.dbeta != nullptr) f.dbeta[c] = (float)s1;
        if (f.dgamma != nullptr) f.dgamma[c] = (float)s2;
        f.coef[c] = (float)(s1 / f.npix);
        f.coef[Cs + c] = (float)(s2 / f.npix);
    }
}
