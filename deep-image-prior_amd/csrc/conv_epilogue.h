// Shared epilogue of the implicit-GEMM convolution kernels (conv_igemm.hip, conv_igemm_dma.hip).
//
// A wave holds MS x NS accumulators of the 32x32x2 fp32 MFMA; accumulator (ms, ns) covers output
// rows 2*sub, 2*sub+1 (sub = wm*MS + ms) x 16 columns of the 8x16-pixel tile and 32 output
// channels:  register r of lane (l31, half) is pixel (row r>>3, col (r&3) + 8*((r>>2)&1) + 4*half),
// channel l31.  A store instruction therefore writes two 128-byte runs.
//
// The first version interleaved bias load, optional read-modify-write and the BatchNorm partial
// statistics per element; hipcc guarded every element with s_waitcnt vmcnt(0), which serialised
// the 64 stores of a wave into 64 memory round trips (measured with s_memtime: 25.6k cycles per
// tile alone on the chip, 58-66k under load = 16 % of a workgroup's lifetime).  Here all loads of
// a block are issued before its first store, stores are never waited for, and the statistics are a
// separate register pass.
#pragma once
#include "dip_common.h"

struct DipEpi {
    float* yt;        // &y[tile origin]
    int row_stride;   // pitch * Cy   (floats between output rows)
    int Cy;           // floats between output pixels
    int ncols;        // channels stored per pixel (columns n < ncols)
    int rows_left;    // Hout - tile row origin
    int cols_left;    // Wout - tile col origin
    bool full;        // whole 8x16 tile inside the image (workgroup-uniform)
    bool accumulate;
    int oy0, ox0;     // tile origin in the output domain
};

__device__ __forceinline__ DipEpi dip_epi_make(const DipConvDesc& d, int ty, int tx, int TH, int TW) {
    DipEpi e;
    const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
    e.yt = d.y + ((size_t)(ty * TH) * pitch + (size_t)tx * TW) * d.Cy;
    e.row_stride = pitch * d.Cy;
    e.Cy = d.Cy;
    e.ncols = d.Cy;
    e.rows_left = d.Hout - ty * TH;
    e.cols_left = d.Wout - tx * TW;
    e.full = (e.rows_left >= TH) && (e.cols_left >= TW);
    e.accumulate = d.accumulate != 0;
    e.oy0 = ty * TH;
    e.ox0 = tx * TW;
    return e;
}

__device__ __forceinline__ bool dip_epi_valid(const DipEpi& e, int sub, int r, int half) {
    const int row = 2 * sub + (r >> 3), col = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
    return e.full || (row < e.rows_left && col < e.cols_left);
}

// a[r] <- a[r] + bias (+ previous y when accumulating); stores it for channel n < Cy.
__device__ __forceinline__ void dip_epi_store16(const DipEpi& e, f32x16& a, int sub, int n, float bias, int half) {
    const bool ncol = n < e.ncols;
    float* p0 = e.yt + (2 * sub) * e.row_stride + (4 * half) * e.Cy + n;
    if (e.accumulate) {
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = (r >> 3) * e.row_stride + ((r & 3) + 8 * ((r >> 2) & 1)) * e.Cy;
            old[r] = (ncol && dip_epi_valid(e, sub, r, half)) ? p0[off] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += bias + old[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] += bias;
    }
    if (ncol) {
        if (e.full) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p0[(r >> 3) * e.row_stride + ((r & 3) + 8 * ((r >> 2) & 1)) * e.Cy] = a[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (dip_epi_valid(e, sub, r, half))
                    p0[(r >> 3) * e.row_stride + ((r & 3) + 8 * ((r >> 2) & 1)) * e.Cy] = a[r];
        }
    }
}

// shifted sums (shift = first valid value): cancellation-free single-pass variance
__device__ __forceinline__ void dip_epi_stats16(const DipEpi& e, const f32x16& a, int sub, int half, float& cnt,
                                                float& k, float& s1, float& s2) {
    if (e.full) {
        if (cnt == 0.f) k = a[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dv = a[r] - k;
            s1 += dv;
            s2 = fmaf(dv, dv, s2);
        }
        cnt += 16.f;
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (dip_epi_valid(e, sub, r, half)) {
                if (cnt == 0.f) k = a[r];
                const float dv = a[r] - k;
                cnt += 1.f;
                s1 += dv;
                s2 = fmaf(dv, dv, s2);
            }
        }
    }
}

// LDS-only workgroup barrier: unlike __syncthreads() it does not drain vmcnt, so the output stores
// issued just before stay in flight.
__device__ __forceinline__ void dip_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fused phase 1 of a BatchNorm(+activation) backward (DipConvDesc.bnb_*): the tile's values (already stored) are the
// gradient g wrt the activated output on a domain padded by bnb_pad; per channel  s1 = sum g*act'(z),
// s2 = sum g*act'(z)*xhat  with y taken at the mirror pixel of a ring position (the reflection fold is linear).
// Lane = channel, so the 16 y loads of an accumulator are the same two 128-byte runs per instruction as its stores.
template <class C, int BN>
__device__ __forceinline__ void dip_conv_epilogue_bnb(const DipConvDesc& d, f32x16 (&acc)[C::MS][C::NS], const DipEpi& e,
                                                      int n0, int wn, int wm, int l31, int half, int tid, int tile,
                                                      float* red) {
    const int pad = d.bnb_pad;
    const int Hi = d.Hout - 2 * pad, Wi = d.Wout - 2 * pad;
    const int Cs = d.bnb_Cs;
    dip_lds_barrier();                     // every wave is done reading the staging buffers (`red` aliases them)
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) {
        const int n = n0 + (wn * C::NS + ns) * 32 + l31;
        const bool nv = n < d.Cout;
        const int nn = nv ? n : 0;
        const float mean = d.bnb_state[nn], rstd = d.bnb_state[Cs + nn], sa = d.bnb_state[2 * Cs + nn],
                    sb = d.bnb_state[3 * Cs + nn];
        float s1 = 0.f, s2 = 0.f;
        // all loads of this column block (MS accumulators x 16 pixels) are issued before the first use: the K loop's
        // registers are dead by now, and one memory round trip per accumulator was measured at +100 us on the 256^2
        // data gradients of a slow-class box (a tile's epilogue is on the critical path of its workgroup slot)
        float yv[C::MS][16];
        bool ok[C::MS][16];
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) {
            const int sub = wm * C::MS + ms;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ok[ms][r] = nv && dip_epi_valid(e, sub, r, half);
                const int oy = e.oy0 + 2 * sub + (r >> 3), ox = e.ox0 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
                const int iy = ok[ms][r] ? dip_reflect(oy - pad, Hi) : 0, ix = ok[ms][r] ? dip_reflect(ox - pad, Wi) : 0;
                yv[ms][r] = d.bnb_y[((size_t)iy * Wi + ix) * d.bnb_Cy + nn];
            }
        }
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = fmaf(sa, yv[ms][r], sb);
                const float gm = ok[ms][r] ? dip_mul_rn(acc[ms][ns][r], dip_act_grad(z, d.bnb_slope)) : 0.f;
                const float xh = (yv[ms][r] - mean) * rstd;
                s1 += gm;
                s2 = fmaf(gm, xh, s2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);          // one column block at a time (register budget)
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (half == 0) {
            float* q = red + ((wm * (C::WN * C::NS * 32)) + (wn * C::NS + ns) * 32 + l31) * 2;
            q[0] = s1; q[1] = s2;
        }
    }
    dip_lds_barrier();
    if (tid < BN) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < C::WM; ++w) {
            const float* q = red + (w * (C::WN * C::NS * 32) + tid) * 2;
            s1 += q[0]; s2 += q[1];
        }
        const int n = n0 + tid;
        if (n < Cs) {
            float* o = d.bnb_partials + (size_t)tile * 2 * Cs + n;
            o[0] = s1; o[Cs] = s2;
        }
    }
}

// Whole-tile epilogue for a wave's acc[MS][NS]; C supplies TH, TW, MS, NS, WN, WM.  `red` is LDS
// scratch of WM * WN*NS*32 * 3 floats (the staging buffers, dead by now).
template <class C, int BN>
__device__ __forceinline__ void dip_conv_epilogue(const DipConvDesc& d, f32x16 (&acc)[C::MS][C::NS], const DipEpi& e,
                                                  int n0, int wn, int wm, int l31, int half, int tid, int tile,
                                                  int CoutP, float* red) {
    float bias[C::NS];
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) {
        const int n = n0 + (wn * C::NS + ns) * 32 + l31;
        bias[ns] = (d.bias != nullptr && n < d.Cout) ? d.bias[n] : 0.f;
    }
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) {
        const int n = n0 + (wn * C::NS + ns) * 32 + l31;
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) dip_epi_store16(e, acc[ms][ns], wm * C::MS + ms, n, bias[ns], half);
    }
    if (d.bnb_y != nullptr) {
        dip_conv_epilogue_bnb<C, BN>(d, acc, e, n0, wn, wm, l31, half, tid, tile, red);
        if (d.stats == nullptr) return;
        dip_lds_barrier();                 // (`red` is about to be reused)
    }
    if (d.stats == nullptr) return;
    dip_lds_barrier();                     // every wave is done reading the staging buffers
#pragma unroll
    for (int ns = 0; ns < C::NS; ++ns) {
        float cn = 0.f, k = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ms = 0; ms < C::MS; ++ms) dip_epi_stats16(e, acc[ms][ns], wm * C::MS + ms, half, cn, k, s1, s2);
        float mean = cn > 0.f ? k + s1 / cn : 0.f;
        float M2 = cn > 0.f ? s2 - s1 * s1 / cn : 0.f;
        const float on = __shfl_xor(cn, 32), om = __shfl_xor(mean, 32), oM = __shfl_xor(M2, 32);
        dip_chan(cn, mean, M2, on, om, oM);
        if (half == 0) {
            float* q = red + ((wm * (C::WN * C::NS * 32)) + (wn * C::NS + ns) * 32 + l31) * 3;
            q[0] = cn; q[1] = mean; q[2] = M2;
        }
    }
    dip_lds_barrier();
    if (tid < BN) {
        float cn = 0.f, mean = 0.f, M2 = 0.f;
#pragma unroll
        for (int w = 0; w < C::WM; ++w) {
            const float* q = red + (w * (C::WN * C::NS * 32) + tid) * 3;
            dip_chan(cn, mean, M2, q[0], q[1], q[2]);
        }
        const int n = n0 + tid;
        if (n < CoutP) {
            float* o = d.stats + (size_t)tile * 3 * CoutP + n;
            o[0] = cn; o[CoutP] = mean; o[2 * CoutP] = M2;
        }
    }
}
