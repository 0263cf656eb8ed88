// 3x3 stride-1 convolution towards <= 4 output channels (vector ALU, LDS-staged halo).
//
// Where it is used: the data gradient of the decoder convs flows into a 132-channel concat tensor
// = 4 skip channels + 128 up-sampled channels (reference models/skip.py:50-53, models/common.py:39).
// On the matrix core the 4 skip columns cost a whole 32-column block (the N = 160 variant of
// conv_igemm_kernel, or a 32-column launch with 28 idle columns); here they cost
// 2*9*Cin*4 FLOP per pixel on the VALU (2.4 GFLOP at 512x512, Cin = 128) and the other 128 columns
// go through conv_igemm_dma_kernel at its full rate.
//
//   y[p][n] = bias[n] + sum_tap sum_c x[src(p, tap)][c] * Wp[(tap*Cin/4 + c/4)*CoutP + n][c%4],  n < 4
//
// One thread = one output pixel of a 16x16 tile, 4 accumulators.  The 18x18-pixel halo of a
// 32-channel chunk is staged in LDS (pixel pitch 36 dwords: the 16 lanes of a ds_read_b128 pass
// fall on 16 distinct 16-byte slots); the weights of a (tap, 4-channel group) are 16 consecutive
// floats of the packed B operand; a chunk's 72 groups (4.6 KB) are staged in LDS next to the halo and
// read as broadcasts (through the scalar cache they missed on every group: 18 KB per workgroup walk
// thrashes it, 85 us per workgroup instead of ~25).
#include "dip_common.h"
#include "dip_group.h"

namespace {

constexpr int T4_TH = 16, T4_TW = 16, T4_HH = T4_TH + 2, T4_HW = T4_TW + 2, T4_NPIX = T4_HH * T4_HW;
constexpr int T4_PITCH = 36;                                   // dwords per halo pixel (32 channels + 4 pad)
constexpr int T4_SLOTS = (T4_NPIX * 8 + 255) / 256;            // 16-byte pieces per thread and chunk

__device__ __forceinline__ int t4_map_src(int v, int n_in, int reflect) {
    if (reflect) v = dip_reflect(v, n_in);
    return (v < 0 || v >= n_in) ? -1 : v;
}

template <bool GRP = false>
__global__ __launch_bounds__(256) void conv_thin4_kernel(const DipConvDesc d_, const int ntx, const int CoutP,
                                                        const int ncols, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    __shared__ __attribute__((aligned(16))) float halo[T4_NPIX * T4_PITCH];
    __shared__ int srcoff[T4_NPIX];
    __shared__ __attribute__((aligned(16))) float wsh[9 * 8 * 16];           // [tap][c4][n][c%4] of the current chunk
    const int tid = threadIdx.x;
    const int ty = blockIdx.x / ntx, tx = blockIdx.x - ty * ntx;
    for (int hp = tid; hp < T4_NPIX; hp += 256) {
        const int hr = hp / T4_HW, hc = hp - hr * T4_HW;
        const int sr = t4_map_src(ty * T4_TH + hr - d.off, d.Hin, d.pad_mode);
        const int sc = t4_map_src(tx * T4_TW + hc - d.off, d.Win, d.pad_mode);
        srcoff[hp] = (sr < 0 || sc < 0) ? -1 : (sr * d.Win + sc);
    }
    __syncthreads();
    int soff[T4_SLOTS];                     // this thread's staging pieces: source float offset or -1
#pragma unroll
    for (int i = 0; i < T4_SLOTS; ++i) {
        const int f = tid + i * 256;
        const int hp = f >> 3;
        const int so = (hp < T4_NPIX) ? srcoff[hp] : -2;
        soff[i] = so >= 0 ? so * d.Cx + (f & 7) * 4 : so;
    }
    const int py = tid >> 4, px = tid & 15;
    const float* hbase = halo + (py * T4_HW + px) * T4_PITCH;
    const int cin4 = d.Cin >> 2;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int cb = 0; cb < d.Cin; cb += 32) {
        const int c4n = min(8, (d.Cin - cb) >> 2);          // 4-channel groups in this chunk
        f32x4 st[T4_SLOTS];
#pragma unroll
        for (int i = 0; i < T4_SLOTS; ++i) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (soff[i] >= 0 && ((tid + i * 256) & 7) < c4n) v = *reinterpret_cast<const f32x4*>(d.x + soff[i] + cb);
            st[i] = v;
        }
        f32x4 wst[2];                       // this thread's pieces of the chunk's 288 weight quads
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = tid + i * 256;    // quad q = (tap*8 + c4)*4 + n
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q < 288) {
                const int grp = q >> 2, tap = grp >> 3, c4 = grp & 7;
                if (c4 < c4n) v = *reinterpret_cast<const f32x4*>(d.wp + ((size_t)(tap * cin4 + (cb >> 2) + c4) * CoutP + (q & 3)) * 4);
            }
            wst[i] = v;
        }
        __syncthreads();                    // previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < T4_SLOTS; ++i) {
            const int f = tid + i * 256;
            if (soff[i] != -2) *reinterpret_cast<f32x4*>(halo + (f >> 3) * T4_PITCH + (f & 7) * 4) = st[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * 256 < 288) *reinterpret_cast<f32x4*>(wsh + (tid + i * 256) * 4) = wst[i];
        __syncthreads();
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const float* hp = hbase + (ky * T4_HW + kx) * T4_PITCH;
            const float* wt = wsh + tap * 128;
#pragma unroll 4
            for (int c4 = 0; c4 < c4n; ++c4) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(hp + c4 * 4);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wt + c4 * 16 + n * 4);   // broadcast read
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[n] = fmaf(g[e], w[e], acc[n]);
                }
            }
        }
    }
    const int oy = ty * T4_TH + py, ox = tx * T4_TW + px;
    const bool inside = oy < d.Hout && ox < d.Wout;
    if (inside) {
        const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
        float* p = d.y + ((size_t)oy * pitch + ox) * d.Cy;
        if (d.bias != nullptr) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < ncols) acc[n] += d.bias[n];
        }
        if (ncols == 4) {
            if (d.accumulate) acc += *reinterpret_cast<const f32x4*>(p);
            *reinterpret_cast<f32x4*>(p) = acc;
        } else {
            for (int n = 0; n < ncols; ++n) {
                acc[n] = d.accumulate ? p[n] + acc[n] : acc[n];
                p[n] = acc[n];
            }
        }
    }
    // fused phase 1 of the BatchNorm backward of these columns (DipConvDesc.bnb_*, see conv_epilogue.h): one row of
    // {sum g*act'(z), sum g*act'(z)*xhat} per workgroup in bnb_partials_thin (channels 0..3)
    if (d.bnb_y != nullptr) {
        f32x4 s1 = f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
        if (inside) {
            const int pad = d.bnb_pad, Hi = d.Hout - 2 * pad, Wi = d.Wout - 2 * pad, Cs = d.bnb_Cs;
            const int iy = dip_reflect(oy - pad, Hi), ix = dip_reflect(ox - pad, Wi);
            const f32x4 yv = *reinterpret_cast<const f32x4*>(d.bnb_y + ((size_t)iy * Wi + ix) * d.bnb_Cy);
            const f32x4 mean = *reinterpret_cast<const f32x4*>(d.bnb_state), rstd = *reinterpret_cast<const f32x4*>(d.bnb_state + Cs),
                        sa = *reinterpret_cast<const f32x4*>(d.bnb_state + 2 * Cs), sb = *reinterpret_cast<const f32x4*>(d.bnb_state + 3 * Cs);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < ncols) {
                    const float z = fmaf(sa[n], yv[n], sb[n]);
                    const float gm = dip_mul_rn(acc[n], dip_act_grad(z, d.bnb_slope));
                    s1[n] = gm;
                    s2[n] = gm * ((yv[n] - mean[n]) * rstd[n]);
                }
            }
        }
        __syncthreads();                    // the halo is dead: reuse it for the tree (256 threads x 8 floats)
        dip_tree_sum8(halo, 1, 256, tid, 0, true, s1, s2);
        if (tid == 0) {
            float* o = d.bnb_partials_thin + (size_t)blockIdx.x * 2 * d.bnb_Cs;
            *reinterpret_cast<f32x4*>(o) = s1;
            *reinterpret_cast<f32x4*>(o + d.bnb_Cs) = s2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The matrix-pipe form (round 6).  The sum over taps is pulled out of the contraction:
//
//   Z[q][tap * 4 + n] = sum_c x[q][c] * W[tap][c][n]          one GEMM, M = source pixels, K = Cin, N = 36 (three 16-column tiles)
//   y[p][n]           = sum_tap Z[src(p, tap)][tap * 4 + n]   nine LDS reads per output
//
// so every input pixel is read ONCE, straight from global memory into the A operand of v_mfma_f32_16x16x4_f32 (a lane's
// float4 = four channels of its pixel = its k-slot of four K steps: the channel order inside the contraction is a
// permutation, applied to A and B alike), and all of B -- 3 tiles x Cin / 4 K steps, one register each -- stays in registers
// for the whole walk.  No operand passes through LDS and the workgroup never meets at a barrier: a WAVE owns a strip of 14
// output columns (16 source columns = one M tile per source row), walks down TH + 2 source rows, keeps the Z rows of the
// last three in a ring of its own in LDS and emits output row r - 2 after source row r.  fp32 products, fp32 accumulation.
// (The form above: 89 / 44 / 41 us for the 512^2 / 256^2 / 128^2 layers of the default net; its halo chunk loop is bound by
// LDS reads.  Knock-out of the three launches: - 118 us per iteration, profiles/r06_knockout.txt.)
constexpr int T4M_SW = 14;                   // output columns per wave
constexpr int T4M_ZP = 36;                   // floats per Z pixel
constexpr int T4M_ZROW = 16 * T4M_ZP;        // floats per Z row of a wave

template <int NJ, bool GRP = false>          // NJ = 16-channel groups (Cin <= 16 NJ; groups beyond Cin are zero)
__global__ __launch_bounds__(256) void conv_thin4_mfma_kernel(const DipConvDesc d_, const int nsx, const int TH, const int CoutP,
                                                             const int ncols, const DipGrpArg<GRP> grp) {
    DIP_GRP_DESC(DipConvDesc, d);
    __shared__ float zring_all[4 * 3 * T4M_ZROW];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p = lane & 15, g = lane >> 4;
    const int w = blockIdx.x * 4 + wv, tyi = w / nsx, sx = w - tyi * nsx;      // walks are numbered row by row of strips
    const int x0 = sx * T4M_SW, y0 = tyi * TH;
    if (y0 >= d.Hout) return;                                        // (no workgroup barrier anywhere below)
    const int rows = min(TH, d.Hout - y0);
    float* zring = zring_all + wv * 3 * T4M_ZROW;
    const int cin4 = d.Cin >> 2;

    // B: lane (p, g) holds column 16 t + p = (tap, n) of K slot g
    float bw[3][NJ][4];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int col = 16 * t + p, tap = col >> 2, n = col & 3;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (col < 36 && n < ncols && 16 * j + 4 * g < d.Cin)
                v = *reinterpret_cast<const f32x4*>(d.wp + ((size_t)(tap * cin4 + 4 * j + g) * CoutP + n) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) bw[t][j][i] = v[i];
        }
    }
    const int sc = t4_map_src(x0 + p - d.off, d.Win, d.pad_mode);     // this lane's source column
    // The A loads are inline asm, unconditional straight-line code (a clamped address; a padding pixel's value is replaced by
    // zero when the row is consumed), and a row is waited for with an explicit vmcnt(NJ) whose operands are its registers:
    // the row behind it stays in flight under its MFMAs (see waitA for why not two).  (Left to hipcc, a branch around a load or the conditional store
    // of the output phase makes it lose count and wait with vmcnt(0) before every row: no overlap at all, 85 us at 512 x 512.
    // Its own waits only see its own loads and stores; with ours in flight as well they wait longer than needed, never
    // shorter.)  Channel groups beyond Cin re-read group 0: their B registers are zero.
    const float* xcol = d.x + (size_t)max(sc, 0) * d.Cx + 4 * g;
    int coff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) coff[j] = (16 * j + 4 * g < d.Cin) ? 16 * j : -4 * g;
    const int nhr = rows + 2;
    auto loadA = [&](int hr, f32x4 (&a)[NJ], bool& ok) {              // (rows past the walk re-read its last row)
        const int sr = t4_map_src(y0 + min(hr, nhr - 1) - d.off, d.Hin, d.pad_mode);
        const float* q = xcol + (size_t)max(sr, 0) * d.Win * d.Cx;
        ok = sr >= 0 && sc >= 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(a[j]) : "v"(q + coff[j]));
    };
    auto waitA = [&](f32x4 (&a)[NJ], bool ok) {
        // vmcnt(NJ), not vmcnt(2 NJ): ONE row stays in flight behind the row being consumed.  With two (the count the issue
        // order gives: the loads of rows hr + 1 and hr + 2 are the only vector-memory loads younger than row hr's) 3 % of the
        // backward passes of a 256 x 256 net came out with wrong thin columns WHEN a chip-filling wgrad_bf3 launch ran beside
        // this kernel (DIP_DEFER_WGRAD=-1; tools/race_loop.py: 48 of 1500 passes, differences 1e-7 .. garbage), none without
        // one (3000 of 3000 bit-identical), none with vmcnt(0) (2000) and none with this form (4000).  Not explained -- the ISA
        // shows the expected order and tools/isa_inflight_check.py finds nothing -- so the margin is kept; it costs nothing measurable (53 us at 512^2, 28 us per launch on average, as before).
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(NJ));
#pragma unroll
        for (int j = 1; j < NJ; ++j) asm volatile("" : "+v"(a[j]));
        if (!ok) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) a[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    const int pxo = lane >> 2, no = lane & 3;                        // output phase: lane = (column of the strip, output channel)
    const int ox = x0 + pxo;
    const bool owrite = pxo < T4M_SW && ox < d.Wout && no < ncols;
    const int pitch = d.y_pitch > 0 ? d.y_pitch : d.Wout;
    const float bias = (d.bias != nullptr && no < ncols) ? d.bias[no] : 0.f;
    auto row = [&](int hr, f32x4 (&a)[NJ], bool ok) {
        waitA(a, ok);
        f32x4 acc[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], bw[t][j][i], acc[t], 0, 0, 0);
        // accumulator register r of lane (p, g) = Z[source column 4 g + r][column 16 t + p]
        float* zs = zring + (hr % 3) * T4M_ZROW;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (16 * t + p < 36) {
#pragma unroll
                for (int r = 0; r < 4; ++r) zs[(4 * g + r) * T4M_ZP + 16 * t + p] = acc[t][r];
            }
        // wave-local hand-over through LDS: the DS queue is in order; the explicit wait keeps the other lanes' reads behind
        // the partial-exec writes above (a wavefront-scope fence would do it too, but hipcc lowers it to vmcnt(0) as well,
        // which drains the two rows of loads in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (hr >= 2) {
            const int oy = y0 + hr - 2;
            float sum = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* zr = zring + ((hr - 2 + ky) % 3) * T4M_ZROW + pxo * T4M_ZP + no;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) sum += zr[kx * T4M_ZP + (ky * 3 + kx) * 4];
            }
            if (owrite) {
                float* o = d.y + ((size_t)oy * pitch + ox) * d.Cy + no;
                sum += bias;
                if (d.accumulate) sum += *o;
                *o = sum;
            }
            asm volatile("" ::: "memory");                            // the next row overwrites the oldest ring slot: its writes
        }                                                             // are issued after these reads have returned (sum is used)
    };
    f32x4 a0[NJ], a1[NJ], a2[NJ];
    bool ok0, ok1, ok2;
    loadA(0, a0, ok0);
    loadA(1, a1, ok1);
    for (int hr = 0; hr < nhr; hr += 3) {
        loadA(hr + 2, a2, ok2);
        row(hr, a0, ok0);
        if (hr + 1 < nhr) {
            loadA(hr + 3, a0, ok0);
            row(hr + 1, a1, ok1);
        }
        if (hr + 2 < nhr) {
            loadA(hr + 4, a1, ok1);
            row(hr + 2, a2, ok2);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");                               // (the rows loaded past the walk)
}

// Rows per walk.  A walk costs (TH + 2) rows of 96 MFMAs and the chip has 1024 SIMDs: with k walks per SIMD the launch lasts
// k (TH_k + 2) rows, TH_k = the smallest height whose walks number <= 1024 k.  k = 1 .. 4 are compared (DIP_THIN4_TH overrides).
int t4m_th(int Hout, int Wout) {
    static const int forced = getenv("DIP_THIN4_TH") ? atoi(getenv("DIP_THIN4_TH")) : 0;
    if (forced > 0) return forced;
    const int strips = dip_cdiv(Wout, T4M_SW);
    int best = 0, best_cost = 1 << 30;
    for (int k = 1; k <= 4; ++k) {
        const int ntile = 1024 * k / strips;
        if (ntile < 1) continue;
        const int th = dip_cdiv(Hout, min(ntile, Hout)), cost = k * (th + 2);
        if (cost < best_cost) { best = th; best_cost = cost; }
    }
    return best > 0 ? best : Hout;
}

}  // namespace

extern "C" int dip_conv_thin4_ntiles(int Hout, int Wout) { return dip_cdiv(Wout, T4_TW) * dip_cdiv(Hout, T4_TH); }

// columns [0, ncols) (ncols <= 4) of a 3x3 stride-1, undilated, transform-free convolution
extern "C" int dip_conv_thin4(const DipConvDesc* dp, int ncols, void* stream) {
    const DipConvDesc& d = *dp;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.ks != 3 || d.stride != 1 || d.dil != 1 || d.tr.a != nullptr || ncols < 1 || ncols > 4 || (d.Cin & 3))
        DIP_FAIL("conv_thin4: unsupported configuration");
    if (d.bnb_y != nullptr && (d.bnb_partials_thin == nullptr || (d.bnb_Cy & 3) || (d.bnb_Cs & 3)))
        DIP_FAIL("conv_thin4: fused BatchNorm-backward partials need bnb_partials_thin and 4-aligned strides");
    static const bool mfma_form = getenv("DIP_THIN4_VALU") == nullptr;
    if (mfma_form && d.bnb_y == nullptr && d.Cin <= 128) {
        const int th = t4m_th(d.Hout, d.Wout), nsx = dip_cdiv(d.Wout, T4M_SW), nwg = dip_cdiv(nsx * dip_cdiv(d.Hout, th), 4);
        const int nj = d.Cin <= 16 ? 1 : d.Cin <= 32 ? 2 : d.Cin <= 64 ? 4 : 8, CoutP = dip_round_up(d.Cout, 32);
#define T4M_LAUNCH(NJ_)                                                                                                            \
    dip_launch_pair<DIP_FAM_THIN>(conv_thin4_mfma_kernel<NJ_, false>, conv_thin4_mfma_kernel<NJ_, true>, dim3(nwg), dim3(256),     \
                                  0, st, d, nsx, th, CoutP, ncols)
        if (nj == 1) T4M_LAUNCH(1);
        else if (nj == 2) T4M_LAUNCH(2);
        else if (nj == 4) T4M_LAUNCH(4);
        else T4M_LAUNCH(8);
#undef T4M_LAUNCH
        DIP_CHECK_LAUNCH();
        return 0;
    }
    const int ntx = dip_cdiv(d.Wout, T4_TW), nty = dip_cdiv(d.Hout, T4_TH);
    dip_launch_pair<DIP_FAM_THIN>(conv_thin4_kernel<false>, conv_thin4_kernel<true>, dim3(ntx * nty), dim3(256), 0, st, d, ntx,
                                  dip_round_up(d.Cout, 32), ncols);
    DIP_CHECK_LAUNCH();
    return 0;
}
