// 1x1 convolution 128 -> 128 (the need1x1_up convs of models/skip.py:88-91 at the high-resolution scales, forward and
// data gradient), weights-resident persistent variant.
//
// Arithmetic intensity is ~32 FLOP/B: the launch sits on the ridge of the MFMA and HBM roofs (8.6 GFLOP and 268 MB at
// 512^2: 55 us of matrix pipe, 43 us of HBM at the measured copy ceiling).  conv_igemm_dma_kernel<1, 128> runs it at
// 106 us: one (32-channel chunk) unit = 64 MFMAs per wave per DMA round trip, so its K loop is latency-bound.  Here
//   * two PERSISTENT workgroups per CU walk 64-pixel tiles (linear pixel index: a 1x1 conv has no halo); one
//     workgroup per CU with 128-pixel tiles was measured first: 198 us -- a lone wave per SIMD exposes every store /
//     DMA wait of its epilogue;
//   * the weights never touch LDS: wave w owns output columns 32w..32w+31, and the B operand of the 32x32x2 fp32 MFMA
//     for ALL 128 input channels is 64 registers per lane, loaded once per workgroup;
//   * the A tile (64 pixels x 128 channels = 32 KB) is copied global -> LDS by LDS-DMA into one of two buffers
//     while the MFMAs of the previous tile run; 16-byte slots are XOR-swizzled with the pixel index on the source side
//     (DMA writes are lane-linear), a lane reads 4 consecutive channels of its pixel with one ds_read_b128 and feeds
//     four MFMAs (the K order is permuted identically on the weight side, as in conv_igemm_dma.hip);
//   * the producer's BatchNorm + LeakyReLU is applied at the fragment read;
//   * epilogue per tile straight from the accumulators (lane = output channel): bias, store, the {count, mean, M2}
//     partials of the consumer BatchNorm (forward) or the fused BatchNorm-backward partials (data gradient,
//     DipConvDesc.bnb_*) -- a wave owns its columns, so no cross-wave reduction is needed.
// Domain (dip_conv1x1_res_eligible): ks 1, stride 1, Cin == 128, Cout <= 128, H % 8 == 0 and W % 16 == 0 (then the
// 128-pixel tiles are as many as the 8x16 tiles the planner sized the statistics buffers for), >= 65536 pixels, one pass.
#include "dip_common.h"
#include "dip_group.h"
#include "conv_epilogue.h"
#include "lds_dma.h"
#include <stdlib.h>

namespace {

constexpr int R_RB = 2;                      // 32-pixel row blocks per tile
constexpr int R_TP = 32 * R_RB;              // pixels per tile
constexpr int R_K = 128;                     // input channels
constexpr int R_ABUF = R_TP * R_K;           // floats per A buffer
constexpr int R_NDMA = R_TP * 32 / 256;      // 16-byte DMA pieces per thread and tile
constexpr int R_LDS_BYTES = (2 * R_ABUF + 2 * R_K) * 4;      // 66.5 KB: two workgroups per CU

template <bool TR>
__global__ __launch_bounds__(256, 2) void conv1x1_res_kernel(const DipConvDesc d, const int nmacro, const int CoutP) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* tra = smem + 2 * R_ABUF;
    float* trb = tra + R_K;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int n = wave * 32 + l31;                        // this lane's output column
    const bool nv = n < d.Cout;

    // ---- weights: B operand of all 16 K groups (8 channels each) in registers ----
    f32x4 wb[16];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        const int c4 = 2 * kk + half;
        wb[kk] = *reinterpret_cast<const f32x4*>(d.wp + ((size_t)c4 * CoutP + min(n, CoutP - 1)) * 4);
    }
    if constexpr (TR) {
        if (tid < R_K) {
            tra[tid] = d.tr.a[tid];
            trb[tid] = d.tr.b[tid];
        }
    }
    const float slope = d.tr.slope;
    const float bias = (d.bias != nullptr && nv) ? d.bias[n] : 0.f;

    // ---- DMA geometry: piece f = tid + i*256 -> pixel hp = f >> 5, slot s = f & 31 holds channel group s ^ (hp & 7) ----
    unsigned voff[R_NDMA];
#pragma unroll
    for (int i = 0; i < R_NDMA; ++i) {
        const int f = tid + i * 256;
        const int hp = f >> 5, c4 = (f & 31) ^ (hp & 7);
        voff[i] = (unsigned)(hp * d.Cx + c4 * 4) * 4u;
    }
    const unsigned lds_base = (unsigned)(size_t)(lptr_t)smem;
    const unsigned lds_piece = (unsigned)wave * 1024u;
    auto dma_tile = [&](int tile, int buf) {        // 64-pixel tile `tile` -> A buffer `buf`
        const float* sb = d.x + (size_t)tile * R_TP * d.Cx;
        const unsigned m0b = lds_base + (unsigned)(buf * R_ABUF) * 4u + lds_piece;
#pragma unroll
        for (int i = 0; i < R_NDMA; ++i) lds_dma16_s(sb, voff[i], m0b + i * 4096u);
    };

    // fused BatchNorm-backward state of this lane's channel
    const bool bnb = d.bnb_y != nullptr;
    const bool fstats = d.stats != nullptr;
    const int nn = nv ? n : 0;
    float b_mean = 0.f, b_rstd = 0.f, b_a = 0.f, b_b = 0.f;
    if (bnb) {
        b_mean = d.bnb_state[nn];
        b_rstd = d.bnb_state[d.bnb_Cs + nn];
        b_a = d.bnb_state[2 * d.bnb_Cs + nn];
        b_b = d.bnb_state[3 * d.bnb_Cs + nn];
    }
    const bool ncol = n < d.Cy;
    // row block rb: pixel 32*rb + l31; its 16-byte slot for channel group c4 is c4 ^ (pixel & 7)
    int abase[R_RB], sw[R_RB];
#pragma unroll
    for (int rb = 0; rb < R_RB; ++rb) {
        const int px = 32 * rb + l31;
        abase[rb] = px * R_K;
        sw[rb] = px & 7;
    }

    // A workgroup walks MACRO tiles of 128 pixels = two 64-pixel halves (one statistics row per macro tile, as many
    // rows as the 8x16-pixel tiles the planner sized the partial buffers for); the halves alternate the A buffers.
    int mt = blockIdx.x;
    int buf = 0;
    if (mt < nmacro) dma_tile(2 * mt, 0);
    for (; mt < nmacro; mt += gridDim.x) {
        float cn = 0.f, k0 = 0.f, s1 = 0.f, s2 = 0.f;      // forward statistics (shifted sums) of the macro tile
        float g1 = 0.f, g2 = 0.f;                         // fused BatchNorm-backward sums
#pragma unroll 1
        for (int h = 0; h < 2; ++h, buf ^= 1) {
            const int tile = 2 * mt + h;
            dma_wait();                      // this tile's DMA has landed (and the previous tile's stores are out)
            __syncthreads();                 // ... for every wave; nobody reads the other buffer any more
            const int next = h == 0 ? tile + 1 : 2 * (mt + (int)gridDim.x);
            if (next < 2 * nmacro) dma_tile(next, buf ^ 1);
            const float* A = As + buf * R_ABUF;
            f32x16 acc[R_RB];
#pragma unroll
            for (int rb = 0; rb < R_RB; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
            f32x4 fa[R_RB], fn[R_RB];
            auto read_a = [&](f32x4 (&f)[R_RB], int kk) {
                const int c4 = 2 * kk + half;
#pragma unroll
                for (int rb = 0; rb < R_RB; ++rb) f[rb] = *reinterpret_cast<const f32x4*>(A + abase[rb] + ((c4 ^ sw[rb]) << 2));
            };
            read_a(fa, 0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                if (kk < 15) read_a(fn, kk + 1);
                if constexpr (TR) {
                    const int c4 = 2 * kk + half;
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(tra + c4 * 4), tb = *reinterpret_cast<const f32x4*>(trb + c4 * 4);
#pragma unroll
                    for (int rb = 0; rb < R_RB; ++rb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) fa[rb][e] = dip_act_leaky(fmaf(ta[e], fa[rb][e], tb[e]), slope);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int rb = 0; rb < R_RB; ++rb)
                        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rb][j], wb[kk][j], acc[rb], 0, 0, 0);
#pragma unroll
                for (int rb = 0; rb < R_RB; ++rb) fa[rb] = fn[rb];
                __builtin_amdgcn_sched_barrier(0);       // one K group's operands in flight at a time (register budget)
            }

            // ---- epilogue: register r of lane (l31, half) = pixel 32*rb + (r & 3) + 8*(r >> 2) + 4*half, column n ----
            const size_t p0 = (size_t)tile * R_TP;
#pragma unroll
            for (int rb = 0; rb < R_RB; ++rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][r] += bias;
                if (ncol) {
                    float* yt = d.y + (p0 + 32 * rb + 4 * half) * d.Cy + n;
#pragma unroll
                    for (int r = 0; r < 16; ++r) yt[(size_t)((r & 3) + 8 * (r >> 2)) * d.Cy] = acc[rb][r];
                }
                if (fstats) {
                    if (cn == 0.f) k0 = acc[rb][0];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float dv = acc[rb][r] - k0;
                        s1 += dv;
                        s2 = fmaf(dv, dv, s2);
                    }
                    cn += 16.f;
                }
                if (bnb) {
                    const float* yb = d.bnb_y + (p0 + 32 * rb + 4 * half) * d.bnb_Cy + nn;
                    float yv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) yv[r] = yb[(size_t)((r & 3) + 8 * (r >> 2)) * d.bnb_Cy];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float z = fmaf(b_a, yv[r], b_b);
                        const float gm = dip_mul_rn(acc[rb][r], dip_act_grad(z, d.bnb_slope));
                        g1 += gm;
                        g2 = fmaf(gm, (yv[r] - b_mean) * b_rstd, g2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // one row block's loads / stores at a time (register budget)
            }
        }
        if (fstats) {
            float mean = k0 + s1 / cn;
            float M2 = s2 - s1 * s1 / cn;
            const float on = __shfl_xor(cn, 32), om = __shfl_xor(mean, 32), oM = __shfl_xor(M2, 32);
            dip_chan(cn, mean, M2, on, om, oM);
            if (half == 0 && n < CoutP) {
                float* o = d.stats + (size_t)mt * 3 * CoutP + n;
                o[0] = nv ? cn : 0.f; o[CoutP] = nv ? mean : 0.f; o[2 * CoutP] = nv ? M2 : 0.f;
            }
        }
        if (bnb) {
            g1 += __shfl_xor(g1, 32);
            g2 += __shfl_xor(g2, 32);
            if (half == 0 && n < d.bnb_Cs) {
                float* o = d.bnb_partials + (size_t)mt * 2 * d.bnb_Cs + n;
                o[0] = nv ? g1 : 0.f; o[d.bnb_Cs] = nv ? g2 : 0.f;
            }
        }
    }
}

}  // namespace

// domain of the weights-resident kernel (see the file header); DIP_CONV_NO_RES1X1=1 switches it off (A/B)
extern "C" int dip_conv1x1_res_eligible(const DipConvDesc* dp) {
    const DipConvDesc& d = *dp;
    static const bool off = getenv("DIP_CONV_NO_RES1X1") != nullptr || getenv("DIP_CONV_NO_DMA") != nullptr;
    if (off) return 0;
    if (d.ks != 1 || d.stride != 1 || d.dil != 1 || d.off != 0 || d.Cin != R_K || d.Cout > 128 || d.Cout < 97) return 0;
    if ((d.Hout % 8) || (d.Wout % 16) || d.Hout != d.Hin || d.Wout != d.Win || d.Hout * d.Wout < 65536) return 0;
    if (d.y_pitch > 0 || d.accumulate || d.ksplit > 1) return 0;
    if (d.tr.a != nullptr && !(d.tr.slope > 0.f)) return 0;          // Swish / ELU producers: the register-staged kernel
    return 1;
}

extern "C" int dip_conv1x1_res(const DipConvDesc* dp, void* stream) {
    const DipConvDesc& d = *dp;
    if (!dip_conv1x1_res_eligible(dp)) DIP_FAIL("conv1x1_res: descriptor outside the kernel's domain");
    static bool attr_set[16] = {};
    if (dip_once_per_device(attr_set)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_res_kernel<true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_res_kernel<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS_BYTES);
        if (e != hipSuccess) { dip_set_error(hipGetErrorString(e)); return (int)e; }
    }
    const int nmacro = d.Hout * d.Wout / (2 * R_TP);           // (the domain guarantees whole macro tiles)
    const int CoutP = dip_round_up(d.Cout, 32);
    static int ncus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (ncus[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) ncus[dev] = prop.multiProcessorCount;
        if (ncus[dev] <= 0) ncus[dev] = 256;
    }
    const int ncu = ncus[dev];
    const int grid = nmacro < 2 * ncu ? nmacro : 2 * ncu;          // two persistent workgroups per CU
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (d.tr.a != nullptr)
        dip_launch(conv1x1_res_kernel<true>, dim3(grid), dim3(256), R_LDS_BYTES, st, d, nmacro, CoutP);
    else
        dip_launch(conv1x1_res_kernel<false>, dim3(grid), dim3(256), R_LDS_BYTES, st, d, nmacro, CoutP);
    DIP_CHECK_LAUNCH();
    return 0;
}
