// Round-5 candidate: weight gradient of the 128 -> 128 1x1 convolutions at >= 256^2
// (models/skip.py:88-91 of the reference; conv_wgrad_kernel<1,1,1,4> today: 127 us at 512^2 = 68 TF against a 55 us roof)
// with the fp32 MFMA fed STRAIGHT FROM GLOBAL MEMORY -- no LDS staging, x and dy read exactly once.
//
//   dW[o][c] = sum_p u[p][c] * dy[p][o],   u = act(a[c] * x[p][c] + b[c]),   K = pixels
//
// v_mfma_f32_32x32x2_f32 takes A[i][k] from lane (i, k = lane / 32) and B[k][j] from lane (j, k): with k = the pixel of a
// PAIR, a lane that loads the float4 of channels 4 * l31 .. + 3 of its pixel holds one A operand for each of FOUR 32 x 32
// row blocks (block e: row i <-> channel 4 i + e), and the float4 of dy the same for four column blocks: 2 x 16-byte loads
// per lane feed 16 MFMAs (1024 cycles).  A wave owns the whole 128 x 128 accumulator (256 registers: 1 wave per SIMD) and
// walks a strided list of pixel pairs with its loads RING stages ahead; the 4 waves of a workgroup (distinct pixels) meet
// in LDS at the end, one 64 KB slab per workgroup (256 slabs at 512^2: 16 MB, summed in a fixed order).
// Bounds at 512^2: MFMA 131072 cycles = 54.6 us @ 2.4 GHz; HBM 268 MB = 43 us @ 6.29 TB/s.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/wgrad1x1_direct.hip -o tools/ubench/bin/wgrad1x1_direct
//   tools/ubench/bin/wgrad1x1_direct          # self-check against fp64 on the host, then times 256^2 and 512^2
//
// NOT YET RUN ON HARDWARE (written after round 4's GPU budget was spent): the self-check below decides.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int RING = 6;            // K steps (pixel pairs) in flight per wave: 12 x 16-byte loads (no spills in the loop at 4..6)

template <int TR>
__global__ __launch_bounds__(512) void wgrad1x1_direct(const float* __restrict__ x, int Cx, const float* __restrict__ dy, int Cdy,
                                                       int npairs, const float* __restrict__ tra, const float* __restrict__ trb,
                                                       float slope, float* __restrict__ partial, float* __restrict__ bias_partial, int mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];          // 2 x 64 KB accumulator tiles + 4 x 128 bias sums
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave >> 1, nh = wave & 1;                              // pair stream of the workgroup, half of the output columns
    const int gw = blockIdx.x * 4 + ws, nw = gridDim.x * 4;

    f32x4 ta = f32x4{1.f, 1.f, 1.f, 1.f}, tb = f32x4{0.f, 0.f, 0.f, 0.f};
    if (TR) {       // (asm as well: a load hipcc tracks would be waited for with vmcnt(0) at its first use INSIDE the loop, every round)
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ta) : "v"(tra + 4 * l31));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(tb) : "v"(trb + 4 * l31));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ta), "+v"(tb));
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};

    // software pipeline: stage s of the ring holds pair q0 + s * nw.  The loads are inline asm with explicit vmcnt waits
    // (left to hipcc, the scheduler sinks all twelve loads of a ring round behind its MFMAs and drains them with vmcnt(0):
    // no load is in flight under an MFMA).  A pair past the end re-reads pair 0 and is multiplied by a zero dy.
    f32x4 xr[RING], dr[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s) xr[s] = dr[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto load = [&](int s, int q) __attribute__((always_inline)) {
        const size_t p = (size_t)(q < npairs ? 2 * q + half : 0);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[s]) : "v"(x + p * Cx + 4 * l31));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dr[s]) : "v"(dy + p * Cdy + 4 * l31));
    };
#pragma unroll
    for (int s = 0; s < RING; ++s) load(s, gw + s * nw);
    for (int q = gw; q < npairs; q += RING * nw) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xr[s]), "+v"(dr[s]) : "n"(2 * (RING - 1)));
            f32x4 u = xr[s];
            f32x4 g = dr[s];
            if (q + s * nw >= npairs) g = f32x4{0.f, 0.f, 0.f, 0.f};
            if (TR) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fmaf(ta[e], u[e], tb[e]);
                    u[e] = fmaxf(t, slope * t);                           // LeakyReLU, slope in (0, 1]
                }
            }
            bsum += g;
            float g0 = nh ? g[2] : g[0], g1 = nh ? g[3] : g[1];
            // everything that reads the stage's registers is done HERE, before they are refilled: the old values are dead at the
            // load, so its destination can be (and, checked in the ISA, is) the register the loop carries -- a copy of a
            // register whose load is still in flight would read garbage
            asm volatile("" : "+v"(u), "+v"(g0), "+v"(g1), "+v"(bsum));
            if (mode != 1) load(s, q + (s + RING) * nw);                  // refill this stage: RING steps ahead (mode 1: knock-out)
            if (mode != 2)
#pragma unroll
            for (int ea = 0; ea < 4; ++ea) {
                acc[ea][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[ea], g0, acc[ea][0], 0, 0, 0);
                acc[ea][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[ea], g1, acc[ea][1], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)");

    // ---- the four streams' accumulators -> one per column half: (1 -> 0, 3 -> 2) then (2 -> 0); tile layout [block][r][lane] ----
    // LDS: 4 tiles of 32 KB (stream pair x column half)
    auto put = [&](float* t) __attribute__((always_inline)) {
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
#pragma unroll
                for (int r = 0; r < 16; ++r) t[((ea * 2 + eb) * 16 + r) * 64 + lane] = acc[ea][eb][r];
    };
    auto add = [&](const float* t) __attribute__((always_inline)) {
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
            for (int eb = 0; eb < 2; ++eb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ea][eb][r] += t[((ea * 2 + eb) * 16 + r) * 64 + lane];
    };
    float* tile = lds + (((ws >> 1) & 1) * 2 + nh) * 8192;
    float* bred = lds + 4 * 8192;
    {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = bsum[e] + __shfl_xor(bsum[e], 32);
        if (half == 0 && nh == 0) *reinterpret_cast<f32x4*>(bred + ws * 128 + 4 * l31) = o;
    }
    if (ws & 1) put(tile);
    __syncthreads();
    if (!(ws & 1)) add(tile);
    __syncthreads();
    if (ws == 2) put(lds + nh * 8192);
    __syncthreads();
    if (ws == 0) {
        add(lds + nh * 8192);
        // slab [c][o]: block (ea, eb), register r, lane: c = 4 * row + ea, row = 8 (r / 4) + 4 half + (r % 4); o = 4 * l31 + 2 nh + eb
        float* slab = partial + (size_t)blockIdx.x * 128 * 128;
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = 4 * (8 * (r >> 2) + 4 * half + (r & 3)) + ea;
                *reinterpret_cast<float2*>(slab + c * 128 + 4 * l31 + 2 * nh) = float2{acc[ea][0][r], acc[ea][1][r]};
            }
    } else if (wave == 2 && bias_partial != nullptr) {
        for (int o = lane; o < 128; o += 64)
            bias_partial[(size_t)blockIdx.x * 128 + o] = (bred[o] + bred[128 + o]) + (bred[256 + o] + bred[384 + o]);
    }
}

// fixed-order sum of the slabs (what dip_wgrad_reduce does in the library, there straight into the OIHW gradient)
__global__ __launch_bounds__(256) void reduce_slabs(const float* __restrict__ partial, int nslab, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 3 < nslab; k += 4) {
        s0 += partial[(size_t)k * n + i]; s1 += partial[(size_t)(k + 1) * n + i];
        s2 += partial[(size_t)(k + 2) * n + i]; s3 += partial[(size_t)(k + 3) * n + i];
    }
    for (; k < nslab; ++k) s0 += partial[(size_t)k * n + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static int g_mode = 0;
static int run(int H, int W, int nwg, bool check_all) {
    const int P = H * W, C = 128, npairs = P / 2;
    std::vector<float> hx((size_t)P * C), hd((size_t)P * C), ha(C), hb(C);
    srand(7);
    for (auto& v : hx) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hd) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 1e-2f;
    for (int c = 0; c < C; ++c) { ha[c] = 0.5f + (float)rand() / RAND_MAX; hb[c] = (float)rand() / RAND_MAX - 0.5f; }
    float *x, *dy, *a, *b, *partial, *bp, *dw, *db;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&dy, hd.size() * 4)); CK(hipMalloc(&a, C * 4)); CK(hipMalloc(&b, C * 4));
    CK(hipMalloc(&partial, (size_t)nwg * C * C * 4)); CK(hipMalloc(&bp, (size_t)nwg * C * 4));
    CK(hipMalloc(&dw, C * C * 4)); CK(hipMalloc(&db, C * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(a, ha.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice));
    const int lds = (4 * 8192 + 512) * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad1x1_direct<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const float slope = 0.2f;
    auto launch = [&]() {
        wgrad1x1_direct<1><<<nwg, 512, lds>>>(x, C, dy, C, npairs, a, b, slope, partial, bp, g_mode);
        reduce_slabs<<<(C * C + 255) / 256, 256>>>(partial, nwg, C * C, dw);
        reduce_slabs<<<1, 256>>>(bp, nwg, C, db);
    };
    launch();
    CK(hipDeviceSynchronize());
    std::vector<float> gw(C * C), gb(C);
    CK(hipMemcpy(gw.data(), dw, C * C * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gb.data(), db, C * 4, hipMemcpyDeviceToHost));
    // fp64 reference: dW[c][o] (this program's slab layout) for all (c, o) or for 8 x 8 of them; u evaluated in fp32 as the kernel does
    double worst = 0.0, scale = 0.0;
    const int step = check_all ? 1 : 16;
    for (int c = 0; c < C; c += step)
        for (int o = 0; o < C; o += step) {
            double s = 0.0;
            for (int p = 0; p < P; ++p) {
                const float t = fmaf(ha[c], hx[(size_t)p * C + c], hb[c]);
                s += (double)fmaxf(t, slope * t) * (double)hd[(size_t)p * C + o];
            }
            worst = fmax(worst, fabs(s - (double)gw[c * C + o]));
            scale = fmax(scale, fabs(s));
        }
    double bworst = 0.0, bscale = 0.0;
    for (int o = 0; o < C; ++o) {
        double s = 0.0;
        for (int p = 0; p < P; ++p) s += (double)hd[(size_t)p * C + o];
        bworst = fmax(bworst, fabs(s - (double)gb[o]));
        bscale = fmax(bscale, fabs(s));
    }
    // fp32 accumulation of P products of size ~|u||dy|: error ~ sqrt(P) * 2^-24 * rms term; bound generously at 1e-4 of the scale
    const bool ok = worst <= 1e-4 * scale + 1e-6 && bworst <= 1e-4 * bscale + 1e-6;
    printf("%dx%d, %d workgroups: max|dW err| %.3e (scale %.3e), max|db err| %.3e (scale %.3e): %s\n", H, W, nwg, worst, scale, bworst,
           bscale, ok ? "OK" : "MISMATCH");
    fflush(stdout);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass) {            // kernel alone, then kernel + reductions
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) {
            if (pass == 0) wgrad1x1_direct<1><<<nwg, 512, lds>>>(x, C, dy, C, npairs, a, b, slope, partial, bp, g_mode);
            else launch();
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = 1e3 * ms / reps, gf = 2.0 * P * C * C / 1e9, mb = 2.0 * P * C * 4 / 1e6;
        printf("   %s: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s of compulsory reads\n", pass == 0 ? "kernel alone      " : "kernel + reductions", us,
               gf / us * 1e3, mb / us);
    }
    hipFree(x); hipFree(dy); hipFree(a); hipFree(b); hipFree(partial); hipFree(bp); hipFree(dw); hipFree(db);
    return ok ? 0 : 1;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    int bad = 0;
    if (getenv("W1_MODE")) { g_mode = atoi(getenv("W1_MODE")); run(256, 256, 256, false); run(512, 512, 256, false); return 0; }
    bad += run(64, 64, 16, true);            // every (c, o) against fp64; 2048 pairs over 64 waves: ragged ring tail
    bad += run(50, 66, 7, true);             // pairs not divisible by the wave count
    bad += run(256, 256, 256, false);
    bad += run(512, 512, 256, false);
    bad += run(512, 512, 128, false);
    printf(bad ? "FAILED\n" : "ALL OK\n");
    return bad;
}
