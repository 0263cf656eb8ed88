// Instruction mix of a weight-gradient K step in which a wave owns 32 input channels x 64 output channels
// (18 accumulators = 288 registers, so ONE workgroup of 4 waves per CU and the accumulators in AGPRs):
// 9 A operands + 2 B operands (ds_read_b32) feed 18 MFMAs -- 0.61 LDS reads per MFMA instead of 1.11.
//   mode 0: operands of step s read right before its MFMAs
//   mode 1: operands of step s+1 read while the MFMAs of step s run
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void wg_loop(const float* in, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 16384; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    f32x16 c[18];
    for (int t = 0; t < 18; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    const float* U = lds;                 // [pixel][64 ch], 4 x 18 pixels
    const float* D = lds + 8192;          // [pixel][128]
    auto rd = [&](int s, float (&a)[9], float (&b)[2]) {
        const int px = 2 * (s & 15) + half;
        const int r = px >> 4, cc = px & 15;
        b[0] = D[px * 128 + wn * 64 + l31];
        b[1] = D[px * 128 + wn * 64 + 32 + l31];
        const float* ub = U + (r * 18 + cc) * 64 + wm * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = ub[((t / 3) * 18 + (t % 3)) * 64];
    };
    // modes 2/3: the same with the accumulators pinned by inline asm: 16 in AGPRs, 2 in arch VGPRs (hipcc left to
    // itself shuffles ~100 v_accvgpr moves per step between the two files)
    f32x16 ca[16], cv[2];
    if (MODE >= 2) {
        for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) ca[t][r] = 0.f;
        for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) cv[t][r] = 0.f;
    }
#define MFMA_A(acc, x, y) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y))
#define MFMA_V(acc, x, y) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))
    auto mm = [&](float (&a)[9], float (&b)[2]) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            MFMA_A(ca[2 * t], a[t], b[0]);
            MFMA_A(ca[2 * t + 1], a[t], b[1]);
        }
        MFMA_V(cv[0], a[8], b[0]);
        MFMA_V(cv[1], a[8], b[1]);
    };
    if (MODE == 2) {
        for (int s = 0; s < iters; ++s) {
            float a[9], b[2];
            rd(s, a, b);
            mm(a, b);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int t = 0; t < 16; ++t) c[t] = ca[t];
        c[16] = cv[0]; c[17] = cv[1];
    } else if (MODE == 3) {
        float a0[9], b0[2], a1[9], b1[2];
        rd(0, a0, b0);
        for (int s = 0; s < iters; s += 2) {
            rd(s + 1, a1, b1);
            mm(a0, b0);
            rd(s + 2, a0, b0);
            mm(a1, b1);
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int t = 0; t < 16; ++t) c[t] = ca[t];
        c[16] = cv[0]; c[17] = cv[1];
    } else if (MODE == 0) {
        for (int s = 0; s < iters; ++s) {
            float a[9], b[2];
            rd(s, a, b);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                c[2 * t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[0], c[2 * t], 0, 0, 0);
                c[2 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[1], c[2 * t + 1], 0, 0, 0);
            }
        }
    } else {
        float a0[9], b0[2], a1[9], b1[2];
        rd(0, a0, b0);
        for (int s = 0; s < iters; s += 2) {
            rd(s + 1, a1, b1);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                c[2 * t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[0], c[2 * t], 0, 0, 0);
                c[2 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[1], c[2 * t + 1], 0, 0, 0);
            }
            rd(s + 2, a0, b0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                c[2 * t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[0], c[2 * t], 0, 0, 0);
                c[2 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[1], c[2 * t + 1], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
    for (int t = 0; t < 18; ++t) for (int r = 0; r < 16; ++r) sum += c[t][r];
    out[tid + blockIdx.x * 256] = sum;
}

template <int MODE>
static void run(const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds_bytes = 90 * 1024;          // > 80 KB: one workgroup per CU, as the real kernel would have
    hipFuncSetAttribute((const void*)wg_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    const int blocks = 256, iters = 8000;
    wg_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, 64);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    wg_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)blocks * 4 * iters * 18 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("mode %d, 1 WG/CU, 18 accumulators per wave: %8.3f ms %7.1f TFLOP/s\n", MODE, ms, tf);
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2e-3f - 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, out); run<1>(in, out); run<2>(in, out); run<3>(in, out);
    return 0;
}
