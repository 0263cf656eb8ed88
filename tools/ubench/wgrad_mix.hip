// Instruction mix of the weight-gradient K step (9 MFMAs, 9 A operands + 1 B operand read from LDS
// with ds_read_b32) at two workgroups per CU:
//   mode 0: operands of step s read right before its MFMAs, A operands through two registers
//           (what hipcc generates for conv_wgrad*_kernel at 240 VGPRs)
//   mode 1: operands of step s+1 read into a second register set while the MFMAs of step s run
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void wg_loop(const float* in, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < 12288; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    f32x16 c[9];
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    const float* U = lds;                 // [pixel][32 ch], 4 x 18 pixels
    const float* D = lds + 4096;          // [pixel][128]
    auto rd = [&](int s, float (&a)[9], float& b) {
        const int px = 2 * (s & 15) + half;
        const int r = px >> 4, cc = px & 15;
        b = D[px * 128 + wave * 32 + l31];
        const float* ub = U + (r * 18 + cc) * 32 + l31;
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = ub[((t / 3) * 18 + (t % 3)) * 32];
    };
    if (MODE == 0) {
        for (int s = 0; s < iters; ++s) {
            float a[9], b;
            rd(s, a, b);
#pragma unroll
            for (int t = 0; t < 9; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, c[t], 0, 0, 0);
        }
    } else {
        float a0[9], b0, a1[9], b1;
        rd(0, a0, b0);
        for (int s = 0; s < iters; s += 2) {
            rd(s + 1, a1, b1);
#pragma unroll
            for (int t = 0; t < 9; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0, c[t], 0, 0, 0);
            rd(s + 2, a0, b0);
#pragma unroll
            for (int t = 0; t < 9; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1, c[t], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int t = 0; t < 9; ++t) for (int r = 0; r < 16; ++r) sum += c[t][r];
    out[tid + blockIdx.x * 256] = sum;
}

template <int MODE>
static void run(const float* in, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds_bytes = 12288 * 4;
    hipFuncSetAttribute((const void*)wg_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    for (int wgs = 1; wgs <= 2; ++wgs) {
        const int blocks = 256 * wgs, iters = 8000;
        wg_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        wg_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)blocks * 4 * iters * 9 * 4096.0 / (ms * 1e-3) / 1e12;
        printf("mode %d, %d WG/CU: %8.3f ms %7.1f TFLOP/s\n", MODE, wgs, ms, tf);
    }
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 512 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2e-3f - 1e-3f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<0>(in, out); run<1>(in, out);
    return 0;
}
