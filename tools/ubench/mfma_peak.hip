// Measured fp32-MFMA ceiling on this box: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent
// accumulators per wave, random (non-zero) operands, no memory traffic in the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(const float* in, float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a0 = in[tid & 1023], a1 = in[(tid + 17) & 1023], b0 = in[(tid + 33) & 1023], b1 = in[(tid + 71) & 1023];
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c3, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    out[tid] = s;
}
int main() {
    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 4096);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2e-3f - 1e-3f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (int cus = 64; cus <= 256; cus *= 4) {
            const int blocks = cus * wgs_per_cu, iters = 4000;
            mfma_loop<<<blocks, 256>>>(in, out, 100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            mfma_loop<<<blocks, 256>>>(in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)blocks * 4 * iters * 64;               // MFMAs
            const double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
            const double ns_per_mfma_per_simd = ms * 1e6 / ((double)iters * 64 * wgs_per_cu);
            printf("blocks=%4d (%d WG/CU on %3d CUs): %8.3f ms  %7.1f TFLOP/s  %5.1f ns per MFMA per SIMD (64 cyc @2.4GHz = 26.7 ns)\n",
                   blocks, wgs_per_cu, cus, ms, tf, ns_per_mfma_per_simd);
        }
    return 0;
}
