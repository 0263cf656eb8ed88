// fp32 products on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 terms
// (a = a1 + a2 + a3: 8 + 8 + 8 significand bits, by truncation) and a*b = sum of the nine cross products a_i*b_j, each
// exact in fp32 (8 x 8 bits), accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- 9 bf16 MFMAs (32 cycles each, K = 16)
// replace 8 fp32 MFMAs (64 cycles each, K = 2): 0.5625 of the matrix-pipe time per fp32 product.
//   part 1: numerics -- C[32x32] = A[32xK] B[Kx32], K = 1152 (the 3x3 x 128-channel layers), against an fp64 evaluation:
//           fp32 MFMA (the current kernels), bf16x9, bf16x6 (the three smallest cross terms dropped)
//   part 2: throughput of the inner loop a conv kernel would run: 64x64 wave tile, operands from LDS (12 ds_read_b128
//           per K = 16 step), 36 MFMAs per step, 1 or 2 workgroups of 4 waves per CU
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float a, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned ua = __float_as_uint(a);
    const unsigned uh = ua & 0xFFFF0000u;
    const float r1 = a - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = uh >> 16; m = um >> 16; l = __float_as_uint(r2) >> 16;
}

// one wave: C = A B with A [32][K] row-major, B [K][32] row-major
__global__ void numerics(const float* A, const float* B, int K, float* C32, float* C9, float* C6) {
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    f32x16 c32, c9, c6;
    for (int r = 0; r < 16; ++r) { c32[r] = 0.f; c9[r] = 0.f; c6[r] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k0 + half], B[(k0 + half) * 32 + l31], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        // lane (l31, half) holds k = k0 + 8 * half + 0..7 of row / column l31
        union { bf16x8 v; unsigned short s[8]; } a[3], b[3];
        for (int e = 0; e < 8; ++e) {
            unsigned h, m, l;
            split3(A[l31 * K + k0 + 8 * half + e], h, m, l);
            a[0].s[e] = h; a[1].s[e] = m; a[2].s[e] = l;
            split3(B[(k0 + 8 * half + e) * 32 + l31], h, m, l);
            b[0].s[e] = h; b[1].s[e] = m; b[2].s[e] = l;
        }
        // smallest terms first
        for (int s = 4; s >= 0; --s)
            for (int i = 0; i < 3; ++i) {
                const int j = s - i;
                if (j < 0 || j > 2) continue;
                c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, c9, 0, 0, 0);
                if (s <= 2) c6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, c6, 0, 0, 0);
            }
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        C32[row * 32 + l31] = c32[r]; C9[row * 32 + l31] = c9[r]; C6[row * 32 + l31] = c6[r];
    }
}

template <int SYNC_EVERY>
__global__ __launch_bounds__(256) void loop_bf16x9(const float* in, float* out, int steps) {
    // LDS image of one K = 16 unit: three bf16 planes of A [128 px][16 k] and B [128 n][16 k], row pitch 48 B (12 dwords)
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 3 * 128 * 12];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5, w = tid >> 6, wm = w >> 1, wn = w & 1;
    for (int i = tid; i < 2 * 3 * 128 * 12; i += 256) lds[i] = __float_as_uint(in[i & 1023]) & 0x3F803F80u;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned* Ab = lds, * Bb = lds + 3 * 128 * 12;
    for (int s = 0; s < steps; ++s) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(Ab + (p * 128 + (wm * 64 + i * 32 + l31)) * 12 + 4 * half);
                b[i][p] = *reinterpret_cast<const bf16x8*>(Bb + (p * 128 + (wn * 64 + i * 32 + l31)) * 12 + 4 * half);
            }
        }
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
            for (int pb = 0; pb < 3; ++pb)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][pa], b[j][pb], acc[i][j], 0, 0, 0);
        if (SYNC_EVERY && (s % SYNC_EVERY) == SYNC_EVERY - 1) __syncthreads();
    }
    float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = sum;
}

int main() {
    const int K = 1152;
    std::vector<float> hA(32 * K), hB(K * 32);
    srand(1);
    auto rnd = [] { double u = 0; for (int i = 0; i < 12; ++i) u += (double)rand() / RAND_MAX; return (float)(u - 6.0); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd() * 0.03f;
    float *A, *B, *C32, *C9, *C6;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C32, 4096); hipMalloc(&C9, 4096); hipMalloc(&C6, 4096);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    numerics<<<1, 64>>>(A, B, K, C32, C9, C6);
    std::vector<float> r32(1024), r9(1024), r6(1024);
    hipMemcpy(r32.data(), C32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(r9.data(), C9, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(r6.data(), C6, 4096, hipMemcpyDeviceToHost);
    double e32 = 0, e9 = 0, e6 = 0, m32 = 0, m9 = 0, m6 = 0, sc = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0, sabs = 0;
            for (int k = 0; k < K; ++k) { ref += (double)hA[i * K + k] * hB[k * 32 + j]; sabs += fabs((double)hA[i * K + k] * hB[k * 32 + j]); }
            const double d32 = r32[i * 32 + j] - ref, d9 = r9[i * 32 + j] - ref, d6 = r6[i * 32 + j] - ref;
            e32 += d32 * d32; e9 += d9 * d9; e6 += d6 * d6;
            m32 = fmax(m32, fabs(d32) / sabs); m9 = fmax(m9, fabs(d9) / sabs); m6 = fmax(m6, fabs(d6) / sabs);
            sc += ref * ref;
        }
    printf("K = %d: rel-L2 error vs fp64: fp32 MFMA %.3e | bf16x9 %.3e | bf16x6 %.3e ;  max |err| / sum|a b|: %.3e | %.3e | %.3e\n", K,
           sqrt(e32 / sc), sqrt(e9 / sc), sqrt(e6 / sc), m32, m9, m6);

    float *in, *out;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 1024);
    float h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX + 1.f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode)
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
            const int blocks = 256 * wgs_per_cu, steps = 4000;
            auto run = [&](int n) {
                if (mode == 0) loop_bf16x9<0><<<blocks, 256>>>(in, out, n);
                else if (mode == 1) loop_bf16x9<9><<<blocks, 256>>>(in, out, n);
                else if (mode == 2) loop_bf16x9<3><<<blocks, 256>>>(in, out, n);
                else loop_bf16x9<1><<<blocks, 256>>>(in, out, n);
            };
            run(100);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            run(steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * steps * 2.0 * 128 * 128 * 16;      // fp32-equivalent
            const char* names[4] = {"no barrier", "barrier every 9 steps", "barrier every 3 steps", "barrier every step"};
            printf("%s, %d WG/CU: %.3f ms -> %.1f TFLOP/s fp32-equivalent (x9 = %.0f TFLOP/s on the bf16 pipe; fp32 MFMA peak 157.3)\n",
                   names[mode], wgs_per_cu, ms, flop / (ms * 1e-3) / 1e12, 9 * flop / (ms * 1e-3) / 1e12);
        }
    return 0;
}
