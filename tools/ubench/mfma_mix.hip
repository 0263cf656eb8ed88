// MFMA issue rate with the conv kernel's instruction mix, on top of tools/ubench/mfma_peak.hip:
//   mode 0: 64 MFMAs per iteration, operands in registers
//   mode 1: + 16 ds_read_b128 per iteration feeding the operands (the kernel's LDS read rate)
//   mode 2: + one workgroup barrier per iteration
//   mode 3: + 14 global_load_lds_dwordx4 per thread per iteration (the kernel's DMA rate)
// Also reports s_memtime ticks (clock64) against the 100 MHz wall clock for one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const void* g, unsigned lds_wave_base) {
    unsigned save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(g), "s"(lds_wave_base) : "memory");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void mix_loop(const float* in, float* out, int iters, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 14336; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    f32x16 c[2][2];
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) c[m][n][r] = 0.f;
    f32x4 a[2], b[2];
    a[0] = a[1] = b[0] = b[1] = f32x4{in[lane], in[lane + 64], in[lane + 128], in[lane + 192]};
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    const float* gsrc = in + (size_t)(blockIdx.x & 255) * 65536 + tid * 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (MODE >= 1) {
                const int base = ((i * 4 + kk) * 256) & 4095;
                a[0] = *reinterpret_cast<const f32x4*>(lds + base + (lane & 31) * 4 + (lane >> 5) * 128);
                a[1] = *reinterpret_cast<const f32x4*>(lds + base + 4096 + (lane & 31) * 4 + (lane >> 5) * 128);
                b[0] = *reinterpret_cast<const f32x4*>(lds + 8192 + base / 2 + wave * 1024 + lane * 4);
                b[1] = *reinterpret_cast<const f32x4*>(lds + 8192 + base / 2 + wave * 1024 + 256 + lane * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        c[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][j], b[n][j], c[m][n], 0, 0, 0);
            if (MODE >= 3 && kk == 0) {
#pragma unroll
                for (int q = 0; q < 14; ++q)
                    dma16(gsrc + (size_t)((i * 14 + q) & 15) * 1024, (unsigned)((14336 + (q & 3) * 1024 + wave * 256) * 4));
            }
        }
        if (MODE >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE >= 2) __syncthreads();
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int n = 0; n < 2; ++n) for (int r = 0; r < 16; ++r) s += c[m][n][r];
    out[threadIdx.x + blockIdx.x * 256] = s;
    if (tid == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int MODE>
static void run(const float* in, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds_bytes = 18432 * 4;                          // 72 KB: 2 WG/CU fit
    hipFuncSetAttribute((const void*)mix_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    for (int wgs = 1; wgs <= 2; ++wgs)
    for (int cus = 64; cus <= 256; cus *= 4) {
        const int blocks = cus * wgs, iters = 4000;
        mix_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, 100, clk);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        mix_loop<MODE><<<blocks, 256, lds_bytes>>>(in, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double tf = (double)blocks * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12;
        printf("mode %d blocks=%4d (%d WG/CU): %8.3f ms %7.1f TFLOP/s %5.1f ns/MFMA/SIMD | clock64 ticks/MFMA %.1f, clock64 %.1f MHz (wall 100 MHz)\n",
               MODE, blocks, wgs, ms, tf, ms * 1e6 / ((double)iters * 64 * wgs), (double)h[0] / ((double)iters * 64),
               (double)h[0] / (double)h[1] * 100.0);
    }
}
int main() {
    float *in, *out; unsigned long long* clk;
    hipMalloc(&in, (size_t)256 * 65536 * 4 + (1 << 20)); hipMalloc(&out, 4 * 256 * 4096); hipMalloc(&clk, 64);
    float* h = (float*)malloc(4096 * 4);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX * 2e-3f - 1e-3f;
    for (size_t o = 0; o < (size_t)256 * 65536; o += 4096) hipMemcpy(in + o, h, 4096 * 4, hipMemcpyHostToDevice);
    run<0>(in, out, clk); run<1>(in, out, clk); run<2>(in, out, clk); run<3>(in, out, clk);
    return 0;
}
