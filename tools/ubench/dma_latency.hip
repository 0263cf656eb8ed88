// Latency of one global_load_dwordx4 vs one global_load_lds_dwordx4 (LDS-DMA), L2-resident data,
// idle chip and with the issuing wave's MFMA queue busy.  Cycles = s_memtime (shader clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const void* g, unsigned m0v) {
    unsigned save;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(save) : "v"(g), "s"(m0v) : "memory");
}

// mode 0: load to VGPR; 1: LDS-DMA; +2: 16 independent MFMAs issued right before the load
template <int MODE>
__global__ __launch_bounds__(256) void lat(const float* in, float* out, unsigned long long* res, int reps, int nld) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 c[4];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) c[m][r] = 0.f;
    const float a = in[tid], b = in[tid + 256];
    unsigned long long tot = 0, mx = 0;
    f32x4 sink = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < reps; ++i) {
        const float* g = in + 1024 + ((size_t)((i * 37 + blockIdx.x * 11) & 255)) * 4096 + tid * 4;   // 4 MB window
        if (MODE & 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) c[j & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[j & 3], 0, 0, 0);
        }
        const unsigned long long t0 = clock64();
        if (MODE & 1) {
            for (int q = 0; q < nld; ++q) dma16(g + q * 1024, (unsigned)((wave * 256 + q * 1024) * 4));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int q = 0; q < nld; ++q) {
                f32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g + q * 1024) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // serial only when nld == 1
                sink += v;
            }
        }
        const unsigned long long t1 = clock64();
        tot += t1 - t0;
        if (t1 - t0 > mx) mx = t1 - t0;
        __syncthreads();
    }
    float s = sink[0] + sink[1] + sink[2] + sink[3] + lds[tid];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += c[m][r];
    out[tid + blockIdx.x * 256] = s;
    if (tid == 0) { res[blockIdx.x * 2] = tot; res[blockIdx.x * 2 + 1] = mx; }
}

template <int MODE>
static void run(const char* name, const float* in, float* out, unsigned long long* res, int blocks, int nld) {
    const int reps = 200;
    lat<MODE><<<blocks, 256, 65536>>>(in, out, res, reps, nld);      // warm L2
    lat<MODE><<<blocks, 256, 65536>>>(in, out, res, reps, nld);
    hipDeviceSynchronize();
    unsigned long long h[2048]; hipMemcpy(h, res, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double mean = 0; unsigned long long mx = 0;
    for (int b = 0; b < blocks; ++b) { mean += (double)h[2 * b] / reps; if (h[2 * b + 1] > mx) mx = h[2 * b + 1]; }
    printf("%-34s blocks=%4d loads/wave=%d: mean %7.0f cycles, max %llu\n", name, blocks, nld, mean / blocks, mx);
}
int main() {
    float *in, *out; unsigned long long* res;
    hipMalloc(&in, (size_t)(1024 + 256 * 4096 + 16384) * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&res, 16 * 2048);
    hipMemset(in, 0, (size_t)(1024 + 256 * 4096 + 16384) * 4);
    hipFuncSetAttribute((const void*)lat<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)lat<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)lat<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)lat<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int blocks : {1, 256, 512}) {
        run<0>("global_load_dwordx4", in, out, res, blocks, 1);
        run<1>("global_load_lds_dwordx4", in, out, res, blocks, 1);
        run<1>("global_load_lds_dwordx4", in, out, res, blocks, 4);
        run<1>("global_load_lds_dwordx4", in, out, res, blocks, 10);
        run<2>("global_load_dwordx4 after 16 MFMA", in, out, res, blocks, 1);
        run<3>("global_load_lds_dwordx4 after 16 MFMA", in, out, res, blocks, 1);
        run<3>("global_load_lds_dwordx4 after 16 MFMA", in, out, res, blocks, 4);
    }
    return 0;
}
