// Do the vector-memory loads of one wave return in order under contention?  conv_thin4_mfma_kernel (csrc/conv_thin4.hip) waited
// with vmcnt(2 NJ) -- exact if they do -- and produced wrong data in 3 % of the launches that ran beside a chip-filling kernel;
// vmcnt(NJ) did not.  This program isolates the assumption: a wave keeps three groups of 8 global_load_dwordx4 in flight over a
// buffer whose every dword holds its own index, waits with vmcnt(16) for the oldest group (the other two are the only younger
// loads), and checks what arrived; a second stream runs a copy kernel over 2 GiB all the while.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/vmcnt_order.hip -o tools/ubench/bin/vmcnt_order && tools/ubench/bin/vmcnt_order
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// buf[i] = i.  A "row" = 8 loads of 1 KB (64 lanes x 16 B), 64 KB apart (8 different DRAM pages), rows 1 MB apart
template <int WAIT>
__global__ __launch_bounds__(256) void probe(const unsigned* __restrict__ buf, size_t ndw, int nrows, unsigned long long* bad, int mfma, int share) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    auto addr = [&](int row, int j) -> const unsigned* {
        // share > 1: `share` neighbouring waves read the SAME lines (a mix of L1 hits and misses inside one wave's queue, as the
        // overlapping strips of conv_thin4 produce); odd rows are skewed by j so that the sharers do not arrive in step
        const size_t wv = share > 1 ? (size_t)(gw / share) : (size_t)gw;
        const size_t base = ((wv + (size_t)row * nw) * 262144 + (size_t)((j + (share > 1 ? (gw % share) * row : 0)) & 7) * 16384 + (size_t)lane * 4) % (ndw - 4);
        return buf + (base & ~(size_t)3);
    };
    u32x4 a0[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a0[j] = a1[j] = a2[j] = u32x4{0, 0, 0, 0};
    auto load = [&](int row, u32x4 (&a)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(a[j]) : "v"(addr(row, j)));
    };
    unsigned long long nbad = 0;
    f32x16 acc = {0};
    auto use = [&](int row, u32x4 (&a)[8]) {
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a[0]) : "n"(WAIT));
#pragma unroll
        for (int j = 1; j < 8; ++j) asm volatile("" : "+v"(a[j]));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned e = (unsigned)(addr(row, j) - buf);
            nbad += (a[j][0] != e) + (a[j][1] != e + 1) + (a[j][2] != e + 2) + (a[j][3] != e + 3);
        }
        for (int m = 0; m < mfma; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(1.f, 1.f, acc, 0, 0, 0);      // time between rows
    };
    load(0, a0);
    load(1, a1);
    for (int r = 0; r < nrows; r += 3) {
        load(r + 2, a2); use(r, a0);
        load(r + 3, a0); use(r + 1, a1);
        load(r + 4, a1); use(r + 2, a2);
    }
    asm volatile("s_waitcnt vmcnt(0)");
    if (acc[0] == 12345.f) nbad += 1;
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void hog(const float4* __restrict__ src, float4* __restrict__ dst, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}

int main() {
    const size_t ndw = (size_t)1 << 29;      // 2 GiB of dwords = its own index (fits 32 bits)
    unsigned* buf; float4 *s, *d; unsigned long long* bad;
    if (hipMalloc(&buf, ndw * 4) || hipMalloc(&s, (size_t)1 << 30) || hipMalloc(&d, (size_t)1 << 30) || hipMalloc(&bad, 8)) { printf("hipMalloc failed\n"); return 1; }
    fill<<<4096, 256>>>(buf, ndw);
    hipMemset(s, 0, (size_t)1 << 30);
    hipStream_t st1, st2; hipStreamCreate(&st1); hipStreamCreate(&st2);
    for (int share = 1; share <= 4; share *= 2)
    for (int contention = 0; contention < 2; ++contention)
        for (int mf = 0; mf <= 24; mf += 24)
            for (int wait = 16; wait >= 8; wait -= 8) {
                hipMemset(bad, 0, 8);
                hipDeviceSynchronize();
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0, st1);
                for (int rep = 0; rep < 20; ++rep) {
                    if (contention) hog<<<2048, 256, 0, st2>>>(s, d, ((size_t)1 << 30) / 16, 1);
                    if (wait == 16) probe<16><<<256, 256, 0, st1>>>(buf, ndw, 600, bad, mf, share);
                    else probe<8><<<256, 256, 0, st1>>>(buf, ndw, 600, bad, mf, share);
                }
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; }
                unsigned long long h = 0; hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
                hipEventRecord(e1, st1); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                printf("share %d, contention %d, %2d MFMAs per row, vmcnt(%2d): %llu wrong dwords of %llu   (%.1f ms per probe launch)\n", share, contention, mf, wait, h,
                       20ull * 256 * 256 * 600 * 32, ms / 20);
            }
    return 0;
}
