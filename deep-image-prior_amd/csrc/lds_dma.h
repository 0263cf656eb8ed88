// LDS-DMA helpers (gfx950 global_load_lds_dwordx4) shared by the asynchronous-staging kernels.
#pragma once
#include "dip_common.h"

typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_dst_wave_uniform) {
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)lds_dst_wave_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(base)
                 : "memory");
}
// saddr form: 64-bit wave-uniform base in SGPRs + 32-bit per-lane byte offset; LDS base in m0.
// Two SALU moves + the load: the per-unit weight DMA costs ~15 instructions per wave.
__device__ __forceinline__ void lds_dma16_s(const void* sbase_in, unsigned voff, unsigned m0val_in) {
    // the operands are workgroup-uniform by construction; tell the register allocator so
    const unsigned long long sb64 = (unsigned long long)sbase_in;
    const unsigned sb_hi = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sb64 >> 32));
    const unsigned sb_lo = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sb64);     // (the builtin returns int)
    const unsigned long long sb_u = ((unsigned long long)sb_hi << 32) | (unsigned long long)sb_lo;
    const void* sbase = (const void*)sb_u;
    const unsigned m0val = __builtin_amdgcn_readfirstlane(m0val_in);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(m0val)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// wait until at most n (wave-uniform, 0..8) of this wave's loads are still in flight
__device__ __forceinline__ void dma_wait_keep(int n) {
    switch (n) {
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

