// lds_ring.hpp -- helpers for a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4).
//
// The DMA is issued from inline asm, so hipcc neither counts it nor waits for it: every wait is a
// hand-counted s_waitcnt vmcnt(N) (cdna_hip_programming.md section 5.7).  vmcnt retires in issue
// order, so "N = number of my DMAs issued after the piece I need" is always safe: any other VMEM
// operation the compiler issues in between only makes the wait longer, never shorter.
// A wave only ever reads ring slots it filled itself, so no barrier is needed for visibility.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// 64 lanes x 16 B from per-lane global pointers -> LDS [lds_dst, lds_dst + 1 KiB) (lane i at +16 i)
__device__ __forceinline__ void lds_dma_1k(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// same, but first waits for this wave's outstanding LDS reads (the slot is being recycled)
__device__ __forceinline__ void lds_dma_1k_after_read(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// wait until at most `younger` of my DMAs are outstanding; exact for the steady state (R-1), a full
// drain otherwise (only the tail of a stream takes that path)
template <int R>
__device__ __forceinline__ void wait_vmcnt_dyn(int younger) {
    if (younger >= R - 1) wait_vmcnt<R - 1>(); else wait_vmcnt0();
}

