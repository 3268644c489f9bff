// Helpers shared by the ping-pong conv kernels (conv_igemm_bf16_pp.hip, conv3x3_dwr_bf16.hip): hidden LDS-DMA, raw barrier.
#pragma once
#include "conv_bf16_args.h"

namespace {

// LDS-DMA piece issued from inline asm: INVISIBLE to hipcc's waitcnt pass on purpose.  With the builtin, the pass makes every
// ds_read wait for every LDS-DMA it believes outstanding (at the loop head: vmcnt(5) ... vmcnt(0) in front of the phase-1 fragment
// reads), which drains exactly the queue this kernel keeps in flight across its barriers.  Hidden, the queue is counted by hand
// (one s_waitcnt vmcnt(6) per chunk); hipcc's own counted waits for ordinary loads / spills only ever see FEWER outstanding
// operations than there are, i.e. they over-wait, never under-wait (memory operations return in order).
// POLICY: cache-policy bits of the load (0 default, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt)
template <int POLICY>
__device__ __forceinline__ void pp_dma16(u32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
    // (m0 is not used by anything else in this kernel: gfx950 DS instructions do not read it)
    if constexpr (POLICY == 1)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else if constexpr (POLICY == 2)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else if constexpr (POLICY == 3)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc0 sc1 lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else if constexpr (POLICY == 4)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 nt lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

#ifndef HN_PP_POLICY_A        // measurement builds: -DHN_PP_POLICY_A=n -DHN_PP_POLICY_B=n
#define HN_PP_POLICY_A 0
#endif
#ifndef HN_PP_POLICY_B
#define HN_PP_POLICY_B 0
#endif

__device__ __forceinline__ u32x4 pp_rsrc(const void* base)
{
    const unsigned long long a = (unsigned long long)base;
    u32x4 r;
    r[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));      // stride 0
    r[2] = 0x7fffffffu;                                                                       // num_records (bytes)
    r[3] = 0x00020000u;
    return r;
}

__device__ __forceinline__ void pp_bar_raw()
{
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

}  // namespace
