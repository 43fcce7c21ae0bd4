// Declarations shared by the attention kernels (attention.hip: per-tile loop; attention_sp.hip: software-pipelined loop).
#pragma once
#include "common.h"

namespace vitg {

struct AttnArgs {
    const void* q; const void* k; const void* v; void* o;
    long q_bs, k_bs, v_bs, o_bs;     // batch strides (elements)
    int q_rs, k_rs, v_rs, o_rs;      // row (token) strides (elements)
    int q_hs, k_hs, v_hs, o_hs;      // head strides (elements)
    int S;                           // number of queries
    int Sk;                          // number of keys (== S unless the keys were all-gathered from other ranks)
    const int* sk_dev;               // optional: the number of keys lives in device memory (<= Sk); lets ONE captured graph serve
                                     // prompts of different length inside a bucket (vitg_attention_dyn)
    int group;                       // q heads per kv head
    int kv_heads, nqb;               // kv heads, q-blocks per head
    int ablate;                      // profiling only (tools/attn_ablate.py): bit0 no QK^T, bit1 no exp, bit2 no PV, bit3 no K/V staging
    float scale_log2e;               // softmax scale * log2(e)
};

// first statement of every attention kernel: adopt the device-side key count (wave-uniform scalar load)
__device__ __forceinline__ void adopt_device_key_count(AttnArgs& a) {
    if (a.sk_dev) a.Sk = min(a.Sk, max(1, __builtin_amdgcn_readfirstlane(*a.sk_dev)));
}

// One LDS-DMA piece: 64 lanes x 16 B from (wave-uniform base + per-lane byte offset) to 1 KiB of LDS at a wave-uniform address.
// VITG_ASM_DMA = 1 (default since round 5: +0.7 % hd 72, +1 % hd 128 in same-call alternation, gpurun_out/r05_dma_1): issued from inline assembly, which hipcc does not count - the caller drains with lds_dma_drain() in front of the barrier that
// publishes the tile.  Why: hipcc orders every LDS read it cannot disambiguate behind a pending LDS-DMA builtin (s_waitcnt vmcnt(0) in front of the
// first V^T read of an interval, in the middle of the interleaved stream - profiles/r05_attention_w4.md), i.e. a wave waits for its staging of the
// NEXT tile before it reads THIS one.  VITG_ASM_DMA = 0: the builtin, hipcc's own waits.
#ifndef VITG_ASM_DMA
#define VITG_ASM_DMA 1
#endif
__device__ __forceinline__ void lds_dma16(const char* base_uniform, unsigned off, char* lds_dst_uniform) {
#if VITG_ASM_DMA
    // both are wave-uniform by contract; readfirstlane makes that provable where the value travelled through a lane-dependent-looking path (free otherwise)
    const unsigned ldsu = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_dst_uniform);
    const uintptr_t pb = (uintptr_t)base_uniform;
    const char* bu = (const char*)(((uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pb >> 32)) << 32) | (uintptr_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pb));      // (unsigned): the builtin returns int
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(ldsu), "v"(off), "s"(bu) : "memory", "m0");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base_uniform + off),
                                     (__attribute__((address_space(3))) void*)lds_dst_uniform, 16, 0, 0);
#endif
}
// The same piece through a BUFFER descriptor (base, extent in bytes): offsets at or past the extent read as zeros in hardware, so a staging stream needs
// neither row clamps nor a tail branch.  The range check covers the VGPR offset (and the immediate), NOT soffset: the whole offset rides in `off`.
__device__ __forceinline__ u32x4 lds_dma_rsrc(const void* base_uniform, long bytes) {
    const uintptr_t pb = (uintptr_t)base_uniform;
    u32x4 r;
    r[0] = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pb);
    r[1] = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pb >> 32)) & 0xffffu;
    r[2] = (unsigned)__builtin_amdgcn_readfirstlane((unsigned)bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void lds_dma16_buf(const u32x4& rsrc, unsigned lds_dst_uniform, int off) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_dst_uniform), "v"(off), "s"(rsrc) : "memory", "m0");
}
__device__ __forceinline__ void lds_dma_drain() {
#if VITG_ASM_DMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

constexpr int ATT_BK = 64;
// NW waves per block, 32 queries per wave: BQ = 32*NW queries share one K/V staging stream (NW = 8: 256 queries, 1 block/CU;
// NW = 4: 128 queries, 2 blocks/CU)

template <int D> struct AttnCfg;
template <> struct AttnCfg<128> { static constexpr int DK = 128, DV = 128, LDK = 128, LDV = 128, CH = 16; };
// K rows: 80 used (d 72..79 zero) at a row stride of 88 elements = 176 B: 176*r mod 256 takes 16 distinct 16-B slots
// over the rows of every ds_read_b128 lane group -> conflict-free (160 B collided rows r and r+8: measured 35 % of LDS
// cycles).  V rows padded to 96 (192 B, conflict-free for the tr reads; the pad is never stored to the output).
template <> struct AttnCfg<72> { static constexpr int DK = 80, DV = 96, LDK = 88, LDV = 96, CH = 9; };

template <int D> __device__ __forceinline__ int k_lds_off(int row, int chunk) {   // bytes
    if constexpr (D == 128) return row * 256 + ((chunk ^ (row & 15)) << 4);
    else return row * 176 + (chunk << 4);
}


// Hardware block b runs on XCD b % 8 (private 4 MiB L2 each).  Logical ids are made XCD-contiguous so that all blocks reading
// the same K/V (the q-blocks of one kv head) sit on ONE XCD and re-read K/V from its L2.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

}  // namespace vitg
