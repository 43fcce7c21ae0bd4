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
