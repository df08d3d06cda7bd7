// thresh_device.h -- device helpers shared by the threshold kernels (threshold4.hip) and the persistent IHT kernel (iht_persist.hip):
// the DPP wave scan and the SWAR nibble-magnitude operations of the one-workgroup radix select (CloverVector4::threshold,
// CloverVector4.h:1913-1975).
#pragma once

#include "common.h"

// inclusive scan over the 64 lanes of a wave with DPP moves (no LDS crossbar): Hillis-Steele inside each row of 16,
// then lane 15 of rows 0 and 2 into rows 1 and 3, then lane 31 into rows 2 and 3
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
#define DPP_ADD(ctrl, row_mask) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, row_mask, 0xF, false)
    DPP_ADD(0x111, 0xF);      // row_shr:1
    DPP_ADD(0x112, 0xF);      // row_shr:2
    DPP_ADD(0x114, 0xF);      // row_shr:4
    DPP_ADD(0x118, 0xF);      // row_shr:8
    DPP_ADD(0x142, 0xA);      // row_bcast:15 -> rows 1, 3
    DPP_ADD(0x143, 0xC);      // row_bcast:31 -> rows 2, 3
#undef DPP_ADD
    return v;
}

__device__ __forceinline__ uint32_t cand_key(float s7, int m) { return __float_as_uint(__builtin_fabsf(s7 * (float)m)); }

// element e of a word sits in nibble e after the two nibbles of every byte are swapped (even elements are stored high)
__device__ __forceinline__ uint32_t swap_nibbles(uint32_t w) { return ((w & 0x0F0F0F0Fu) << 4) | ((w >> 4) & 0x0F0F0F0Fu); }
// |two's complement nibble| for all 8 nibbles: 0..8, no carries between nibbles
__device__ __forceinline__ uint32_t abs_nibbles(uint32_t w)
{
    const uint32_t sgn = (w >> 3) & 0x11111111u;
    return (w ^ (sgn * 0xFu)) + sgn;
}
// bit 3 of every nibble whose value (0..8) is >= t, t in 0..9
__device__ __forceinline__ uint32_t ge_nibbles(uint32_t ab, uint32_t t)
{
    if (t == 0) return 0x88888888u;
    if (t > 8) return 0u;
    return (ab + (8u - t) * 0x11111111u) & 0x88888888u;
}
// bit 3 of each of the first `count` nibbles (count 0..8)
__device__ __forceinline__ uint32_t first_nibbles(uint32_t count) { return count >= 8 ? 0x88888888u : 0x88888888u & ((1u << (4 * count)) - 1u); }

// one word's contribution to its block's table: field m += #(|nibble| == m) over the first `valid` elements of the word
__device__ __forceinline__ unsigned long long th4_count_word(uint32_t w, uint32_t valid)
{
    const uint32_t ab = abs_nibbles(swap_nibbles(w));
    unsigned long long acc = 0;
#pragma unroll
    for (uint32_t e = 0; e < 8; e++)
        if (e < valid) acc += 1ull << (7u * ((ab >> (4 * e)) & 0xFu));
    return acc;
}

