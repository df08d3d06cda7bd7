// dot_common.h -- the single-launch reduction both dot kernels end with (CloverVector4::dot FAST in vector4.hip, CloverVector8::dot FAST in
// mixed8.hip): fixed-order workgroup sum, hand-over of the workgroup partials through zero-initialised slots, and the collector
// workgroup that runs the fixed final tree.  See k_v4_dot_fast1 for why it is built without any atomic read-modify-write.
#pragma once

#include "common.h"

#define DOT_FAST_THREADS 256

__device__ __forceinline__ float block_sum_256(float v, float *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// grid <= DOT_FAST_THREADS * DOT_MAX_SLOTS_PER_THREAD workgroups (the callers' grids are at most 4 per CU)
#define DOT_MAX_SLOTS_PER_THREAD 8

// Called by EVERY thread of every workgroup with the workgroup's partial t (thread 0's value counts).  Thread 0 publishes {valid = 1, t} in
// slot blockIdx.x with one agent-scope 8-byte store -- partial and flag arrive together -- and the LAST workgroup of the grid (dispatched
// last, so every other one is running or done: no deadlock whatever else shares the device) collects the slots: thread k adds slots k,
// k + 256, ... in that order, then block_sum_256 -- one fixed tree, whatever order the workgroups finish in.  The collector clears every
// slot it has read, so the slots are zero again when the kernel ends (clv_internal_sync_slots' contract; a captured graph replays).
__device__ __forceinline__ void dot_hand_over_and_collect(float t, unsigned long long *slots, float *__restrict__ out, float *sh)
{
    if (threadIdx.x == 0)
        __hip_atomic_store(&slots[blockIdx.x], (1ull << 32) | __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x != gridDim.x - 1) return;
    // collector: every round requests all of this thread's outstanding slots at once
    constexpr int MAXS = DOT_MAX_SLOTS_PER_THREAD;
    const int count = (int)gridDim.x;
    float part[MAXS];
    uint32_t need = 0;
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
        part[k] = 0.0f;
        if ((int)threadIdx.x + DOT_FAST_THREADS * k < count) need |= 1u << k;
    }
    uint32_t spins = 0;
    unsigned long long t_start = 0;
    while (need) {
        unsigned long long v[MAXS];
#pragma unroll
        for (int k = 0; k < MAXS; k++)
            if (need & (1u << k)) v[k] = __hip_atomic_load(&slots[threadIdx.x + DOT_FAST_THREADS * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < MAXS; k++)
            if ((need & (1u << k)) && (v[k] >> 32)) {
                part[k] = __uint_as_float((uint32_t)v[k]);
                need &= ~(1u << k);
            }
        if (need) {
            __builtin_amdgcn_s_sleep(1);
            // bounded: HIP does not promise that every other workgroup of the grid has been dispatched when the last one runs (it holds on
            // this hardware); 4 s on the 100 MHz wall clock without a slot turn a scheduling anomaly into an error instead of a hang
            if ((++spins & 4095u) == 0) {
                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                if (!t_start) t_start = now;
                else if (now - t_start > 400000000ull) __builtin_trap();
            }
        }
    }
    float acc2 = 0.0f;
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
        acc2 += part[k];                                                       // slots beyond `count` contribute +0.0f: acc2 + 0 == acc2 bit for bit
        // zero again for the next launch: plain stores (nobody reads the slots any more in this launch; the end of the kernel writes them back)
        if ((int)threadIdx.x + DOT_FAST_THREADS * k < count) slots[threadIdx.x + DOT_FAST_THREADS * k] = 0ull;
    }
    __syncthreads();                                                           // sh is reused
    const float r = block_sum_256(acc2, sh);
    if (threadIdx.x == 0) *out = r;
}
