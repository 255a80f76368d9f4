// so100_balance.hpp -- which env sits in which lane slot of the persistent rollout kernel's workgroups.
//
// The multi-wave kernels synchronise their waves at workgroup barriers, so a workgroup's substep costs what its SLOWEST env
// costs: a lane whose finger pads touch the table runs a contact Newton (several row passes) while its neighbours wait, and a
// launch ends with its slowest workgroup.  Under the bench's random policy the workgroups of one launch differ by 1.9x (272 k ..
// 525 k cycles per step, profiles/r02_final_phase_ref_random.txt: 7 of 16 envs in contact in the median workgroup, 10-12 in the
// slowest ones).  Envs are independent and results do not depend on the lane an env is computed in, so before every rollout launch
// the envs that were in pad contact at the end of the previous chunk (state row contact_stat) are DEALT OUT over the workgroups like
// cards -- contact envs first, then the rest -- instead of sitting wherever their index puts them.  Every workgroup then carries the
// same number of contact-prone envs (+-1).  One 1024-thread workgroup builds the map (a block scan over <= 16384 envs: ~3 us).
//
// slot s = workgroup (s / epw), lane (s % epw);  slot_env[s] = env index, or -1 for a slot without an env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so100 {

constexpr int BALANCE_THREADS = 1024;
constexpr int BALANCE_MAX_ENVS = 16384;                     // the persistent kernel's range (one workgroup per CU x 64 envs)

// cstat: the contact_stat state row (int32 bit patterns: contacts | dropped << 8) -- "busy" = had a pad contact in the last env step
__global__ void __launch_bounds__(BALANCE_THREADS) so100_build_slot_map(int n, int epw, int nwg, const int32_t* __restrict__ cstat, int32_t* __restrict__ slot_env) {
    __shared__ int wsum[BALANCE_THREADS/64];
    __shared__ int total_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = BALANCE_MAX_ENVS/BALANCE_THREADS;   // consecutive envs per thread
    const int e0 = tid*PER;
    int flags = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int e = e0 + k;
        const bool busy = e < n && (cstat[e] & 255) != 0;
        flags |= busy ? 1 << k : 0; mine += busy ? 1 : 0;
    }
    // exclusive scan of `mine` over the block: wave scan (DPP-free: shuffles), then the 16 wave totals through LDS
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) wsum[wave] = incl;
    for (int s = tid; s < nwg*epw; s += BALANCE_THREADS) slot_env[s] = -1;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < BALANCE_THREADS/64; w++) { const int v = wsum[w]; wsum[w] = t; t += v; } total_s = t; }
    __syncthreads();
    const int busy_before = wsum[wave] + incl - mine, total_busy = total_s;
    int rb = busy_before;                                    // rank among the busy envs
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int e = e0 + k;
        if (e >= n) break;
        const bool busy = (flags >> k) & 1;
        const int r = busy ? rb : total_busy + (e - rb);     // busy envs take ranks 0 .. B-1, the others B .. n-1 (both in env order)
        rb += busy ? 1 : 0;
        slot_env[(r % nwg)*epw + r / nwg] = e;               // dealt round-robin over the workgroups: r / nwg <= (n - 1) / nwg < epw
    }
}

}  // namespace so100
