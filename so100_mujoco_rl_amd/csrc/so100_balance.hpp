// so100_balance.hpp -- which env sits in which lane slot of the persistent rollout kernel's workgroups.
//
// The multi-wave kernels synchronise their waves at workgroup barriers, so a workgroup's substep costs what its SLOWEST env
// costs: a lane whose finger pads touch the table runs a contact Newton (several row passes) while its neighbours wait, and a
// launch ends with its slowest workgroup (round 2, bench's random policy: 272 k .. 525 k cycles per step over the workgroups of one launch).
// Envs are independent and results do not depend on the lane an env is computed in, so before every rollout launch the envs are
// DEALT OUT over the workgroups like a sorted deck: in the order of their CONTACT LOAD (state row contact_load: running average of the
// substeps per launch the env spent in contact; the correlation of that count between consecutive 64-step launches is 0.64), heaviest
// first, round-robin.  Every workgroup then carries one env of every load tier.  (Round 3's first version dealt by "in contact at the
// end of the last chunk", which predicts only 29 % of the next chunk's contact substeps: +1 %.)
// One 1024-thread workgroup builds the map: a counting sort over 64 load buckets (LDS histogram, scan, scatter), ~4 us for 16384 envs.
//
// slot s = workgroup (s / epw), lane (s % epw);  slot_env[s] = env index, or -1 for a slot without an env.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace so100 {

constexpr int BALANCE_THREADS = 1024;
constexpr int BALANCE_MAX_ENVS = 16384;                     // the persistent kernel's range (one workgroup per CU x 64 envs)
constexpr int BALANCE_BUCKETS = 64;

// load: the contact_load state row (int32 bit patterns)
__global__ void __launch_bounds__(BALANCE_THREADS) so100_build_slot_map(int n, int epw, int nwg, const int32_t* __restrict__ load, int32_t* __restrict__ slot_env) {
    __shared__ int hist[BALANCE_BUCKETS], start[BALANCE_BUCKETS];
    const int tid = threadIdx.x;
    if (tid < BALANCE_BUCKETS) hist[tid] = 0;
    for (int s = tid; s < nwg*epw; s += BALANCE_THREADS) slot_env[s] = -1;
    __syncthreads();
    // bucket 0 = heaviest: a load is at most the substeps of a launch (1024 for 64 steps x 16), 16 per bucket
    auto bucket = [](int l) { const int b = (l & 0xFFFF) >> 4; return BALANCE_BUCKETS - 1 - (b > BALANCE_BUCKETS - 1 ? BALANCE_BUCKETS - 1 : b); };
    for (int e = tid; e < n; e += BALANCE_THREADS) atomicAdd(&hist[bucket(load[e])], 1);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int b = 0; b < BALANCE_BUCKETS; b++) { start[b] = t; t += hist[b]; } }
    __syncthreads();
    for (int e = tid; e < n; e += BALANCE_THREADS) {
        const int r = atomicAdd(&start[bucket(load[e])], 1);      // rank of the env in the sorted deck (the order inside a bucket does not matter)
        slot_env[(r % nwg)*epw + r / nwg] = e;                    // dealt round-robin over the workgroups: r / nwg <= (n - 1) / nwg < epw
    }
}

}  // namespace so100
