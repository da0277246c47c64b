// Pieces shared by the convolution epilogue and the BatchNorm kernels: the "last block to arrive finishes the
// reduction" hand-off, and the row reductions that turn per-block partial sums into BatchNorm statistics.
#pragma once
#include "common.h"

// A value another workgroup of the same launch will read: stored write-through (agent scope), so that no
// release fence - a write-back of the whole L2 on this part - is needed before the ticket.
__device__ __forceinline__ void ried_publish(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// True, in every thread, in exactly one block of the grid: the one whose ticket is the last.  Everything the
// other blocks stored with ried_publish() before calling is visible to that block's plain loads afterwards.  No
// block waits for another, so the hand-off cannot deadlock whatever the residency of the grid.  The counter must
// be zero before the launch; the last block puts it back to zero.
//   writers: write-through stores drained by every wave (vmcnt) -> block barrier -> ticket (relaxed, agent)
//   last:    one agent-scope acquire -> block barrier -> plain loads
__device__ __forceinline__ bool ried_last_block(unsigned* counter, unsigned nblocks, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int last = t == nblocks - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// Grid-wide barrier for launches whose blocks are all resident at once (the callers bound the grid well below the
// chip's wave slots, and kernels running beside it on other streams finish on their own, so every block does get
// a slot).  counter: zero before the launch, left at nblocks - a word of an arena that is cleared once per pass.
// What the blocks exchange across it must travel as agent-scope atomics on both sides (float atomic adds before,
// ried_peek() after): no fence is issued, so plain stores are NOT ordered by it.  The wait is bounded: a block
// that gives up counts itself in ried_barrier_timeouts and goes on, so the grid always drains.
static __device__ unsigned ried_barrier_timeouts_dev;      // one per translation unit
#define RIED_BARRIER_MAX_POLLS (1 << 22)
__device__ __forceinline__ void ried_grid_barrier(unsigned* counter, unsigned nblocks) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int polls = 0;
    while (t < nblocks && polls < RIED_BARRIER_MAX_POLLS) {
      __builtin_amdgcn_s_sleep(1);
      t = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ++polls;
    }
    if (t < nblocks) __hip_atomic_fetch_add(&ried_barrier_timeouts_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
}
// A value other blocks of this launch produced with agent-scope atomics before ried_grid_barrier().
__device__ __forceinline__ float ried_peek(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct BnFinalize {          // train-mode BatchNorm2d statistics from [nrows][2][C] partial rows
  unsigned* counter;         // null: no fused finalize
  float* save;               // [2][C]: mean, 1/sqrt(var + eps)
  float* running_mean;       // may be null
  float* running_var;
  long long* nbt;            // num_batches_tracked, may be null
  float count, eps, momentum;
  int nrows;
};

// One 256-thread block: for every channel, sum the rows and write the statistics.  lds: 512 floats.
__device__ __forceinline__ void bn_finalize_rows(const float* __restrict__ stats, int C, const BnFinalize& f,
                                                 float* lds) {
  const int cl = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int r = ty; r < f.nrows; r += 8) {
        a += stats[(long long)r * 2 * C + c];
        b += stats[(long long)r * 2 * C + C + c];
      }
    }
    lds[ty * 32 + cl] = a;
    lds[256 + ty * 32 + cl] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0 += lds[k * 32 + cl]; s1 += lds[256 + k * 32 + cl]; }
      float mean = s0 / f.count;
      float var = fmaxf(s1 / f.count - mean * mean, 0.f);
      f.save[c] = mean;
      f.save[C + c] = rsqrtf(var + f.eps);
      if (f.running_mean) {
        float unb = f.count > 1.f ? var * (f.count / (f.count - 1.f)) : var;
        f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * mean;
        f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * unb;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0 && f.nbt) f.nbt[0] += 1;
}

// One 256-thread block: partial rows [nrows][2][C] of the BatchNorm backward sums -> fin [2][C] and the
// parameter gradients.  lds: 512 floats.
__device__ __forceinline__ void bn_bwd_finish_rows(const float* __restrict__ part, int nrows, int C,
                                                   float* __restrict__ fin, float* dgamma, float* dbeta,
                                                   int accumulate, float* lds) {
  const int cl = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int c0 = 0; c0 < C; c0 += 32) {
    const int c = c0 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int r = ty; r < nrows; r += 8) {
        a += part[(long long)r * 2 * C + c];
        b += part[(long long)r * 2 * C + C + c];
      }
    }
    lds[ty * 32 + cl] = a;
    lds[256 + ty * 32 + cl] = b;
    __syncthreads();
    if (ty == 0 && c < C) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { s0 += lds[k * 32 + cl]; s1 += lds[256 + k * 32 + cl]; }
      fin[c] = s0; fin[C + c] = s1;
      if (dgamma) {
        dbeta[c] = accumulate ? dbeta[c] + s0 : s0;
        dgamma[c] = accumulate ? dgamma[c] + s1 : s1;
      }
    }
    __syncthreads();
  }
}
