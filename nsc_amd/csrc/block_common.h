// Device helpers shared by the gated-block kernels (block.hip: exact fp32 MFMA; block_split.hip: bf16 matrix cores on split operands):
// probe stamps, cache-policy bits of the streaming traffic, and the neighbour flags of the pair launches (described in block.hip).
#pragma once
#include "block_args.h"

#ifdef NSC_PROBES
// phase stamps of workgroup 0 / waves 0 and 4 (s_memtime), read back with nsc_probe_read (block.hip) / nsc_probe_read_split
// (block_split.hip): profiling builds only.  One array per translation unit.
static __device__ unsigned long long nsc_dbg_stamps[128];     // [0, 64): wave 0, [64, 128): wave 4 of workgroup 0
#define NSC_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) nsc_dbg_stamps[(threadIdx.x >> 8) * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define NSC_STAMP(i) do { } while (0)
#endif
#if defined(NSC_EXP) && (NSC_EXP & 128)
#define NSC_AUX_COHERENT 0        /* timing experiment (NOT coherent): ordinary loads */
#else
#define NSC_AUX_COHERENT 0x11     /* raw buffer load aux bits on gfx940+: bit 0 = sc0, bit 4 = sc1 */
#endif
// Cache policy of the streaming traffic of the forward / data-gradient kernels (profiles/r04h_store_flavours.txt).  Stores of
// tensors nobody reads before the backward pass / the tail of the step (saved activations, da, dz1) are NONTEMPORAL: they do not sit
// dirty in the eight L2s until the end-of-kernel write-back (-1.0 % of the step; written through with sc0 sc1: -0.5 %).  The tile
// prefetch loads are nontemporal too (-0.7 %).  Measured and not kept: the block's main output (out / dx, read by the next launch)
// nontemporal (+-0); nontemporal operand loads in the weight-gradient kernels (+0.3 %) and in the convs outside the blocks (+-0.1 %).
#define NSC_AUX_LATE 2            /* raw buffer aux bits: bit 1 = nt */
#define NSC_AUX_STREAM 2
__device__ __forceinline__ void nsc_store4_late(float* gp, const f32x4& v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(gp)); }

__device__ __forceinline__ void nsc_pair_publish(int* flags) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores are acknowledged: they are in memory
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// timeouts: ONE caller-owned int that the library only ever adds to (never cleared here): a neighbour wait that gave up means the
// second block read unpublished data, and the count must survive whatever zeroes the per-launch flags (the engine: every step).
#define NSC_PAIR_MAX_WG 256       /* flags[0, 256): one per workgroup of a pair launch (grid <= 256) */
__device__ __forceinline__ void nsc_pair_wait(int* flags, int* timeouts) {
  if (threadIdx.x < 2) {
    const int nb = (int)blockIdx.x + (threadIdx.x == 0 ? -1 : 1);
    if (nb >= 0 && nb < (int)gridDim.x) {
      int it = 0;
      while (__hip_atomic_load(flags + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(4);
        if (++it > (1 << 20)) {                       // ~0.1 s: never in a healthy launch
          atomicAdd(timeouts, 1);
          break;
        }
      }
    }
  }
  nsc_lds_barrier();      // (not __syncthreads(): its vmcnt(0) would drain the second block's weight loads, which were issued before
                          //  this wait precisely so that they and the first tile's loads - issued right after it - are in flight together)
}

