// Run-to-run bit-reproducible cross-block reductions (the f32 and bf16x3 parity modes).
//
// The reference's CPU path is deterministic (guided_diffusion/nn.py:17-19 GroupNorm32, unet.py:182-213 convs on ATen's fixed
// summation order).  Floating-point atomics are not: the order in which blocks arrive changes the rounding, and the guided output
// is discontinuous in the UNet output at the clamp boundary (condition.py:231), so 1e-7 of summation-order noise decides O(1)
// differences downstream.  In the parity modes every cross-block sum is therefore made in a FIXED order:
//   1. each block stores its partial result in its own slot of a slab (plain stores, device scope),
//   2. blocks count themselves on a per-output counter; the block that arrives last (whichever it is) re-reads ALL slots and adds
//      them in slot order -- the same additions in the same order on every run -- and resets the counter for the next launch.
// No extra launch, no dependence on which block is last.  The bf16 throughput mode keeps its atomics.
#pragma once
#include <hip/hip_runtime.h>

namespace kdip {

// workspace of the deterministic reductions of one handle (one stream at a time): a slab the launches reuse one after the other,
// and counters that are zero between launches (allocated zeroed; the last block of a launch writes its counter back to zero)
struct DetWs {
  void* slab = nullptr; size_t slab_bytes = 0;
  unsigned* cnt = nullptr; int ncnt = 0;
};

#ifdef __HIPCC__
// device-scope stores / loads of slab slots: they bypass the non-coherent levels (each XCD has its own L2), independent of what
// the fences below already guarantee
__device__ __forceinline__ void det_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void det_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float det_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double det_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Block-uniform: true in exactly one block of the `expected` blocks that call it with this counter -- the last to arrive --, after
// every slot written by any of them (before its call) is visible to that block.  All threads of the block must call it.
__device__ __forceinline__ bool det_last_block(unsigned* counter, unsigned expected) {
  __shared__ unsigned det_flag_;
  __threadfence();                       // release: this thread's slot stores are visible device-wide before the count
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = old + 1u == expected ? 1u : 0u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    det_flag_ = last;
  }
  __syncthreads();
  const bool last = det_flag_ != 0u;
  if (last) __threadfence();             // acquire: the other blocks' slots
  return last;
}
#endif

}  // namespace kdip
