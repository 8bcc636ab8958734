// Run-to-run bit-reproducible cross-block reductions (the f32 and bf16x3 parity modes).
//
// The reference's CPU path is deterministic (guided_diffusion/nn.py:17-19 GroupNorm32, unet.py:182-213 convs on ATen's fixed
// summation order).  Floating-point atomics are not: the order in which blocks arrive changes the rounding, and the guided output
// is discontinuous in the UNet output at the clamp boundary (condition.py:231), so 1e-7 of summation-order noise decides O(1)
// differences downstream.  In the parity modes every cross-block sum is therefore made in a FIXED order, in two stages:
//   1. each block of the producing kernel stores its partial result in its own slot of a slab (plain stores);
//   2. a small finish kernel (one block per image) adds the slots in a fixed tree -- the same additions in the same order on every
//      run -- and WRITES the result.  The kernel boundary is the only synchronisation: no fences, no counters, no atomics.
// (An in-kernel variant -- the last block to arrive does the final sum -- was built first: it needs an agent-scope release fence
// in every block (an L2 write-back per block: each XCD has its own L2) and serialises the final sum behind the slowest block; the
// dominant 480 us conv took 1100 us with it.)  The bf16 throughput mode keeps its floating-point atomics.
#pragma once
#include <stddef.h>

namespace kdip {

// slab of one launch's block partials (the handle's scratch arena: transient, one stream at a time)
struct DetWs {
  void* slab = nullptr; size_t slab_bytes = 0;
};

// A fused-statistics finish pass that the producing conv did NOT launch (ConvStats::defer): the slab (which must then outlive the launch: the
// handle's persist arena) and the launch geometry the finish needs.  The consumer of the sums runs it -- alone (conv_stats_finish) or together with
// the GroupNorm coefficient kernel it feeds (conv_stats_finish_coef: one launch instead of two in the dependency chain between two convs).
struct DetPending {
  const void* slab = nullptr; int tpi = 0, nv = 0, cpg = 0, gy = 1;
};

}  // namespace kdip
