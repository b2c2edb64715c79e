// Dynamic work distribution for the persistent scalar-multiplication kernels (round 4).
//
// These kernels used to launch exactly as many workgroups as the chip holds (256 CUs x 4) and give every lane the same number of units with a grid
// stride.  Measured (tools/block_times_probe.py on a diagnostic build, profiles/r04_ab_measurements.txt set six): all 1 024 workgroups start at
// t = 0 and - with IDENTICAL work - end anywhere between 49 and 132 ms (median 89 ms).  The four waves that share a SIMD do not share its issue
// slots evenly: the favoured wave runs up to 2.7x faster and leaves early, the SIMD then runs on three, two, one wave (a lone wave cannot fill
// the multiplier pipeline: 2 waves per SIMD reach 0.91 of the throughput of 4), and at the end of a launch most of the chip idles - the time
// average of resident waves is 68 %, which is also what SQ_WAVE_CYCLES says.  Launching more, smaller workgroups helps (8 / 16 / 32 per CU:
// 124.9 / 122.0 / 120.5 against 131.5 ms for the headline) but shrinks the per-lane batch that shares an inversion and multiplies the per-lane
// workspaces.  So the grid stays persistent and the WORK moves: every wave draws chunks of 64 x u consecutive units from one atomic counter
// until the batch is exhausted, so fast waves simply do more chunks and all waves finish within about one chunk's time of each other.  The
// chunks are SMALL (two units per lane for the headline kernel, one at the very end); the kernels keep their results in a per-lane buffer
// across chunks and flush it - one shared inversion - whenever it is full, so the chunk size costs no inversion amortisation.  (A first version
// drew chunks as large as the inversion batch and shrank them with the work left: the small chunks then paid an inversion each and a slow
// wave's last large chunk was the tail - headline 128.3 against 130.8 ms, P-384 and the fixed-base kernel slower than static.)
// The exit condition is reached by every wave: a draw at or beyond n ends its loop.  The counter is zeroed on the stream before the launch.
// Not used by the constant-time kernels: which wave computes which unit would then depend on timing (never on data), and their evidence -
// instruction counters identical across scalar sets - is cleaner with the static assignment.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace ecgpu {

struct WaveSched {
  unsigned long long* counter;      // device memory, zero at launch
  unsigned long long n;             // units of the call
  unsigned int waves;               // persistent waves of the launch (gridDim.x * 4)
  unsigned int chunk_units;         // per-lane units of a full chunk
  unsigned int shrink;              // 1: chunks shrink towards one unit per lane when little is left (kernels that buffer their results and
                                    // flush them with one shared inversion whenever the buffer is full: the chunk size is free of the batch size);
                                    // 0: every chunk is a full pass (kernels whose pass shares a TABLE inversion: a small pass would pay for it again)
};

// [lo, hi) of the next chunk of this wave (wave-uniform); false when the batch is exhausted.  Called by all lanes of the wave together.
__device__ __forceinline__ bool wave_next_chunk(const WaveSched& s, size_t& lo, size_t& hi) {
  unsigned long long start = 0, take = 0;
  if ((threadIdx.x & 63u) == 0) {
    unsigned long long u = s.chunk_units;
    if (s.shrink) {
      const unsigned long long cur = __hip_atomic_load(s.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long rem = cur < s.n ? s.n - cur : 0ull;
      const unsigned long long g = rem / ((unsigned long long)s.waves * 64ull * 2ull);      // half an even share of what is left
      u = g < 1ull ? 1ull : (g > u ? u : g);
    }
    take = u * 64ull;
    start = atomicAdd(s.counter, take);
  }
  const unsigned int s_lo = __builtin_amdgcn_readfirstlane((unsigned int)start), s_hi = __builtin_amdgcn_readfirstlane((unsigned int)(start >> 32));
  const unsigned int t_lo = __builtin_amdgcn_readfirstlane((unsigned int)take);
  start = ((unsigned long long)s_hi << 32) | s_lo;
  lo = (size_t)start;
  hi = (size_t)(start + t_lo);
  if (hi > s.n) hi = (size_t)s.n;
  return start < s.n;
}

}  // namespace ecgpu
