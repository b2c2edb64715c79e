// Measured issue peak of the integer multiplier on this GPU, for bench.py's roofline object ("peak
// measured by a dependent-free v_mad_u64_u32 microbenchmark on the same GPU").  MEASUREMENT TOOL, not part of the
// product: built into tools/ubench/libecpeak.so by __graft_entry__.build(), loaded by bench.py only.
//   mode 0: v_mad_u64_u32 alone, independent accumulator chains            -> the multiplier's issue rate
//   mode 1: v_mad_u64_u32 + v_addc_co_u32 (one exact 32x32+96 multiply-accumulate) -> what a column accumulation can reach
// Occupancy is pinned to 4 waves per SIMD (the scalar-multiplication kernels' 128-VGPR budget).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32;
typedef uint64_t u64;
#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256, 4) peak_kernel(u32* out, int iters, u32 a, u32 b) {
  u64 acc0 = threadIdx.x, acc1 = threadIdx.x + 1, acc2 = threadIdx.x + 2, acc3 = threadIdx.x + 3;
  u32 h0 = 0, h1 = 0;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\t"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");)
    } else {
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc0), "+v"(h0) : "v"(a), "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc1), "+v"(h1) : "v"(a), "v"(b) : "vcc");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc0 ^ (u32)(acc1 >> 32) ^ (u32)acc2 ^ (u32)(acc3 >> 32) ^ h0 ^ h1;
}

template <int MODE>
static int run(int device_cus, int macs_per_iter, double* tmacs) {
  const int blocks = device_cus * 4, iters = 4000;
  u32* out = nullptr;
  if (hipMalloc(&out, (size_t)blocks * 256 * 4) != hipSuccess) return -1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; r++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(peak_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u, 67890u);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) { hipFree(out); return -2; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(out);
  *tmacs = (double)blocks * 256 * iters * macs_per_iter / (best * 1e-3) / 1e12;
  return 0;
}

// TMAC/s (32x32-bit multiply-accumulates per second / 1e12) over the whole chip
extern "C" int ecpeak_measure(int device, double* mad_only_tmacs, double* mad_addc_tmacs) {
  if (hipSetDevice(device) != hipSuccess) return -1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
  int rc = run<0>(prop.multiProcessorCount, 64, mad_only_tmacs);
  if (rc) return rc;
  return run<1>(prop.multiProcessorCount, 32, mad_addc_tmacs);
}
