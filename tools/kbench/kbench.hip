// Standalone A/B harness for the k256 fast kernel: builds in ~30 s per variant (compile-time
// switches via -D), runs on device-generated inputs, prints time and an output checksum so that
// variants can be compared for speed and equality.  Development tool, not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.hpp"
using namespace ecgpu;
#ifndef KB_WAVES
#define KB_WAVES 4
#endif
#ifndef KB_BATCH
#define KB_BATCH 16
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)
int main(int argc, char** argv) {
  int lg = argc > 1 ? atoi(argv[1]) : 22;
  int reps = argc > 2 ? atoi(argv[2]) : 5;
  size_t n = (size_t)1 << lg;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  u32 *ds, *dp, *dout; uint8_t* dinf;
  CK(hipMalloc(&ds, n * 32)); CK(hipMalloc(&dp, n * 64)); CK(hipMalloc(&dout, n * 64)); CK(hipMalloc(&dinf, n));
  hipLaunchKernelGGL((synth_scalars_kernel<CurveK256>), dim3(cus * 8), dim3(256), 0, 0, (u64)0xEC5CA1A5ull, (u64)0, ds, n);
  hipLaunchKernelGGL((synth_points_kernel<CurveK256>), dim3(cus * 8), dim3(256), 0, 0, (u64)0xEC5CA1A5ull, (u64)0, dp, n);
  CK(hipDeviceSynchronize());
  TabSlotK256* ws; CK(hipMalloc(&ws, (size_t)cus * KB_WAVES * 256 * K256_TAB_SLOTS * sizeof(TabSlotK256)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k256_mul_fast_kernel<KB_BATCH, KB_WAVES>), dim3(cus * KB_WAVES), dim3(256), 0, 0, ds, dp, 0, dout, 0, dinf, n, ws);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0 && ms < best) best = ms;
  }
  std::vector<u32> h(n * 16);
  CK(hipMemcpy(h.data(), dout, n * 64, hipMemcpyDeviceToHost));
  unsigned long long cs = 0;
  for (size_t i = 0; i < n * 16; i++) cs = cs * 1000003ull + h[i];
  printf("%s waves=%d batch=%d n=2^%d best %.3f ms  %.2f M/s  checksum %016llx\n", argc > 3 ? argv[3] : "", KB_WAVES, KB_BATCH, lg, best, n / best / 1e3, cs);
  return 0;
}
