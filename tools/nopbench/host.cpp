// Loads a code object and times kloop: ./host file.hsaco [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
  int iters = argc > 2 ? atoi(argv[2]) : 2000;
  hipModule_t m; hipFunction_t f;
  CK(hipModuleLoad(&m, argv[1]));
  CK(hipModuleGetFunction(&f, m, "kloop"));
  const int per_cu = argc > 3 ? atoi(argv[3]) : 4;      // workgroups of 4 waves per CU = waves per SIMD
  const int blocks = 256 * per_cu, threads = 256;
  size_t n = (size_t)blocks * threads;
  std::vector<unsigned> h(n * 8);
  for (size_t i = 0; i < n * 8; i++) h[i] = 0x9E3779B9u * (unsigned)(i + 1) + 12345u;
  unsigned *a, *b, *o;
  CK(hipMalloc(&a, n * 32)); CK(hipMalloc(&b, n * 32)); CK(hipMalloc(&o, n * 32));
  CK(hipMemcpy(a, h.data(), n * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), n * 32, hipMemcpyHostToDevice));
  void* args[] = {&a, &b, &o, &iters};
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0));
    CK(hipModuleLaunchKernel(f, blocks, 1, 1, threads, 1, 1, 0, 0, args, nullptr));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> out(16);
    CK(hipMemcpy(out.data(), o, 64, hipMemcpyDeviceToHost));
    printf("%s waves/SIMD=%d iters=%d  %.3f ms  %.2f G modmul/s  chk=%08x%08x\n", argv[1], per_cu, iters, ms, 3.0 * iters * n / ms / 1e6, out[0], out[9]);
  }
  return 0;
}
