// Instruction-mix microbenchmark: cost of v_mad_u64_u32 alone and paired with other VALU instructions,
// at 4 waves per SIMD (the occupancy of the scalar-multiplication kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256, 4) mix(u32* out, int iters, u32 a, u32 b) {
  u64 acc0 = threadIdx.x, acc1 = threadIdx.x + 1;
  u32 h0 = 0, h1 = 0, m0 = a, m1 = b, m2 = 3, m3 = 4;
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {   // mad only, two independent chains
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(acc0), "+v"(acc1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 1) {   // mad + addc (the MAC)
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc0), "+v"(h0) : "v"(m0), "v"(m1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc1), "+v"(h1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 2) {   // mad + v_mov
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mov_b32 %1, %2" : "+v"(acc0), "=v"(h0) : "v"(m0), "v"(m1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mov_b32 %1, %3" : "+v"(acc1), "=v"(h1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 3) {   // mad + v_add_u32 (no carry out)
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %2" : "+v"(acc0), "+v"(h0) : "v"(m0), "v"(m1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %1, %3" : "+v"(acc1), "+v"(h1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 4) {   // mad + addc + 2 v_mov
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %2, %4" : "+v"(acc0), "+v"(h0), "=v"(m2) : "v"(m0), "v"(m1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %2, %4" : "+v"(acc1), "+v"(h1), "=v"(m3) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 5) {   // addc only (carry in and out)
      REP16(asm volatile("v_addc_co_u32 %0, vcc, %2, %0, vcc\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(h0), "+v"(h1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 6) {   // v_mov only
      REP16(asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(h0), "=v"(h1) : "v"(m0), "v"(m1));)
    } else if (MODE == 7) {   // mad + v_add_co (carry out only)
      REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_co_u32 %1, vcc, %1, %2" : "+v"(acc0), "+v"(h0) : "v"(m0), "v"(m1) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_co_u32 %1, vcc, %1, %3" : "+v"(acc1), "+v"(h1) : "v"(m0), "v"(m1) : "vcc");)
    } else if (MODE == 8) {   // v_mul_lo + v_mul_hi + add_co + addc x2 (product via separate halves)
      REP16(asm volatile("v_mul_lo_u32 %2, %4, %5\n\tv_mul_hi_u32 %3, %4, %5\n\tv_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(h0), "+v"(h1), "=&v"(m2), "=&v"(m3) : "v"(m0), "v"(m1) : "vcc");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)acc0 ^ (u32)(acc1 >> 32) ^ h0 ^ h1 ^ m2 ^ m3;
}
template <int MODE>
static void run(const char* name, int per_iter) {
  u32* out; hipMalloc(&out, 1024 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mix<MODE>, dim3(1024), dim3(256), 0, 0, out, iters, 12345u, 67890u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // per SIMD: 4 waves x iters x per_iter instructions
  const double instr = 4.0 * iters * per_iter;
  printf("%-44s %8.3f ms  %6.2f cycles per wave-instr per SIMD (2.4 GHz)  %6.2f cycles per group\n", name, best, best * 1e-3 * 2.4e9 / instr,
         best * 1e-3 * 2.4e9 / (4.0 * iters * 32));
  hipFree(out);
}
int main() {
  run<0>("mad only", 32);
  run<1>("mad + addc", 64);
  run<2>("mad + v_mov", 64);
  run<3>("mad + v_add_u32", 64);
  run<7>("mad + v_add_co_u32", 64);
  run<4>("mad + addc + 2 v_mov", 128);
  run<5>("addc only", 32);
  run<6>("v_mov only", 32);
  run<8>("mul_lo + mul_hi + add_co + addc (per 4)", 64);
  return 0;
}
