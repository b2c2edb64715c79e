// Device side of the s_nop cost experiment: a dependent chain of k256 multiplications / squarings per lane.
#include "fe_k256.hpp"
#ifndef KWAVES
#define KWAVES 4
#endif
using namespace ecgpu;
extern "C" __global__ void __launch_bounds__(256, KWAVES) kloop(const u32* a, const u32* b, u32* o, int iters) {
  FeK256 x, y, r;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; i++) { x.v[i] = a[t * 8 + i]; y.v[i] = b[t * 8 + i]; }
  r = x;
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    k256::mul(r, r, y);
    k256::sqr(x, r);
    k256::mul(r, x, r);
    k256::add(y, y, x);
  }
  for (int i = 0; i < 8; i++) o[t * 8 + i] = r.v[i];
}
