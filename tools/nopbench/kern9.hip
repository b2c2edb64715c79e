// Prototype of a k256 multiplication on 9 unsaturated limbs of 29 bits (carry-free product columns), timed as a
// dependent chain like kern.hip.  Not used by the library: an experiment for DESIGN.md section 2.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32; typedef uint64_t u64;
#define M29 0x1FFFFFFFu
// limbs < 2^30 in, limbs < 2^29 + small out.  2^261 = 2^5 * 2^256 = 2^37 + 31264 (mod p)
__device__ __forceinline__ void mul9(u32* r, const u32* a, const u32* b) {
  u32 h[9];
  u64 acc = 0;
  // high columns 9..16 first (digits h[0..7]), h[8] = what is left
#pragma unroll
  for (int k = 9; k <= 16; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) { const int j = k - i; if (j >= 0 && j < 9) acc += (u64)a[i] * b[j]; }
    h[k - 9] = (u32)acc & M29; acc >>= 29;
  }
  const u64 htop = acc;            // digit 17 and beyond, < 2^36
  // column 8 carries into column 9: handled by doing the low columns first would need h; instead fold the low-half carry at the end
  u32 t[9];
  acc = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) acc += (u64)a[i] * b[k - i];
    acc += (u64)h[k < 8 ? k : 8] * 0u;                      // placeholder keeps the shape; real fold below
    t[k] = (u32)acc & M29; acc >>= 29;
  }
  // acc = carry out of column 8 (belongs to column 9): add to the high digits' value
  // value = T + 2^261 * (H + acc) with H = h[0..7] + htop * 2^(29*8)
  // fold: e_k = t_k + 31264 * H_k + 256 * H_{k-1}
  u64 H[9];
#pragma unroll
  for (int k = 0; k < 8; k++) H[k] = h[k];
  H[8] = htop;
  H[0] += acc;
  u64 c = 0;
#pragma unroll
  for (int k = 0; k < 9; k++) {
    u64 e = (u64)t[k] + H[k] * 31264u + (k ? (H[k - 1] << 8) : 0) + c;
    r[k] = (u32)e & M29; c = e >> 29;
  }
  // limb 9: c + H[8] << 8 -> fold once more into limbs 0, 1, 2
  u64 top = c + (H[8] << 8);
  u64 e0 = (u64)r[0] + top * 31264u;
  r[0] = (u32)e0 & M29;
  u64 e1 = (u64)r[1] + (top << 8) + (e0 >> 29);
  r[1] = (u32)e1 & M29;
  r[2] += (u32)(e1 >> 29);
}
extern "C" __global__ void __launch_bounds__(256, 4) kloop(const u32* a, const u32* b, u32* o, int iters) {
  u32 x[9], y[9], r[9];
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < 8; i++) { x[i] = a[t * 8 + i] & M29; y[i] = b[t * 8 + i] & M29; }
  x[8] = 1; y[8] = 2;
  for (int i = 0; i < 9; i++) r[i] = x[i];
#pragma unroll 1
  for (int it = 0; it < iters; it++) {
    mul9(r, r, y);
    mul9(x, r, r);
    mul9(r, x, r);
#pragma unroll
    for (int i = 0; i < 9; i++) y[i] = (y[i] + x[i]) & M29;      // keep limbs bounded (stand-in for a lazy add + rare normalise)
  }
  for (int i = 0; i < 8; i++) o[t * 8 + i] = r[i] ^ (r[8] << i);
}
