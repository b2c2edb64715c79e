// VALU instruction-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: the scalar-multiplication hot path is bound by integer multiply
// throughput, whose rate is not documented.  This program measures, per SIMD,
// the issue cost (cycles per wave64 instruction) of every candidate primitive
// at 1/2/4/8 waves per SIMD, so that the field-multiplication design (limb
// width, mad vs fp64 tricks) is chosen from measurements, and so that
// `roofline.peak` in bench.py is a measured number.
//
// Build:  hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
// Run:    ./valu_rates            (prints one line per op x occupancy)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

enum Op {
  OP_ADD_U32, OP_FMA_F32, OP_MAD_U64_U32, OP_MAD_U64_U32_DEP, OP_MUL_LO_U32, OP_MUL_HI_U32,
  OP_MAD_U32_U24, OP_MUL_HI_U32_U24, OP_ADDC_CHAIN, OP_ADD3_U32, OP_LSHL_ADD_U64,
  OP_FMA_F64, OP_ADD_F64, OP_DOT2_U32_U16, OP_DOT4_U32_U8, OP_ALIGNBIT, OP_CNDMASK,
  OP_LSHRREV_B64, OP_AND_OR, OP_MIX_MAD1_ADDC1, OP_MIX_MAD1_ADD2, OP_MIX_MAD1_ADD4, OP_MIX_MAD1_ADD8,
  OP_MAD_U32_U16, OP_MIX_FMA64_ADD2, OP_MAD_U64_SGPRCARRY, OP_COUNT
};

static const char* op_name[OP_COUNT] = {
  "v_add_u32", "v_fma_f32", "v_mad_u64_u32", "v_mad_u64_u32(dependent)", "v_mul_lo_u32", "v_mul_hi_u32",
  "v_mad_u32_u24", "v_mul_hi_u32_u24", "v_add_co+v_addc_co", "v_add3_u32", "v_lshl_add_u64",
  "v_fma_f64", "v_add_f64", "v_dot2_u32_u16", "v_dot4_u32_u8", "v_alignbit_b32", "v_cndmask_b32",
  "v_lshrrev_b64", "v_and_or_b32", "mix 1mad+1addc", "mix 1mad+2add", "mix 1mad+4add", "mix 1mad+8add",
  "v_mad_u32_u16", "mix 1fma64+2add", "v_mad_u64_u32(sgpr carry-out)"
};
// VALU instructions issued per "unit" (one macro expansion below), for cycles/instr.
static const int op_instrs[OP_COUNT] = {8,8,8,8,8,8, 8,8,8,8,8, 8,8,8,8,8,8, 8,8, 16, 24, 40, 72, 8, 24, 8};
// multiply-accumulates (32x32) per unit, for MAC-rate reporting (0 = n/a)
static const int op_macs[OP_COUNT]   = {0,0,8,8,8,8, 0,0,0,0,0, 0,0,0,0,0,0, 0,0, 8,8,8,8, 0,0,8};

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) bench(unsigned* out, int iters, unsigned long long* cyc) {
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned a = tid * 2654435761u + 12345u, b = tid * 40503u + 977u;
  unsigned x[8];
  unsigned long long q[8];
  double d[8];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { x[i] = a + i; q[i] = (unsigned long long)(b + i) << 7; d[i] = (double)(a + i); f[i] = (float)(b + i); }
  double da = (double)(a & 0xfffff), db = (double)(b & 0xfffff);
  float fa = (float)(a & 0xfff), fb = 1.0f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if constexpr (OP == OP_ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
        R8(X)
#undef X
      } else if constexpr (OP == OP_FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f[i]) : "v"(fa), "v"(fb));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MAD_U64_U32) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MAD_U64_SGPRCARRY) {
#define X(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(q[i]) : "v"(a), "v"(b) : "s20", "s21");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MAD_U64_U32_DEP) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[0]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MUL_HI_U32) {
#define X(i) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(x[i]) : "v"(a));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MAD_U32_U24) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MAD_U32_U16) {
#define X(i) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MUL_HI_U32_U24) {
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(x[i]) : "v"(a));
        R8(X)
#undef X
      } else if constexpr (OP == OP_ADDC_CHAIN) {
        // four (add_co, addc_co) pairs = 8 instructions, the carry travels through vcc
#define X(i) asm volatile("v_add_co_u32 %0, vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, %3, %1, vcc" : "+v"(x[2*i]), "+v"(x[2*i+1]) : "v"(a), "v"(b) : "vcc");
        X(0) X(1) X(2) X(3)
#undef X
      } else if constexpr (OP == OP_ADD3_U32) {
#define X(i) asm volatile("v_add3_u32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_LSHL_ADD_U64) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(q[i]) : "v"(q[(i+1)&7]));
        R8(X)
#undef X
      } else if constexpr (OP == OP_FMA_F64) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
        R8(X)
#undef X
      } else if constexpr (OP == OP_ADD_F64) {
#define X(i) asm volatile("v_add_f64 %0, %1, %0" : "+v"(d[i]) : "v"(da));
        R8(X)
#undef X
      } else if constexpr (OP == OP_DOT2_U32_U16) {
#define X(i) asm volatile("v_dot2_u32_u16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_DOT4_U32_U8) {
#define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_ALIGNBIT) {
#define X(i) asm volatile("v_alignbit_b32 %0, %1, %0, 13" : "+v"(x[i]) : "v"(a));
        R8(X)
#undef X
      } else if constexpr (OP == OP_CNDMASK) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_LSHRREV_B64) {
#define X(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[i]));
        R8(X)
#undef X
      } else if constexpr (OP == OP_AND_OR) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        R8(X)
#undef X
      } else if constexpr (OP == OP_MIX_MAD1_ADDC1) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(q[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MIX_MAD1_ADD2) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1" : "+v"(q[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MIX_MAD1_ADD4) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1" : "+v"(q[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MIX_MAD1_ADD8) {
#define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1\n\tv_add_u32 %1, %2, %1\n\tv_add_u32 %1, %3, %1" : "+v"(q[i]), "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
        R8(X)
#undef X
      } else if constexpr (OP == OP_MIX_FMA64_ADD2) {
#define X(i) asm volatile("v_fma_f64 %0, %2, %3, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %1, %5, %1" : "+v"(d[i]), "+v"(x[i]) : "v"(da), "v"(db), "v"(a), "v"(b));
        R8(X)
#undef X
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) acc += x[i] + (unsigned)q[i] + (unsigned)(q[i] >> 32) + (unsigned)d[i] + (unsigned)f[i];
  out[tid] = acc;
  if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

typedef void (*kern_t)(unsigned*, int, unsigned long long*);
template <int OP> struct Table { static void fill(kern_t* t) { t[OP] = bench<OP>; Table<OP + 1>::fill(t); } };
template <> struct Table<OP_COUNT> { static void fill(kern_t*) {} };

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("# device %s  CUs %d  clockRate %d kHz\n", prop.name, cus, prop.clockRate);
  kern_t tab[OP_COUNT]; Table<0>::fill(tab);
  const int max_blocks = cus * 8;
  unsigned* out; unsigned long long* cyc;
  CK(hipMalloc(&out, (size_t)max_blocks * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)max_blocks * 4 * 8));
  std::vector<unsigned long long> h(max_blocks * 4);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("# op | waves/SIMD | cycles per wave-instr per SIMD (s_memtime, 100MHz ticks scaled?) | wall ms | Ginstr/s chip | GMAC/s chip\n");
  for (int op = 0; op < OP_COUNT; op++) {
    for (int w = 1; w <= 8; w *= 2) {
      int blocks = cus * w;  // 256-thread blocks: w blocks per CU -> w waves per SIMD
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, 50, cyc);  // warm-up
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(tab[op], dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < blocks * 4; i++) sum += (double)h[i];
      double ticks = sum / (blocks * 4);                       // s_memtime ticks per wave
      double n_instr = (double)iters * 4 * op_instrs[op];      // wave-instructions per wave
      double total_instr = n_instr * blocks * 4;               // wave-instructions, whole chip
      double ginstr = total_instr / (ms * 1e-3) / 1e9;         // wave-instr/s
      double simd_cycles_at_2p4 = (ms * 1e-3) * 2.4e9;         // cycles elapsed at 2.4 GHz
      double cyc_per_instr_simd = simd_cycles_at_2p4 / (n_instr * w);  // per SIMD, if clock were 2.4 GHz
      double gmac = op_macs[op] ? (double)iters * 4 * op_macs[op] * 64.0 * blocks * 4 / (ms * 1e-3) / 1e9 : 0.0;
      printf("%-32s w=%d  memtime_ticks/instr/wave=%7.3f  wall=%8.3f ms  cyc@2.4GHz/instr/SIMD=%6.3f  Gwaveinstr/s=%8.2f  GMAC/s=%9.1f\n",
             op_name[op], w, ticks / n_instr, ms, cyc_per_instr_simd, ginstr, gmac);
    }
  }
  return 0;
}
