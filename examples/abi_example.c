/* Minimal C caller of the boundary (include/ecgpu.h): what a foreign-language binding does, without Python.
 * Builds with any C compiler against rustcrypto-elliptic-curves_amd/lib/libecgpu.so:
 *     make -C examples            (gcc abi_example.c -I../include -L../rustcrypto-elliptic-curves_amd/lib -lecgpu)
 * Runs on a box with an MI355X: multiplies the secp256k1 generator by 1, 2, 3 (MulByGenerator::mul_by_generator,
 * k256/src/arithmetic/mul.rs:415-440), checks 2 G against the well-known value and G + 2 G = 3 G through the complete
 * addition (k256/src/arithmetic/projective.rs:96-161) and ProjectivePoint equality, then signs and verifies one ECDSA
 * prehash.  Exit status 0 on success, 2 when no usable GPU is present (there is no CPU fallback), 1 on a wrong result. */
#include <stdio.h>
#include <string.h>
#include "ecgpu.h"

static void hex(const char* tag, const uint8_t* b, size_t n) {
  printf("%s", tag);
  for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
  printf("\n");
}

int main(void) {
  ecgpu_ctx* ctx = NULL;
  int rc = ecgpu_create(&ctx, 0);
  if (rc != ECGPU_OK) {
    fprintf(stderr, "ecgpu_create failed with %d: no usable gfx950 device (the library has no CPU path)\n", rc);
    return 2;
  }
  printf("%s\n", ecgpu_version());
  uint8_t k[3][32];
  memset(k, 0, sizeof(k));
  k[0][31] = 1; k[1][31] = 2; k[2][31] = 3;
  uint8_t xyz[3][96];
  /* exact (X, Y, Z) of the reference's schedule, constant-time */
  rc = ecgpu_mul_batch(ctx, ECGPU_K256, &k[0][0], NULL, ECGPU_PT_AFFINE, &xyz[0][0], ECGPU_PT_PROJECTIVE, NULL, 3, ECGPU_MEM_HOST, ECGPU_EXACT_REFERENCE);
  if (rc) { fprintf(stderr, "mul_batch: %s\n", ecgpu_last_error(ctx)); return 1; }
  uint8_t xy[3][64], inf[3];
  rc = ecgpu_batch_normalize(ctx, ECGPU_K256, &xyz[0][0], &xy[0][0], inf, 3, ECGPU_MEM_HOST);
  if (rc) { fprintf(stderr, "batch_normalize: %s\n", ecgpu_last_error(ctx)); return 1; }
  hex("1 G x = ", xy[0], 32);
  hex("2 G x = ", xy[1], 32);
  static const uint8_t two_g_x[32] = {0xc6, 0x04, 0x7f, 0x94, 0x41, 0xed, 0x7d, 0x6d, 0x30, 0x45, 0x40, 0x6e, 0x95, 0xc0, 0x7c, 0xd8,
                                      0x5c, 0x77, 0x8e, 0x4b, 0x8c, 0xef, 0x3c, 0xa7, 0xab, 0xac, 0x09, 0xb9, 0x5c, 0x70, 0x9e, 0xe5};
  if (memcmp(xy[1], two_g_x, 32) != 0 || inf[1] != 0) { fprintf(stderr, "2 G is wrong\n"); return 1; }
  /* G + 2 G == 3 G as group elements */
  uint8_t sum[96], eq = 0;
  rc = ecgpu_point_add_batch(ctx, ECGPU_K256, xyz[0], xyz[1], sum, 1, ECGPU_MEM_HOST);
  if (!rc) rc = ecgpu_point_eq_batch(ctx, ECGPU_K256, sum, xyz[2], &eq, 1, ECGPU_MEM_HOST);
  if (rc || eq != 1) { fprintf(stderr, "G + 2 G != 3 G (%s)\n", ecgpu_last_error(ctx)); return 1; }
  /* sign a prehash with d = 3, nonce 2 (public test values: the throughput schedule is fine), verify it with Q = 3 G */
  uint8_t z[32], sig[64], rid = 0, ok = 0;
  for (int i = 0; i < 32; i++) z[i] = (uint8_t)(i * 7 + 1);
  rc = ecgpu_ecdsa_sign_batch(ctx, ECGPU_K256, k[2], k[1], z, sig, &rid, &ok, 1, ECGPU_MEM_HOST, ECGPU_ECDSA_LOW_S | ECGPU_PUBLIC_SCALARS);
  if (rc || !ok) { fprintf(stderr, "sign failed (%s)\n", ecgpu_last_error(ctx)); return 1; }
  hex("r || s  = ", sig, 64);
  ok = 0;
  rc = ecgpu_ecdsa_verify_batch(ctx, ECGPU_K256, z, sig, xy[2], &ok, 1, ECGPU_MEM_HOST, ECGPU_ECDSA_LOW_S);
  if (rc || !ok) { fprintf(stderr, "verify rejected a valid signature (%s)\n", ecgpu_last_error(ctx)); return 1; }
  sig[40] ^= 1;
  rc = ecgpu_ecdsa_verify_batch(ctx, ECGPU_K256, z, sig, xy[2], &ok, 1, ECGPU_MEM_HOST, ECGPU_ECDSA_LOW_S);
  if (rc || ok) { fprintf(stderr, "verify accepted a corrupted signature\n"); return 1; }
  ecgpu_destroy(ctx);
  printf("abi example ok\n");
  return 0;
}
