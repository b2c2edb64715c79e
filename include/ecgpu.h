/* ecgpu - batched elliptic-curve arithmetic on MI355X (gfx950): C ABI.
 *
 * Drop-in boundary for the data-parallel hot path of risc0/RustCrypto-elliptic-curves
 * (field mul/sqr/add/sub -> complete point add/double -> fixed/variable-base scalar
 * multiplication -> linear combination / MSM over k256, p256, p384).  The reference exposes
 * that path only as Rust traits (no FFI exists); each entry point below names the trait item(s)
 * it stands in for.  INTEGRATION.md shows the `extern "C"` block and the trait impls a
 * maintainer would add on the Rust side.
 *
 * Conventions
 *  - Plain C, no callbacks, caller-owned buffers, batch first.  No Rust struct memory crosses
 *    the boundary (the reference's types are not repr(C)): everything is canonical big-endian
 *    bytes exactly as `to_bytes` / `to_repr` produce them.
 *      field element / scalar : NB bytes (32 for k256/p256, 48 for p384), value < modulus
 *      affine point           : x || y (2*NB); the identity is x = y = 0, mirroring
 *                               AffinePoint::IDENTITY (k256 affine.rs:51-55, primeorder
 *                               affine.rs:48-52).  Entry points that produce affine points also
 *                               fill an optional `inf` byte array (0/1 per element) = the
 *                               `infinity` field of AffinePoint.
 *      projective point       : X || Y || Z (3*NB), homogeneous (x = X/Z), identity (0, 1, 0)
 *                               (k256 projective.rs:38-50, primeorder projective.rs:37-52).
 *  - `mem` says where the caller's buffers live: ECGPU_MEM_HOST (the library stages them
 *    through HBM) or ECGPU_MEM_DEVICE (pointers into this GPU's HBM, 4-byte aligned - ecgpu_msm
 *    additionally wants its AFFINE points 16-byte aligned; the call is asynchronous on the
 *    context's stream).
 *  - Every function returns 0 on success or a negative ecgpu_status; ecgpu_last_error() gives
 *    the text.  Arithmetic on valid inputs cannot fail.  Decoding failures (scalar >= n,
 *    coordinate >= p, point not on the curve) are reported per element by the *_validate
 *    calls, mirroring CtOption::is_none of from_repr / from_bytes / from_encoded_point.
 *  - Re-entrant; one context per device; calls on different contexts may run concurrently.
 *  - There is no CPU fallback: without a usable GPU every call fails with ECGPU_ERR_NO_DEVICE.
 */
#ifndef ECGPU_H
#define ECGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ecgpu_ctx ecgpu_ctx;

typedef enum ecgpu_curve {
  ECGPU_K256 = 0, /* secp256k1  k256/src/lib.rs:76-104 */
  ECGPU_P256 = 1, /* NIST P-256 p256/src/lib.rs:74-108 */
  ECGPU_P384 = 2  /* NIST P-384 p384/src/lib.rs:50-64 */
} ecgpu_curve;

typedef enum ecgpu_mem { ECGPU_MEM_HOST = 0, ECGPU_MEM_DEVICE = 1 } ecgpu_mem;

typedef enum ecgpu_status {
  ECGPU_OK = 0,
  ECGPU_ERR_ARG = -1,       /* bad curve / op / format / NULL pointer */
  ECGPU_ERR_NO_DEVICE = -2, /* no GPU, or the device is not gfx950 */
  ECGPU_ERR_RUNTIME = -3,   /* HIP runtime error (text in ecgpu_last_error) */
  ECGPU_ERR_UNSUPPORTED = -4
} ecgpu_status;

typedef enum ecgpu_point_format {
  ECGPU_PT_AFFINE = 0,     /* x || y, identity = all zero */
  ECGPU_PT_PROJECTIVE = 1  /* X || Y || Z */
} ecgpu_point_format;

/* FieldElement operations (k256/src/arithmetic/field.rs:56, p256/src/arithmetic/field.rs:43,
 * p384/src/arithmetic/field.rs:47-63).  Results are canonical (`normalize().to_bytes()`). */
typedef enum ecgpu_field_op {
  ECGPU_FE_MUL = 0, /* Mul        field.rs:164-169 / p256 field.rs:293-319 / fiat_p384_mul   */
  ECGPU_FE_SQR = 1, /* square     field.rs:172-174                                           */
  ECGPU_FE_ADD = 2, /* Add        field_5x52.rs:264-272 / p256 field.rs:118-134              */
  ECGPU_FE_SUB = 3, /* Sub        field.rs:425-451 / p256 field.rs:142-147                   */
  ECGPU_FE_NEG = 4, /* Neg / negate field_5x52.rs:252-260                                    */
  ECGPU_FE_INV = 5, /* invert     field.rs:187-216 / p256 field.rs:357-382 (0 -> 0, see doc) */
  ECGPU_FE_SQRT = 6 /* sqrt       field.rs:220-255; non-residue -> all 0xFF                  */
} ecgpu_field_op;

/* flags for the scalar-multiplication entry points */
enum {
  /* Follow the reference's algorithm step for step (GLV + signed radix-16 tables for k256,
   * 4-bit window for p256/p384) so that a PROJECTIVE result is the very (X, Y, Z) triple the
   * reference returns.  Without it the library is free to pick the fastest schedule and only
   * the group element (hence the affine result) is specified. */
  /* The reference schedule is also its constant-time one: table entries are picked by the reference's masked scan over
   * the whole table (k256 mul.rs:92-127, primeorder projective.rs:132-137), formulas are complete, and nothing - no
   * branch, no address - depends on a digit of the scalar.  It is the schedule to use for secret scalars (ECDH, signing
   * nonces); staged host copies of the scalars are cleared before the call returns.  The field arithmetic under the kernels
   * that run on secret scalars has no data-dependent branch either: the Montgomery fields of P-256 / P-384 never had one, and the
   * secp256k1 kernels of this kind are compiled in their own translation unit in which the rare carry paths of the saturated-limb
   * additions, subtractions and folds (probability 2^-26 .. 2^-32 per operation; real branches in the throughput kernels) execute
   * unconditionally (csrc/ops_k256_ct.hip; branch listing and counters: profiles/r04_k256_ct_branches.txt; cost 5-10 %).
   * What remains outside the claim: the scalar-field arithmetic of ecgpu_ecdsa_sign_batch's last step (s = k^-1 (z + r d) mod n) is
   * masked but not audited instruction by instruction, and GPU hardware gives no timing guarantees of its own.
   * The throughput schedules (default for ecgpu_mul_batch / ecgpu_lincomb_batch / ecgpu_msm) skip zero digits, index
   * tables by digits and branch on exceptional cases: bulk PUBLIC data only. */
  ECGPU_EXACT_REFERENCE = 1u,
  /* ECDSA on secp256k1 as the reference configures it: verification rejects s > n/2
   * (k256/src/ecdsa.rs:199-207), signing normalises s to the low half and flips the recovery
   * parity (k256/src/ecdsa.rs:182-196).  The NIST curves are used without it. */
  ECGPU_ECDSA_LOW_S = 2u,
  /* ecgpu_ecdsa_sign_batch only: the nonces are not secret (test vectors, benchmarks, deterministic replays of public
   * data), so k G may run on the throughput fixed-base schedule (digit-indexed table reads, ~4x faster).  Without it
   * signing runs k G constant-time (see ecgpu_ecdsa_sign_batch). */
  ECGPU_PUBLIC_SCALARS = 4u,
  /* ecgpu_mul_batch / ecgpu_lincomb_batch: the scalars are secret and only the group element is wanted (key generation:
   * PublicKey::from_secret_scalar is d G).  With points == NULL the multiplication runs on the constant-time fixed-base
   * kernel signing uses (every table entry read, one masked XYZZ mixed addition per window; the result is the same point, not the reference's
   * (X, Y, Z)); with a variable base point (ECDH: elliptic_curve::ecdh::diffie_hellman) dedicated constant-time
   * kernels run.  P-256 / P-384 (csrc/varbase_ct.hpp): signed 4-bit digits of min(k, n - k) by branch-free recoding, a masked scan
   * over all eight entries of a per-lane affine table, Jacobian doublings and a mixed addition for every digit whose two special
   * operands (empty accumulator, zero digit) are resolved by masks; the fold keeps every addition off the formula's exceptional
   * cases for every k (the file has the argument) - 1.8x / 1.9x the reference schedule.  secp256k1 (csrc/varbase_ct_k256.hpp): the
   * reference's GLV split and recoding on Jacobian formulas over one per-lane common-Z table (one masked scan per window serves both
   * halves, beta x multiplied in); the bounds of the split keep every addition off the exceptional cases (the file has the
   * argument: the accumulator's two coordinates stay below the GLV lattice's shortest vector) - 1.5x the reference schedule.
   * Staged host copies of the scalars are cleared as with ECGPU_EXACT_REFERENCE.
   * Precondition, as for the reference's types: the points are points of the curve.  ecgpu_mul_batch does not validate them (ecgpu_validate_points /
   * ecgpu_ecdh_batch do); for a point that is NOT on the curve the element's own result is unspecified, but it cannot disturb any other element of the
   * batch (a zero denominator is flagged and kept out of the shared inversions), and the instruction stream does not depend on it. */
  ECGPU_SECRET_SCALARS = 8u
};

/* ---- context ------------------------------------------------------------------------------
 * A context owns grow-only workspaces and, per curve, the generator tables it has needed so far (built on the device on
 * first use, kept until ecgpu_destroy): 270 KB for batches below 2^18, 36 MB from 2^18, 436 MB from 2^21, 5.9 GB
 * (13.7 GB for P-384) from 2^23 and 21.5 GB (secp256k1, P-256) from 2^24 results per call; the first call of a size class
 * pays the build (7 / 30 / 85 ms for the last three).  A table that cannot be allocated (hipErrorOutOfMemory) or would
 * exceed ECGPU_OPT_FB_MEMORY_BUDGET is not an error: the call steps down to the next narrower table (26 -> 24 -> 20 -> 16 -> 8 bits) and the context remembers the
 * width that fitted as its cap (ECGPU_OPT_FB_MAX_WINDOW shows and resets it).  ecgpu_fb_table_bytes reports what a context
 * holds.  No entry point reads the process environment; the tuning knobs are per-context options (ecgpu_set_option). */
int ecgpu_create(ecgpu_ctx** ctx, int device_index);
void ecgpu_destroy(ecgpu_ctx* ctx);
/* Streams.  A new context launches on a stream of its own, created as a BLOCKING stream: it synchronises with the legacy
 * default stream (handle 0, PyTorch's default stream) the way every ordinary HIP stream does, so a caller that prepares
 * inputs on the default stream and calls the library without further ado is ordered correctly.
 *   ecgpu_set_stream(ctx, s)      all later launches go to the caller's hipStream_t s.  s = NULL is the legacy default
 *                                 stream itself, as for every HIP API (it is NOT "the context's own stream").
 *   ecgpu_use_own_stream(ctx)     back to the context's own stream.
 * Calls on one context share its scratch buffers and lazily built tables, so they are serialised by the context's lock
 * and, across a stream switch, by an event: everything queued on the previous stream is ordered before anything queued
 * on the new one.  The previous stream must still be alive at the switch; if recording on it fails the library falls
 * back to hipDeviceSynchronize and installs the new stream all the same. */
int ecgpu_set_stream(ecgpu_ctx* ctx, void* hip_stream);
int ecgpu_use_own_stream(ecgpu_ctx* ctx);
int ecgpu_synchronize(ecgpu_ctx* ctx);
/* Text of the context's last error.  The pointer form reads the context's buffer without a lock (single-threaded use);
 * ecgpu_last_error_copy copies it under the lock that writers take, for contexts shared between threads. */
const char* ecgpu_last_error(const ecgpu_ctx* ctx);
int ecgpu_last_error_copy(ecgpu_ctx* ctx, char* buf, size_t cap);

/* Per-context tuning and test knobs (none is needed for correct results; defaults in brackets).  Values outside the
 * listed sets are refused with ECGPU_ERR_ARG. */
typedef enum ecgpu_option {
  ECGPU_OPT_FB_WINDOW = 0,        /* pin the generator-table window width: 8 | 16 | 20 | 24 | 26, 0 = the size rule [0] */
  ECGPU_OPT_FB_MAX_WINDOW = 1,    /* widest generator table the size rule may build: 8 | 16 | 20 | 24 | 26 [26]; lowered by
                                     the library itself when an allocation fails, setting it again lifts that */
  ECGPU_OPT_MSM_WINDOW_BITS = 2,  /* bucket-method window: 16 | 19, 0 = the size rule [0] */
  ECGPU_OPT_MSM_SLAB_TERMS = 3,   /* terms per slab of a large sum, 1024 .. the window's maximum, 0 = that maximum [0] */
  ECGPU_OPT_MSM_SMALL_PATH = 4,   /* 1: sums below 5 * 2^14 terms run as scalar multiplications and a tree sum, 0: always buckets [1] */
  ECGPU_OPT_MSM_ROUNDS = 5,       /* bucket-sum runs per resident lane, 1 .. 64, 0 = default [0] */
  ECGPU_OPT_K256_WAVES = 6,       /* occupancy variant of the secp256k1 variable-base kernel: 3 | 4 waves per SIMD [4] */
  ECGPU_OPT_FB_MEMORY_BUDGET = 7, /* bytes of device memory the generator tables of ONE curve may take in this context,
                                     0 = no budget [0].  A table that would exceed it is treated exactly like one whose
                                     allocation failed: the call steps down to the next narrower width. */
  ECGPU_OPT_LINCOMB_TERM_BY_TERM = 8, /* 1: combinations of 3 .. 1024 terms run as one reference-schedule multiplication per term folded with
                                     the complete addition (the form used until round 4: measurements, cross-checks), 0: the shared-doubling
                                     schedule of csrc/straus.hpp [0] */
  ECGPU_OPT_COUNT_ = 9
} ecgpu_option;
int ecgpu_set_option(ecgpu_ctx* ctx, int option, int64_t value);
int ecgpu_get_option(ecgpu_ctx* ctx, int option, int64_t* value);
/* Device memory held by the generator tables of `curve` in this context (all widths built so far), and the widest window
 * among them (0: none built yet).  Either output may be NULL. */
int ecgpu_fb_table_bytes(ecgpu_ctx* ctx, int curve, size_t* bytes, int* widest_window);
const char* ecgpu_version(void);
/* NB for a curve (32 / 32 / 48), 0 for an unknown curve. */
size_t ecgpu_field_bytes(int curve);
/* Page-locked host memory for ECGPU_MEM_HOST batches.  Host-buffer calls work with any memory.  Batches of 2^21 elements or
 * more are streamed through the device in chunks with upload, kernels and download overlapped (chunk sizes grow from an eighth of
 * a kernel pass to a whole pass and the last chunk is small again, so fill and drain are short); page-locked buffers (these, or
 * memory the caller registered with hipHostRegister) are read and written by DMA directly, pageable ones are copied through a small
 * pool of page-locked bounce buffers by helper threads.  secp256k1 variable base, 2^24 pairs: see DESIGN.md section 7 / bench.py
 * `host_io` for the PCIe-inclusive rates.  ecgpu_msm from host memory streams its terms the same way in parts of 2^23. */
int ecgpu_host_alloc(ecgpu_ctx* ctx, size_t bytes, void** out);
int ecgpu_host_free(ecgpu_ctx* ctx, void* p);

/* How a host-buffer batch of n elements is cut into chunks when one whole pass of its kernel takes pass_units elements
 * (clamped to 2^20 .. 2^23): sizes grow from pass / 8 to pass, the last chunk is at most pass / 8.  Fills sizes[0 .. cap) and
 * returns the number of chunks; needs no device.  (Batches below 2^21 elements are staged whole.) */
int ecgpu_host_chunk_schedule(size_t n, size_t pass_units, size_t* sizes, size_t cap);

/* Diagnostic: size and contents of a context's grow-only device workspaces (which = 0: per-lane table workspace of the
 * variable-base kernels, 1: intermediate scalars / points of the ECDSA / ECDH pipelines, 2: MSM workspace, 16 + i: staging slot i of
 * host-buffer calls; 16 + 6 + 6 s + a is argument a of pipeline slot s).  *bytes = the workspace's size (0 if it does not exist
 * yet); if host_copy is not NULL the first min(cap, size) bytes are copied into it after the context's stream has drained.
 * For audits of secret hygiene: the tests read the workspaces back after secret-scalar calls and check that no result, prefix
 * product or staged secret is left (tests/test_gpu_ct_varbase.py). */
int ecgpu_debug_workspace(ecgpu_ctx* ctx, int which, void* host_copy, size_t cap, size_t* bytes);

/* HIP-event timer on the context's stream: bracket launches, read milliseconds. */
int ecgpu_timer_start(ecgpu_ctx* ctx);
int ecgpu_timer_stop(ecgpu_ctx* ctx, float* milliseconds);

/* ---- base field ---------------------------------------------------------------------------- */
/* out[i] = a[i] (op) b[i]; b is ignored (may be NULL) for unary ops. */
int ecgpu_field_op_batch(ecgpu_ctx* ctx, int curve, int op, const uint8_t* a, const uint8_t* b,
                         uint8_t* out, size_t n, int mem);

/* ---- group law: ProjectivePoint::{add, add_mixed, double}  -------------------------------------
 * k256 projective.rs:96-161, :164-221, :225-274; primeorder point_arithmetic.rs:209-238, :247-277,
 * :286-317.  Complete formulas: total on every input, results are the exact (X, Y, Z) of the
 * reference (canonical bytes). */
int ecgpu_point_add_batch(ecgpu_ctx* ctx, int curve, const uint8_t* p_xyz, const uint8_t* q_xyz,
                          uint8_t* out_xyz, size_t n, int mem);
int ecgpu_point_add_mixed_batch(ecgpu_ctx* ctx, int curve, const uint8_t* p_xyz, const uint8_t* q_xy,
                                uint8_t* out_xyz, size_t n, int mem);
int ecgpu_point_double_batch(ecgpu_ctx* ctx, int curve, const uint8_t* p_xyz, uint8_t* out_xyz,
                             size_t n, int mem);

/* ConstantTimeEq / PartialEq for ProjectivePoint (k256 projective.rs:421-446: cross-multiplied coordinates;
 * primeorder projective.rs:191-198: equality of the affine forms): eq[i] = 1 iff p[i] and q[i] are the same group
 * element, whatever their representatives. */
int ecgpu_point_eq_batch(ecgpu_ctx* ctx, int curve, const uint8_t* p_xyz, const uint8_t* q_xyz, uint8_t* eq,
                         size_t n, int mem);

/* BatchNormalize::batch_normalize / to_affine (k256 projective.rs:73-84, :325-379; primeorder
 * projective.rs:62-74, :346-413): out_xy[i] = affine(p[i]); out_inf may be NULL. */
int ecgpu_batch_normalize(ecgpu_ctx* ctx, int curve, const uint8_t* p_xyz, uint8_t* out_xy,
                          uint8_t* out_inf, size_t n, int mem);

/* ---- scalar multiplication ------------------------------------------------------------------
 * out[i] = scalars[i] * points[i]          `&P * &k`   k256 mul.rs:442-481, primeorder projective.rs:106-150
 * points == NULL: out[i] = scalars[i] * G   MulByGenerator::mul_by_generator  k256 mul.rs:415-440,
 *                                           primeorder projective.rs:422-431
 * point_format / out_format: ecgpu_point_format.  out_inf (optional) only for affine output. */
int ecgpu_mul_batch(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points,
                    int point_format, uint8_t* out, int out_format, uint8_t* out_inf, size_t n,
                    int mem, unsigned flags);

/* Scalars are taken as `Reduce<U256>::reduce` takes them (one conditional subtraction of n, k256 scalar.rs:700-713):
 * the arithmetic never fails.  Scalar::from_repr instead REJECTS values >= n (k256 scalar.rs:365-368); callers that
 * decode untrusted scalars use the *_checked forms, which also fill scalar_ok[i] = 1 iff every scalar of element i is
 * below n (CtOption::is_some); out[i] of an element with scalar_ok[i] = 0 is the result for the reduced scalar and is
 * to be discarded. */
int ecgpu_mul_batch_checked(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points,
                            int point_format, uint8_t* out, int out_format, uint8_t* out_inf, uint8_t* scalar_ok,
                            size_t n, int mem, unsigned flags);

/* n independent linear combinations of `terms` terms each:
 *   out[i] = sum_j scalars[i*terms + j] * points[i*terms + j]
 * terms = 1, 2: throughput or exact-reference schedules as for ecgpu_mul_batch.  3 <= terms <= 1024: groups of up to 16 terms share
 * the doublings of one window loop over per-term affine tables (csrc/straus.hpp; ~890 instead of ~2 000 field multiplications per
 * secp256k1 term) - the group element is specified, a PROJECTIVE result is the representative (x : y : 1).  With
 * ECGPU_EXACT_REFERENCE secp256k1 runs the reference's own interleaved schedule (k256 mul.rs:342-393) for any length up to 1024, so the
 * (X, Y, Z) of lincomb_ext over a slice is exact; for P-256 / P-384 it is refused above two terms (their lincomb over slices lives in
 * the external elliptic-curve crate: nothing in tree to be exact to).
 * LinearCombination::lincomb (terms = 2) / LinearCombinationExt::lincomb_ext (k256 mul.rs:313-393;
 * primeorder default projective.rs:415-420). */
int ecgpu_lincomb_batch(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points,
                        int point_format, size_t terms, uint8_t* out, int out_format, uint8_t* out_inf,
                        size_t n, int mem, unsigned flags);

int ecgpu_lincomb_batch_checked(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points,
                                int point_format, size_t terms, uint8_t* out, int out_format, uint8_t* out_inf,
                                uint8_t* scalar_ok, size_t n, int mem, unsigned flags);

/* One multi-scalar multiplication out = sum_i scalars[i] * points[i] over n terms (Pippenger
 * bucket method; the large-N form of lincomb_ext over a slice, k256 mul.rs:325-340).
 * `out` is one point in out_format. */
int ecgpu_msm(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, const uint8_t* points, int point_format,
              size_t n, uint8_t* out, int out_format, int mem);

/* ---- decoding checks (CtOption::is_none mirrors) ---------------------------------------------
 * ok[i] = 1 iff scalars[i] < n   (Scalar::from_repr, k256 scalar.rs:365-368) */
int ecgpu_validate_scalars(ecgpu_ctx* ctx, int curve, const uint8_t* scalars, uint8_t* ok, size_t n, int mem);
/* ok[i] = 1 iff the affine point decodes: both coordinates < p and on the curve, or the identity
 * (AffinePoint::from_encoded_point, k256 affine.rs:241-270, primeorder affine.rs:161-195) */
int ecgpu_validate_points(ecgpu_ctx* ctx, int curve, const uint8_t* points_xy, uint8_t* ok, size_t n, int mem);
/* DecompressPoint::decompress (k256 affine.rs:184-202, primeorder affine.rs:129-150):
 * out_xy[i] = (x, y) with y parity y_is_odd[i]; ok[i] = 0 (and zeros) when x >= p or no root. */
int ecgpu_decompress_batch(ecgpu_ctx* ctx, int curve, const uint8_t* x, const uint8_t* y_is_odd,
                           uint8_t* out_xy, uint8_t* ok, size_t n, int mem);

/* GroupEncoding::to_bytes / from_bytes (k256 affine.rs:213-238, primeorder affine.rs:256-278): the fixed-width
 * compressed SEC1 form, 1 + field_bytes bytes per point: 0x02 | 0x03 then x; the identity is all zeros.
 * to_bytes takes affine or projective points (projective input is batch-normalised on the device);
 * from_bytes gives out_xy (zeros for the identity) and ok[i] = 0 for a bad tag, x >= p or no root; like the
 * reference it also accepts tag 0x05 (compact: the even root). */
int ecgpu_to_bytes_batch(ecgpu_ctx* ctx, int curve, const uint8_t* points, int point_format, uint8_t* out,
                         size_t n, int mem);
int ecgpu_from_bytes_batch(ecgpu_ctx* ctx, int curve, const uint8_t* in, uint8_t* out_xy, uint8_t* ok,
                           size_t n, int mem);

/* ToEncodedPoint::to_encoded_point(compress) / FromEncodedPoint::from_encoded_point (k256 affine.rs:241-284, primeorder
 * affine.rs:161-195, 340-358) with fixed-width records so that a batch is one dense array:
 *   compress = 0: 1 + 2 field_bytes per point, 0x04 || x || y;   compress = 1: 1 + field_bytes, 0x02 | 0x03 || x.
 * The identity, one byte 0x00 in SEC1, is tag 0x00 followed by zero padding.
 * ecgpu_sec1_decode_batch reads records of `record_bytes` = 1 + 2 field_bytes (tags 0x04; 0x02 / 0x03 / 0x05 with x and
 * zero padding; 0x00 and zeros) or 1 + field_bytes (what ecgpu_from_bytes_batch takes) and applies the reference's checks:
 * coordinates below p, the curve equation for 0x04, a square root for the compressed forms.  ok[i] = 0 and zeros otherwise. */
int ecgpu_sec1_encode_batch(ecgpu_ctx* ctx, int curve, const uint8_t* points, int point_format, int compress, uint8_t* out,
                            size_t n, int mem);
int ecgpu_sec1_decode_batch(ecgpu_ctx* ctx, int curve, const uint8_t* in, size_t record_bytes, uint8_t* out_xy, uint8_t* ok,
                            size_t n, int mem);

/* ---- ECDSA over the path (the callers of mul_by_generator / lincomb) ---------------------------
 * VerifyPrimitive::verify_prehashed / SignPrimitive::try_sign_prehashed; the primitives are the
 * external ecdsa 0.16.9 hazmat functions entered from k256/src/ecdsa.rs:182-209,
 * p256/src/ecdsa.rs:72-75, p384/src/ecdsa.rs:69-72.
 *   prehash     n x field_bytes: the message digest after bits2field (left-most field_bytes bytes,
 *               zero-padded on the left when shorter)
 *   sig_rs      n x 2 field_bytes: r || s, the fixed-size Signature::to_bytes form
 *   pubkeys_xy  n x 2 field_bytes affine x || y
 * ok[i] = 1 iff signature i verifies.  Like the reference, r or s outside [1, n-1], a public key
 * that is off the curve, non-canonical or the identity, and R = identity all give 0. */
int ecgpu_ecdsa_verify_batch(ecgpu_ctx* ctx, int curve, const uint8_t* prehash, const uint8_t* sig_rs,
                             const uint8_t* pubkeys_xy, uint8_t* ok, size_t n, int mem, unsigned flags);
/* sig_rs[i] = (r, s) with R = k G, r = x(R) mod n, s = k^-1 (z + r d) mod n; recovery_id[i] (optional) =
 * y_is_odd(R) | x_is_reduced << 1; ok[i] = 0 (and a zero signature) when d or k is outside [1, n-1]
 * or r = 0 or s = 0, where the reference returns Err.
 * The nonce is secret, so k G runs constant-time: by default on a fixed-base kernel that reads every entry of its
 * 5-bit-window table and keeps the digit's one by masks, one XYZZ mixed addition (8M + 2S) per window whose special operands (empty
 * accumulator, zero digit) are masks and which the bounds on the digits keep off its exceptional cases (csrc/fixedbase_ct.hpp) - 6-7x the
 * speed of the reference schedule on P-256 / P-384, whose mul_by_generator is the generic variable-base
 * multiplication); ECGPU_EXACT_REFERENCE in `flags` selects the reference's own mul_by_generator schedule (constant-time
 * as well), ECGPU_PUBLIC_SCALARS the digit-indexed throughput schedule.  The signatures are identical in all three.
 * Staged host copies of d and k are cleared before the call returns. */
int ecgpu_ecdsa_sign_batch(ecgpu_ctx* ctx, int curve, const uint8_t* secret_d, const uint8_t* nonce_k,
                           const uint8_t* prehash, uint8_t* sig_rs, uint8_t* recovery_id, uint8_t* ok,
                           size_t n, int mem, unsigned flags);

/* Public-key recovery, VerifyingKey::recover_from_prehash (external ecdsa crate; exercised by k256/src/ecdsa.rs:259-336):
 * R = decompress(r, or r + n when recovery_id bit 1 is set; y parity = bit 0), Q = -(z r^-1) G + (s r^-1) R.
 * pubkeys_xy[i] = Q and ok[i] = 1, or zeros and ok[i] = 0 where the reference returns Err: r or s outside [1, n-1],
 * recovery id above 3, x not below p or without a square root, Q the identity, and - with ECGPU_ECDSA_LOW_S, as the
 * reference's final verify_prehash does for secp256k1 - s in the high half. */
int ecgpu_ecdsa_recover_batch(ecgpu_ctx* ctx, int curve, const uint8_t* prehash, const uint8_t* sig_rs,
                              const uint8_t* recovery_id, uint8_t* pubkeys_xy, uint8_t* ok, size_t n, int mem,
                              unsigned flags);

/* ---- ECDH over the path -------------------------------------------------------------------------------------------------
 * elliptic_curve::ecdh::diffie_hellman for a batch (external elliptic-curve 0.13.8; re-exported at k256/src/ecdh.rs:41,
 * p256/src/ecdh.rs, p384/src/ecdh.rs): shared_x[i] = x((pubkeys[i] * secret[i]).to_affine()), the bytes SharedSecret holds
 * (k256 ecdh.rs:51-55).  ok[i] = 0 and zeros where the reference's types could not have been built: a secret scalar that
 * is zero or >= n (NonZeroScalar), a public key that is non-canonical, off the curve or the identity (PublicKey).
 * The multiplication is the ECGPU_SECRET_SCALARS one (constant-time); staged copies of the secrets AND of the shared
 * values are cleared before the call returns, the intermediate products do not stay in the context's workspace. */
int ecgpu_ecdh_batch(ecgpu_ctx* ctx, int curve, const uint8_t* secret_scalars, const uint8_t* pubkeys_xy, uint8_t* shared_x,
                     uint8_t* ok, size_t n, int mem);

/* ---- BIP340 Schnorr over secp256k1 -----------------------------------------------------------------
 * The elliptic-curve part of VerifyingKey::verify_prehash (k256/src/schnorr/verifying.rs:62-93): with the challenge
 * e = tagged_hash("BIP0340/challenge", r || P.x || m) supplied by the caller (32 bytes, reduced mod n here),
 * ok[i] = 1 iff R = s G - e P is finite, has even y and x(R) = r.  Decoding rules of the reference apply: r in
 * [1, p), s in [1, n) (k256/src/schnorr.rs:142-160), key x < p with a curve point (verifying.rs:39-45). */
int ecgpu_schnorr_verify_batch(ecgpu_ctx* ctx, int curve, const uint8_t* pubkeys_x, const uint8_t* sig_rs,
                               const uint8_t* challenges, uint8_t* ok, size_t n, int mem);

/* ---- hash to curve (RFC 9380 suites *_XMD:SHA-*_SSWU_RO_) ------------------------------------------------------
 * MapToCurve::map_to_curve and the sum of GroupDigest::hash_from_bytes (k256|p256|p384/src/arithmetic/hash2curve.rs):
 * u holds n x count field elements (canonical big-endian, already reduced: hash_to_field / FromOkm is host glue);
 * count = 1: out[i] = map_to_curve(u[i]);  count = 2: out[i] = map_to_curve(u[2i]) + map_to_curve(u[2i+1]). */
int ecgpu_map_to_curve_batch(ecgpu_ctx* ctx, int curve, const uint8_t* u, int count, uint8_t* out_xy,
                             uint8_t* out_inf, size_t n, int mem);

/* ---- device groups: one call, several GPUs -------------------------------------------------------------------------------
 * The reference's bulk entry points are single calls over slices (LinearCombinationExt::lincomb_ext, k256 mul.rs:325-340; `&P * &k`
 * element by element, mul.rs:442-481), so the split over devices lives on this side of the boundary:
 *   - independent batches: member i of the group takes the contiguous index range ecgpu_shard_range(n, size, i) on its own
 *     context, host thread and stream; generator tables are per device; NO collective;
 *   - one split sum (ecgpu_group_msm): every member runs the whole bucket method over its range of terms -> one projective point
 *     per device -> all-gather of those points (RCCL ncclAllGather over xGMI when the devices are distinct; elliptic-curve addition
 *     is not an RCCL reduction operator, so all-reduce does not apply) -> the leader (member 0) adds them up.  RCCL is loaded with
 *     dlopen at the first split sum; a group whose devices are not distinct (two contexts on one card) or a system without RCCL
 *     gathers through host memory (96 / 144 bytes per device).  ecgpu_group_gather_path says which way the last sum went.
 * Results are those of the single-context call on the whole batch, byte for byte (for a split sum: the same group element, output
 * formatted as ecgpu_msm formats it).  A group owns one context per member (ecgpu_group_context: options, generator-table
 * queries); calls on one group are serialised. */
typedef struct ecgpu_group ecgpu_group;
enum { ECGPU_GROUP_NO_RCCL = 1u };            /* ecgpu_group_create flags: gather partial sums through host memory even if RCCL could be used */
int ecgpu_group_create(ecgpu_group** group, const int* devices, int n_devices, unsigned flags);
void ecgpu_group_destroy(ecgpu_group* group);
int ecgpu_group_size(const ecgpu_group* group);
ecgpu_ctx* ecgpu_group_context(ecgpu_group* group, int index);
const char* ecgpu_group_last_error(const ecgpu_group* group);
const char* ecgpu_group_gather_path(const ecgpu_group* group);
int ecgpu_group_synchronize(ecgpu_group* group);
/* first / count of part `index` when n elements are cut into `parts` balanced contiguous ranges (needs no device) */
int ecgpu_shard_range(size_t n, int parts, int index, size_t* first, size_t* count);
/* ecgpu_mul_batch / ecgpu_lincomb_batch over HOST buffers, split over the group's devices (each member streams its range through
 * its own host pipeline) */
int ecgpu_group_mul_batch(ecgpu_group* group, int curve, const uint8_t* scalars, const uint8_t* points, int point_format, uint8_t* out,
                          int out_format, uint8_t* out_inf, size_t n, unsigned flags);
int ecgpu_group_lincomb_batch(ecgpu_group* group, int curve, const uint8_t* scalars, const uint8_t* points, int point_format, size_t terms,
                              uint8_t* out, int out_format, uint8_t* out_inf, size_t n, unsigned flags);
/* the same over device-resident shards: scalars[i], points[i], out[i], out_inf[i] are pointers into member i's HBM, counts[i] its
 * number of elements; asynchronous on every member's stream (ecgpu_group_synchronize).  points / out_inf may be NULL. */
int ecgpu_group_lincomb_sharded(ecgpu_group* group, int curve, const uint8_t* const* scalars, const uint8_t* const* points, int point_format,
                                size_t terms, uint8_t* const* out, int out_format, uint8_t* const* out_inf, const size_t* counts, unsigned flags);
/* ONE multi-scalar multiplication over n terms in host memory, split over the group (the 8-GPU form of lincomb_ext over a slice);
 * `out` is one point in host memory */
int ecgpu_group_msm(ecgpu_group* group, int curve, const uint8_t* scalars, const uint8_t* points, int point_format, size_t n, uint8_t* out,
                    int out_format);
/* the same with the terms already resident: member i sums counts[i] terms at scalars[i] / points[i] in its own HBM; `out` in host
 * memory */
int ecgpu_group_msm_sharded(ecgpu_group* group, int curve, const uint8_t* const* scalars, const uint8_t* const* points, int point_format,
                            const size_t* counts, uint8_t* out, int out_format);

/* ---- synthetic inputs for benchmarks (device memory only) ------------------------------------
 * Fill device buffers with the counter-based streams specified in oracle/synth.py:
 * scalars[i] = reduce(stream 0), points[i] = try-and-increment decompress of streams 1.. .
 * `first_index` lets ranks generate disjoint slices of one global batch. */
int ecgpu_synth_scalars(ecgpu_ctx* ctx, int curve, uint64_t seed, uint64_t first_index, uint8_t* d_scalars, size_t n);
int ecgpu_synth_points(ecgpu_ctx* ctx, int curve, uint64_t seed, uint64_t first_index, uint8_t* d_points_xy, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ECGPU_H */
