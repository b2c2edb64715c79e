//! UNBUILT SOURCE (no Rust toolchain in the build image or on the GPU box: this crate has never been compiled).
//!
//! Layer 1 of the Rust side of the boundary: the complete `extern "C"` block of `include/ecgpu.h` (one declaration
//! per exported function; `tests/test_rust_shim_sync.py` keeps names and arities in step with the header) and a safe
//! `Context` whose methods work on the wire format - canonical big-endian byte slices - so that it depends on no
//! reference crate.  Layer 2, the trait-shaped impls over the reference's own types, lives in `../integration/`:
//! those modules read the private coordinates of `ProjectivePoint`, so they belong inside `k256` / `primeorder`.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_uint, c_void};

#[repr(C)]
pub struct ecgpu_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct ecgpu_group {
    _private: [u8; 0],
}

pub const ECGPU_K256: c_int = 0;
pub const ECGPU_P256: c_int = 1;
pub const ECGPU_P384: c_int = 2;
pub const ECGPU_MEM_HOST: c_int = 0;
pub const ECGPU_MEM_DEVICE: c_int = 1;
pub const ECGPU_PT_AFFINE: c_int = 0;
pub const ECGPU_PT_PROJECTIVE: c_int = 1;
pub const ECGPU_OK: c_int = 0;
pub const ECGPU_ERR_ARG: c_int = -1;
pub const ECGPU_ERR_NO_DEVICE: c_int = -2;
pub const ECGPU_ERR_RUNTIME: c_int = -3;
pub const ECGPU_ERR_UNSUPPORTED: c_int = -4;
pub const ECGPU_FE_MUL: c_int = 0;
pub const ECGPU_FE_SQR: c_int = 1;
pub const ECGPU_FE_ADD: c_int = 2;
pub const ECGPU_FE_SUB: c_int = 3;
pub const ECGPU_FE_NEG: c_int = 4;
pub const ECGPU_FE_INV: c_int = 5;
pub const ECGPU_FE_SQRT: c_int = 6;
/// the reference's own schedule, constant-time table scans included: exact (X, Y, Z), and the one for secret scalars
pub const ECGPU_EXACT_REFERENCE: c_uint = 1;
/// the k256 low-s rules of k256/src/ecdsa.rs:182-207
pub const ECGPU_ECDSA_LOW_S: c_uint = 2;
/// signing only: the nonces are public, k G may use the throughput fixed-base schedule
pub const ECGPU_PUBLIC_SCALARS: c_uint = 4;
/// secret scalars, group element only: constant-time fixed base for k G (key generation); for a variable base (ECDH) the
/// constant-time kernels of csrc/varbase_ct.hpp (P-256 / P-384) and csrc/varbase_ct_k256.hpp (secp256k1)
pub const ECGPU_SECRET_SCALARS: c_uint = 8;
/// `ecgpu_option`: per-context tuning / test knobs (the library never reads the process environment)
pub const ECGPU_OPT_FB_WINDOW: c_int = 0;
pub const ECGPU_OPT_FB_MAX_WINDOW: c_int = 1;
pub const ECGPU_OPT_MSM_WINDOW_BITS: c_int = 2;
pub const ECGPU_OPT_MSM_SLAB_TERMS: c_int = 3;
pub const ECGPU_OPT_MSM_SMALL_PATH: c_int = 4;
pub const ECGPU_OPT_MSM_ROUNDS: c_int = 5;
pub const ECGPU_OPT_K256_WAVES: c_int = 6;
pub const ECGPU_OPT_FB_MEMORY_BUDGET: c_int = 7;
pub const ECGPU_OPT_LINCOMB_TERM_BY_TERM: c_int = 8;
pub const ECGPU_OPT_COUNT_: c_int = 9;
/// `ecgpu_group_create` flag: gather the partial sums of a split sum through host memory even where RCCL could be used
pub const ECGPU_GROUP_NO_RCCL: c_uint = 1;

#[link(name = "ecgpu")]
extern "C" {
    pub fn ecgpu_create(ctx: *mut *mut ecgpu_ctx, device_index: c_int) -> c_int;
    pub fn ecgpu_destroy(ctx: *mut ecgpu_ctx);
    pub fn ecgpu_set_stream(ctx: *mut ecgpu_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn ecgpu_use_own_stream(ctx: *mut ecgpu_ctx) -> c_int;
    pub fn ecgpu_synchronize(ctx: *mut ecgpu_ctx) -> c_int;
    pub fn ecgpu_last_error(ctx: *const ecgpu_ctx) -> *const c_char;
    pub fn ecgpu_last_error_copy(ctx: *mut ecgpu_ctx, buf: *mut c_char, cap: usize) -> c_int;
    pub fn ecgpu_set_option(ctx: *mut ecgpu_ctx, option: c_int, value: i64) -> c_int;
    pub fn ecgpu_get_option(ctx: *mut ecgpu_ctx, option: c_int, value: *mut i64) -> c_int;
    pub fn ecgpu_fb_table_bytes(ctx: *mut ecgpu_ctx, curve: c_int, bytes: *mut usize, widest_window: *mut c_int) -> c_int;
    pub fn ecgpu_version() -> *const c_char;
    pub fn ecgpu_field_bytes(curve: c_int) -> usize;
    pub fn ecgpu_host_alloc(ctx: *mut ecgpu_ctx, bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn ecgpu_host_free(ctx: *mut ecgpu_ctx, p: *mut c_void) -> c_int;
    pub fn ecgpu_host_chunk_schedule(n: usize, pass_units: usize, sizes: *mut usize, cap: usize) -> c_int;
    pub fn ecgpu_debug_workspace(ctx: *mut ecgpu_ctx, which: c_int, host_copy: *mut c_void, cap: usize, bytes: *mut usize) -> c_int;
    pub fn ecgpu_timer_start(ctx: *mut ecgpu_ctx) -> c_int;
    pub fn ecgpu_timer_stop(ctx: *mut ecgpu_ctx, milliseconds: *mut f32) -> c_int;
    pub fn ecgpu_field_op_batch(ctx: *mut ecgpu_ctx, curve: c_int, op: c_int, a: *const u8, b: *const u8, out: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_point_add_batch(ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, q_xyz: *const u8, out_xyz: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_point_add_mixed_batch(ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, q_xy: *const u8, out_xyz: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_point_double_batch(ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, out_xyz: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_point_eq_batch(ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, q_xyz: *const u8, eq: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_batch_normalize(ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, out_xy: *mut u8, out_inf: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_mul_batch(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, out: *mut u8, out_format: c_int,
                           out_inf: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_mul_batch_checked(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, out: *mut u8,
                                   out_format: c_int, out_inf: *mut u8, scalar_ok: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_lincomb_batch(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, terms: usize, out: *mut u8,
                               out_format: c_int, out_inf: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_lincomb_batch_checked(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, terms: usize,
                                       out: *mut u8, out_format: c_int, out_inf: *mut u8, scalar_ok: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_msm(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, n: usize, out: *mut u8, out_format: c_int,
                     mem: c_int) -> c_int;
    pub fn ecgpu_validate_scalars(ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, ok: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_validate_points(ctx: *mut ecgpu_ctx, curve: c_int, points_xy: *const u8, ok: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_decompress_batch(ctx: *mut ecgpu_ctx, curve: c_int, x: *const u8, y_is_odd: *const u8, out_xy: *mut u8, ok: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_to_bytes_batch(ctx: *mut ecgpu_ctx, curve: c_int, points: *const u8, point_format: c_int, out: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_from_bytes_batch(ctx: *mut ecgpu_ctx, curve: c_int, input: *const u8, out_xy: *mut u8, ok: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_sec1_encode_batch(ctx: *mut ecgpu_ctx, curve: c_int, points: *const u8, point_format: c_int, compress: c_int, out: *mut u8, n: usize,
                                   mem: c_int) -> c_int;
    pub fn ecgpu_sec1_decode_batch(ctx: *mut ecgpu_ctx, curve: c_int, input: *const u8, record_bytes: usize, out_xy: *mut u8, ok: *mut u8, n: usize,
                                   mem: c_int) -> c_int;
    pub fn ecgpu_ecdsa_verify_batch(ctx: *mut ecgpu_ctx, curve: c_int, prehash: *const u8, sig_rs: *const u8, pubkeys_xy: *const u8, ok: *mut u8, n: usize,
                                    mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_ecdsa_sign_batch(ctx: *mut ecgpu_ctx, curve: c_int, secret_d: *const u8, nonce_k: *const u8, prehash: *const u8, sig_rs: *mut u8,
                                  recovery_id: *mut u8, ok: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_ecdsa_recover_batch(ctx: *mut ecgpu_ctx, curve: c_int, prehash: *const u8, sig_rs: *const u8, recovery_id: *const u8, pubkeys_xy: *mut u8,
                                     ok: *mut u8, n: usize, mem: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_ecdh_batch(ctx: *mut ecgpu_ctx, curve: c_int, secret_scalars: *const u8, pubkeys_xy: *const u8, shared_x: *mut u8, ok: *mut u8, n: usize,
                            mem: c_int) -> c_int;
    pub fn ecgpu_schnorr_verify_batch(ctx: *mut ecgpu_ctx, curve: c_int, pubkeys_x: *const u8, sig_rs: *const u8, challenges: *const u8, ok: *mut u8, n: usize,
                                      mem: c_int) -> c_int;
    pub fn ecgpu_map_to_curve_batch(ctx: *mut ecgpu_ctx, curve: c_int, u: *const u8, count: c_int, out_xy: *mut u8, out_inf: *mut u8, n: usize, mem: c_int) -> c_int;
    pub fn ecgpu_group_create(group: *mut *mut ecgpu_group, devices: *const c_int, n_devices: c_int, flags: c_uint) -> c_int;
    pub fn ecgpu_group_destroy(group: *mut ecgpu_group);
    pub fn ecgpu_group_size(group: *const ecgpu_group) -> c_int;
    pub fn ecgpu_group_context(group: *mut ecgpu_group, index: c_int) -> *mut ecgpu_ctx;
    pub fn ecgpu_group_last_error(group: *const ecgpu_group) -> *const c_char;
    pub fn ecgpu_group_gather_path(group: *const ecgpu_group) -> *const c_char;
    pub fn ecgpu_group_synchronize(group: *mut ecgpu_group) -> c_int;
    pub fn ecgpu_shard_range(n: usize, parts: c_int, index: c_int, first: *mut usize, count: *mut usize) -> c_int;
    pub fn ecgpu_group_mul_batch(group: *mut ecgpu_group, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, out: *mut u8, out_format: c_int,
                                 out_inf: *mut u8, n: usize, flags: c_uint) -> c_int;
    pub fn ecgpu_group_lincomb_batch(group: *mut ecgpu_group, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, terms: usize, out: *mut u8,
                                     out_format: c_int, out_inf: *mut u8, n: usize, flags: c_uint) -> c_int;
    pub fn ecgpu_group_lincomb_sharded(group: *mut ecgpu_group, curve: c_int, scalars: *const *const u8, points: *const *const u8, point_format: c_int, terms: usize,
                                       out: *const *mut u8, out_format: c_int, out_inf: *const *mut u8, counts: *const usize, flags: c_uint) -> c_int;
    pub fn ecgpu_group_msm(group: *mut ecgpu_group, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int, n: usize, out: *mut u8,
                           out_format: c_int) -> c_int;
    pub fn ecgpu_group_msm_sharded(group: *mut ecgpu_group, curve: c_int, scalars: *const *const u8, points: *const *const u8, point_format: c_int,
                                   counts: *const usize, out: *mut u8, out_format: c_int) -> c_int;
    pub fn ecgpu_synth_scalars(ctx: *mut ecgpu_ctx, curve: c_int, seed: u64, first_index: u64, d_scalars: *mut u8, n: usize) -> c_int;
    pub fn ecgpu_synth_points(ctx: *mut ecgpu_ctx, curve: c_int, seed: u64, first_index: u64, d_points_xy: *mut u8, n: usize) -> c_int;
}

/// A status other than `ECGPU_OK`, with the library's text for it.
#[derive(Debug)]
pub struct Error {
    pub code: i32,
    pub message: String,
}

/// One context per device (`ecgpu_create`); the byte-level mirror of the header.  Host buffers only: device-resident
/// callers use the raw functions with `ECGPU_MEM_DEVICE`.
pub struct Context(*mut ecgpu_ctx);

// The library serialises calls on one context with its own lock (ecgpu.h: "Re-entrant; one context per device"), and
// `check` reads the error text through ecgpu_last_error_copy, which takes the lock the writers of that buffer take: a
// `&Context` may be used from several threads.  (The text is the context's LAST error: with concurrent failing calls a
// thread may read its neighbour's message; the status code it got back is always its own.)
unsafe impl Send for Context {}
unsafe impl Sync for Context {}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { ecgpu_destroy(self.0) }
    }
}

impl Context {
    pub fn new(device: i32) -> Result<Self, Error> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { ecgpu_create(&mut p, device) };
        if rc == ECGPU_OK { Ok(Context(p)) } else { Err(Error { code: rc, message: "ecgpu_create failed (no gfx950 device? there is no CPU fallback)".into() }) }
    }
    pub fn raw(&self) -> *mut ecgpu_ctx { self.0 }
    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == ECGPU_OK { return Ok(()); }
        let mut buf = [0 as c_char; 512];
        unsafe { ecgpu_last_error_copy(self.0, buf.as_mut_ptr(), buf.len()) };
        let msg = unsafe { std::ffi::CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message: msg })
    }
    pub fn field_bytes(curve: c_int) -> usize { unsafe { ecgpu_field_bytes(curve) } }
    fn arg(ok: bool) -> Result<(), Error> {
        if ok { Ok(()) } else { Err(Error { code: ECGPU_ERR_ARG, message: "slice lengths do not match the batch".into() }) }
    }
    fn pt_bytes(nb: usize, fmt: c_int) -> usize { if fmt == ECGPU_PT_PROJECTIVE { 3 * nb } else { 2 * nb } }

    /// FieldElement op on `n` elements: `a`, `b` (binary ops) and the result are `n * NB` bytes.
    pub fn field_op(&self, curve: c_int, op: c_int, a: &[u8], b: Option<&[u8]>) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && a.len() % nb == 0 && b.map_or(true, |b| b.len() == a.len()))?;
        let n = a.len() / nb;
        let mut out = vec![0u8; a.len()];
        self.check(unsafe { ecgpu_field_op_batch(self.0, curve, op, a.as_ptr(), b.map_or(core::ptr::null(), |b| b.as_ptr()), out.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    /// ProjectivePoint + ProjectivePoint, X || Y || Z per point.
    pub fn point_add(&self, curve: c_int, p: &[u8], q: &[u8]) -> Result<Vec<u8>, Error> {
        let w = 3 * Self::field_bytes(curve);
        Self::arg(w != 0 && p.len() % w == 0 && q.len() == p.len())?;
        let mut out = vec![0u8; p.len()];
        self.check(unsafe { ecgpu_point_add_batch(self.0, curve, p.as_ptr(), q.as_ptr(), out.as_mut_ptr(), p.len() / w, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    /// ProjectivePoint + AffinePoint (x || y, zeros = identity).
    pub fn point_add_mixed(&self, curve: c_int, p: &[u8], q_xy: &[u8]) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && p.len() % (3 * nb) == 0 && q_xy.len() == p.len() / 3 * 2)?;
        let mut out = vec![0u8; p.len()];
        self.check(unsafe { ecgpu_point_add_mixed_batch(self.0, curve, p.as_ptr(), q_xy.as_ptr(), out.as_mut_ptr(), p.len() / (3 * nb), ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    pub fn point_double(&self, curve: c_int, p: &[u8]) -> Result<Vec<u8>, Error> {
        let w = 3 * Self::field_bytes(curve);
        Self::arg(w != 0 && p.len() % w == 0)?;
        let mut out = vec![0u8; p.len()];
        self.check(unsafe { ecgpu_point_double_batch(self.0, curve, p.as_ptr(), out.as_mut_ptr(), p.len() / w, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    /// `ct_eq` of projective points: one flag per pair.
    pub fn point_eq(&self, curve: c_int, p: &[u8], q: &[u8]) -> Result<Vec<u8>, Error> {
        let w = 3 * Self::field_bytes(curve);
        Self::arg(w != 0 && p.len() % w == 0 && q.len() == p.len())?;
        let mut eq = vec![0u8; p.len() / w];
        self.check(unsafe { ecgpu_point_eq_batch(self.0, curve, p.as_ptr(), q.as_ptr(), eq.as_mut_ptr(), eq.len(), ECGPU_MEM_HOST) })?;
        Ok(eq)
    }
    /// BatchNormalize: X || Y || Z -> (x || y, infinity flags).
    pub fn batch_normalize(&self, curve: c_int, p: &[u8]) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && p.len() % (3 * nb) == 0)?;
        let n = p.len() / (3 * nb);
        let (mut xy, mut inf) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_batch_normalize(self.0, curve, p.as_ptr(), xy.as_mut_ptr(), inf.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((xy, inf))
    }
    /// n independent linear combinations of `terms` terms; `points = None` (terms = 1) multiplies the generator.
    /// Returns (points in `out_format`, infinity flags for affine output, scalar_ok flags when `checked`).
    #[allow(clippy::too_many_arguments)]
    pub fn lincomb(&self, curve: c_int, scalars: &[u8], points: Option<&[u8]>, point_format: c_int, terms: usize, out_format: c_int, flags: c_uint,
                   checked: bool) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && terms >= 1 && scalars.len() % (nb * terms) == 0)?;
        let n = scalars.len() / (nb * terms);
        Self::arg(points.map_or(terms == 1, |p| p.len() == n * terms * Self::pt_bytes(nb, point_format)))?;
        let mut out = vec![0u8; n * Self::pt_bytes(nb, out_format)];
        let mut inf = vec![0u8; n];
        let mut ok = vec![1u8; n];
        let pp = points.map_or(core::ptr::null(), |p| p.as_ptr());
        let rc = unsafe {
            if checked {
                ecgpu_lincomb_batch_checked(self.0, curve, scalars.as_ptr(), pp, point_format, terms, out.as_mut_ptr(), out_format, inf.as_mut_ptr(), ok.as_mut_ptr(), n,
                                            ECGPU_MEM_HOST, flags)
            } else {
                ecgpu_lincomb_batch(self.0, curve, scalars.as_ptr(), pp, point_format, terms, out.as_mut_ptr(), out_format, inf.as_mut_ptr(), n, ECGPU_MEM_HOST, flags)
            }
        };
        self.check(rc)?;
        Ok((out, inf, ok))
    }
    /// One sum over all terms (bucket method), the large-N form of `lincomb_ext` over a slice.
    pub fn msm(&self, curve: c_int, scalars: &[u8], points: &[u8], point_format: c_int, out_format: c_int) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && scalars.len() % nb == 0 && points.len() == scalars.len() / nb * Self::pt_bytes(nb, point_format))?;
        let mut out = vec![0u8; Self::pt_bytes(nb, out_format)];
        self.check(unsafe { ecgpu_msm(self.0, curve, scalars.as_ptr(), points.as_ptr(), point_format, scalars.len() / nb, out.as_mut_ptr(), out_format, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    pub fn validate_scalars(&self, curve: c_int, scalars: &[u8]) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && scalars.len() % nb == 0)?;
        let mut ok = vec![0u8; scalars.len() / nb];
        self.check(unsafe { ecgpu_validate_scalars(self.0, curve, scalars.as_ptr(), ok.as_mut_ptr(), ok.len(), ECGPU_MEM_HOST) })?;
        Ok(ok)
    }
    pub fn validate_points(&self, curve: c_int, xy: &[u8]) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && xy.len() % (2 * nb) == 0)?;
        let mut ok = vec![0u8; xy.len() / (2 * nb)];
        self.check(unsafe { ecgpu_validate_points(self.0, curve, xy.as_ptr(), ok.as_mut_ptr(), ok.len(), ECGPU_MEM_HOST) })?;
        Ok(ok)
    }
    /// DecompressPoint::decompress -> (x || y, ok flags)
    pub fn decompress(&self, curve: c_int, x: &[u8], y_is_odd: &[u8]) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && x.len() == y_is_odd.len() * nb)?;
        let n = y_is_odd.len();
        let (mut xy, mut ok) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_decompress_batch(self.0, curve, x.as_ptr(), y_is_odd.as_ptr(), xy.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((xy, ok))
    }
    /// GroupEncoding::to_bytes (compressed SEC1, NB + 1 bytes per point)
    pub fn to_bytes(&self, curve: c_int, points: &[u8], point_format: c_int) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        let w = Self::pt_bytes(nb, point_format);
        Self::arg(nb != 0 && points.len() % w == 0)?;
        let n = points.len() / w;
        let mut out = vec![0u8; n * (nb + 1)];
        self.check(unsafe { ecgpu_to_bytes_batch(self.0, curve, points.as_ptr(), point_format, out.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    /// GroupEncoding::from_bytes -> (x || y, ok flags)
    pub fn from_bytes(&self, curve: c_int, encoded: &[u8]) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && encoded.len() % (nb + 1) == 0)?;
        let n = encoded.len() / (nb + 1);
        let (mut xy, mut ok) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_from_bytes_batch(self.0, curve, encoded.as_ptr(), xy.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((xy, ok))
    }
    /// VerifyPrimitive::verify_prehashed for a batch: one accept flag per (prehash, r || s, x || y)
    pub fn ecdsa_verify(&self, curve: c_int, prehash: &[u8], sig_rs: &[u8], pubkeys_xy: &[u8], flags: c_uint) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && prehash.len() % nb == 0 && sig_rs.len() == 2 * prehash.len() && pubkeys_xy.len() == 2 * prehash.len())?;
        let mut ok = vec![0u8; prehash.len() / nb];
        self.check(unsafe { ecgpu_ecdsa_verify_batch(self.0, curve, prehash.as_ptr(), sig_rs.as_ptr(), pubkeys_xy.as_ptr(), ok.as_mut_ptr(), ok.len(), ECGPU_MEM_HOST, flags) })?;
        Ok(ok)
    }
    /// SignPrimitive::try_sign_prehashed for a batch -> (r || s, recovery ids, ok flags); constant-time k G unless
    /// `ECGPU_PUBLIC_SCALARS` is set
    pub fn ecdsa_sign(&self, curve: c_int, secret_d: &[u8], nonce_k: &[u8], prehash: &[u8], flags: c_uint) -> Result<(Vec<u8>, Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && secret_d.len() % nb == 0 && nonce_k.len() == secret_d.len() && prehash.len() == secret_d.len())?;
        let n = secret_d.len() / nb;
        let (mut sig, mut rec, mut ok) = (vec![0u8; 2 * nb * n], vec![0u8; n], vec![0u8; n]);
        self.check(unsafe {
            ecgpu_ecdsa_sign_batch(self.0, curve, secret_d.as_ptr(), nonce_k.as_ptr(), prehash.as_ptr(), sig.as_mut_ptr(), rec.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST,
                                   flags)
        })?;
        Ok((sig, rec, ok))
    }
    /// VerifyingKey::recover_from_prehash for a batch -> (x || y, ok flags)
    pub fn ecdsa_recover(&self, curve: c_int, prehash: &[u8], sig_rs: &[u8], recovery_id: &[u8], flags: c_uint) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && prehash.len() == recovery_id.len() * nb && sig_rs.len() == 2 * prehash.len())?;
        let n = recovery_id.len();
        let (mut xy, mut ok) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe {
            ecgpu_ecdsa_recover_batch(self.0, curve, prehash.as_ptr(), sig_rs.as_ptr(), recovery_id.as_ptr(), xy.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST, flags)
        })?;
        Ok((xy, ok))
    }
    /// elliptic_curve::ecdh::diffie_hellman for a batch -> (x of the shared points, ok flags); constant-time multiplication
    pub fn ecdh(&self, curve: c_int, secret_scalars: &[u8], pubkeys_xy: &[u8]) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && secret_scalars.len() % nb == 0 && pubkeys_xy.len() == 2 * secret_scalars.len())?;
        let n = secret_scalars.len() / nb;
        let (mut shared, mut ok) = (vec![0u8; nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_ecdh_batch(self.0, curve, secret_scalars.as_ptr(), pubkeys_xy.as_ptr(), shared.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((shared, ok))
    }
    /// the elliptic-curve part of BIP340 verification (challenges = tagged hashes, 32 bytes each)
    pub fn schnorr_verify(&self, pubkeys_x: &[u8], sig_rs: &[u8], challenges: &[u8]) -> Result<Vec<u8>, Error> {
        Self::arg(pubkeys_x.len() % 32 == 0 && sig_rs.len() == 2 * pubkeys_x.len() && challenges.len() == pubkeys_x.len())?;
        let mut ok = vec![0u8; pubkeys_x.len() / 32];
        self.check(unsafe { ecgpu_schnorr_verify_batch(self.0, ECGPU_K256, pubkeys_x.as_ptr(), sig_rs.as_ptr(), challenges.as_ptr(), ok.as_mut_ptr(), ok.len(), ECGPU_MEM_HOST) })?;
        Ok(ok)
    }
    /// MapToCurve::map_to_curve (count = 1) or Q0 + Q1 of hash_from_bytes (count = 2) -> (x || y, infinity flags)
    pub fn map_to_curve(&self, curve: c_int, u: &[u8], count: c_int) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && (count == 1 || count == 2) && u.len() % (nb * count as usize) == 0)?;
        let n = u.len() / (nb * count as usize);
        let (mut xy, mut inf) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_map_to_curve_batch(self.0, curve, u.as_ptr(), count, xy.as_mut_ptr(), inf.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((xy, inf))
    }
    /// ToEncodedPoint::to_encoded_point(compress) in fixed-width records (identity: tag 0x00 and zero padding)
    pub fn sec1_encode(&self, curve: c_int, points: &[u8], point_format: c_int, compress: bool) -> Result<Vec<u8>, Error> {
        let nb = Self::field_bytes(curve);
        let w = Self::pt_bytes(nb, point_format);
        Self::arg(nb != 0 && points.len() % w == 0)?;
        let n = points.len() / w;
        let mut out = vec![0u8; n * (1 + if compress { nb } else { 2 * nb })];
        self.check(unsafe { ecgpu_sec1_encode_batch(self.0, curve, points.as_ptr(), point_format, compress as c_int, out.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok(out)
    }
    /// FromEncodedPoint::from_encoded_point on records of 1 + NB or 1 + 2 NB bytes -> (x || y, ok flags)
    pub fn sec1_decode(&self, curve: c_int, encoded: &[u8], record_bytes: usize) -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Self::field_bytes(curve);
        Self::arg(nb != 0 && (record_bytes == 1 + nb || record_bytes == 1 + 2 * nb) && encoded.len() % record_bytes == 0)?;
        let n = encoded.len() / record_bytes;
        let (mut xy, mut ok) = (vec![0u8; 2 * nb * n], vec![0u8; n]);
        self.check(unsafe { ecgpu_sec1_decode_batch(self.0, curve, encoded.as_ptr(), record_bytes, xy.as_mut_ptr(), ok.as_mut_ptr(), n, ECGPU_MEM_HOST) })?;
        Ok((xy, ok))
    }
    pub fn set_option(&self, option: c_int, value: i64) -> Result<(), Error> { self.check(unsafe { ecgpu_set_option(self.0, option, value) }) }
    pub fn use_own_stream(&self) -> Result<(), Error> { self.check(unsafe { ecgpu_use_own_stream(self.0) }) }
    /// All later launches go to the caller's `hipStream_t`.  Since library version 0.3 a NULL handle is the legacy default stream
    /// itself, as for every HIP API - it is NOT "back to the context's own stream" (that is `use_own_stream`).
    ///
    /// # Safety
    /// `hip_stream` must be NULL or a live stream of this context's device, and must outlive its use by the context.
    pub unsafe fn set_stream(&self, hip_stream: *mut c_void) -> Result<(), Error> { self.check(ecgpu_set_stream(self.0, hip_stream)) }
    /// The legacy default stream (handle 0): ordered with every blocking stream of the process.
    pub fn use_default_stream(&self) -> Result<(), Error> { self.check(unsafe { ecgpu_set_stream(self.0, core::ptr::null_mut()) }) }
    pub fn synchronize(&self) -> Result<(), Error> { self.check(unsafe { ecgpu_synchronize(self.0) }) }
}

/// A device group (`ecgpu_group_create`): the single-call entry points split over several GPUs.  Independent batches take
/// contiguous index ranges per device (no collective); one split sum runs the bucket method per device, all-gathers one point per
/// device (RCCL over xGMI when the devices are distinct) and folds on the first device.  Host buffers only.
pub struct Group(*mut ecgpu_group);
unsafe impl Send for Group {}
unsafe impl Sync for Group {}          // calls on one group are serialised by the library

impl Drop for Group {
    fn drop(&mut self) {
        unsafe { ecgpu_group_destroy(self.0) }
    }
}

impl Group {
    /// `devices` may repeat a device (several contexts on one card).
    pub fn new(devices: &[i32], flags: c_uint) -> Result<Self, Error> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { ecgpu_group_create(&mut p, devices.as_ptr(), devices.len() as c_int, flags) };
        if rc == ECGPU_OK { Ok(Group(p)) } else { Err(Error { code: rc, message: "ecgpu_group_create failed (no gfx950 device? there is no CPU fallback)".into() }) }
    }
    pub fn raw(&self) -> *mut ecgpu_group { self.0 }
    pub fn size(&self) -> usize { unsafe { ecgpu_group_size(self.0) as usize } }
    fn check(&self, rc: c_int) -> Result<(), Error> {
        if rc == ECGPU_OK { return Ok(()); }
        let msg = unsafe { std::ffi::CStr::from_ptr(ecgpu_group_last_error(self.0)) }.to_string_lossy().into_owned();
        Err(Error { code: rc, message: msg })
    }
    /// `Context::lincomb` over the group: n independent linear combinations, member i computes the range `ecgpu_shard_range(n, size, i)`.
    #[allow(clippy::too_many_arguments)]
    pub fn lincomb(&self, curve: c_int, scalars: &[u8], points: Option<&[u8]>, point_format: c_int, terms: usize, out_format: c_int, flags: c_uint)
                   -> Result<(Vec<u8>, Vec<u8>), Error> {
        let nb = Context::field_bytes(curve);
        Context::arg(nb != 0 && terms >= 1 && scalars.len() % (nb * terms) == 0)?;
        let n = scalars.len() / (nb * terms);
        Context::arg(points.map_or(terms == 1, |p| p.len() == n * terms * Context::pt_bytes(nb, point_format)))?;
        let mut out = vec![0u8; n * Context::pt_bytes(nb, out_format)];
        let mut inf = vec![0u8; n];
        let pp = points.map_or(core::ptr::null(), |p| p.as_ptr());
        self.check(unsafe { ecgpu_group_lincomb_batch(self.0, curve, scalars.as_ptr(), pp, point_format, terms, out.as_mut_ptr(), out_format, inf.as_mut_ptr(), n, flags) })?;
        Ok((out, inf))
    }
    /// One sum over all terms, split over the group's devices: the 8-GPU form of `lincomb_ext` over a slice.
    pub fn msm(&self, curve: c_int, scalars: &[u8], points: &[u8], point_format: c_int, out_format: c_int) -> Result<Vec<u8>, Error> {
        let nb = Context::field_bytes(curve);
        Context::arg(nb != 0 && scalars.len() % nb == 0 && points.len() == scalars.len() / nb * Context::pt_bytes(nb, point_format))?;
        let mut out = vec![0u8; Context::pt_bytes(nb, out_format)];
        self.check(unsafe { ecgpu_group_msm(self.0, curve, scalars.as_ptr(), points.as_ptr(), point_format, scalars.len() / nb, out.as_mut_ptr(), out_format) })?;
        Ok(out)
    }
}
