//! UNBUILT SOURCE (no Rust toolchain offline) - the `extern "C"` block mirrors include/ecgpu.h and
//! the wrappers adapt slices of the reference's own types to the batch ABI.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_uint, c_void};
use elliptic_curve::sec1::ToEncodedPoint;
use k256::{AffinePoint, ProjectivePoint, Scalar};

#[repr(C)]
pub struct ecgpu_ctx {
    _private: [u8; 0],
}
pub const ECGPU_K256: c_int = 0;
pub const ECGPU_MEM_HOST: c_int = 0;
pub const ECGPU_PT_AFFINE: c_int = 0;

#[link(name = "ecgpu")]
extern "C" {
    pub fn ecgpu_create(ctx: *mut *mut ecgpu_ctx, device_index: c_int) -> c_int;
    pub fn ecgpu_destroy(ctx: *mut ecgpu_ctx);
    pub fn ecgpu_last_error(ctx: *const ecgpu_ctx) -> *const c_char;
    pub fn ecgpu_set_stream(ctx: *mut ecgpu_ctx, hip_stream: *mut c_void) -> c_int;
    pub fn ecgpu_mul_batch(
        ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int,
        out: *mut u8, out_format: c_int, out_inf: *mut u8, n: usize, mem: c_int, flags: c_uint,
    ) -> c_int;
    pub fn ecgpu_lincomb_batch(
        ctx: *mut ecgpu_ctx, curve: c_int, scalars: *const u8, points: *const u8, point_format: c_int,
        terms: usize, out: *mut u8, out_format: c_int, out_inf: *mut u8, n: usize, mem: c_int, flags: c_uint,
    ) -> c_int;
    pub fn ecgpu_batch_normalize(
        ctx: *mut ecgpu_ctx, curve: c_int, p_xyz: *const u8, out_xy: *mut u8, out_inf: *mut u8, n: usize, mem: c_int,
    ) -> c_int;
    pub fn ecgpu_ecdsa_verify_batch(
        ctx: *mut ecgpu_ctx, curve: c_int, prehash: *const u8, sig_rs: *const u8, pubkeys_xy: *const u8, ok: *mut u8,
        n: usize, mem: c_int, flags: c_uint,
    ) -> c_int;
    pub fn ecgpu_ecdsa_sign_batch(
        ctx: *mut ecgpu_ctx, curve: c_int, secret_d: *const u8, nonce_k: *const u8, prehash: *const u8, sig_rs: *mut u8,
        recovery_id: *mut u8, ok: *mut u8, n: usize, mem: c_int, flags: c_uint,
    ) -> c_int;
}

/// `ECGPU_ECDSA_LOW_S`: the k256 rules of k256/src/ecdsa.rs:182-207.
pub const ECGPU_ECDSA_LOW_S: c_uint = 2;

pub struct Gpu(*mut ecgpu_ctx);

impl Gpu {
    pub fn new(device: i32) -> Result<Self, i32> {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { ecgpu_create(&mut p, device) };
        if rc == 0 { Ok(Gpu(p)) } else { Err(rc) }
    }

    /// Bulk form of `MulByGenerator::mul_by_generator` (k256/src/arithmetic/mul.rs:415-440).
    pub fn mul_by_generator(&self, scalars: &[Scalar]) -> Result<Vec<AffinePoint>, i32> {
        self.mul_impl(scalars, None)
    }

    /// Bulk form of `&P * &k` (k256/src/arithmetic/mul.rs:455-461).
    pub fn mul(&self, scalars: &[Scalar], points: &[AffinePoint]) -> Result<Vec<AffinePoint>, i32> {
        assert_eq!(scalars.len(), points.len());
        self.mul_impl(scalars, Some(points))
    }

    fn mul_impl(&self, scalars: &[Scalar], points: Option<&[AffinePoint]>) -> Result<Vec<AffinePoint>, i32> {
        let n = scalars.len();
        let mut s = Vec::with_capacity(32 * n);
        for k in scalars { s.extend_from_slice(&k.to_bytes()); }            // Scalar::to_bytes, scalar.rs:94-96
        let p: Option<Vec<u8>> = points.map(|ps| {
            let mut v = Vec::with_capacity(64 * n);
            for a in ps {
                let e = a.to_encoded_point(false);                           // affine.rs:272-284
                match (e.x(), e.y()) {
                    (Some(x), Some(y)) => { v.extend_from_slice(x); v.extend_from_slice(y); }
                    _ => v.extend_from_slice(&[0u8; 64]),                    // identity = zeros
                }
            }
            v
        });
        let mut out = vec![0u8; 64 * n];
        let mut inf = vec![0u8; n];
        let rc = unsafe {
            ecgpu_mul_batch(self.0, ECGPU_K256, s.as_ptr(), p.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()),
                            ECGPU_PT_AFFINE, out.as_mut_ptr(), ECGPU_PT_AFFINE, inf.as_mut_ptr(), n, ECGPU_MEM_HOST, 0)
        };
        if rc != 0 { return Err(rc); }
        Ok((0..n).map(|i| decode_affine(&out[64 * i..64 * i + 64], inf[i])).collect())
    }
}

fn decode_affine(xy: &[u8], inf: u8) -> AffinePoint {
    use elliptic_curve::sec1::FromEncodedPoint;
    if inf != 0 { return AffinePoint::IDENTITY; }
    let e = k256::EncodedPoint::from_affine_coordinates(xy[..32].into(), xy[32..].into(), false);
    AffinePoint::from_encoded_point(&e).unwrap()
}

impl Drop for Gpu {
    fn drop(&mut self) { unsafe { ecgpu_destroy(self.0) } }
}
