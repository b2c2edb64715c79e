//! UNBUILT SOURCE.  Layer 2 of the Rust side for secp256k1: a module a maintainer would add as
//! `k256/src/arithmetic/gpu.rs` (inside the crate: it reads the private coordinates of `ProjectivePoint`).
//! Bulk, trait-shaped entry points over the reference's own types, backed by `ecgpu_sys::Context`.
//!
//! Wire format = what the reference's own serialisers produce: `FieldElement::to_bytes` after `normalize`
//! (field.rs:118-120, field_5x52.rs:96-131), `Scalar::to_bytes` (scalar.rs:94-96); nothing here depends on the in-memory
//! layout of the types (which is not ABI-stable, field_impl.rs:23-28).
use alloc::vec::Vec;
use ecgpu_sys::{Context, Error, Group, ECGPU_EXACT_REFERENCE, ECGPU_K256, ECGPU_PT_AFFINE, ECGPU_PT_PROJECTIVE};

use crate::{AffinePoint, FieldBytes, ProjectivePoint, Scalar};
use super::FieldElement;

fn put_projective(buf: &mut Vec<u8>, p: &ProjectivePoint) {
    buf.extend_from_slice(&p.x.normalize().to_bytes());
    buf.extend_from_slice(&p.y.normalize().to_bytes());
    buf.extend_from_slice(&p.z.normalize().to_bytes());
}
fn get_field(b: &[u8]) -> FieldElement {
    // the library only ever returns canonical values (< p): from_bytes cannot fail on them
    FieldElement::from_bytes(FieldBytes::from_slice(b)).unwrap()
}
fn get_projective(b: &[u8]) -> ProjectivePoint {
    ProjectivePoint { x: get_field(&b[..32]), y: get_field(&b[32..64]), z: get_field(&b[64..96]) }
}
fn get_affine(xy: &[u8], infinity: u8) -> AffinePoint {
    if infinity != 0 { AffinePoint::IDENTITY } else { AffinePoint { x: get_field(&xy[..32]), y: get_field(&xy[32..64]), infinity: 0 } }
}
fn put_scalars<'a>(it: impl Iterator<Item = &'a Scalar>) -> Vec<u8> {
    let mut v = Vec::new();
    for k in it { v.extend_from_slice(&k.to_bytes()); }
    v
}

/// Bulk `MulByGenerator::mul_by_generator` (mul.rs:415-440): out[i] = scalars[i] * G, the exact (X, Y, Z) the CPU path
/// returns (constant-time schedule: the scalars may be secret).
pub fn mul_by_generator_batch(gpu: &Context, scalars: &[Scalar]) -> Result<Vec<ProjectivePoint>, Error> {
    let s = put_scalars(scalars.iter());
    let (out, _, _) = gpu.lincomb(ECGPU_K256, &s, None, ECGPU_PT_AFFINE, 1, ECGPU_PT_PROJECTIVE, ECGPU_EXACT_REFERENCE, false)?;
    Ok(out.chunks_exact(96).map(get_projective).collect())
}

/// Bulk `&ProjectivePoint * &Scalar` (mul.rs:442-481) on PUBLIC data, throughput schedule: the group elements as affine
/// points (use `flags = ECGPU_EXACT_REFERENCE` through `Context::lincomb` for secret scalars or exact triples).
pub fn mul_batch(gpu: &Context, points: &[ProjectivePoint], scalars: &[Scalar]) -> Result<Vec<AffinePoint>, Error> {
    assert_eq!(points.len(), scalars.len());
    let s = put_scalars(scalars.iter());
    let mut p = Vec::with_capacity(96 * points.len());
    for q in points { put_projective(&mut p, q); }
    let (out, inf, _) = gpu.lincomb(ECGPU_K256, &s, Some(&p), ECGPU_PT_PROJECTIVE, 1, ECGPU_PT_AFFINE, 0, false)?;
    Ok(out.chunks_exact(64).zip(inf.iter()).map(|(xy, i)| get_affine(xy, *i)).collect())
}

/// `LinearCombinationExt<[(ProjectivePoint, Scalar)]>::lincomb_ext` (mul.rs:325-340) at MSM scale: one bucket-method sum
/// over the whole slice, split over the devices of a `Group` (every device sums a contiguous range of the terms, one point per
/// device is all-gathered - RCCL over xGMI - and folded on the first device; a group of one device is the single-GPU call).
/// The group element is that of the CPU path; the representative is (x : y : 1).
pub fn lincomb_ext_gpu(gpus: &Group, points_and_scalars: &[(ProjectivePoint, Scalar)]) -> Result<ProjectivePoint, Error> {
    let s = put_scalars(points_and_scalars.iter().map(|(_, k)| k));
    let mut p = Vec::with_capacity(96 * points_and_scalars.len());
    for (q, _) in points_and_scalars { put_projective(&mut p, q); }
    let out = gpus.msm(ECGPU_K256, &s, &p, ECGPU_PT_PROJECTIVE, ECGPU_PT_PROJECTIVE)?;
    Ok(get_projective(&out))
}

/// Bulk `&ProjectivePoint * &Scalar` over a `Group`: contiguous index ranges per device, no collective.
pub fn mul_batch_group(gpus: &Group, points: &[ProjectivePoint], scalars: &[Scalar]) -> Result<Vec<AffinePoint>, Error> {
    assert_eq!(points.len(), scalars.len());
    let s = put_scalars(scalars.iter());
    let mut p = Vec::with_capacity(96 * points.len());
    for q in points { put_projective(&mut p, q); }
    let (out, inf) = gpus.lincomb(ECGPU_K256, &s, Some(&p), ECGPU_PT_PROJECTIVE, 1, ECGPU_PT_AFFINE, 0)?;
    Ok(out.chunks_exact(64).zip(inf.iter()).map(|(xy, i)| get_affine(xy, *i)).collect())
}

/// `LinearCombination::lincomb(x, k, y, l)` (mul.rs:313-323) for many independent pairs: out[i] = x_i k_i + y_i l_i.
pub fn lincomb_batch(gpu: &Context, terms: &[[(ProjectivePoint, Scalar); 2]]) -> Result<Vec<AffinePoint>, Error> {
    let s = put_scalars(terms.iter().flat_map(|t| t.iter().map(|(_, k)| k)));
    let mut p = Vec::with_capacity(192 * terms.len());
    for t in terms { for (q, _) in t { put_projective(&mut p, q); } }
    let (out, inf, _) = gpu.lincomb(ECGPU_K256, &s, Some(&p), ECGPU_PT_PROJECTIVE, 2, ECGPU_PT_AFFINE, 0, false)?;
    Ok(out.chunks_exact(64).zip(inf.iter()).map(|(xy, i)| get_affine(xy, *i)).collect())
}

/// `BatchNormalize<[ProjectivePoint]>::batch_normalize` (projective.rs:337-348): one shared inversion per 16 points on
/// the device instead of one `BatchInvert` over the slice; identities come back as `AffinePoint::IDENTITY` (:361-364).
pub fn batch_normalize_gpu(gpu: &Context, points: &[ProjectivePoint]) -> Result<Vec<AffinePoint>, Error> {
    let mut p = Vec::with_capacity(96 * points.len());
    for q in points { put_projective(&mut p, q); }
    let (xy, inf) = gpu.batch_normalize(ECGPU_K256, &p)?;
    Ok(xy.chunks_exact(64).zip(inf.iter()).map(|(c, i)| get_affine(c, *i)).collect())
}

/// `ProjectivePoint::ct_eq` (projective.rs:421-446) for many pairs.
pub fn eq_batch(gpu: &Context, a: &[ProjectivePoint], b: &[ProjectivePoint]) -> Result<Vec<bool>, Error> {
    assert_eq!(a.len(), b.len());
    let (mut pa, mut pb) = (Vec::with_capacity(96 * a.len()), Vec::with_capacity(96 * b.len()));
    for q in a { put_projective(&mut pa, q); }
    for q in b { put_projective(&mut pb, q); }
    Ok(gpu.point_eq(ECGPU_K256, &pa, &pb)?.into_iter().map(|f| f != 0).collect())
}
