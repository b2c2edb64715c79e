//! UNBUILT SOURCE.  Layer 2 of the Rust side for the primeorder curves (P-256, P-384): a module a maintainer would add
//! as `primeorder/src/gpu.rs` (inside the crate: `ProjectivePoint<C>`'s coordinates are `pub(crate)`).
//!
//! Coordinates travel as canonical big-endian bytes, `PrimeField::to_repr` of `C::FieldElement` (for these curves that
//! leaves the Montgomery domain first: p256 field.rs:281-284, fiat_p384_from_montgomery); scalars as `to_repr` too
//! (p384's `Scalar` is in Montgomery form internally, scalar.rs:335-339 - `to_repr` canonicalises it).
use alloc::vec::Vec;
use ecgpu_sys::{Context, Error, ECGPU_EXACT_REFERENCE, ECGPU_P256, ECGPU_P384, ECGPU_PT_AFFINE, ECGPU_PT_PROJECTIVE};
use elliptic_curve::{ff::PrimeField, FieldBytes, Scalar};

use crate::{AffinePoint, PrimeCurveParams, ProjectivePoint};

/// The library's curve id for a `PrimeCurveParams` implementor.
pub trait GpuCurve: PrimeCurveParams {
    const ECGPU_ID: i32;
}
// in p256 / p384:   impl primeorder::gpu::GpuCurve for NistP256 { const ECGPU_ID: i32 = ECGPU_P256; }
//                   impl primeorder::gpu::GpuCurve for NistP384 { const ECGPU_ID: i32 = ECGPU_P384; }
const _: (i32, i32) = (ECGPU_P256, ECGPU_P384);

fn nb<C: GpuCurve>() -> usize { Context::field_bytes(C::ECGPU_ID) }
fn put_projective<C: GpuCurve>(buf: &mut Vec<u8>, p: &ProjectivePoint<C>) {
    buf.extend_from_slice(p.x.to_repr().as_ref());
    buf.extend_from_slice(p.y.to_repr().as_ref());
    buf.extend_from_slice(p.z.to_repr().as_ref());
}
fn get_field<C: GpuCurve>(b: &[u8]) -> C::FieldElement {
    C::FieldElement::from_repr(FieldBytes::<C>::clone_from_slice(b)).unwrap()      // canonical by construction
}
fn get_projective<C: GpuCurve>(b: &[u8]) -> ProjectivePoint<C> {
    let n = nb::<C>();
    ProjectivePoint { x: get_field::<C>(&b[..n]), y: get_field::<C>(&b[n..2 * n]), z: get_field::<C>(&b[2 * n..3 * n]) }
}
fn get_affine<C: GpuCurve>(xy: &[u8], infinity: u8) -> AffinePoint<C> {
    let n = nb::<C>();
    if infinity != 0 { AffinePoint::IDENTITY } else { AffinePoint { x: get_field::<C>(&xy[..n]), y: get_field::<C>(&xy[n..2 * n]), infinity: 0 } }
}
fn put_scalars<'a, C: GpuCurve>(it: impl Iterator<Item = &'a Scalar<C>>) -> Vec<u8> {
    let mut v = Vec::new();
    for k in it { v.extend_from_slice(k.to_repr().as_ref()); }
    v
}

/// Bulk `MulByGenerator::mul_by_generator` (projective.rs:422-431, "TODO: precomputed basepoint tables" - the device has
/// them): throughput schedule for PUBLIC scalars, affine results.
pub fn mul_by_generator_batch<C: GpuCurve>(gpu: &Context, scalars: &[Scalar<C>]) -> Result<Vec<AffinePoint<C>>, Error> {
    let s = put_scalars::<C>(scalars.iter());
    let (out, inf, _) = gpu.lincomb(C::ECGPU_ID, &s, None, ECGPU_PT_AFFINE, 1, ECGPU_PT_AFFINE, 0, false)?;
    Ok(out.chunks_exact(2 * nb::<C>()).zip(inf.iter()).map(|(xy, i)| get_affine::<C>(xy, *i)).collect())
}

/// Bulk `ProjectivePoint::mul` (projective.rs:106-150) exactly as the CPU path computes it - same (X, Y, Z), constant-time
/// 15-way table scans - for secret scalars (ECDH, nonces).
pub fn mul_exact_batch<C: GpuCurve>(gpu: &Context, points: &[ProjectivePoint<C>], scalars: &[Scalar<C>]) -> Result<Vec<ProjectivePoint<C>>, Error> {
    assert_eq!(points.len(), scalars.len());
    let s = put_scalars::<C>(scalars.iter());
    let mut p = Vec::with_capacity(3 * nb::<C>() * points.len());
    for q in points { put_projective::<C>(&mut p, q); }
    let (out, _, _) = gpu.lincomb(C::ECGPU_ID, &s, Some(&p), ECGPU_PT_PROJECTIVE, 1, ECGPU_PT_PROJECTIVE, ECGPU_EXACT_REFERENCE, false)?;
    Ok(out.chunks_exact(3 * nb::<C>()).map(get_projective::<C>).collect())
}

/// The default `LinearCombination::lincomb` (x * k + y * l, projective.rs:415-420) for many independent pairs.
pub fn lincomb_batch<C: GpuCurve>(gpu: &Context, terms: &[[(ProjectivePoint<C>, Scalar<C>); 2]]) -> Result<Vec<AffinePoint<C>>, Error> {
    let s = put_scalars::<C>(terms.iter().flat_map(|t| t.iter().map(|(_, k)| k)));
    let mut p = Vec::with_capacity(6 * nb::<C>() * terms.len());
    for t in terms { for (q, _) in t { put_projective::<C>(&mut p, q); } }
    let (out, inf, _) = gpu.lincomb(C::ECGPU_ID, &s, Some(&p), ECGPU_PT_PROJECTIVE, 2, ECGPU_PT_AFFINE, 0, false)?;
    Ok(out.chunks_exact(2 * nb::<C>()).zip(inf.iter()).map(|(xy, i)| get_affine::<C>(xy, *i)).collect())
}

/// `BatchNormalize<[ProjectivePoint<C>]>::batch_normalize` (projective.rs:363-379).
pub fn batch_normalize_gpu<C: GpuCurve>(gpu: &Context, points: &[ProjectivePoint<C>]) -> Result<Vec<AffinePoint<C>>, Error> {
    let mut p = Vec::with_capacity(3 * nb::<C>() * points.len());
    for q in points { put_projective::<C>(&mut p, q); }
    let (xy, inf) = gpu.batch_normalize(C::ECGPU_ID, &p)?;
    Ok(xy.chunks_exact(2 * nb::<C>()).zip(inf.iter()).map(|(c, i)| get_affine::<C>(c, *i)).collect())
}

/// `ProjectivePoint<C>: ConstantTimeEq` (projective.rs:191-198, equality of the affine forms) for many pairs.
pub fn eq_batch<C: GpuCurve>(gpu: &Context, a: &[ProjectivePoint<C>], b: &[ProjectivePoint<C>]) -> Result<Vec<bool>, Error> {
    assert_eq!(a.len(), b.len());
    let (mut pa, mut pb) = (Vec::new(), Vec::new());
    for q in a { put_projective::<C>(&mut pa, q); }
    for q in b { put_projective::<C>(&mut pb, q); }
    Ok(gpu.point_eq(C::ECGPU_ID, &pa, &pb)?.into_iter().map(|f| f != 0).collect())
}
