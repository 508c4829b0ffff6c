//! rust/dock_gpu/src/generic.rs — GENERIC drop-ins for the arkworks calls the reference makes through type parameters.
//!
//! The reference never names `G1Affine` at its hot call sites: `Pairs<'_, '_, G, G::ScalarField>::msm` (utils/src/pairs.rs:143-156),
//! `OwnedPairs<G, _>::msm` (owned_pairs.rs:93-106), `RandomizedMultChecker<G>::verify` (randomized_mult_checker.rs:93-101) and
//! `calculate_coeff<G>` (legogroth16/src/prover.rs:585-594) are generic over `G: AffineRepr`; `RandomizedPairingChecker<E>`
//! (randomized_pairing_check.rs:116-214), `create_proof…<E>` (prover.rs:267-383) and `verify_qap_proof<E>` (verifier.rs:62-84) over
//! `E: Pairing`.  A drop-in therefore has to be generic too.  The functions below have the signature of the arkworks call they replace,
//! decide by `TypeId` whether the type parameter is one the library serves (BLS12-381 G1 / G2 / the pairing), reinterpret the slices as the
//! concrete types of `crate` (same type, so the reinterpretation is the identity) and fall through to arkworks for every other curve —
//! `rust/patches/*.diff` are the one-line edits that route the reference's call sites here behind `feature = "gpu"`.
//!
//! `AffineRepr` and `Pairing` are `'static` (ark-ec 0.4 `AffineRepr: Eq + 'static + …`, `Pairing: Sized + 'static + …`), which is what
//! `TypeId::of` needs.  Not compiled in the build image (no Rust toolchain): `tests/test_rust_shim_consistency.py` checks this file's `extern`
//! usage against `include/dock_gpu.h`; `cargo test` on a machine with cargo compiles and runs `tests/parity.rs::generic_*`.
use core::any::TypeId;

use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::bls12::{G1Prepared as ArkG1Prepared, G2Prepared as ArkG2Prepared};
use ark_ec::pairing::{MillerLoopOutput, Pairing, PairingOutput};
use ark_ec::{AffineRepr, VariableBaseMSM};
use ark_ff::{BigInt, PrimeField};
use ark_std::vec::Vec;

type Cfg = ark_bls12_381::Config;

/// Pairs below this count stay on arkworks in the generic Miller loop: three pairs take 0.53 ms through the library against ≈ 0.9 ms on one
/// host core, 1024 pairs 0.72 ms against 86 ms on 64 threads; below eight the call is launch latency.
pub const MIN_PAIRS_GPU: usize = 3;

#[inline(always)]
fn same<A: 'static, B: 'static>() -> bool { TypeId::of::<A>() == TypeId::of::<B>() }
/// `&[A]` as `&[B]` — only ever called after `same::<A, B>()` (then it is the identity)
#[inline(always)]
unsafe fn cast_slice<A, B>(s: &[A]) -> &[B] {
    debug_assert_eq!(core::mem::size_of::<A>(), core::mem::size_of::<B>());
    core::slice::from_raw_parts(s.as_ptr() as *const B, s.len())
}
/// `A` as `B` by value — only ever called after `same::<A, B>()`
#[inline(always)]
unsafe fn cast_val<A, B>(a: A) -> B {
    debug_assert_eq!(core::mem::size_of::<A>(), core::mem::size_of::<B>());
    let b = core::mem::transmute_copy::<A, B>(&a);
    core::mem::forget(a);
    b
}

// ---- VariableBaseMSM ------------------------------------------------------------------------------------------------------------------------
/// `G::Group::msm_unchecked(bases, scalars)` — utils/src/pairs.rs:145-147, owned_pairs.rs:95-97, randomized_mult_checker.rs:100
pub fn msm_unchecked<G: AffineRepr>(bases: &[G], scalars: &[G::ScalarField]) -> G::Group {
    if same::<G, G1Affine>() {
        // G == G1Affine, hence G::ScalarField == Fr and G::Group == G1Projective
        return unsafe { cast_val::<G1Projective, G::Group>(crate::msm_unchecked_g1(cast_slice::<G, G1Affine>(bases), cast_slice::<G::ScalarField, Fr>(scalars))) };
    }
    if same::<G, G2Affine>() {
        return unsafe { cast_val::<G2Projective, G::Group>(crate::msm_unchecked_g2(cast_slice::<G, G2Affine>(bases), cast_slice::<G::ScalarField, Fr>(scalars))) };
    }
    G::Group::msm_unchecked(bases, scalars)
}
/// `G::Group::msm(bases, scalars)`: `Err(min(len))` when the lengths differ, like arkworks (ark-ec 0.4 `VariableBaseMSM::msm`)
pub fn msm<G: AffineRepr>(bases: &[G], scalars: &[G::ScalarField]) -> Result<G::Group, usize> {
    if bases.len() != scalars.len() { return Err(bases.len().min(scalars.len())); }
    Ok(msm_unchecked(bases, scalars))
}
/// `G::Group::msm_bigint(bases, bigints)` — legogroth16/src/prover.rs:286,299,363,592; utils/src/pairs.rs:153-155, owned_pairs.rs:103-105
pub fn msm_bigint<G: AffineRepr>(bases: &[G], bigints: &[<G::ScalarField as PrimeField>::BigInt]) -> G::Group {
    if same::<G, G1Affine>() {
        return unsafe { cast_val::<G1Projective, G::Group>(crate::msm_bigint_g1(cast_slice::<G, G1Affine>(bases), cast_slice::<<G::ScalarField as PrimeField>::BigInt, BigInt<4>>(bigints))) };
    }
    if same::<G, G2Affine>() {
        return unsafe { cast_val::<G2Projective, G::Group>(crate::msm_bigint_g2(cast_slice::<G, G2Affine>(bases), cast_slice::<<G::ScalarField as PrimeField>::BigInt, BigInt<4>>(bigints))) };
    }
    G::Group::msm_bigint(bases, bigints)
}
/// the name `north_star` uses for the same thing (`variable_base_msm(&[G], &[G::ScalarField]) -> G::Group`)
#[inline]
pub fn variable_base_msm<G: AffineRepr>(bases: &[G], scalars: &[G::ScalarField]) -> G::Group { msm_unchecked(bases, scalars) }

// ---- Pairing --------------------------------------------------------------------------------------------------------------------------------
/// `E::multi_miller_loop(a, b)` with arkworks' own signature — utils/src/randomized_pairing_check.rs:134,169-170,194,207;
/// legogroth16/src/verifier.rs:69-76.  The operands arrive prepared (that is what the signature promises), so for BLS12-381 they go to
/// `dgpu_multi_miller_loop_mixed` as its prepared half; a call site that still HOLDS the affine points should call
/// `multi_miller_loop_affine` instead (no 19.6 KB of line coefficients per pair across PCIe, preparation inside the kernel's own chain).
pub fn multi_miller_loop<E: Pairing>(a: impl IntoIterator<Item = impl Into<E::G1Prepared>>, b: impl IntoIterator<Item = impl Into<E::G2Prepared>>) -> MillerLoopOutput<E> {
    let a: Vec<E::G1Prepared> = a.into_iter().map(Into::into).collect();
    let b: Vec<E::G2Prepared> = b.into_iter().map(Into::into).collect();
    if same::<E, Bls12_381>() && a.len() == b.len() && a.len() >= MIN_PAIRS_GPU {
        // E == Bls12_381: E::G1Prepared == bls12::G1Prepared<Config> (a newtype around the affine point), E::G2Prepared == bls12::G2Prepared<Config>
        let pa: &[ArkG1Prepared<Cfg>] = unsafe { cast_slice(&a) };
        let pb: &[ArkG2Prepared<Cfg>] = unsafe { cast_slice(&b) };
        let g1: Vec<G1Affine> = pa.iter().map(|p| p.0).collect();
        let out = crate::multi_miller_loop_mixed(&[], &[], &g1, pb);
        return unsafe { cast_val::<MillerLoopOutput<Bls12_381>, MillerLoopOutput<E>>(out) };
    }
    E::multi_miller_loop(a, b)          // (arkworks panics on unequal lengths: so does this path)
}
/// the same over affine operands on both sides: what `RandomizedPairingChecker::verify` holds when it is built with `rust/patches`'
/// affine pending list, and what `E::multi_pairing(&[G1Affine], &[G2Affine])` call sites pass (67 of them, e.g. bbs_plus/src/signature.rs:284)
pub fn multi_miller_loop_affine<E: Pairing>(a: &[E::G1Affine], b: &[E::G2Affine]) -> MillerLoopOutput<E> {
    assert_eq!(a.len(), b.len(), "multi_miller_loop: lengths differ");
    if same::<E, Bls12_381>() && a.len() >= MIN_PAIRS_GPU {
        let out = crate::multi_miller_loop(unsafe { cast_slice::<E::G1Affine, G1Affine>(a) }, unsafe { cast_slice::<E::G2Affine, G2Affine>(b) });
        return unsafe { cast_val::<MillerLoopOutput<Bls12_381>, MillerLoopOutput<E>>(out) };
    }
    E::multi_miller_loop(a.iter().copied(), b.iter().copied())
}
/// `E::multi_miller_loop(a_i.mul_bigint(m), b_i)` — the scalings of `RandomizedPairingChecker::add_multiple_sources[_and_target]` and the loop
/// they feed (randomized_pairing_check.rs:125-134,152-170) as ONE call for a caller that holds the G2 operands affine: on the device the scaling
/// chains run beside the line chain of the `b_i` (`dgpu_multi_miller_loop_scaled`); arkworks for other curves and below the size where a launch pays
pub fn multi_miller_loop_scaled<E: Pairing>(a: &[E::G1Affine], m: <E::ScalarField as PrimeField>::BigInt, b: &[E::G2Affine]) -> MillerLoopOutput<E> {
    assert_eq!(a.len(), b.len(), "multi_miller_loop: lengths differ");
    if same::<E, Bls12_381>() && a.len() >= MIN_PAIRS_GPU {
        let m4: BigInt<4> = unsafe { cast_val(m) };
        let out = crate::multi_miller_loop_scaled(unsafe { cast_slice::<E::G1Affine, G1Affine>(a) }, &[m4], unsafe { cast_slice::<E::G2Affine, G2Affine>(b) });
        return unsafe { cast_val::<MillerLoopOutput<Bls12_381>, MillerLoopOutput<E>>(out) };
    }
    use ark_ec::CurveGroup;
    let scaled: Vec<E::G1Affine> = E::G1::normalize_batch(&a.iter().map(|p| p.mul_bigint(m)).collect::<Vec<_>>());
    E::multi_miller_loop(scaled, b.iter().copied())
}
/// `E::final_exponentiation(f)` — randomized_pairing_check.rs:213, verifier.rs:78
pub fn final_exponentiation<E: Pairing>(f: MillerLoopOutput<E>) -> Option<PairingOutput<E>> {
    if same::<E, Bls12_381>() {
        let r = crate::final_exponentiation(unsafe { cast_val::<MillerLoopOutput<E>, MillerLoopOutput<Bls12_381>>(f) });
        return r.map(|x| unsafe { cast_val::<PairingOutput<Bls12_381>, PairingOutput<E>>(x) });
    }
    E::final_exponentiation(f)
}
/// `E::multi_pairing(a, b)`
pub fn multi_pairing<E: Pairing>(a: &[E::G1Affine], b: &[E::G2Affine]) -> PairingOutput<E> {
    final_exponentiation(multi_miller_loop_affine::<E>(a, b)).expect("Miller output of valid operands is never zero")
}
/// `E::G2Prepared::from(q)` for a batch — randomized_pairing_check.rs:132,163,188-189; legogroth16/src/verifier.rs:22-23
pub fn g2_prepare<E: Pairing>(qs: &[E::G2Affine]) -> Vec<E::G2Prepared> {
    if same::<E, Bls12_381>() && qs.len() >= MIN_PAIRS_GPU {
        let v: Vec<ArkG2Prepared<Cfg>> = crate::g2_prepare(unsafe { cast_slice::<E::G2Affine, G2Affine>(qs) });
        return unsafe { cast_val::<Vec<ArkG2Prepared<Cfg>>, Vec<E::G2Prepared>>(v) };
    }
    qs.iter().map(|q| E::G2Prepared::from(*q)).collect()
}

/// `a.mul_bigint(m)` for a batch of G1 points and ONE scalar — the `cfg_iter!(a).map(|a| a.mul_bigint(m))` of
/// randomized_pairing_check.rs:126-129,152-158 (`dgpu_g1_scale_batch`: GLV, four lanes per point); arkworks for other curves
pub fn scale_batch_g1<E: Pairing>(points: &[E::G1Affine], m: <E::ScalarField as PrimeField>::BigInt, negate: bool) -> Vec<E::G1Affine> {
    if same::<E, Bls12_381>() && points.len() >= 64 {
        let m4: BigInt<4> = unsafe { cast_val(m) };
        if let Some(v) = crate::g1_scale_batch(unsafe { cast_slice::<E::G1Affine, G1Affine>(points) }, &m4, negate) {
            return unsafe { cast_val::<Vec<G1Affine>, Vec<E::G1Affine>>(v) };
        }
    }
    use ark_ec::CurveGroup;
    let p: Vec<E::G1> = points.iter().map(|a| { let t = a.mul_bigint(m); if negate { -t } else { t } }).collect();
    E::G1::normalize_batch(&p)
}

// ---- LegoGroth16: the prover and the witness map as the reference's generic functions see them -----------------------------------------------------
/// where the prover's h comes from
pub enum H<'a, F: PrimeField> {
    /// the coefficients `QAP::witness_map` returned — `create_proof_and_committed_witnesses_with_assignment`'s `h: &[E::ScalarField]` (prover.rs:267-276)
    Coefficients(&'a [F]),
    /// the constraint system's matrices (`cs.to_matrices()`: a, b, c), its number of instance variables and of constraints — `create_proof_with_reduction`
    /// (prover.rs:153-180): the circuit becomes resident (found again by the content hash of its matrices), the witness map runs inside the prover call
    Matrices(&'a [Vec<(F, usize)>], &'a [Vec<(F, usize)>], &'a [Vec<(F, usize)>], usize, usize),
}
/// Circuits below this many constraints stay on the CPU prover (a device proof is a few milliseconds whatever the size)
pub const MIN_CONSTRAINTS_GPU: usize = 1 << 12;

/// `create_proof_and_committed_witnesses_with_assignment::<E, QAP>` (legogroth16/src/prover.rs:267-383) as one call: (A, B, C, D), or None — another
/// curve, a small circuit, the library declined — and the caller runs the reference's own body.  The members of `ProvingKeyCommon<E>` / `VerifyingKey<E>`
/// come one by one because this crate cannot name legogroth16's types.  `input_assignment` includes the leading 1.
#[allow(clippy::too_many_arguments)]
pub fn legogroth16_create_proof<E: Pairing>(
    alpha_g1: &E::G1Affine, beta_g1: &E::G1Affine, delta_g1: &E::G1Affine, eta_delta_inv_g1: &E::G1Affine, eta_gamma_inv_g1: &E::G1Affine,
    beta_g2: &E::G2Affine, delta_g2: &E::G2Affine, gamma_abc_g1: &[E::G1Affine], commit_witness_count: usize,
    a_query: &[E::G1Affine], b_g1_query: &[E::G1Affine], b_g2_query: &[E::G2Affine], h_query: &[E::G1Affine], l_query: &[E::G1Affine],
    h: H<E::ScalarField>, input_assignment: &[E::ScalarField], witness_assignment: &[E::ScalarField],
    r: E::ScalarField, s: E::ScalarField, v: E::ScalarField,
) -> Option<(E::G1Affine, E::G2Affine, E::G1Affine, E::G1Affine)> {
    if !same::<E, Bls12_381>() || a_query.len() < MIN_CONSTRAINTS_GPU { return None; }
    // E == Bls12_381: every associated type is the concrete one of `crate`
    let pk = unsafe { crate::host::HostProvingKey {
        alpha_g1: *cast_ref::<E::G1Affine, G1Affine>(alpha_g1), beta_g1: *cast_ref::<E::G1Affine, G1Affine>(beta_g1), delta_g1: *cast_ref::<E::G1Affine, G1Affine>(delta_g1),
        eta_delta_inv_g1: *cast_ref::<E::G1Affine, G1Affine>(eta_delta_inv_g1), eta_gamma_inv_g1: *cast_ref::<E::G1Affine, G1Affine>(eta_gamma_inv_g1),
        beta_g2: *cast_ref::<E::G2Affine, G2Affine>(beta_g2), delta_g2: *cast_ref::<E::G2Affine, G2Affine>(delta_g2),
        gamma_abc_g1: cast_slice::<E::G1Affine, G1Affine>(gamma_abc_g1), commit_witness_count,
        a_query: cast_slice::<E::G1Affine, G1Affine>(a_query), b_g1_query: cast_slice::<E::G1Affine, G1Affine>(b_g1_query), b_g2_query: cast_slice::<E::G2Affine, G2Affine>(b_g2_query),
        h_query: cast_slice::<E::G1Affine, G1Affine>(h_query), l_query: cast_slice::<E::G1Affine, G1Affine>(l_query),
    } };
    let (inst, wit) = unsafe { (cast_slice::<E::ScalarField, Fr>(input_assignment), cast_slice::<E::ScalarField, Fr>(witness_assignment)) };
    let (r, s, v): (Fr, Fr, Fr) = unsafe { (cast_val(r), cast_val(s), cast_val(v)) };
    let out = match h {
        H::Coefficients(c) => crate::host::create_proof_host(&pk, crate::host::HSource::Coefficients(unsafe { cast_slice::<E::ScalarField, Fr>(c) }), inst, wit, r, s, v),
        H::Matrices(a, b, c, num_inputs, num_constraints) => {
            let (a, b, c) = unsafe { (cast_slice::<Vec<(E::ScalarField, usize)>, Vec<(Fr, usize)>>(a), cast_slice::<Vec<(E::ScalarField, usize)>, Vec<(Fr, usize)>>(b), cast_slice::<Vec<(E::ScalarField, usize)>, Vec<(Fr, usize)>>(c)) };
            let circuit = crate::host::resident_circuit(a, b, c, inst.len() + wit.len(), num_inputs, num_constraints)?;
            crate::host::create_proof_host(&pk, crate::host::HSource::Circuit(&circuit), inst, wit, r, s, v)
        }
    }?;
    Some(unsafe { (cast_val::<G1Affine, E::G1Affine>(out.0), cast_val::<G2Affine, E::G2Affine>(out.1), cast_val::<G1Affine, E::G1Affine>(out.2), cast_val::<G1Affine, E::G1Affine>(out.3)) })
}
/// `&A` as `&B` — only ever called after `same::<…>()` established that A and B are one type
#[inline(always)]
unsafe fn cast_ref<A, B>(a: &A) -> &B {
    debug_assert_eq!(core::mem::size_of::<A>(), core::mem::size_of::<B>());
    &*(a as *const A as *const B)
}
/// `LibsnarkReduction::witness_map_from_matrices::<F, D>` (legogroth16/src/r1cs_to_qap.rs:150-210): the D coefficients of h as `Vec<F>`, or None
/// (another field, a small circuit, the library declined) — the caller then runs the reference's own body
pub fn witness_map_from_matrices<F: PrimeField>(a: &[Vec<(F, usize)>], b: &[Vec<(F, usize)>], c: &[Vec<(F, usize)>], num_inputs: usize, num_constraints: usize, full_assignment: &[F]) -> Option<Vec<F>> {
    if !same::<F, Fr>() { return None; }
    let h: Vec<Fr> = unsafe { crate::host::witness_map_from_matrices(cast_slice::<Vec<(F, usize)>, Vec<(Fr, usize)>>(a), cast_slice::<Vec<(F, usize)>, Vec<(Fr, usize)>>(b),
                                                                    cast_slice::<Vec<(F, usize)>, Vec<(Fr, usize)>>(c), num_inputs, num_constraints, cast_slice::<F, Fr>(full_assignment)) }?;
    Some(unsafe { cast_val::<Vec<Fr>, Vec<F>>(h) })
}

// ---- fixed-base batch multiplication: utils/src/msm.rs:8-62 ---------------------------------------------------------------------------------------
/// `FixedBase::msm(.., &table, elements)` as `WindowTable::multiply_many` and `multiply_field_elems_with_same_group_elem` use it: `base` times every
/// element, or None (another group, a short batch, the library declined).  G = G1Projective / G2Projective of BLS12-381 are served.
pub fn fixed_base_msm<G: ark_ec::CurveGroup>(base: &G::Affine, elements: &[G::ScalarField]) -> Option<Vec<G>> {
    if elements.len() < 256 { return None; }
    if same::<G, G1Projective>() {
        let v = crate::host::fixed_base_g1(unsafe { cast_ref::<G::Affine, G1Affine>(base) }, unsafe { cast_slice::<G::ScalarField, Fr>(elements) })?;
        let p: Vec<G1Projective> = v.iter().map(|a| a.into_group()).collect();
        return Some(unsafe { cast_val::<Vec<G1Projective>, Vec<G>>(p) });
    }
    if same::<G, G2Projective>() {
        let v = crate::host::fixed_base_g2(unsafe { cast_ref::<G::Affine, G2Affine>(base) }, unsafe { cast_slice::<G::ScalarField, Fr>(elements) })?;
        let p: Vec<G2Projective> = v.iter().map(|a| a.into_group()).collect();
        return Some(unsafe { cast_val::<Vec<G2Projective>, Vec<G>>(p) });
    }
    None
}
