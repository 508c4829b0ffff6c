//! rust/dock_gpu/tests/parity.rs — the pin against REAL arkworks that the build image cannot run (no Rust toolchain there).
//!
//!   DOCK_GPU_LIB_DIR=<repo>/crypto_amd cargo test --release            (one MI355X visible)
//!
//! 1. the library == arkworks on seeded inputs: `msm_bigint` / `msm_unchecked` over G1 and G2 (edge cases included), the RAW Fp12 output of
//!    `multi_miller_loop`, `final_exponentiation` (arkworks' chain raises to 3 (p^12 - 1) / r: only same-chain code is comparable),
//!    every coefficient of `G2Prepared::from`, the verifier's mixed call;
//! 2. `write_golden` stores what ARKWORKS computed under <repo>/tests/golden/ark/*.json (schema: tests/golden/ark/README.md).  Those files
//!    are then checked everywhere without Rust: `tests/test_ark_golden.py` compares the CPU oracle with them (pinning the oracle: parity
//!    "partial" -> "green") and, on a GPU box, the library through its C ABI.
use ark_bls12_381::{Bls12_381, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::pairing::Pairing;
use ark_ec::{AffineRepr, CurveGroup, VariableBaseMSM};
use ark_ff::{BigInt, PrimeField, UniformRand, Zero};
use ark_std::rand::{rngs::StdRng, SeedableRng};
use dock_gpu::*;
use std::fmt::Write as _;

fn setup() { assert!(init(0, 1 << 16), "no MI355X / libdock_gpu.so"); unsafe { dgpu_set_min_gpu_n(1); } }
fn g1s(rng: &mut StdRng, n: usize) -> Vec<G1Affine> { G1Projective::normalize_batch(&(0..n).map(|_| G1Projective::rand(rng)).collect::<Vec<_>>()) }
fn g2s(rng: &mut StdRng, n: usize) -> Vec<G2Affine> { G2Projective::normalize_batch(&(0..n).map(|_| G2Projective::rand(rng)).collect::<Vec<_>>()) }
fn frs(rng: &mut StdRng, n: usize) -> Vec<Fr> { (0..n).map(|_| Fr::rand(rng)).collect() }
fn big(s: &[Fr]) -> Vec<BigInt<4>> { s.iter().map(|x| x.into_bigint()).collect() }

#[test]
fn msm_equals_arkworks() {
    setup();
    let mut rng = StdRng::seed_from_u64(0x5EED0001);
    for &n in &[1usize, 2, 31, 32, 33, 255, 600, 1 << 12, (1 << 13) + 1, 1 << 16] {
        let (b1, b2, s) = (g1s(&mut rng, n), g2s(&mut rng, n.min(1 << 12)), frs(&mut rng, n));
        let sb = big(&s);
        assert_eq!(msm_bigint_g1(&b1, &sb).into_affine(), G1Projective::msm_bigint(&b1, &sb).into_affine(), "G1 n = {n}");
        assert_eq!(msm_unchecked_g1(&b1, &s).into_affine(), G1Projective::msm_unchecked(&b1, &s).into_affine());
        let m = b2.len();
        assert_eq!(msm_bigint_g2(&b2, &sb[..m]).into_affine(), G2Projective::msm_bigint(&b2, &sb[..m]).into_affine(), "G2 n = {m}");
        // the truncation the prover relies on (prover.rs:286): one scalar more than bases
        if n > 1 { assert_eq!(msm_bigint_g1(&b1[..n - 1], &sb).into_affine(), G1Projective::msm_bigint(&b1[..n - 1], &sb[..n - 1]).into_affine()); }
    }
    // edge cases: identity bases, zero / one / r - 1 scalars, P and -P, all-equal scalars
    let n = 300;
    let mut b = g1s(&mut rng, n);
    let mut s = frs(&mut rng, n);
    b[3] = G1Affine::identity(); b[7] = (-b[6].into_group()).into_affine();
    s[0] = Fr::zero(); s[1] = Fr::from(1u64); s[2] = -Fr::from(1u64); s[7] = s[6];
    assert_eq!(msm_unchecked_g1(&b, &s).into_affine(), G1Projective::msm_unchecked(&b, &s).into_affine());
    let same = vec![s[9]; n];
    assert_eq!(msm_unchecked_g1(&b, &same).into_affine(), G1Projective::msm_unchecked(&b, &same).into_affine());
    let r1 = ResidentG1::upload(&b, None);
    assert_eq!(r1.msm_bigint(1, &big(&s[1..])).into_affine(), G1Projective::msm_bigint(&b[1..], &big(&s[1..])).into_affine());
}

#[test]
fn pairing_equals_arkworks() {
    setup();
    let mut rng = StdRng::seed_from_u64(0x5EED0002);
    for &n in &[1usize, 2, 3, 4, 5, 64, 130, 1024] {
        let (mut p, mut q) = (g1s(&mut rng, n), g2s(&mut rng, n));
        if n >= 5 { p[1] = G1Affine::identity(); q[4] = G2Affine::identity(); }
        let f = multi_miller_loop(&p, &q);
        let g = Bls12_381::multi_miller_loop(p.iter().copied(), q.iter().copied());
        assert_eq!(f.0, g.0, "raw Miller output, n = {n}");
        assert_eq!(final_exponentiation(f).unwrap().0, Bls12_381::final_exponentiation(g).unwrap().0);
        assert_eq!(multi_pairing(&p, &q), Bls12_381::multi_pairing(p.iter().copied(), q.iter().copied()));
    }
    let q = g2s(&mut rng, 7);
    let (mine, theirs): (Vec<G2Prepared>, Vec<G2Prepared>) = (g2_prepare(&q), q.iter().map(|x| G2Prepared::from(*x)).collect());
    for (a, b) in mine.iter().zip(theirs.iter()) { assert_eq!(a.infinity, b.infinity); assert_eq!(a.ell_coeffs, b.ell_coeffs); }
    // the verifier's call shape (verifier.rs:69-76): one affine pair, two prepared
    let p = g1s(&mut rng, 3);
    let f = multi_miller_loop_mixed(&p[..1], &q[..1], &p[1..], &theirs[1..3]);
    let g = Bls12_381::multi_miller_loop(p.iter().copied(), theirs[..3].iter().cloned());
    assert_eq!(f.0, g.0);
}

// ---- golden files: ARKWORKS' results, consumed by tests/test_ark_golden.py ---------------------------------------------------------------
fn hexw(w: &[u64]) -> String { let mut s = String::with_capacity(w.len() * 16); for x in w { write!(s, "{:016x}", x).unwrap(); } s }
fn norm_g1(p: G1Projective) -> Vec<u64> { // the ABI's normalised Jacobian: (x, y, 1) or (1, 1, 0)
    let a = p.into_affine();
    let one = ark_bls12_381::Fq::from(1u64);
    let mut w = Vec::new();
    if a.is_zero() { w.extend_from_slice(&one.0 .0); w.extend_from_slice(&one.0 .0); w.extend_from_slice(&[0u64; 6]); }
    else { w.extend_from_slice(&a.x.0 .0); w.extend_from_slice(&a.y.0 .0); w.extend_from_slice(&one.0 .0); }
    w
}
fn norm_g2(p: G2Projective) -> Vec<u64> {
    let a = p.into_affine();
    let one = ark_bls12_381::Fq::from(1u64);
    let mut w = Vec::new();
    let one2 = |w: &mut Vec<u64>| { w.extend_from_slice(&one.0 .0); w.extend_from_slice(&[0u64; 6]); };
    if a.is_zero() { one2(&mut w); one2(&mut w); w.extend_from_slice(&[0u64; 12]); }
    else { for c in [&a.x.c0, &a.x.c1, &a.y.c0, &a.y.c1] { w.extend_from_slice(&c.0 .0); } one2(&mut w); }
    w
}
fn scal_words(s: &[BigInt<4>]) -> Vec<u64> { s.iter().flat_map(|b| b.0).collect() }

#[test]
#[ignore = "writes tests/golden/ark/*.json: run with `cargo test --release -- --ignored write_golden`"]
fn write_golden() {
    let mut rng = StdRng::seed_from_u64(0x5EED00A2);
    let dir = concat!(env!("CARGO_MANIFEST_DIR"), "/../../tests/golden/ark");
    std::fs::create_dir_all(dir).unwrap();
    let head = "\"schema\": \"dock_gpu/ark-golden/1\", \"producer\": \"ark-ec ^0.4.1 / ark-ff ^0.4.1 / ark-bls12-381 ^0.4.0 (rust/dock_gpu/tests/parity.rs write_golden)\"";
    // MSM
    let mut cases = Vec::new();
    for &n in &[1usize, 2, 33, 300, 1025] {
        let (b1, s) = (g1s(&mut rng, n), big(&frs(&mut rng, n)));
        let (xy, inf) = pack_g1(&b1);
        cases.push(format!("{{\"kind\": \"msm_g1\", \"n\": {n}, \"bases\": \"{}\", \"inf\": \"{}\", \"scalars\": \"{}\", \"out\": \"{}\"}}",
                           hexw(&xy), inf.iter().map(|b| b.to_string()).collect::<String>(), hexw(&scal_words(&s)), hexw(&norm_g1(G1Projective::msm_bigint(&b1, &s)))));
        let m = n.min(300);
        let b2 = g2s(&mut rng, m);
        let (xy2, inf2) = pack_g2(&b2);
        cases.push(format!("{{\"kind\": \"msm_g2\", \"n\": {m}, \"bases\": \"{}\", \"inf\": \"{}\", \"scalars\": \"{}\", \"out\": \"{}\"}}",
                           hexw(&xy2), inf2.iter().map(|b| b.to_string()).collect::<String>(), hexw(&scal_words(&s[..m])), hexw(&norm_g2(G2Projective::msm_bigint(&b2, &s[..m])))));
    }
    std::fs::write(format!("{dir}/msm.json"), format!("{{{head}, \"cases\": [\n{}\n]}}\n", cases.join(",\n"))).unwrap();
    // pairings
    let mut cases = Vec::new();
    for &n in &[1usize, 2, 3, 4, 5, 9, 64] {
        let (p, q) = (g1s(&mut rng, n), g2s(&mut rng, n));
        let f = Bls12_381::multi_miller_loop(p.iter().copied(), q.iter().copied());
        let e = Bls12_381::final_exponentiation(f).unwrap();
        cases.push(format!("{{\"kind\": \"miller_loop\", \"n\": {n}, \"p\": \"{}\", \"q\": \"{}\", \"out\": \"{}\", \"final_exponentiation\": \"{}\"}}",
                           hexw(&pack_g1(&p).0), hexw(&pack_g2(&q).0), hexw(&fq12_to_words(&f.0)), hexw(&fq12_to_words(&e.0))));
    }
    let q = g2s(&mut rng, 2);
    for x in q.iter() {
        let pre = G2Prepared::from(*x);
        let mut w = Vec::new();
        for (c0, c1, c2) in pre.ell_coeffs.iter() { for c in [c0, c1, c2] { w.extend_from_slice(&c.c0 .0 .0); w.extend_from_slice(&c.c1 .0 .0); } }
        cases.push(format!("{{\"kind\": \"g2_prepared\", \"q\": \"{}\", \"coeffs\": \"{}\"}}", hexw(&pack_g2(&[*x]).0), hexw(&w)));
    }
    std::fs::write(format!("{dir}/pairing.json"), format!("{{{head}, \"cases\": [\n{}\n]}}\n", cases.join(",\n"))).unwrap();
}

// ---- the generic drop-ins (src/generic.rs): what rust/patches routes the reference's generic call sites to ----------------------------------
/// a function that is generic the way the reference's call sites are (utils/src/pairs.rs:143-147): it cannot name G1Affine
fn pairs_msm<G: AffineRepr>(left: &[G], right: &[G::ScalarField]) -> G::Group { dock_gpu::generic::msm_unchecked(left, right) }
fn coeff<G: AffineRepr>(query: &[G], assignment: &[<G::ScalarField as PrimeField>::BigInt]) -> G::Group { dock_gpu::generic::msm_bigint::<G>(&query[1..], assignment) }
fn checker_verify<E: Pairing>(a: Vec<E::G1Prepared>, b: Vec<E::G2Prepared>) -> Option<ark_ec::pairing::PairingOutput<E>> {
    dock_gpu::generic::final_exponentiation::<E>(dock_gpu::generic::multi_miller_loop::<E>(a, b))
}

#[test]
fn generic_wrappers_dispatch_and_fall_through() {
    setup();
    let mut rng = StdRng::seed_from_u64(0x5EED0007);
    for &n in &[3usize, 64, 600, 1 << 13] {
        let (b1, b2, s) = (g1s(&mut rng, n), g2s(&mut rng, n.min(1 << 11)), frs(&mut rng, n));
        let sb = big(&s);
        // BLS12-381: served by the library, equal to arkworks
        assert_eq!(pairs_msm::<G1Affine>(&b1, &s).into_affine(), G1Projective::msm_unchecked(&b1, &s).into_affine());
        assert_eq!(pairs_msm::<G2Affine>(&b2, &s[..b2.len()]).into_affine(), G2Projective::msm_unchecked(&b2, &s[..b2.len()]).into_affine());
        assert_eq!(coeff::<G1Affine>(&b1, &sb).into_affine(), G1Projective::msm_bigint(&b1[1..], &sb).into_affine());
        assert_eq!(dock_gpu::generic::msm::<G1Affine>(&b1, &s[..n - 1]), Err(n - 1));
        let m = n.min(1 << 10);
        let pa: Vec<<Bls12_381 as Pairing>::G1Prepared> = b1[..m].iter().map(|p| (*p).into()).collect();
        let pb: Vec<<Bls12_381 as Pairing>::G2Prepared> = b2[..m.min(b2.len())].iter().map(|q| (*q).into()).collect();
        let pa = pa[..pb.len()].to_vec();
        assert_eq!(checker_verify::<Bls12_381>(pa.clone(), pb.clone()), Bls12_381::final_exponentiation(Bls12_381::multi_miller_loop(pa, pb)));
        assert_eq!(dock_gpu::generic::multi_pairing::<Bls12_381>(&b1[..pb_len(&b2, m)], &b2[..pb_len(&b2, m)]), Bls12_381::multi_pairing(&b1[..pb_len(&b2, m)], &b2[..pb_len(&b2, m)]));
        let g = dock_gpu::generic::scale_batch_g1::<Bls12_381>(&b1, sb[0], true);
        for (p, q) in b1.iter().zip(g.iter()) { assert_eq!((-p.mul_bigint(sb[0])).into_affine(), *q); }
    }
}
fn pb_len(b2: &[G2Affine], m: usize) -> usize { m.min(b2.len()) }

/// another curve goes straight through to arkworks: the same generic code path with a type the library does not serve
#[cfg(feature = "other-curve-test")]
#[test]
fn generic_wrappers_leave_other_curves_alone() {
    use ark_bn254::{Fr as BnFr, G1Affine as BnG1, G1Projective as BnG1P};
    let mut rng = StdRng::seed_from_u64(0x5EED0008);
    let b: Vec<BnG1> = BnG1P::normalize_batch(&(0..600).map(|_| BnG1P::rand(&mut rng)).collect::<Vec<_>>());
    let s: Vec<BnFr> = (0..600).map(|_| BnFr::rand(&mut rng)).collect();
    assert_eq!(pairs_msm::<BnG1>(&b, &s), BnG1P::msm_unchecked(&b, &s));
}
