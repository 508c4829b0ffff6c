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

// ---- round 6: the resident-bases cache, the host-side wrappers (src/host.rs), the whole-call entry points ---------------------------------------------
use dock_gpu::host::{self, cache, HSource, HostProvingKey, R1cs, ShardedG1, ShardedG2, WindowTableG1, WindowTableG2};

/// the same `&[G1Affine]` twice, then again: the second call makes it resident, later calls hit; a refilled buffer is noticed, an announced edit too
#[test]
fn resident_bases_cache_same_slice_twice_and_mutated_slice() {
    setup();
    cache::clear(); assert!(cache::set_min_n(1 << 12));
    let mut rng = StdRng::seed_from_u64(0x5EED0009);
    let n = 40_000usize;
    let mut b = g1s(&mut rng, n);
    let s0 = cache::stats();
    for k in 0..4 {
        let s = big(&frs(&mut rng, n));
        assert_eq!(msm_bigint_g1(&b, &s).into_affine(), G1Projective::msm_bigint(&b, &s).into_affine(), "call {k}");
        assert_eq!(msm_bigint_g1(&b[1..], &s).into_affine(), G1Projective::msm_bigint(&b[1..], &s[..n - 1]).into_affine(), "&query[1..], call {k}");
    }
    let s1 = cache::stats();
    assert!(s1.fills > s0.fills && s1.hits >= s0.hits + 5, "{:?} -> {:?}", s0, s1);
    // another key in the same buffer: the answer must be the new key's, at once
    let other = g1s(&mut rng, n);
    b.copy_from_slice(&other);
    let s = big(&frs(&mut rng, n));
    assert_eq!(msm_bigint_g1(&b, &s).into_affine(), G1Projective::msm_bigint(&other, &s).into_affine());
    assert!(cache::stats().stale > s1.stale);
    // an in-place edit of one record, announced (or caught by the exact mode)
    for _ in 0..3 { let _ = msm_bigint_g1(&b, &s); }
    b[n / 2] = b[7];
    assert!(cache::invalidate(&b));
    assert_eq!(msm_bigint_g1(&b, &s).into_affine(), G1Projective::msm_bigint(&b, &s).into_affine());
    assert!(cache::verify_every_record());
    for _ in 0..3 { let _ = msm_bigint_g1(&b, &s); }
    b[n / 3] = b[9];
    assert_eq!(msm_bigint_g1(&b, &s).into_affine(), G1Projective::msm_bigint(&b, &s).into_affine());
    assert!(cache::verify_every_record() && cache::set_min_n(1 << 16));       // (the library's defaults)
    assert!(!host::error_string(DGPU_E_BADARG).is_empty() && host::device_count() >= 1 && host::context_count() >= 1 && host::set_device(0));
    // G2 and the Montgomery forms through the same entry points
    let (b2, sf) = (g2s(&mut rng, 5000), frs(&mut rng, 5000));
    assert_eq!(msm_unchecked_g2(&b2, &sf).into_affine(), G2Projective::msm_unchecked(&b2, &sf).into_affine());
    let r2 = ResidentG2::upload(&b2, None);
    assert_eq!(r2.msm_bigint(1, &big(&sf[1..])).into_affine(), G2Projective::msm_bigint(&b2[1..], &big(&sf[1..])).into_affine());
}

#[test]
fn sharded_msm_window_tables_and_serde() {
    // two contexts on the one GPU of a test box: every code path of the multi-GPU form except a second physical device
    assert!(host::init_devices(&[0, 0], 1 << 16), "no MI355X / libdock_gpu.so");
    unsafe { dgpu_set_min_gpu_n(1); }
    let mut rng = StdRng::seed_from_u64(0x5EED000A);
    let n = 20_000usize;
    let (b, b2, s) = (g1s(&mut rng, n), g2s(&mut rng, 3000), big(&frs(&mut rng, n)));
    let want = G1Projective::msm_bigint(&b, &s).into_affine();
    assert_eq!(msm_bigint_g1_sharded(&b, &s, 0).into_affine(), want);
    // the unmodified one-shot call sharding itself over the two contexts (each caches its chunk at the second sighting)
    assert!(host::set_auto_shard_min_n(1 << 12) && cache::set_min_n(1 << 12));
    for _ in 0..3 { assert_eq!(msm_bigint_g1(&b, &s).into_affine(), want); }
    assert!(host::set_auto_shard_min_n(0) && cache::set_min_n(1 << 16));
    let sh = ShardedG1::upload(&b, 0, true).expect("sharded upload");
    assert_eq!(sh.shards(), 2);
    assert_eq!(sh.msm_bigint(&s).into_affine(), want);
    let rs = sh.upload_scalars(&s).expect("sharded scalars");
    assert_eq!(sh.msm_resident(&rs).expect("resident").into_affine(), want);
    let sh2 = ShardedG2::upload(&b2, 0, false).expect("sharded upload G2");
    assert_eq!(sh2.msm_bigint(&s).into_affine(), G2Projective::msm_bigint(&b2, &s[..3000]).into_affine());
    // the multi-process form's fold (what follows the all-gather of the ranks' partial points)
    let parts = [G1Projective::msm_bigint(&b[..n / 2], &s[..n / 2]), G1Projective::msm_bigint(&b[n / 2..], &s[n / 2..]), G1Projective::zero()];
    assert_eq!(fold_g1(&parts).into_affine(), want);
    // fixed base: WindowTable / multiply_field_elems_with_same_group_elem (utils/src/msm.rs:8-62)
    let f = frs(&mut rng, 1000);
    let (g1, g2) = (b[0], b2[0]);
    let t1 = WindowTableG1::new(&g1).expect("table");
    let prod = t1.multiply_many(&f).expect("products");
    for (x, p) in f.iter().zip(prod.iter()) { assert_eq!((g1 * x).into_affine(), *p); }
    assert_eq!(fixed_base_g1(&g1, &f).unwrap(), prod);
    let hb = t1.multiply_many_to_bases(&f).expect("products as a bases handle");
    let mut out = [0u64; 18];
    assert_eq!(unsafe { dgpu_msm_g1_handle(hb, 0, s.as_ptr() as *const u64, 1000, 0, out.as_mut_ptr()) }, DGPU_OK);
    unsafe { dgpu_bases_free(hb); }
    let t2 = WindowTableG2::new(&g2).expect("table G2");
    let prod2 = t2.multiply_many(&f[..100]).expect("products G2");
    for (x, p) in f.iter().zip(prod2.iter()) { assert_eq!((g2 * x).into_affine(), *p); }
    assert_eq!(fixed_base_g2(&g2, &f[..100]).unwrap(), prod2);
    assert_eq!(dock_gpu::generic::fixed_base_msm::<G1Projective>(&g1, &f).unwrap().iter().map(|p| p.into_affine()).collect::<Vec<_>>(), prod);
    // canonical (de)serialisation == ark-serialize
    use ark_serialize::CanonicalSerialize;
    for compressed in [true, false] {
        let mut want_bytes = Vec::new();
        for p in b[..50].iter() { if compressed { p.serialize_compressed(&mut want_bytes).unwrap() } else { p.serialize_uncompressed(&mut want_bytes).unwrap() } }
        let got = serialize_g1(&b[..50], compressed).unwrap();
        assert_eq!(got, want_bytes);
        assert_eq!(deserialize_g1(&got, 50, compressed, true).unwrap(), b[..50].to_vec());
        let mut want2 = Vec::new();
        for p in b2[..20].iter() { if compressed { p.serialize_compressed(&mut want2).unwrap() } else { p.serialize_uncompressed(&mut want2).unwrap() } }
        let got2 = serialize_g2(&b2[..20], compressed).unwrap();
        assert_eq!(got2, want2);
        assert_eq!(deserialize_g2(&got2, 20, compressed, true).unwrap(), b2[..20].to_vec());
    }
}

#[test]
fn scalings_and_the_scaled_miller_loop() {
    setup();
    let mut rng = StdRng::seed_from_u64(0x5EED000B);
    let (p, q) = (g1s(&mut rng, 130), g2s(&mut rng, 130));
    let m = frs(&mut rng, 130);
    let mb = big(&m);
    let scaled = g1_scale_batch(&p, &mb[0], false).expect("scalings");
    for (x, y) in p.iter().zip(scaled.iter()) { assert_eq!(x.mul_bigint(mb[0]).into_affine(), *y); }
    let want = Bls12_381::multi_miller_loop(p.iter().zip(m.iter()).map(|(x, k)| (*x * k).into_affine()), q.iter().copied());
    assert_eq!(multi_miller_loop_scaled(&p, &mb, &q).0, want.0);
    let want1 = Bls12_381::multi_miller_loop(p.iter().map(|x| (*x * m[0]).into_affine()), q.iter().copied());
    assert_eq!(multi_miller_loop_scaled(&p, &mb[..1], &q).0, want1.0);
}

// ---- LegoGroth16: witness map, prover, verifier ---------------------------------------------------------------------------------------------------------
/// the reference's witness map (legogroth16/src/r1cs_to_qap.rs:150-210) restated over ark-poly for the comparison
fn cpu_witness_map(a: &[Vec<(Fr, usize)>], b: &[Vec<(Fr, usize)>], c: &[Vec<(Fr, usize)>], num_inputs: usize, num_constraints: usize, z: &[Fr]) -> Vec<Fr> {
    use ark_ff::Field;
    use ark_poly::{EvaluationDomain, GeneralEvaluationDomain};
    let domain = GeneralEvaluationDomain::<Fr>::new(num_constraints + num_inputs).unwrap();
    let d = domain.size();
    let eval = |row: &Vec<(Fr, usize)>| row.iter().map(|(k, i)| *k * z[*i]).sum::<Fr>();
    let (mut va, mut vb, mut vc) = (vec![Fr::zero(); d], vec![Fr::zero(); d], vec![Fr::zero(); d]);
    for i in 0..num_constraints { va[i] = eval(&a[i]); vb[i] = eval(&b[i]); vc[i] = eval(&c[i]); }
    va[num_constraints..num_constraints + num_inputs].copy_from_slice(&z[..num_inputs]);
    let coset = domain.get_coset(Fr::GENERATOR).unwrap();
    for v in [&mut va, &mut vb, &mut vc] { domain.ifft_in_place(v); coset.fft_in_place(v); }
    let zinv = domain.evaluate_vanishing_polynomial(Fr::GENERATOR).inverse().unwrap();
    let mut ab: Vec<Fr> = va.iter().zip(vb.iter()).zip(vc.iter()).map(|((x, y), w)| (*x * y - w) * zinv).collect();
    coset.ifft_in_place(&mut ab);
    ab
}
/// x_i = x_{i-1}^2 + i: m constraints, one public input — (matrices a, b, c, full assignment, num_inputs)
#[allow(clippy::type_complexity)]
fn square_chain(m: usize) -> (Vec<Vec<(Fr, usize)>>, Vec<Vec<(Fr, usize)>>, Vec<Vec<(Fr, usize)>>, Vec<Fr>, usize) {
    let one = Fr::from(1u64);
    let mut z = vec![one, Fr::from(3u64)];                              // (1, x_0 public)
    let (mut a, mut b, mut c) = (Vec::new(), Vec::new(), Vec::new());
    for i in 0..m {
        let prev = z[1 + i];
        z.push(prev * prev + Fr::from(i as u64 + 1));
        a.push(vec![(one, 1 + i)]); b.push(vec![(one, 1 + i)]);
        c.push(vec![(one, 2 + i), (-Fr::from(i as u64 + 1), 0)]);       // x_{i-1}^2 = x_i - (i + 1)
    }
    (a, b, c, z, 2)
}
struct SyntheticKey { alpha_g1: G1Affine, beta_g1: G1Affine, delta_g1: G1Affine, eta_delta_inv_g1: G1Affine, eta_gamma_inv_g1: G1Affine, beta_g2: G2Affine, delta_g2: G2Affine,
                      gamma_abc_g1: Vec<G1Affine>, cw: usize, a_query: Vec<G1Affine>, b_g1_query: Vec<G1Affine>, b_g2_query: Vec<G2Affine>, h_query: Vec<G1Affine>, l_query: Vec<G1Affine> }
/// what create_proof_and_committed_witnesses_with_assignment computes (prover.rs:267-383), with arkworks on the CPU, for ANY key material
fn cpu_proof(k: &SyntheticKey, h: &[Fr], inst: &[Fr], wit: &[Fr], r: Fr, s: Fr, v: Fr) -> (G1Affine, G2Affine, G1Affine, G1Affine) {
    let assignment: Vec<BigInt<4>> = inst[1..].iter().chain(wit.iter()).map(|x| x.into_bigint()).collect();
    let aux: Vec<BigInt<4>> = wit[k.cw..].iter().map(|x| x.into_bigint()).collect();
    let hb: Vec<BigInt<4>> = h.iter().map(|x| x.into_bigint()).collect();
    let coeff1 = |init: G1Projective, q: &[G1Affine], vk: G1Affine| init + q[0] + G1Projective::msm_bigint(&q[1..], &assignment) + vk;
    let g_a = coeff1(k.delta_g1 * r, &k.a_query, k.alpha_g1);
    let g1_b = if r.is_zero() { G1Projective::zero() } else { coeff1(k.delta_g1 * s, &k.b_g1_query, k.beta_g1) };
    let g2_b = k.delta_g2 * s + k.b_g2_query[0] + G2Projective::msm_bigint(&k.b_g2_query[1..], &assignment) + k.beta_g2;
    let g_c = g_a * s + g1_b * r - k.delta_g1 * (r * s) + G1Projective::msm_bigint(&k.l_query, &aux) + G1Projective::msm_bigint(&k.h_query, &hb) - k.eta_delta_inv_g1 * v;
    let committed: Vec<BigInt<4>> = wit[..k.cw].iter().map(|x| x.into_bigint()).collect();
    let g_d = G1Projective::msm_bigint(&k.gamma_abc_g1[inst.len()..inst.len() + k.cw], &committed) + k.eta_gamma_inv_g1 * v;
    (g_a.into_affine(), g2_b.into_affine(), g_c.into_affine(), g_d.into_affine())
}

#[test]
fn witness_map_and_the_prover_for_a_host_held_key() {
    setup();
    cache::clear(); assert!(cache::set_min_n(1 << 10));
    let mut rng = StdRng::seed_from_u64(0x5EED000C);
    let m = (1usize << 13) - 2;
    let (a, b, c, z, num_inputs) = square_chain(m);
    let want_h = cpu_witness_map(&a, &b, &c, num_inputs, m, &z);
    // the drop-in of witness_map_from_matrices (circuit resident by content hash), twice: the second call finds the circuit
    for _ in 0..2 { assert_eq!(host::witness_map_from_matrices(&a, &b, &c, num_inputs, m, &z).expect("witness map"), want_h); }
    assert_eq!(dock_gpu::generic::witness_map_from_matrices::<Fr>(&a, &b, &c, num_inputs, m, &z).unwrap(), want_h);
    let circuit = host::resident_circuit(&a, &b, &c, z.len(), num_inputs, m).expect("resident circuit");
    assert_eq!(circuit.witness_map(&z).unwrap(), want_h);
    assert_eq!(R1cs::upload(&a, &b, &c, z.len(), num_inputs, m).unwrap().witness_map(&z).unwrap(), want_h);
    // a synthetic key of the circuit's shape (the prover's equations hold for any key material)
    let nv = z.len();
    let cw = 2usize;
    let g = g1s(&mut rng, 6);
    let g2 = g2s(&mut rng, 2);
    let key = SyntheticKey { alpha_g1: g[0], beta_g1: g[1], delta_g1: g[2], eta_delta_inv_g1: g[3], eta_gamma_inv_g1: g[4], beta_g2: g2[0], delta_g2: g2[1],
                             gamma_abc_g1: g1s(&mut rng, num_inputs + cw), cw, a_query: g1s(&mut rng, nv), b_g1_query: g1s(&mut rng, nv), b_g2_query: g2s(&mut rng, nv),
                             h_query: g1s(&mut rng, want_h.len() - 1), l_query: g1s(&mut rng, nv - num_inputs - cw) };
    let (inst, wit) = (&z[..num_inputs], &z[num_inputs..]);
    let (r, s, v) = (Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng));
    let want = cpu_proof(&key, &want_h, inst, wit, r, s, v);
    let hpk = HostProvingKey { alpha_g1: key.alpha_g1, beta_g1: key.beta_g1, delta_g1: key.delta_g1, eta_delta_inv_g1: key.eta_delta_inv_g1, eta_gamma_inv_g1: key.eta_gamma_inv_g1,
                               beta_g2: key.beta_g2, delta_g2: key.delta_g2, gamma_abc_g1: &key.gamma_abc_g1, commit_witness_count: cw,
                               a_query: &key.a_query, b_g1_query: &key.b_g1_query, b_g2_query: &key.b_g2_query, h_query: &key.h_query, l_query: &key.l_query };
    // first proof: views uploaded for the call; second: the cache makes them resident; third: warm — with h from the host and with the circuit resident
    for k in 0..3 {
        assert_eq!(create_proof_host(&hpk, HSource::Coefficients(&want_h), inst, wit, r, s, v).expect("prove_host"), want, "proof {k}, h from the host");
        assert_eq!(create_proof_host(&hpk, HSource::Circuit(&circuit), inst, wit, r, s, v).expect("prove_host"), want, "proof {k}, circuit resident");
    }
    assert!(cache::stats().fills >= 5);
    // r = 0 (no B in G1, prover.rs:330)
    assert_eq!(create_proof_host(&hpk, HSource::Coefficients(&want_h), inst, wit, Fr::zero(), s, v).unwrap(), cpu_proof(&key, &want_h, inst, wit, Fr::zero(), s, v));
    // the generic form the patched prover calls
    let gen = dock_gpu::generic::legogroth16_create_proof::<Bls12_381>(&key.alpha_g1, &key.beta_g1, &key.delta_g1, &key.eta_delta_inv_g1, &key.eta_gamma_inv_g1, &key.beta_g2, &key.delta_g2,
        &key.gamma_abc_g1, cw, &key.a_query, &key.b_g1_query, &key.b_g2_query, &key.h_query, &key.l_query, dock_gpu::generic::H::Matrices(&a, &b, &c, num_inputs, m), inst, wit, r, s, v);
    assert_eq!(gen.expect("generic prover"), want);
    // the opt-in form with explicit handles (GpuProvingKey) gives the same proof
    let gpk = GpuProvingKey::upload(key.alpha_g1, key.beta_g1, key.delta_g1, key.eta_delta_inv_g1, key.eta_gamma_inv_g1, key.beta_g2, key.delta_g2, &key.gamma_abc_g1, cw,
                                    &key.a_query, &key.b_g1_query, &key.b_g2_query, &key.h_query, &key.l_query).expect("key upload");
    assert_eq!(create_proof_gpu(&gpk, circuit.handle(), &z, num_inputs, r, s, v).expect("prove"), want);
    assert!(cache::set_min_n(1 << 16));
}

/// a tiny hash-chain transcript for the aggregation round trip (both sides use the same one; the reference's is merlin)
#[derive(Clone)]
struct TestTranscript(u64);
impl TranscriptBytes for TestTranscript {
    fn append_message_bytes(&mut self, label: &[u8], bytes: &[u8]) {
        for b in label.iter().chain(bytes.iter()) { self.0 = (self.0 ^ *b as u64).wrapping_mul(0x100000001b3).rotate_left(23); }
    }
    fn challenge_fr(&mut self, label: &[u8]) -> Fr {
        self.append_message_bytes(label, b"challenge");
        let mut w = [0u8; 32];
        for (i, c) in w.chunks_mut(8).enumerate() { c.copy_from_slice(&self.0.wrapping_add(i as u64).wrapping_mul(0x9E3779B97F4A7C15).to_le_bytes()); }
        Fr::from_le_bytes_mod_order(&w)
    }
}

#[test]
fn verifier_batch_verifier_and_aggregation() {
    setup();
    let mut rng = StdRng::seed_from_u64(0x5EED000D);
    use ark_ec::Group;
    // a Groth16 key with known discrete logs and n valid proofs of it: e(A, B) = e(alpha, beta) e(C, delta) e(S, gamma), S = gamma_abc[0] + x gamma_abc[1]
    let (g, h) = (G1Projective::generator(), G2Projective::generator());
    let (al, be, ga, de, k0, k1) = (Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng));
    let (alpha_g1, beta_g2, gamma_g2, delta_g2) = ((g * al).into_affine(), (h * be).into_affine(), (h * ga).into_affine(), (h * de).into_affine());
    let gamma_abc = vec![(g * k0).into_affine(), (g * k1).into_affine()];
    let n = 8usize;
    let mut proofs = Vec::new();
    let mut inputs = Vec::new();
    for _ in 0..n {
        let (a, b, x) = (Fr::rand(&mut rng), Fr::rand(&mut rng), Fr::rand(&mut rng));
        use ark_ff::Field;
        let cc = (a * b - al * be - (k0 + x * k1) * ga) * de.inverse().unwrap();
        proofs.push(((g * a).into_affine(), (h * b).into_affine(), (g * cc).into_affine(), G1Affine::identity()));
        inputs.push(vec![x]);
    }
    let alpha_beta = Bls12_381::pairing(alpha_g1, beta_g2).0;
    let neg = |q: G2Affine| G2Prepared::from((-q.into_group()).into_affine());
    let pvk = GpuPreparedVerifyingKey::new(&alpha_beta, &neg(delta_g2), &neg(gamma_g2), &gamma_abc);
    for (p, x) in proofs.iter().zip(inputs.iter()) { assert_eq!(verify_proof_gpu(&pvk, &p.0, &p.1, &p.2, &p.3, x), Some(true)); }
    assert_eq!(verify_proof_gpu(&pvk, &proofs[0].0, &proofs[0].1, &proofs[1].2, &proofs[0].3, &inputs[0]), Some(false));
    assert_eq!(verify_proofs_batch_gpu(&pvk, &proofs, &inputs, Fr::rand(&mut rng)), Some(true));
    let mut bad = proofs.clone(); bad[3].2 = bad[2].2;
    assert_eq!(verify_proofs_batch_gpu(&pvk, &bad, &inputs, Fr::rand(&mut rng)), Some(false));
    // SnarkPack: aggregate the n proofs under a fake SRS (known alpha, beta), verify the aggregate
    let (sa, sb) = (Fr::rand(&mut rng), Fr::rand(&mut rng));
    let pow = |base: Fr, k: usize| { let mut v = Vec::with_capacity(k); let mut c = Fr::from(1u64); for _ in 0..k { v.push(c); c *= base; } v };
    let g_alpha: Vec<G1Affine> = pow(sa, 2 * n).iter().map(|e| (g * e).into_affine()).collect();
    let g_beta: Vec<G1Affine> = pow(sb, 2 * n).iter().map(|e| (g * e).into_affine()).collect();
    let h_alpha: Vec<G2Affine> = pow(sa, n).iter().map(|e| (h * e).into_affine()).collect();
    let h_beta: Vec<G2Affine> = pow(sb, n).iter().map(|e| (h * e).into_affine()).collect();
    let srs = GpuProverSrs::new(n, &g_alpha, &g_beta, &h_alpha, &h_beta, &h_alpha, &h_beta, &g_alpha[n..], &g_beta[n..]).expect("srs");
    let (pa, pb, pc): (Vec<G1Affine>, Vec<G2Affine>, Vec<G1Affine>) = (proofs.iter().map(|p| p.0).collect(), proofs.iter().map(|p| p.1).collect(), proofs.iter().map(|p| p.2).collect());
    let mut tp = TestTranscript(1);
    let words = aggregate_proofs_gpu(&srs, &mut tp, &pa, &pb, &pc, None).expect("aggregate");
    let mut tv = TestTranscript(1);
    let ok = verify_aggregate_proof_gpu(&g.into_affine(), &h.into_affine(), &g_alpha[1], &g_beta[1], &h_alpha[1], &h_beta[1], n, &alpha_g1, &beta_g2, &gamma_g2, &delta_g2, &gamma_abc,
                                        &inputs, &words, 0, None, Fr::rand(&mut rng), &mut tv, true);
    assert_eq!(ok, Some(true));
    assert_eq!(tp.0, tv.0, "prover and verifier leave the transcript in the same state");
    let mut tampered = words.clone(); let last = tampered.len() - 1; tampered[last] ^= 1;
    let mut tv2 = TestTranscript(1);
    assert_ne!(verify_aggregate_proof_gpu(&g.into_affine(), &h.into_affine(), &g_alpha[1], &g_beta[1], &h_alpha[1], &h_beta[1], n, &alpha_g1, &beta_g2, &gamma_g2, &delta_g2, &gamma_abc,
                                          &inputs, &tampered, 0, None, Fr::rand(&mut rng), &mut tv2, true), Some(true));
}
