//! rust/dock_gpu/src/lib.rs — the thin FFI shim `north_star` asks for: host code stays Rust, the hot path runs behind the C ABI of
//! `include/dock_gpu.h` (libdock_gpu.so: hand-written HIP kernels for gfx950).
//!
//! Every function here is a drop-in for one arkworks call the reference makes (file:line of docknetwork/crypto in the comments) and
//! falls back to that arkworks call on ANY non-zero return code, so the reference's contract "an MSM / a Miller loop cannot fail" holds:
//! `DGPU_E_TOO_SMALL` (n below the measured crossover), `DGPU_E_NODEVICE`, `DGPU_E_BADARG` (a scalar with bit 255 set), an allocation
//! failure — the caller never sees them.
//!
//! NOT compiled in the image this repository is built in (no Rust toolchain there).  `cargo test` here, on a machine with cargo and one
//! MI355X, runs `tests/parity.rs`: the one-command pin of the library against real arkworks and the producer of `tests/golden/ark/*.json`
//! (consumed by `tests/test_ark_golden.py` on the GPU box and, for the CPU oracle, everywhere).
//!
//! Layout facts this file relies on (ark-ff / ark-ec 0.4):
//!   * `Fp<MontBackend<_, N>, N>(pub BigInt<N>, PhantomData)`, `BigInt<N>(pub [u64; N])`: `x.0 .0` are the Montgomery limbs the ABI takes;
//!   * `Affine<P> { pub x, pub y, pub infinity: bool }` is a plain (not repr(C)) struct: field offsets are read with `offset_of!`, the
//!     slice goes over as it lies in memory (`dgpu_msm_*_strided`);
//!   * `Projective<P> { pub x, pub y, pub z }` Jacobian, identity <=> z = 0: what the ABI returns (normalised: z = 1 or 0);
//!   * `Fq12 { c0: Fq6 { c0, c1, c2: Fq2 { c0, c1 } }, c1: Fq6 }`: 12 x 6 limbs in exactly the ABI's order;
//!   * `G2Prepared<P> { pub ell_coeffs: Vec<(Fq2, Fq2, Fq2)>, pub infinity: bool }`, 68 triples for BLS12-381.
#![allow(clippy::missing_safety_doc)]

use ark_bls12_381::{Bls12_381, Fq, Fq12, Fq2, Fq6, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::bls12::G2Prepared as ArkG2Prepared;
use ark_ec::pairing::{MillerLoopOutput, Pairing, PairingOutput};
use ark_ec::{AffineRepr, VariableBaseMSM};
use ark_ff::{BigInt, PrimeField};
use ark_std::vec::Vec;

pub mod generic;
pub mod host;

pub type G2Prepared = ArkG2Prepared<ark_bls12_381::Config>;

pub mod ffi;
pub use ffi::*;
pub use host::*;          // (devices, cache, sharded MSMs, window tables, witness map, the host-key prover, serde: src/host.rs)

pub const G2_PREPARED_WORDS: usize = DGPU_G2_PREPARED_WORDS;

/// bind this process to HIP device `device` (one process per GPU) and size every slot for one-shot calls of up to `max_n` terms, so that
/// no rayon worker's first call allocates on the device.  Call it from start-up code, before other threads exist: it first asks the ROCm runtime for
/// eight hardware queues (`dgpu_runtime_hints`: exports GPU_MAX_HW_QUEUES=8 unless the variable is set; too late — and harmless — once HIP is up).
pub fn init(device: i32, max_n: usize) -> bool {
    unsafe {
        let _ = dgpu_runtime_hints(DGPU_HINT_EIGHT_HW_QUEUES);
        dgpu_init(device) == DGPU_OK && dgpu_reserve_g1(max_n) == DGPU_OK && dgpu_reserve_g2(max_n) == DGPU_OK
    }
}

// ---- the caller's `&[G1Affine]` / `&[G2Affine]` as they lie in memory ------------------------------------------------------------------------
const G1_STRIDE: usize = core::mem::size_of::<G1Affine>();
const G1_X: usize = core::mem::offset_of!(G1Affine, x);
const G1_Y: usize = core::mem::offset_of!(G1Affine, y);
const G1_INF: usize = core::mem::offset_of!(G1Affine, infinity);
const G2_STRIDE: usize = core::mem::size_of::<G2Affine>();
const G2_X: usize = core::mem::offset_of!(G2Affine, x);
const G2_Y: usize = core::mem::offset_of!(G2Affine, y);
const G2_INF: usize = core::mem::offset_of!(G2Affine, infinity);

fn fq(w: &[u64]) -> Fq { Fq::new_unchecked(BigInt::new(w[..6].try_into().unwrap())) }
fn fq2(w: &[u64]) -> Fq2 { Fq2::new(fq(&w[0..6]), fq(&w[6..12])) }
fn g1_from_xyz(w: &[u64; 18]) -> G1Projective { G1Projective::new_unchecked(fq(&w[0..6]), fq(&w[6..12]), fq(&w[12..18])) }
fn g2_from_xyz(w: &[u64; 36]) -> G2Projective { G2Projective::new_unchecked(fq2(&w[0..12]), fq2(&w[12..24]), fq2(&w[24..36])) }
pub fn fq12_from_words(w: &[u64; 72]) -> Fq12 {
    let f6 = |o: usize| Fq6::new(fq2(&w[o..o + 12]), fq2(&w[o + 12..o + 24]), fq2(&w[o + 24..o + 36]));
    Fq12::new(f6(0), f6(36))
}
pub fn fq12_to_words(f: &Fq12) -> [u64; 72] {
    let mut w = [0u64; 72];
    let c = [&f.c0.c0, &f.c0.c1, &f.c0.c2, &f.c1.c0, &f.c1.c1, &f.c1.c2];
    for (k, e) in c.iter().enumerate() {
        w[12 * k..12 * k + 6].copy_from_slice(&e.c0 .0 .0);
        w[12 * k + 6..12 * k + 12].copy_from_slice(&e.c1 .0 .0);
    }
    w
}

/// packed form for the pairing entry points (operand counts are small there): x | y words, a flag byte per point
pub fn pack_g1(ps: &[G1Affine]) -> (Vec<u64>, Vec<u8>) {
    let (mut xy, mut inf) = (Vec::with_capacity(ps.len() * 12), Vec::with_capacity(ps.len()));
    for p in ps {
        match p.xy() {
            Some((x, y)) => { xy.extend_from_slice(&x.0 .0); xy.extend_from_slice(&y.0 .0); inf.push(0u8) }
            None => { xy.extend_from_slice(&[0u64; 12]); inf.push(1u8) }
        }
    }
    (xy, inf)
}
pub fn pack_g2(qs: &[G2Affine]) -> (Vec<u64>, Vec<u8>) {
    let (mut xy, mut inf) = (Vec::with_capacity(qs.len() * 24), Vec::with_capacity(qs.len()));
    for q in qs {
        match q.xy() {
            Some((x, y)) => { for c in [&x.c0, &x.c1, &y.c0, &y.c1] { xy.extend_from_slice(&c.0 .0); } inf.push(0u8) }
            None => { xy.extend_from_slice(&[0u64; 24]); inf.push(1u8) }
        }
    }
    (xy, inf)
}

// ---- variable-base MSM ---------------------------------------------------------------------------------------------------------------
/// drop-in for `G1Projective::msm_bigint(bases, scalars)` — legogroth16/src/prover.rs:286,299,363,592; utils/src/pairs.rs:153-155
pub fn msm_bigint_g1(bases: &[G1Affine], scalars: &[BigInt<4>]) -> G1Projective {
    let n = bases.len().min(scalars.len()); // arkworks truncates; prover.rs:286 relies on it (h_query has D - 1 points against D scalars)
    let mut out = [0u64; 18];
    let rc = unsafe { dgpu_msm_g1_strided(bases.as_ptr() as *const _, G1_STRIDE, G1_X, G1_Y, G1_INF, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) };
    if rc != DGPU_OK { return G1Projective::msm_bigint(&bases[..n], &scalars[..n]); }
    g1_from_xyz(&out)
}
/// drop-in for `G::Group::msm_unchecked(bases, &[Fr])` — utils/src/pairs.rs:145-147, owned_pairs.rs:95-97, randomized_mult_checker.rs:100:
/// the `&[Fr]` slice goes over as it is (Montgomery limbs), `Fr::into_bigint` runs on the device
pub fn msm_unchecked_g1(bases: &[G1Affine], scalars: &[Fr]) -> G1Projective {
    let n = bases.len().min(scalars.len());
    let mut out = [0u64; 18];
    let rc = unsafe { dgpu_msm_g1_strided(bases.as_ptr() as *const _, G1_STRIDE, G1_X, G1_Y, G1_INF, scalars.as_ptr() as *const u64, n, 1, out.as_mut_ptr()) };
    if rc != DGPU_OK { return G1Projective::msm_unchecked(&bases[..n], &scalars[..n]); }
    g1_from_xyz(&out)
}
/// `G2Projective::msm_bigint` — legogroth16/src/prover.rs:344 -> :592 (b_g2_query)
pub fn msm_bigint_g2(bases: &[G2Affine], scalars: &[BigInt<4>]) -> G2Projective {
    let n = bases.len().min(scalars.len());
    let mut out = [0u64; 36];
    let rc = unsafe { dgpu_msm_g2_strided(bases.as_ptr() as *const _, G2_STRIDE, G2_X, G2_Y, G2_INF, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) };
    if rc != DGPU_OK { return G2Projective::msm_bigint(&bases[..n], &scalars[..n]); }
    g2_from_xyz(&out)
}
pub fn msm_unchecked_g2(bases: &[G2Affine], scalars: &[Fr]) -> G2Projective {
    let n = bases.len().min(scalars.len());
    let mut out = [0u64; 36];
    let rc = unsafe { dgpu_msm_g2_strided(bases.as_ptr() as *const _, G2_STRIDE, G2_X, G2_Y, G2_INF, scalars.as_ptr() as *const u64, n, 1, out.as_mut_ptr()) };
    if rc != DGPU_OK { return G2Projective::msm_unchecked(&bases[..n], &scalars[..n]); }
    g2_from_xyz(&out)
}

/// a proving-key query resident in HBM (uploaded once per key; `precompute` turns it into the window table the 2^20-term MSMs run on)
pub struct ResidentG1 { handle: u64, host: Vec<G1Affine> }
impl ResidentG1 {
    pub fn upload(bases: &[G1Affine], table_window_bits: Option<i32>) -> Self {
        let mut handle = 0u64;
        let ok = unsafe { dgpu_bases_upload_g1_strided(bases.as_ptr() as *const _, G1_STRIDE, G1_X, G1_Y, G1_INF, bases.len(), &mut handle) } == DGPU_OK;
        if ok { if let Some(c) = table_window_bits { unsafe { dgpu_bases_precompute_g1(handle, c); } } }
        ResidentG1 { handle: if ok { handle } else { 0 }, host: bases.to_vec() }
    }
    pub fn handle(&self) -> u64 { self.handle }
    /// `msm_bigint(&query[offset..], scalars)` — `calculate_coeff` uses offset 1 (prover.rs:592)
    pub fn msm_bigint(&self, offset: usize, scalars: &[BigInt<4>]) -> G1Projective {
        let n = (self.host.len() - offset).min(scalars.len());
        let mut out = [0u64; 18];
        let rc = if self.handle == 0 { -1 } else { unsafe { dgpu_msm_g1_handle(self.handle, offset, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) } };
        if rc != DGPU_OK { return G1Projective::msm_bigint(&self.host[offset..offset + n], &scalars[..n]); }
        g1_from_xyz(&out)
    }
}
impl Drop for ResidentG1 { fn drop(&mut self) { if self.handle != 0 { unsafe { dgpu_bases_free(self.handle); } } } }

/// the same for G2 (b_g2_query)
pub struct ResidentG2 { handle: u64, host: Vec<G2Affine> }
impl ResidentG2 {
    pub fn upload(bases: &[G2Affine], table_window_bits: Option<i32>) -> Self {
        let mut handle = 0u64;
        let ok = unsafe { dgpu_bases_upload_g2_strided(bases.as_ptr() as *const _, G2_STRIDE, G2_X, G2_Y, G2_INF, bases.len(), &mut handle) } == DGPU_OK;
        if ok { if let Some(c) = table_window_bits { unsafe { dgpu_bases_precompute_g2(handle, c); } } }
        ResidentG2 { handle: if ok { handle } else { 0 }, host: bases.to_vec() }
    }
    pub fn handle(&self) -> u64 { self.handle }
    pub fn msm_bigint(&self, offset: usize, scalars: &[BigInt<4>]) -> G2Projective {
        let n = (self.host.len() - offset).min(scalars.len());
        let mut out = [0u64; 36];
        let rc = if self.handle == 0 { -1 } else { unsafe { dgpu_msm_g2_handle(self.handle, offset, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) } };
        if rc != DGPU_OK { return G2Projective::msm_bigint(&self.host[offset..offset + n], &scalars[..n]); }
        g2_from_xyz(&out)
    }
}
impl Drop for ResidentG2 { fn drop(&mut self) { if self.handle != 0 { unsafe { dgpu_bases_free(self.handle); } } } }

// ---- pairings ------------------------------------------------------------------------------------------------------------------------
/// drop-in for `Bls12_381::multi_miller_loop(a, b)` over affine operands — utils/src/randomized_pairing_check.rs:134,169-170,207.
/// Lengths must agree (arkworks' zip_eq panics otherwise); pairs with an identity member are skipped by the library like arkworks does.
pub fn multi_miller_loop(a: &[G1Affine], b: &[G2Affine]) -> MillerLoopOutput<Bls12_381> {
    assert_eq!(a.len(), b.len(), "multi_miller_loop: lengths differ");
    let ((p, pi), (q, qi)) = (pack_g1(a), pack_g2(b));
    let skip: Vec<u8> = pi.iter().zip(qi.iter()).map(|(x, y)| x | y).collect();
    let mut out = [0u64; 72];
    let rc = unsafe { dgpu_multi_miller_loop(p.as_ptr(), q.as_ptr(), skip.as_ptr(), a.len(), out.as_mut_ptr()) };
    if rc != DGPU_OK { return Bls12_381::multi_miller_loop(a.iter().copied(), b.iter().copied()); }
    MillerLoopOutput(fq12_from_words(&out))
}
/// `G2Prepared::from(q)` for a batch — randomized_pairing_check.rs:132,163,188-189; legogroth16/src/verifier.rs:22-23
pub fn g2_prepare(qs: &[G2Affine]) -> Vec<G2Prepared> {
    let (q, qi) = pack_g2(qs);
    let mut co = ark_std::vec![0u64; qs.len() * G2_PREPARED_WORDS];
    let mut inf = ark_std::vec![0u8; qs.len()];
    let rc = unsafe { dgpu_g2_prepare(q.as_ptr(), qi.as_ptr(), qs.len(), co.as_mut_ptr(), inf.as_mut_ptr()) };
    if rc != DGPU_OK { return qs.iter().map(|q| G2Prepared::from(*q)).collect(); }
    (0..qs.len()).map(|i| {
        if inf[i] != 0 { return G2Prepared { ell_coeffs: Vec::new(), infinity: true }; }
        let w = &co[i * G2_PREPARED_WORDS..(i + 1) * G2_PREPARED_WORDS];
        G2Prepared { ell_coeffs: (0..68).map(|s| (fq2(&w[36 * s..]), fq2(&w[36 * s + 12..]), fq2(&w[36 * s + 24..]))).collect(), infinity: false }
    }).collect()
}
fn prepared_words(q: &G2Prepared, out: &mut Vec<u64>) {
    if q.infinity || q.ell_coeffs.len() != 68 { out.extend(core::iter::repeat(0u64).take(G2_PREPARED_WORDS)); return; }
    for (c0, c1, c2) in q.ell_coeffs.iter() { for c in [c0, c1, c2] { out.extend_from_slice(&c.c0 .0 .0); out.extend_from_slice(&c.c1 .0 .0); } }
}
/// the verifier's call — legogroth16/src/verifier.rs:69-76: `[proof.b.into(), pvk.delta_g2_neg_pc, pvk.gamma_g2_neg_pc]`: some G2 operands
/// affine (prepared inside the call, their chain pipelined), the others already `G2Prepared`
pub fn multi_miller_loop_mixed(a_aff: &[G1Affine], b_aff: &[G2Affine], a_prep: &[G1Affine], b_prep: &[G2Prepared]) -> MillerLoopOutput<Bls12_381> {
    assert_eq!(a_aff.len(), b_aff.len()); assert_eq!(a_prep.len(), b_prep.len());
    let ((p, pi), (q, qi)) = (pack_g1(a_aff), pack_g2(b_aff));
    let skip_aff: Vec<u8> = pi.iter().zip(qi.iter()).map(|(x, y)| x | y).collect();
    let (pp, ppi) = pack_g1(a_prep);
    let mut co = Vec::with_capacity(b_prep.len() * G2_PREPARED_WORDS);
    for b in b_prep { prepared_words(b, &mut co); }
    let skip_prep: Vec<u8> = ppi.iter().zip(b_prep.iter()).map(|(x, b)| x | (b.infinity as u8)).collect();
    let mut out = [0u64; 72];
    let rc = unsafe { dgpu_multi_miller_loop_mixed(p.as_ptr(), q.as_ptr(), skip_aff.as_ptr(), a_aff.len(), pp.as_ptr(), co.as_ptr(), skip_prep.as_ptr(), a_prep.len(), out.as_mut_ptr()) };
    if rc != DGPU_OK {
        let g1 = a_aff.iter().chain(a_prep.iter()).copied();
        let g2 = b_aff.iter().map(|q| G2Prepared::from(*q)).chain(b_prep.iter().cloned());
        return Bls12_381::multi_miller_loop(g1, g2);
    }
    MillerLoopOutput(fq12_from_words(&out))
}
/// `multi_miller_loop(a_i.mul_bigint(m_i), b_i)` in ONE call — utils/src/randomized_pairing_check.rs:125-134 (the scalings of
/// `add_multiple_sources_and_target`, then the loop): the scaling chains run beside the chain of the `b_i` on the device.  `m`: one scalar per
/// pair, or a single one for all.  Falls back to arkworks on any error.
pub fn multi_miller_loop_scaled(a: &[G1Affine], m: &[BigInt<4>], b: &[G2Affine]) -> MillerLoopOutput<Bls12_381> {
    assert_eq!(a.len(), b.len()); assert!(m.len() == a.len() || m.len() == 1);
    let ((p, pi), (q, qi)) = (pack_g1(a), pack_g2(b));
    let skip: Vec<u8> = pi.iter().zip(qi.iter()).map(|(x, y)| x | y).collect();
    let sc: Vec<u64> = m.iter().flat_map(|k| k.0).collect();
    let mut out = [0u64; 72];
    let rc = unsafe { dgpu_multi_miller_loop_scaled(p.as_ptr(), sc.as_ptr(), if m.len() == a.len() { 4 } else { 0 }, q.as_ptr(), skip.as_ptr(), a.len(),
                                                    core::ptr::null(), core::ptr::null(), core::ptr::null(), 0, out.as_mut_ptr()) };
    if rc != DGPU_OK {
        use ark_ec::CurveGroup;
        let scaled: Vec<G1Affine> = a.iter().enumerate().map(|(i, p)| p.mul_bigint(m[if m.len() == 1 { 0 } else { i }]).into_affine()).collect();
        return Bls12_381::multi_miller_loop(scaled, b.iter().copied());
    }
    MillerLoopOutput(fq12_from_words(&out))
}
/// drop-in for `Bls12_381::final_exponentiation(f)` — randomized_pairing_check.rs:213, verifier.rs:78 (host code inside the library)
pub fn final_exponentiation(f: MillerLoopOutput<Bls12_381>) -> Option<PairingOutput<Bls12_381>> {
    let w = fq12_to_words(&f.0);
    let mut out = [0u64; 72];
    match unsafe { dgpu_final_exponentiation(w.as_ptr(), out.as_mut_ptr()) } {
        DGPU_OK => Some(PairingOutput(fq12_from_words(&out))),
        -5 => None, // DGPU_E_ZERO: arkworks returns None for f = 0
        _ => Bls12_381::final_exponentiation(f),
    }
}
/// `a.mul_bigint(m)` (or its negative) for every point of a batch, ONE scalar — utils/src/randomized_pairing_check.rs:125-129,152-158.
/// None when the library declined (the caller then scales on the CPU).
pub fn g1_scale_batch(points: &[G1Affine], m: &BigInt<4>, negate: bool) -> Option<Vec<G1Affine>> {
    let (p, pi) = pack_g1(points);
    let neg = ark_std::vec![negate as u8; points.len()];
    let mut out = ark_std::vec![0u64; points.len() * 12];
    let mut oinf = ark_std::vec![0u8; points.len()];
    let rc = unsafe { dgpu_g1_scale_batch(p.as_ptr(), pi.as_ptr(), m.0.as_ptr(), 0, neg.as_ptr(), points.len(), out.as_mut_ptr(), oinf.as_mut_ptr()) };
    if rc != DGPU_OK { return None; }
    Some((0..points.len()).map(|i| g1_affine(out[12 * i..12 * i + 12].try_into().unwrap(), oinf[i])).collect())
}
/// `Bls12_381::multi_pairing(a, b)` — 67 call sites, e.g. bbs_plus/src/signature.rs:284, legogroth16/src/aggregation/commitment.rs:30-31
pub fn multi_pairing(a: &[G1Affine], b: &[G2Affine]) -> PairingOutput<Bls12_381> {
    final_exponentiation(multi_miller_loop(a, b)).expect("Miller output of valid operands is never zero")
}

// ---- the LegoGroth16 prover as one call (legogroth16/src/prover.rs:153-180 -> :267-383) ------------------------------------------------------
/// the proving key's five queries resident in HBM (tables) + its O(1) elements in the ABI's packed form
pub struct GpuProvingKey {
    pub a: ResidentG1, pub b_g1: ResidentG1, pub h: ResidentG1, pub l: ResidentG1,
    b_g2_handle: u64,
    small_g1: Vec<u64>,  // alpha, beta, delta, eta/delta, eta/gamma, a0, b1_0, then gamma_abc_g1
    small_g2: Vec<u64>,  // beta, delta, b2_0
    gamma_abc_len: usize,
    pub commit_witness_count: usize,
}
pub const TABLE_C_WITNESS: i32 = DGPU_TABLE_C_WITNESS; // the queries that meet the witness
impl GpuProvingKey {
    #[allow(clippy::too_many_arguments)]
    pub fn upload(alpha_g1: G1Affine, beta_g1: G1Affine, delta_g1: G1Affine, eta_delta_inv_g1: G1Affine, eta_gamma_inv_g1: G1Affine,
                  beta_g2: G2Affine, delta_g2: G2Affine, gamma_abc_g1: &[G1Affine], commit_witness_count: usize,
                  a_query: &[G1Affine], b_g1_query: &[G1Affine], b_g2_query: &[G2Affine], h_query: &[G1Affine], l_query: &[G1Affine]) -> Option<Self> {
        let mut b2 = 0u64;
        if unsafe { dgpu_bases_upload_g2_strided(b_g2_query.as_ptr() as *const _, G2_STRIDE, G2_X, G2_Y, G2_INF, b_g2_query.len(), &mut b2) } != DGPU_OK { return None; }
        unsafe { dgpu_bases_precompute_g2(b2, TABLE_C_WITNESS); }
        let mut s1: Vec<G1Affine> = ark_std::vec![alpha_g1, beta_g1, delta_g1, eta_delta_inv_g1, eta_gamma_inv_g1, a_query[0], b_g1_query[0]];
        s1.extend_from_slice(gamma_abc_g1);
        Some(GpuProvingKey {
            a: ResidentG1::upload(a_query, Some(TABLE_C_WITNESS)), b_g1: ResidentG1::upload(b_g1_query, Some(TABLE_C_WITNESS)),
            h: ResidentG1::upload(h_query, Some(0)), l: ResidentG1::upload(l_query, Some(TABLE_C_WITNESS)),
            b_g2_handle: b2, small_g1: pack_g1(&s1).0, small_g2: pack_g2(&[beta_g2, delta_g2, b_g2_query[0]]).0,
            gamma_abc_len: gamma_abc_g1.len(), commit_witness_count,
        })
    }
    fn raw(&self) -> DgpuLegoPk {
        let g1 = |k: usize| unsafe { self.small_g1.as_ptr().add(12 * k) };
        let g2 = |k: usize| unsafe { self.small_g2.as_ptr().add(24 * k) };
        DgpuLegoPk { a_query: self.a.handle(), b_g1_query: self.b_g1.handle(), b_g2_query: self.b_g2_handle, h_query: self.h.handle(), l_query: self.l.handle(),
                     alpha_g1: g1(0), beta_g1: g1(1), delta_g1: g1(2), eta_delta_inv_g1: g1(3), eta_gamma_inv_g1: g1(4), beta_g2: g2(0), delta_g2: g2(1),
                     a0: g1(5), b1_0: g1(6), b2_0: g2(2), gamma_abc_g1: g1(7), gamma_abc_len: self.gamma_abc_len, commit_witness_count: self.commit_witness_count }
    }
}
impl Drop for GpuProvingKey { fn drop(&mut self) { if self.b_g2_handle != 0 { unsafe { dgpu_bases_free(self.b_g2_handle); } } } }

fn g1_affine(w: &[u64; 12], inf: u8) -> G1Affine { if inf != 0 { G1Affine::identity() } else { G1Affine::new_unchecked(fq(&w[0..6]), fq(&w[6..12])) } }
fn g2_affine(w: &[u64; 24], inf: u8) -> G2Affine { if inf != 0 { G2Affine::identity() } else { G2Affine::new_unchecked(fq2(&w[0..12]), fq2(&w[12..24])) } }

/// `create_proof_with_reduction` as ONE call: z = (1, instance..., witness...) as `&[Fr]`, `circuit` a `dgpu_r1cs_upload` handle.
/// Returns (A, B, C, D), or None when the library declined (the caller then runs the reference's CPU prover).
pub fn create_proof_gpu(pk: &GpuProvingKey, circuit: u64, z: &[Fr], n_inst: usize, r: Fr, s: Fr, v: Fr) -> Option<(G1Affine, G2Affine, G1Affine, G1Affine)> {
    let (mut a, mut b, mut c, mut d, mut inf) = ([0u64; 12], [0u64; 24], [0u64; 12], [0u64; 12], [0u8; 4]);
    let (rb, sb, vb) = (r.into_bigint(), s.into_bigint(), v.into_bigint());
    let raw = pk.raw();
    let rc = unsafe { dgpu_legogroth16_prove(&raw, circuit, 0, z.as_ptr() as *const u64, z.len(), n_inst, 1, rb.0.as_ptr(), sb.0.as_ptr(), vb.0.as_ptr(),
                                             a.as_mut_ptr(), b.as_mut_ptr(), c.as_mut_ptr(), d.as_mut_ptr(), inf.as_mut_ptr()) };
    if rc != DGPU_OK { return None; }
    Some((g1_affine(&a, inf[0]), g2_affine(&b, inf[1]), g1_affine(&c, inf[2]), g1_affine(&d, inf[3])))
}

// ---- the LegoGroth16 verifier as one call (legogroth16/src/verifier.rs:62-99) ----------------------------------------------------------------
/// the PreparedVerifyingKey's members in the ABI's form (made once per key): e(alpha, beta), -delta and -gamma prepared, gamma_abc_g1
pub struct GpuPreparedVerifyingKey { alpha_beta: [u64; 72], delta_neg: Vec<u64>, gamma_neg: Vec<u64>, gamma_abc: Vec<u64>, gamma_abc_len: usize }
impl GpuPreparedVerifyingKey {
    pub fn new(alpha_g1_beta_g2: &Fq12, delta_g2_neg_pc: &G2Prepared, gamma_g2_neg_pc: &G2Prepared, gamma_abc_g1: &[G1Affine]) -> Self {
        let (mut d, mut g) = (Vec::new(), Vec::new());
        prepared_words(delta_g2_neg_pc, &mut d); prepared_words(gamma_g2_neg_pc, &mut g);
        GpuPreparedVerifyingKey { alpha_beta: fq12_to_words(alpha_g1_beta_g2), delta_neg: d, gamma_neg: g, gamma_abc: pack_g1(gamma_abc_g1).0, gamma_abc_len: gamma_abc_g1.len() }
    }
}
/// `verify_proof(pvk, proof, public_inputs)`: Some(true / false), or None when the library declined (then: the reference's CPU verifier)
pub fn verify_proof_gpu(pvk: &GpuPreparedVerifyingKey, a: &G1Affine, b: &G2Affine, c: &G1Affine, d: &G1Affine, public_inputs: &[Fr]) -> Option<bool> {
    let (pa, ia) = pack_g1(&[*a, *c, *d]);
    let (pb, ib) = pack_g2(&[*b]);
    let inf = [ia[0], ib[0], ia[1], ia[2]];
    let mut ok = -1i32;
    let rc = unsafe { dgpu_legogroth16_verify(pvk.alpha_beta.as_ptr(), pvk.delta_neg.as_ptr(), pvk.gamma_neg.as_ptr(), pvk.gamma_abc.as_ptr(), pvk.gamma_abc_len,
                                              pa.as_ptr(), pb.as_ptr(), pa[12..].as_ptr(), pa[24..].as_ptr(), inf.as_ptr(),
                                              public_inputs.as_ptr() as *const u64, public_inputs.len(), 1, &mut ok) };
    if rc != DGPU_OK { return None; }
    Some(ok == 1)
}

/// Many proofs of ONE verifying key in one call: the classical Groth16 batch check (what the reference reaches through
/// `RandomizedPairingChecker::add_multiple_sources_and_target` per proof + one lazy `verify()`, utils/src/randomized_pairing_check.rs:116-138,204-214,
/// with the pairs that share -delta / -gamma merged before the pairing).  `proofs[i] = (a, b, c, d)`, `public_inputs[i]` the i-th proof's inputs,
/// `random` drawn AFTER the proofs are fixed and non-zero.  Some(all valid) or None when the library declined (then: verify one by one / the checker).
pub fn verify_proofs_batch_gpu(pvk: &GpuPreparedVerifyingKey, proofs: &[(G1Affine, G2Affine, G1Affine, G1Affine)], public_inputs: &[Vec<Fr>], random: Fr) -> Option<bool> {
    let n = proofs.len();
    if public_inputs.len() != n { return None; }
    let k = public_inputs.first().map_or(0, |r| r.len());
    if public_inputs.iter().any(|r| r.len() != k) { return Some(false); }
    let a = pack_g1(&proofs.iter().map(|p| p.0).collect::<Vec<_>>()).0;
    let b = pack_g2(&proofs.iter().map(|p| p.1).collect::<Vec<_>>()).0;
    let c = pack_g1(&proofs.iter().map(|p| p.2).collect::<Vec<_>>()).0;
    let d = pack_g1(&proofs.iter().map(|p| p.3).collect::<Vec<_>>()).0;
    let pubs: Vec<u64> = public_inputs.iter().flat_map(|r| r.iter().flat_map(|x| x.into_bigint().0)).collect();
    let rnd = random.into_bigint();
    let mut ok = -1i32;
    let rc = unsafe { dgpu_legogroth16_verify_batch(pvk.alpha_beta.as_ptr(), pvk.delta_neg.as_ptr(), pvk.gamma_neg.as_ptr(), pvk.gamma_abc.as_ptr(), pvk.gamma_abc_len,
                                                    a.as_ptr(), b.as_ptr(), c.as_ptr(), d.as_ptr(), n, if k == 0 { core::ptr::null() } else { pubs.as_ptr() }, k, 0, rnd.0.as_ptr(), &mut ok) };
    if rc != DGPU_OK { return None; }
    Some(ok == 1)
}

// ---- SnarkPack aggregation as one call each way (legogroth16/src/aggregation/groth16/prover.rs:47-147, verifier.rs:36-100) -----------------------
/// What the library needs of the caller's `impl Transcript` (utils/src/transcript.rs:45-63): `append` hands over the `serialize_compressed` bytes
/// the library produced itself (bit for bit what `Transcript::append(label, &element)` would write), `challenge_scalar` is the trait's.
pub trait TranscriptBytes {
    fn append_message_bytes(&mut self, label: &[u8], bytes: &[u8]);
    fn challenge_fr(&mut self, label: &[u8]) -> Fr;
}
unsafe extern "C" fn tr_append<T: TranscriptBytes>(ctx: *mut core::ffi::c_void, label: *const u8, label_len: usize, bytes: *const u8, len: usize) {
    let t = &mut *(ctx as *mut T);
    t.append_message_bytes(core::slice::from_raw_parts(label, label_len), core::slice::from_raw_parts(bytes, len));
}
unsafe extern "C" fn tr_challenge<T: TranscriptBytes>(ctx: *mut core::ffi::c_void, label: *const u8, label_len: usize, out: *mut u64) {
    let t = &mut *(ctx as *mut T);
    let c = t.challenge_fr(core::slice::from_raw_parts(label, label_len)).into_bigint();
    core::ptr::copy_nonoverlapping(c.0.as_ptr(), out, 4);
}
fn bind_transcript<T: TranscriptBytes>(t: &mut T) -> DgpuTranscript {
    DgpuTranscript { ctx: t as *mut T as *mut core::ffi::c_void, append_message: tr_append::<T>, challenge_scalar: tr_challenge::<T> }
}
/// ProverSRS in the ABI's form (made once per specialised SRS): the four power tables and both commitment keys as packed words
pub struct GpuProverSrs { n: usize, tabs: [Vec<u64>; 8] }
impl GpuProverSrs {
    #[allow(clippy::too_many_arguments)]
    /// None unless the members have the lengths the C side reads: 2 n powers in the two G1 tables, n in the two G2 tables, n elements in each commitment key
    /// (`ProverSRS::specialize`, legogroth16/src/aggregation/srs.rs) — the library copies 2 n / n packed points out of these buffers
    pub fn new(n: usize, g_alpha_powers_table: &[G1Affine], g_beta_powers_table: &[G1Affine], h_alpha_powers_table: &[G2Affine], h_beta_powers_table: &[G2Affine],
               vkey_a: &[G2Affine], vkey_b: &[G2Affine], wkey_a: &[G1Affine], wkey_b: &[G1Affine]) -> Option<Self> {
        if n == 0 || !n.is_power_of_two() { return None; }
        if g_alpha_powers_table.len() != 2 * n || g_beta_powers_table.len() != 2 * n || h_alpha_powers_table.len() != n || h_beta_powers_table.len() != n { return None; }
        if vkey_a.len() != n || vkey_b.len() != n || wkey_a.len() != n || wkey_b.len() != n { return None; }
        Some(GpuProverSrs { n, tabs: [pack_g1(g_alpha_powers_table).0, pack_g1(g_beta_powers_table).0, pack_g2(h_alpha_powers_table).0, pack_g2(h_beta_powers_table).0,
                                 pack_g2(vkey_a).0, pack_g2(vkey_b).0, pack_g1(wkey_a).0, pack_g1(wkey_b).0] })
    }
}
/// `aggregate_proofs(srs, transcript, proofs)`: the aggregate proof as the ABI's flat words (include/dock_gpu.h gives the layout: the fields of
/// AggregateProof in declaration order, so `AggregateProof { com_ab: .., .. }` is rebuilt by walking it with fq12_from_words / the unpackers), or
/// None when the library declined (then: the reference's CPU aggregator).  `d`: the commitments of LegoGroth16 proofs (legogroth16/prover.rs:38-127).
///
/// The library absorbs into and squeezes from the transcript as the protocol goes, and it can fail half-way (an allocation, a HIP error):
/// it therefore works on a CLONE, and the caller's transcript advances only when the call succeeded.  On `None` the caller's transcript is
/// exactly what it was, so the CPU aggregator it falls back to produces a proof every verifier can reproduce.
pub fn aggregate_proofs_gpu<T: TranscriptBytes + Clone>(srs: &GpuProverSrs, transcript: &mut T, a: &[G1Affine], b: &[G2Affine], c: &[G1Affine], d: Option<&[G1Affine]>) -> Option<Vec<u64>> {
    let n = a.len();
    if n != srs.n || b.len() != n || c.len() != n || d.map_or(false, |x| x.len() != n) { return None; }
    let cap = unsafe { dgpu_snarkpack_proof_words(n, d.is_some() as i32) };
    if cap == 0 { return None; }
    let (pa, pb, pc) = (pack_g1(a).0, pack_g2(b).0, pack_g1(c).0);
    let pd = d.map(|x| pack_g1(x).0);
    let view = DgpuSnarkpackProverSrs { n: srs.n, g_alpha_powers_table: srs.tabs[0].as_ptr(), g_beta_powers_table: srs.tabs[1].as_ptr(), h_alpha_powers_table: srs.tabs[2].as_ptr(),
                                        h_beta_powers_table: srs.tabs[3].as_ptr(), vkey_a: srs.tabs[4].as_ptr(), vkey_b: srs.tabs[5].as_ptr(), wkey_a: srs.tabs[6].as_ptr(), wkey_b: srs.tabs[7].as_ptr() };
    let mut work = transcript.clone();
    let tr = bind_transcript(&mut work);
    let mut out = vec![0u64; cap];
    let mut len = 0usize;
    let rc = unsafe { dgpu_snarkpack_aggregate(&view, pa.as_ptr(), pb.as_ptr(), pc.as_ptr(), pd.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()), n, &tr, out.as_mut_ptr(), cap, &mut len) };
    if rc != DGPU_OK { return None; }
    *transcript = work;
    out.truncate(len);
    Some(out)
}
/// `verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, rng, transcript, None)`: Some(valid) or None when the library declined.
/// variant 0 Groth16, 1 LegoGroth16, 2 LegoGroth16 proofs under the Groth16 aggregator with `d_list` (using_groth16.rs:45-128); `random`: the
/// scalar RandomizedPairingChecker::new_using_rng would draw (non-zero: a zero would scale every equation after the first out of the check —
/// the library answers DGPU_E_BADARG, i.e. None here).
///
/// `validate`: the proof came from an untrusted source as raw words.  The reference only ever verifies an `AggregateProof` that
/// `CanonicalDeserialize` (`Validate::Yes`) has checked; with `validate = true` (use it unless the words were produced by
/// `aggregate_proofs_gpu` in this process) the library checks every G1 / G2 member for curve and subgroup membership and every GT member for
/// order r before it verifies, and answers Some(false) for a proof that fails.  Like the aggregator, the call works on a clone of the
/// transcript and commits it only on success.
#[allow(clippy::too_many_arguments)]
pub fn verify_aggregate_proof_gpu<T: TranscriptBytes + Clone>(g: &G1Affine, h: &G2Affine, g_alpha: &G1Affine, g_beta: &G1Affine, h_alpha: &G2Affine, h_beta: &G2Affine, srs_n: usize,
                                                      alpha_g1: &G1Affine, beta_g2: &G2Affine, gamma_g2: &G2Affine, delta_g2: &G2Affine, gamma_abc_g1: &[G1Affine],
                                                      public_inputs: &[Vec<Fr>], proof_words: &[u64], variant: i32, d_list: Option<&[G1Affine]>, random: Fr, transcript: &mut T, validate: bool) -> Option<bool> {
    // the C side reads 12 * n_rows words of d_list in variant 2 and none otherwise: a shorter list would be an out-of-bounds read from safe Rust
    if (variant == 2) != d_list.is_some() || !(0..=2).contains(&variant) { return None; }
    if d_list.map_or(false, |x| x.len() != public_inputs.len()) { return Some(false); }
    let g1s = pack_g1(&[*g, *g_alpha, *g_beta, *alpha_g1]).0;
    let g2s = pack_g2(&[*h, *h_alpha, *h_beta, *beta_g2, *gamma_g2, *delta_g2]).0;
    let abc = pack_g1(gamma_abc_g1).0;
    let l = public_inputs.first().map_or(0, |r| r.len());
    if public_inputs.iter().any(|r| r.len() != l) { return Some(false); }
    let pubs: Vec<u64> = public_inputs.iter().flat_map(|r| r.iter().flat_map(|x| x.into_bigint().0)).collect();
    let dl = d_list.map(|x| pack_g1(x).0);
    let s = DgpuSnarkpackVerifierSrs { n: srs_n, g: g1s.as_ptr(), h: g2s.as_ptr(), g_alpha: g1s[12..].as_ptr(), g_beta: g1s[24..].as_ptr(), h_alpha: g2s[24..].as_ptr(), h_beta: g2s[48..].as_ptr() };
    let k = DgpuGroth16Vk { alpha_g1: g1s[36..].as_ptr(), beta_g2: g2s[72..].as_ptr(), gamma_g2: g2s[96..].as_ptr(), delta_g2: g2s[120..].as_ptr(), gamma_abc_g1: abc.as_ptr(), gamma_abc_len: gamma_abc_g1.len() };
    let mut work = transcript.clone();
    let tr = bind_transcript(&mut work);
    let rnd = random.into_bigint();
    let flags = if validate { DGPU_SNARKPACK_VALIDATE_GT | DGPU_SNARKPACK_VALIDATE_POINTS } else { 0 };
    let mut ok = 0i32;
    let rc = unsafe { dgpu_snarkpack_verify(&s, &k, if l == 0 { core::ptr::null() } else { pubs.as_ptr() }, public_inputs.len(), l, proof_words.as_ptr(), proof_words.len(), variant,
                                            dl.as_ref().map_or(core::ptr::null(), |v| v.as_ptr()), rnd.0.as_ptr(), &tr, flags, &mut ok) };
    if rc != DGPU_OK { return None; }
    *transcript = work;
    Some(ok == 1)
}
