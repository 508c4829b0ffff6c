//! rust/dock_gpu/src/host.rs — the parts of the C ABI a Rust HOST drives besides the per-call drop-ins of lib.rs: the process set-up (several GPUs in one
//! process, the resident-bases cache), MSMs sharded over the GPUs of a node (BASELINE config 5 from Rust), the fixed-base batch multiplication behind
//! `WindowTable` (utils/src/msm.rs:8-62), the R1CS -> QAP witness map (legogroth16/src/r1cs_to_qap.rs:150-210), the prover for a key held as arkworks
//! slices (legogroth16/src/prover.rs:267-383 -> `dgpu_legogroth16_prove_host`) and the canonical (de)serialisation of points.
//!
//! Every wrapper answers `None` / `false` when the library declines, so that the caller stays on the arkworks path it replaced.
//! NOT compiled in the build image (no Rust toolchain there); `tests/parity.rs` has a case per wrapper.
use crate::ffi::*;
use crate::{g1_affine, g1_from_xyz, g2_affine, g2_from_xyz, pack_g1, pack_g2};
use ark_bls12_381::{Fr, G1Affine, G1Projective, G2Affine, G2Projective};
use ark_ec::AffineRepr;
use ark_ff::{BigInt, PrimeField};
use ark_std::vec::Vec;
use std::sync::{Arc, Mutex};

// ---- process set-up ---------------------------------------------------------------------------------------------------------------------------
/// several GPUs in ONE process (a Rust host is one process): context k runs on HIP device `devices[k]`.  Start-up code, like `init`.
pub fn init_devices(devices: &[i32], max_n_per_device: usize) -> bool {
    unsafe {
        let _ = dgpu_runtime_hints(DGPU_HINT_EIGHT_HW_QUEUES);
        if dgpu_init_device_list(devices.as_ptr(), devices.len() as i32) != DGPU_OK { return false; }
        for k in 0..devices.len() as i32 {
            if dgpu_set_device(k) != DGPU_OK || dgpu_reserve_g1(max_n_per_device) != DGPU_OK || dgpu_reserve_g2(max_n_per_device) != DGPU_OK { return false; }
        }
        dgpu_set_device(0) == DGPU_OK
    }
}
/// With several device contexts: `msm_bigint_g1` & co. of at least `n` terms shard themselves over all of them (each device caches its own chunk of a key);
/// 0 = off.  BASELINE config 5's shape — 2^24 terms over the 8 GPUs of a node — from the unmodified `msm_bigint(&[G1Affine], ..)` call.
pub fn set_auto_shard_min_n(n: usize) -> bool { unsafe { dgpu_set_auto_shard_min_n(n) == DGPU_OK } }
pub fn device_count() -> i32 { unsafe { dgpu_device_count() } }
pub fn context_count() -> i32 { unsafe { dgpu_context_count() } }
/// the calling thread's device context for the entry points that take host pointers (thread-local, like hipSetDevice)
pub fn set_device(context: i32) -> bool { unsafe { dgpu_set_device(context) == DGPU_OK } }
pub fn error_string(code: i32) -> &'static str {
    unsafe { core::ffi::CStr::from_ptr(dgpu_strerror(code)).to_str().unwrap_or("?") }
}

/// the resident-bases cache behind `msm_bigint_g1` & co. (include/dock_gpu.h `dgpu_set_bases_cache_*`): on by default — the second call with the same
/// `&[G1Affine]` makes it a device-resident table, later calls send the scalars only.  A `&[G1Affine]` cannot change while it is borrowed; what a host
/// does to a cached `Vec` BETWEEN calls is caught by the default `verify_every_record` mode (every record compared beside the MSM).  `verify_samples(24)` trades that
/// for a sampled check — for keys that never change — and then an in-place edit has to be announced with `invalidate`.
pub mod cache {
    use super::*;
    #[derive(Debug, Clone, Copy, Default)]
    pub struct Stats { pub hits: u64, pub misses: u64, pub fills: u64, pub stale: u64, pub evictions: u64, pub bytes: u64, pub budget: u64, pub entries: u64 }
    pub fn set_bytes(bytes: usize) -> bool { unsafe { dgpu_set_bases_cache_bytes(bytes) == DGPU_OK } }
    pub fn set_min_n(n: usize) -> bool { unsafe { dgpu_set_bases_cache_min_n(n) == DGPU_OK } }
    pub fn verify_samples(samples: i32) -> bool { unsafe { dgpu_set_bases_cache_verify(samples) == DGPU_OK } }
    pub fn verify_every_record() -> bool { verify_samples(DGPU_CACHE_VERIFY_FULL) }
    pub fn invalidate<T>(slice: &[T]) -> bool {
        unsafe { dgpu_bases_cache_invalidate(slice.as_ptr() as *const core::ffi::c_void, core::mem::size_of_val(slice)) == DGPU_OK }
    }
    pub fn clear() { unsafe { dgpu_bases_cache_clear(); } }
    pub fn stats() -> Stats {
        let mut w = [0u64; 8];
        unsafe { dgpu_bases_cache_stats(w.as_mut_ptr()); }
        Stats { hits: w[0], misses: w[1], fills: w[2], stale: w[3], evictions: w[4], bytes: w[5], budget: w[6], entries: w[7] }
    }
}

// ---- MSMs sharded over the GPUs of the process (BASELINE config 5) ---------------------------------------------------------------------------------
/// a proving-key query resident across the process's device contexts (point chunks: context k holds the contiguous chunk k)
pub struct ShardedG1 { handle: u64, host: Vec<G1Affine> }
impl ShardedG1 {
    /// ngpus = 0: every initialised context.  `table`: precompute the per-device window tables (what the 2^24-term MSM of config 5 runs on)
    pub fn upload(bases: &[G1Affine], ngpus: i32, table: bool) -> Option<Self> {
        let (xy, inf) = pack_g1(bases);
        let mut h = 0u64;
        if unsafe { dgpu_bases_upload_g1_sharded(xy.as_ptr(), inf.as_ptr(), bases.len(), ngpus, &mut h) } != DGPU_OK { return None; }
        if table { unsafe { dgpu_bases_precompute_g1(h, 0); } }
        Some(ShardedG1 { handle: h, host: bases.to_vec() })
    }
    pub fn handle(&self) -> u64 { self.handle }
    pub fn shards(&self) -> i32 { let mut c = 0i32; unsafe { dgpu_shard_count(self.handle, &mut c); } c }
    /// `msm_bigint(bases, scalars)`: every device multiplies its chunk by the matching scalars (32 B per term over its own PCIe link), the partial
    /// points (144 B each) are folded on the host — the same group element as the single-device call
    pub fn msm_bigint(&self, scalars: &[BigInt<4>]) -> G1Projective {
        use ark_ec::VariableBaseMSM;
        let n = self.host.len().min(scalars.len());
        let mut out = [0u64; 18];
        if unsafe { dgpu_msm_g1_sharded_handle(self.handle, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) } != DGPU_OK {
            return G1Projective::msm_bigint(&self.host[..n], &scalars[..n]);
        }
        g1_from_xyz(&out)
    }
    /// scalars resident beside the bases (config 5's timed region: inputs pre-sharded): upload once, multiply any number of times
    pub fn upload_scalars(&self, scalars: &[BigInt<4>]) -> Option<ShardedScalars> {
        let mut h = 0u64;
        if unsafe { dgpu_scalars_upload_sharded(scalars.as_ptr() as *const u64, scalars.len().min(self.host.len()), 0, self.handle, &mut h) } != DGPU_OK { return None; }
        Some(ShardedScalars { handle: h })
    }
    pub fn msm_resident(&self, scalars: &ShardedScalars) -> Option<G1Projective> {
        let mut out = [0u64; 18];
        if unsafe { dgpu_msm_g1_sharded_resident(self.handle, scalars.handle, out.as_mut_ptr()) } != DGPU_OK { return None; }
        Some(g1_from_xyz(&out))
    }
}
impl Drop for ShardedG1 { fn drop(&mut self) { unsafe { dgpu_bases_free(self.handle); } } }
pub struct ShardedScalars { handle: u64 }
impl Drop for ShardedScalars { fn drop(&mut self) { unsafe { dgpu_scalars_free(self.handle); } } }
/// the same for G2 (b_g2_query of a sharded key)
pub struct ShardedG2 { handle: u64, host: Vec<G2Affine> }
impl ShardedG2 {
    pub fn upload(bases: &[G2Affine], ngpus: i32, table: bool) -> Option<Self> {
        let (xy, inf) = pack_g2(bases);
        let mut h = 0u64;
        if unsafe { dgpu_bases_upload_g2_sharded(xy.as_ptr(), inf.as_ptr(), bases.len(), ngpus, &mut h) } != DGPU_OK { return None; }
        if table { unsafe { dgpu_bases_precompute_g2(h, 0); } }
        Some(ShardedG2 { handle: h, host: bases.to_vec() })
    }
    pub fn handle(&self) -> u64 { self.handle }
    pub fn msm_bigint(&self, scalars: &[BigInt<4>]) -> G2Projective {
        use ark_ec::VariableBaseMSM;
        let n = self.host.len().min(scalars.len());
        let mut out = [0u64; 36];
        if unsafe { dgpu_msm_g2_sharded_handle(self.handle, scalars.as_ptr() as *const u64, n, 0, out.as_mut_ptr()) } != DGPU_OK {
            return G2Projective::msm_bigint(&self.host[..n], &scalars[..n]);
        }
        g2_from_xyz(&out)
    }
}
impl Drop for ShardedG2 { fn drop(&mut self) { unsafe { dgpu_bases_free(self.handle); } } }
/// one-shot form: host bases and scalars in, every device pulls its own chunk (`dgpu_msm_g1_sharded`)
pub fn msm_bigint_g1_sharded(bases: &[G1Affine], scalars: &[BigInt<4>], ngpus: i32) -> G1Projective {
    use ark_ec::VariableBaseMSM;
    let n = bases.len().min(scalars.len());
    let (xy, inf) = pack_g1(&bases[..n]);
    let mut out = [0u64; 18];
    if unsafe { dgpu_msm_g1_sharded(xy.as_ptr(), inf.as_ptr(), scalars.as_ptr() as *const u64, n, ngpus, out.as_mut_ptr()) } != DGPU_OK {
        return G1Projective::msm_bigint(&bases[..n], &scalars[..n]);
    }
    g1_from_xyz(&out)
}
/// multi-process form (one rank per GPU): the fold of the ranks' partial results after the all-gather (EC addition is not an RCCL reduction)
pub fn fold_g1(partials: &[G1Projective]) -> G1Projective {
    use ark_ec::CurveGroup;
    let one = ark_bls12_381::Fq::from(1u64);
    let mut w = Vec::with_capacity(partials.len() * 18);
    for p in partials {
        let a = p.into_affine();
        match a.xy() {
            Some((x, y)) => { w.extend_from_slice(&x.0 .0); w.extend_from_slice(&y.0 .0); w.extend_from_slice(&one.0 .0); }
            None => { w.extend_from_slice(&one.0 .0); w.extend_from_slice(&one.0 .0); w.extend_from_slice(&[0u64; 6]); }
        }
    }
    let mut out = [0u64; 18];
    if unsafe { dgpu_fold_g1(w.as_ptr(), partials.len(), out.as_mut_ptr()) } != DGPU_OK { return partials.iter().copied().sum(); }
    g1_from_xyz(&out)
}

// ---- fixed-base batch multiplication: `WindowTable` (utils/src/msm.rs:8-62) and the CRS generator's six query computations ------------------------
/// `WindowTable::<G1Projective>::new(_, base)`: 32 x 255 multiples of the base resident on the device
pub struct WindowTableG1 { handle: u64 }
impl WindowTableG1 {
    pub fn new(base: &G1Affine) -> Option<Self> {
        let (xy, inf) = pack_g1(core::slice::from_ref(base));
        if inf[0] != 0 { return None; }
        let mut h = 0u64;
        if unsafe { dgpu_window_table_g1(xy.as_ptr(), &mut h) } != DGPU_OK { return None; }
        Some(WindowTableG1 { handle: h })
    }
    /// `multiply_many(&[Fr])` (utils/src/msm.rs:40-42) — the products as affine points (the reference normalises them next: generator.rs:424-431)
    pub fn multiply_many(&self, elements: &[Fr]) -> Option<Vec<G1Affine>> {
        let n = elements.len();
        let (mut xy, mut inf) = (ark_std::vec![0u64; n * 12], ark_std::vec![0u8; n]);
        if unsafe { dgpu_window_table_mul_g1(self.handle, elements.as_ptr() as *const u64, n, 1, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
        Some((0..n).map(|i| g1_affine(xy[12 * i..12 * i + 12].try_into().unwrap(), inf[i])).collect())
    }
    /// the same products left on the device as an MSM bases handle: a CRS query goes from the generator to the prover without crossing PCIe
    pub fn multiply_many_to_bases(&self, elements: &[Fr]) -> Option<u64> {
        let mut h = 0u64;
        if unsafe { dgpu_window_table_mul_to_bases_g1(self.handle, elements.as_ptr() as *const u64, elements.len(), 1, &mut h) } != DGPU_OK { return None; }
        Some(h)
    }
}
impl Drop for WindowTableG1 { fn drop(&mut self) { unsafe { dgpu_window_table_free(self.handle); } } }
pub struct WindowTableG2 { handle: u64 }
impl WindowTableG2 {
    pub fn new(base: &G2Affine) -> Option<Self> {
        let (xy, inf) = pack_g2(core::slice::from_ref(base));
        if inf[0] != 0 { return None; }
        let mut h = 0u64;
        if unsafe { dgpu_window_table_g2(xy.as_ptr(), &mut h) } != DGPU_OK { return None; }
        Some(WindowTableG2 { handle: h })
    }
    pub fn multiply_many(&self, elements: &[Fr]) -> Option<Vec<G2Affine>> {
        let n = elements.len();
        let (mut xy, mut inf) = (ark_std::vec![0u64; n * 24], ark_std::vec![0u8; n]);
        if unsafe { dgpu_window_table_mul_g2(self.handle, elements.as_ptr() as *const u64, n, 1, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
        Some((0..n).map(|i| g2_affine(xy[24 * i..24 * i + 24].try_into().unwrap(), inf[i])).collect())
    }
}
impl Drop for WindowTableG2 { fn drop(&mut self) { unsafe { dgpu_window_table_free(self.handle); } } }
/// `multiply_field_elems_with_same_group_elem(base, elements)` (utils/src/msm.rs:55-62): table + products + free in one call
pub fn fixed_base_g1(base: &G1Affine, elements: &[Fr]) -> Option<Vec<G1Affine>> {
    let (b, binf) = pack_g1(core::slice::from_ref(base));
    if binf[0] != 0 { return Some(ark_std::vec![G1Affine::identity(); elements.len()]); }
    let n = elements.len();
    let (mut xy, mut inf) = (ark_std::vec![0u64; n * 12], ark_std::vec![0u8; n]);
    if unsafe { dgpu_fixed_base_g1(b.as_ptr(), elements.as_ptr() as *const u64, n, 1, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
    Some((0..n).map(|i| g1_affine(xy[12 * i..12 * i + 12].try_into().unwrap(), inf[i])).collect())
}
pub fn fixed_base_g2(base: &G2Affine, elements: &[Fr]) -> Option<Vec<G2Affine>> {
    let (b, binf) = pack_g2(core::slice::from_ref(base));
    if binf[0] != 0 { return Some(ark_std::vec![G2Affine::identity(); elements.len()]); }
    let n = elements.len();
    let (mut xy, mut inf) = (ark_std::vec![0u64; n * 24], ark_std::vec![0u8; n]);
    if unsafe { dgpu_fixed_base_g2(b.as_ptr(), elements.as_ptr() as *const u64, n, 1, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
    Some((0..n).map(|i| g2_affine(xy[24 * i..24 * i + 24].try_into().unwrap(), inf[i])).collect())
}

// ---- R1CS -> QAP witness map (legogroth16/src/r1cs_to_qap.rs:150-210) ---------------------------------------------------------------------------------
/// one constraint matrix of ark-relations' `ConstraintMatrices` (`Matrix<F> = Vec<Vec<(F, usize)>>`) as the CSR arrays the ABI takes, with a 64-bit hash of
/// its whole contents computed on the way (the copy reads every coefficient anyway)
pub struct Csr { pub rowptr: Vec<u64>, pub cols: Vec<u32>, pub vals: Vec<u64>, pub hash: u64 }
pub fn flatten(m: &[Vec<(Fr, usize)>]) -> Csr {
    let nnz: usize = m.iter().map(|r| r.len()).sum();
    let (mut rowptr, mut cols, mut vals) = (Vec::with_capacity(m.len() + 1), Vec::with_capacity(nnz), Vec::with_capacity(nnz * 4));
    let mut h = 0xcbf29ce484222325u64 ^ (m.len() as u64);
    let mix = |h: &mut u64, w: u64| { *h = (*h ^ w).wrapping_mul(0x100000001b3).rotate_left(29); };
    rowptr.push(0u64);
    for row in m {
        for (coeff, col) in row {
            cols.push(*col as u32);
            vals.extend_from_slice(&coeff.0 .0);                       // Montgomery limbs as they lie in memory (montgomery = 1 at the ABI)
            mix(&mut h, *col as u64);
            for w in coeff.0 .0 { mix(&mut h, w); }
        }
        rowptr.push(cols.len() as u64);
        mix(&mut h, cols.len() as u64);
    }
    Csr { rowptr, cols, vals, hash: h }
}
/// a circuit's three matrices resident on the device (they are fixed per circuit; only the assignment changes per proof)
pub struct R1cs { handle: u64, pub num_vars: usize, pub num_inputs: usize, pub num_constraints: usize }
impl R1cs {
    /// `R1cs::upload(&matrices.a, &matrices.b, &matrices.c, ..)` — `ConstraintMatrices<Fr>` of ark-relations (this crate does not depend on it: the
    /// matrices come as the slices they are)
    pub fn upload(a: &[Vec<(Fr, usize)>], b: &[Vec<(Fr, usize)>], c: &[Vec<(Fr, usize)>], num_vars: usize, num_inputs: usize, num_constraints: usize) -> Option<Self> {
        Self::upload_csr(&flatten(a), &flatten(b), &flatten(c), num_vars, num_inputs, num_constraints)
    }
    pub fn upload_csr(a: &Csr, b: &Csr, c: &Csr, num_vars: usize, num_inputs: usize, num_constraints: usize) -> Option<Self> {
        let mut h = 0u64;
        let rc = unsafe { dgpu_r1cs_upload(a.rowptr.as_ptr(), a.cols.as_ptr(), a.vals.as_ptr(), a.cols.len(), b.rowptr.as_ptr(), b.cols.as_ptr(), b.vals.as_ptr(), b.cols.len(),
                                           c.rowptr.as_ptr(), c.cols.as_ptr(), c.vals.as_ptr(), c.cols.len(), num_vars, num_inputs, num_constraints, 1, &mut h) };
        if rc != DGPU_OK { return None; }
        Some(R1cs { handle: h, num_vars, num_inputs, num_constraints })
    }
    pub fn handle(&self) -> u64 { self.handle }
    /// `witness_map_from_matrices(.., full_assignment)` on the resident circuit: the D coefficients of h as `Vec<Fr>` (what the reference's function
    /// returns: the conversion to Montgomery form runs on the device, DGPU_WM_H_MONTGOMERY)
    pub fn witness_map(&self, full_assignment: &[Fr]) -> Option<Vec<Fr>> {
        if full_assignment.len() != self.num_vars { return None; }
        let d = (self.num_constraints + self.num_inputs).next_power_of_two().max(2);
        let mut h = ark_std::vec![Fr::from(0u64); d];
        let mut len = 0usize;
        let rc = unsafe { dgpu_witness_map_r1cs(self.handle, full_assignment.as_ptr() as *const u64, self.num_vars, 1 | DGPU_WM_H_MONTGOMERY,
                                                h.as_mut_ptr() as *mut u64, core::ptr::null_mut(), &mut len) };
        if rc != DGPU_OK || len != d { return None; }
        Some(h)
    }
}
impl Drop for R1cs { fn drop(&mut self) { unsafe { dgpu_r1cs_free(self.handle); } } }

/// The circuits this process has proved with, by the content hash of their matrices: `QAP::witness_map` calls `cs.to_matrices()` per proof
/// (r1cs_to_qap.rs:57-80), so the matrices arrive as fresh `Vec`s every time — their CONTENTS repeat.  Flattening reads every coefficient anyway; the hash
/// it produces on the way (64 bits over all three matrices and the shape) finds the resident copy, and the 100+ MB of coefficients do not cross PCIe
/// again.  Four circuits are kept (least recently used goes).  The prover chooses its own circuits: the hash guards against accidents, not adversaries.
struct CircuitEntry { key: (u64, u64, u64, usize, usize, usize), circuit: Arc<R1cs>, used: u64 }
static CIRCUITS: Mutex<(Vec<CircuitEntry>, u64)> = Mutex::new((Vec::new(), 0));
pub fn resident_circuit(a: &[Vec<(Fr, usize)>], b: &[Vec<(Fr, usize)>], c: &[Vec<(Fr, usize)>], num_vars: usize, num_inputs: usize, num_constraints: usize) -> Option<Arc<R1cs>> {
    let (fa, fb, fc) = (flatten(a), flatten(b), flatten(c));
    let key = (fa.hash, fb.hash, fc.hash, num_vars, num_inputs, num_constraints);
    {
        let mut g = CIRCUITS.lock().ok()?;
        g.1 += 1;
        let now = g.1;
        if let Some(e) = g.0.iter_mut().find(|e| e.key == key) { e.used = now; return Some(e.circuit.clone()); }
    }
    let fresh = Arc::new(R1cs::upload_csr(&fa, &fb, &fc, num_vars, num_inputs, num_constraints)?);
    let mut g = CIRCUITS.lock().ok()?;
    g.1 += 1;
    let now = g.1;
    if g.0.len() >= 4 { let (i, _) = g.0.iter().enumerate().min_by_key(|(_, e)| e.used)?; g.0.swap_remove(i); }
    g.0.push(CircuitEntry { key, circuit: fresh.clone(), used: now });
    Some(fresh)
}
/// drop-in for `LibsnarkReduction::witness_map_from_matrices` (r1cs_to_qap.rs:150-210): None when the library declined (the caller runs the CPU map)
pub fn witness_map_from_matrices(a: &[Vec<(Fr, usize)>], b: &[Vec<(Fr, usize)>], c: &[Vec<(Fr, usize)>], num_inputs: usize, num_constraints: usize, full_assignment: &[Fr]) -> Option<Vec<Fr>> {
    if num_constraints < (1 << 12) { return None; }              // small circuits: the CPU map is microseconds-to-milliseconds, a device round trip is not
    resident_circuit(a, b, c, full_assignment.len(), num_inputs, num_constraints)?.witness_map(full_assignment)
}

// ---- the LegoGroth16 prover for a key held the way the reference holds it ---------------------------------------------------------------------------------
const G1_STRIDE: usize = core::mem::size_of::<G1Affine>();
const G2_STRIDE: usize = core::mem::size_of::<G2Affine>();
fn view_g1(q: &[G1Affine]) -> DgpuBasesView {
    DgpuBasesView { p: q.as_ptr() as *const core::ffi::c_void, stride: G1_STRIDE, x_off: core::mem::offset_of!(G1Affine, x), y_off: core::mem::offset_of!(G1Affine, y),
                    inf_off: core::mem::offset_of!(G1Affine, infinity), n: q.len() }
}
fn view_g2(q: &[G2Affine]) -> DgpuBasesView {
    DgpuBasesView { p: q.as_ptr() as *const core::ffi::c_void, stride: G2_STRIDE, x_off: core::mem::offset_of!(G2Affine, x), y_off: core::mem::offset_of!(G2Affine, y),
                    inf_off: core::mem::offset_of!(G2Affine, infinity), n: q.len() }
}
/// the members of `ProvingKeyCommon<Bls12_381>` + `VerifyingKey<Bls12_381>` the prover reads (legogroth16/src/data_structures.rs:55-70,151-168), borrowed
pub struct HostProvingKey<'a> {
    pub alpha_g1: G1Affine, pub beta_g1: G1Affine, pub delta_g1: G1Affine, pub eta_delta_inv_g1: G1Affine, pub eta_gamma_inv_g1: G1Affine,
    pub beta_g2: G2Affine, pub delta_g2: G2Affine, pub gamma_abc_g1: &'a [G1Affine], pub commit_witness_count: usize,
    pub a_query: &'a [G1Affine], pub b_g1_query: &'a [G1Affine], pub b_g2_query: &'a [G2Affine], pub h_query: &'a [G1Affine], pub l_query: &'a [G1Affine],
}
/// where h comes from: the coefficients `QAP::witness_map` returned (create_proof_with_assignment's contract, prover.rs:237-265) or a resident circuit
/// (create_proof_with_reduction, prover.rs:153-180: the witness map runs inside the call, h never leaves the device)
pub enum HSource<'a> { Coefficients(&'a [Fr]), Circuit(&'a R1cs) }
/// `create_proof_and_committed_witnesses_with_assignment` (prover.rs:267-383) as ONE call: (A, B, C, D), or None when the library declined (the caller
/// then runs the reference's CPU body).  `input_assignment` includes the leading 1 (arkworks' instance_assignment).  The key's queries are resolved by the
/// library's resident-bases cache: a key's first proof uploads them for the call, its second makes them resident tables, later proofs send only the
/// assignment (and h).
pub fn create_proof_host(pk: &HostProvingKey, h: HSource, input_assignment: &[Fr], witness_assignment: &[Fr], r: Fr, s: Fr, v: Fr) -> Option<(G1Affine, G2Affine, G1Affine, G1Affine)> {
    if pk.a_query.is_empty() || pk.b_g1_query.is_empty() || pk.b_g2_query.is_empty() || input_assignment.is_empty() { return None; }
    let mut s1: Vec<G1Affine> = ark_std::vec![pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.eta_delta_inv_g1, pk.eta_gamma_inv_g1, pk.a_query[0], pk.b_g1_query[0]];
    s1.extend_from_slice(pk.gamma_abc_g1);
    let small_g1 = pack_g1(&s1).0;
    let small_g2 = pack_g2(&[pk.beta_g2, pk.delta_g2, pk.b_g2_query[0]]).0;
    let g1 = |k: usize| unsafe { small_g1.as_ptr().add(12 * k) };
    let g2 = |k: usize| unsafe { small_g2.as_ptr().add(24 * k) };
    let raw = DgpuLegoPkHost {
        a_query: view_g1(pk.a_query), b_g1_query: view_g1(pk.b_g1_query), b_g2_query: view_g2(pk.b_g2_query), h_query: view_g1(pk.h_query), l_query: view_g1(pk.l_query),
        alpha_g1: g1(0), beta_g1: g1(1), delta_g1: g1(2), eta_delta_inv_g1: g1(3), eta_gamma_inv_g1: g1(4), beta_g2: g2(0), delta_g2: g2(1),
        a0: g1(5), b1_0: g1(6), b2_0: g2(2), gamma_abc_g1: g1(7), gamma_abc_len: pk.gamma_abc_g1.len(), commit_witness_count: pk.commit_witness_count,
    };
    let (r1cs, hp, hl) = match h {
        HSource::Coefficients(c) => (0u64, c.as_ptr() as *const u64, c.len()),
        HSource::Circuit(c) => (c.handle(), core::ptr::null(), 0usize),
    };
    let (rb, sb, vb) = (r.into_bigint(), s.into_bigint(), v.into_bigint());
    let (mut a, mut b, mut c, mut d, mut inf) = ([0u64; 12], [0u64; 24], [0u64; 12], [0u64; 12], [0u8; 4]);
    let rc = unsafe { dgpu_legogroth16_prove_host(&raw, r1cs, hp, hl, 1, input_assignment.as_ptr() as *const u64, input_assignment.len(),
                                                  witness_assignment.as_ptr() as *const u64, witness_assignment.len(), 1, rb.0.as_ptr(), sb.0.as_ptr(), vb.0.as_ptr(),
                                                  a.as_mut_ptr(), b.as_mut_ptr(), c.as_mut_ptr(), d.as_mut_ptr(), inf.as_mut_ptr()) };
    if rc != DGPU_OK { return None; }
    Some((g1_affine(&a, inf[0]), g2_affine(&b, inf[1]), g1_affine(&c, inf[2]), g1_affine(&d, inf[3])))
}

// ---- canonical (de)serialisation (ark-serialize's format for BLS12-381 points: Zcash / IETF) ------------------------------------------------------------
pub fn serialize_g1(points: &[G1Affine], compressed: bool) -> Option<Vec<u8>> {
    let (xy, inf) = pack_g1(points);
    let mut out = ark_std::vec![0u8; points.len() * if compressed { 48 } else { 96 }];
    if unsafe { dgpu_g1_serialize(xy.as_ptr(), inf.as_ptr(), points.len(), compressed as i32, out.as_mut_ptr()) } != DGPU_OK { return None; }
    Some(out)
}
/// `validate`: arkworks' Validate::Yes (curve and subgroup membership); None: a malformed encoding or a point that fails the checks
pub fn deserialize_g1(bytes: &[u8], n: usize, compressed: bool, validate: bool) -> Option<Vec<G1Affine>> {
    if bytes.len() != n * if compressed { 48 } else { 96 } { return None; }
    let (mut xy, mut inf) = (ark_std::vec![0u64; n * 12], ark_std::vec![0u8; n]);
    let mode = (compressed as i32) | if validate { 0 } else { DGPU_SERDE_NO_VALIDATE };
    if unsafe { dgpu_g1_deserialize(bytes.as_ptr(), n, mode, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
    Some((0..n).map(|i| g1_affine(xy[12 * i..12 * i + 12].try_into().unwrap(), inf[i])).collect())
}
pub fn serialize_g2(points: &[G2Affine], compressed: bool) -> Option<Vec<u8>> {
    let (xy, inf) = pack_g2(points);
    let mut out = ark_std::vec![0u8; points.len() * if compressed { 96 } else { 192 }];
    if unsafe { dgpu_g2_serialize(xy.as_ptr(), inf.as_ptr(), points.len(), compressed as i32, out.as_mut_ptr()) } != DGPU_OK { return None; }
    Some(out)
}
pub fn deserialize_g2(bytes: &[u8], n: usize, compressed: bool, validate: bool) -> Option<Vec<G2Affine>> {
    if bytes.len() != n * if compressed { 96 } else { 192 } { return None; }
    let (mut xy, mut inf) = (ark_std::vec![0u64; n * 24], ark_std::vec![0u8; n]);
    let mode = (compressed as i32) | if validate { 0 } else { DGPU_SERDE_NO_VALIDATE };
    if unsafe { dgpu_g2_deserialize(bytes.as_ptr(), n, mode, xy.as_mut_ptr(), inf.as_mut_ptr()) } != DGPU_OK { return None; }
    Some((0..n).map(|i| g2_affine(xy[24 * i..24 * i + 24].try_into().unwrap(), inf[i])).collect())
}
