// crypto_amd/csrc/dock_qap.hip — dgpu_witness_map (include/dock_gpu.h): the R1CS -> QAP witness map on the GPU.
// Replaces LibsnarkReduction::witness_map_from_matrices (/root/reference/legogroth16/src/r1cs_to_qap.rs:150-210), the step
// immediately before the prover's h_query MSM (legogroth16/src/prover.rs:281-286).  The result can stay in HBM as a scalars
// handle and be fed straight to dgpu_msm_g1_resident — no D2H/H2D between the transform and the MSM.
#include "dock_ctx.hpp"
#include "host_field.hpp"
#include "qap_launch.hip.h"

namespace {
using namespace dock;
using hostf::FrH;

struct Csr { const uint64_t *rowptr; const uint32_t *cols; const uint64_t *vals; size_t nnz; };

int32_t get_domain(Slot &sl, int logn, NttDomain &out) {
    {
        std::lock_guard<std::mutex> lk(gs.mu);
        auto it = cur().ntt_domains.find(logn);
        if (it != cur().ntt_domains.end()) { out = it->second; return DGPU_OK; }
    }
    const size_t D = (size_t)1 << logn, H = D >> 1 ? D >> 1 : 1;
    FrH w = FrH::root_of_unity(logn), wi = w.inv(), gk = FrH::from_u64(7), gi = gk.inv(), dinv = FrH::from_u64(D).inv(), one = FrH::from_u64(1);
    FrH gd = gk; for (int k = 0; k < logn; k++) gd = gd * gd;
    FrH zinv = gd.sub_one().inv();
    uint64_t consts[7][4];
    w.to_canonical(consts[0]); wi.to_canonical(consts[1]); gk.to_canonical(consts[2]); gi.to_canonical(consts[3]);
    dinv.to_canonical(consts[4]); one.to_canonical(consts[5]); zinv.to_canonical(consts[6]);
    NttDomain d;
    void *dc = nullptr;
    const size_t esz = ntt::FR_WORDS * 4;
    if (dev_malloc(&dc, sizeof consts) != hipSuccess || dev_malloc(&d.tw_f, 2 * H * esz) != hipSuccess || dev_malloc(&d.tw_i, 2 * H * esz) != hipSuccess ||
        dev_malloc(&d.pw_f, D * esz) != hipSuccess || dev_malloc(&d.pw_i, D * esz) != hipSuccess || dev_malloc(&d.zinv, 32) != hipSuccess ||
        dev_malloc(&d.pwr_f, D * esz) != hipSuccess || dev_malloc(&d.pwr_i, D * esz) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(dc, consts, sizeof consts, hipMemcpyHostToDevice, s));
    const uint32_t *c32 = (const uint32_t *)dc;
    ntt::launch_fr_powers(s, c32 + 0 * 8, c32 + 5 * 8, H, (uint32_t *)d.tw_f);    // w^k
    ntt::launch_fr_powers(s, c32 + 1 * 8, c32 + 5 * 8, H, (uint32_t *)d.tw_i);    // w^-k
    ntt::launch_tw_compact(s, (uint32_t *)d.tw_f, H); ntt::launch_tw_compact(s, (uint32_t *)d.tw_i, H);   // per-stage tables behind the full ones
    ntt::launch_fr_powers(s, c32 + 2 * 8, c32 + 4 * 8, D, (uint32_t *)d.pw_f);    // g^k / D
    ntt::launch_fr_powers(s, c32 + 3 * 8, c32 + 4 * 8, D, (uint32_t *)d.pw_i);    // g^-k / D
    ntt::launch_bitrev_table(s, (const uint32_t *)d.pw_f, (uint32_t *)d.pwr_f, logn);   // the same factors in the order of bit-reversed data (coalesced)
    ntt::launch_bitrev_table(s, (const uint32_t *)d.pw_i, (uint32_t *)d.pwr_i, logn);
    HIPCHK(hipMemcpyAsync(d.zinv, consts[6], 32, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    (void)hipFree(dc);
    std::lock_guard<std::mutex> lk(gs.mu);
    auto ins = cur().ntt_domains.emplace(logn, d);
    if (!ins.second) { void *ps[] = {d.tw_f, d.tw_i, d.pw_f, d.pw_i, d.zinv, d.pwr_f, d.pwr_i}; for (void *p : ps) (void)hipFree(p); }   // another thread won the race
    out = ins.first->second;
    return DGPU_OK;
}

// device-resident R1CS (three CSR matrices, coefficients already in the internal Fr form): uploaded once per circuit
struct DevCsr { uint64_t *rowptr = nullptr; uint32_t *cols = nullptr; uint32_t *vals = nullptr; size_t nnz = 0; };
struct DevR1cs { DevCsr m[3]; size_t num_vars = 0, num_inputs = 0, num_constraints = 0; };

void free_r1cs(DevR1cs *r) {
    if (!r) return;
    for (auto &c : r->m) { if (c.rowptr) (void)hipFree(c.rowptr); if (c.cols) (void)hipFree(c.cols); if (c.vals) (void)hipFree(c.vals); }
    delete r;
}
int32_t upload_matrix(Slot &sl, const Csr &m, size_t rows, int mont, DevCsr &out) {
    const size_t nnz = m.nnz ? m.nnz : 1;
    int32_t rc;
    if ((rc = sl.q[0].ensure(nnz * 32))) return rc;
    if (dev_malloc((void **)&out.rowptr, (rows + 1) * 8) != hipSuccess || dev_malloc((void **)&out.cols, nnz * 4) != hipSuccess ||
        dev_malloc((void **)&out.vals, nnz * ntt::FR_WORDS * 4) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_OOM; }
    out.nnz = nnz;
    hipStream_t s = sl.stream;
    HIPCHK(hipMemcpyAsync(out.rowptr, m.rowptr, (rows + 1) * 8, hipMemcpyHostToDevice, s));
    if (m.nnz) { HIPCHK(hipMemcpyAsync(out.cols, m.cols, m.nnz * 4, hipMemcpyHostToDevice, s)); HIPCHK(hipMemcpyAsync(sl.q[0].p, m.vals, m.nnz * 32, hipMemcpyHostToDevice, s)); }
    ntt::launch_fr_load(s, sl.q[0].as<uint32_t>(), m.nnz, mont, out.vals, nnz);
    HIPCHK(hipStreamSynchronize(s));     // q[0] is reused by the next matrix
    return DGPU_OK;
}

// assignment: host scalars (uploaded here), or d_assignment != nullptr: canonical scalars already resident on this device
int32_t witness_map_device(Slot &sl, const DevR1cs &r, const uint64_t *assignment, int32_t montgomery, uint64_t *out_h, uint64_t *out_handle, size_t *out_len, const uint32_t *d_assignment = nullptr) {
    int logn = 0; while (((size_t)1 << logn) < r.num_constraints + r.num_inputs) logn++;
    if (logn < 1) logn = 1;
    if (logn > 28) return DGPU_E_BADARG;
    const size_t D = (size_t)1 << logn;
    NttDomain dom; int32_t rc;
    if ((rc = get_domain(sl, logn, dom))) return rc;
    const size_t esz = ntt::FR_WORDS * 4;
    Buf &zw = sl.q[12], &qa = sl.q[14], &qb = sl.q[15], &qc = sl.digits, &hw = sl.entries;   // digits/entries: reused scratch
    if (!d_assignment && (rc = zw.ensure(r.num_vars * 32))) return rc;
    if ((rc = qa.ensure(D * esz))) return rc;
    if ((rc = qb.ensure(D * esz))) return rc;
    if ((rc = qc.ensure(D * esz))) return rc;
    // the h scalars are written where they stay: the caller's resident vector (a recycled buffer) if one is asked for, else scratch
    void *kept = nullptr;
    if (out_handle) { if (!(kept = scalar_alloc(scalar_bytes(D)))) return DGPU_E_OOM; }
    else if ((rc = hw.ensure(D * 32))) return rc;
    uint32_t *const h_words = kept ? (uint32_t *)kept : hw.as<uint32_t>();
    hipStream_t s = sl.stream;
    struct Giveback { void *&p; size_t bytes; hipStream_t st; ~Giveback() { if (p) { (void)hipStreamSynchronize(st); scalar_release(cur_index(), p, bytes); } } } giveback{kept, scalar_bytes(D), s};   // on any early return
    uint32_t *arr[3] = {qa.as<uint32_t>(), qb.as<uint32_t>(), qc.as<uint32_t>()};
    {
        StageTimer st(sl, "qap.matvec");
        if (!d_assignment) HIPCHK(hipMemcpyAsync(zw.p, assignment, r.num_vars * 32, hipMemcpyHostToDevice, s));
        const uint32_t *zsrc = d_assignment ? d_assignment : zw.as<uint32_t>();
        for (int k = 0; k < 3; k++)
            ntt::launch_csr_eval(s, r.m[k].rowptr, r.m[k].cols, r.m[k].vals, r.m[k].nnz, zsrc, d_assignment ? 0 : (montgomery & 1), r.num_vars, r.num_constraints, k == 0 ? r.num_inputs : 0, arr[k], D);
    }
    {
        StageTimer st(sl, "qap.ntt");
        ntt::launch_ntt_batch(s, arr, 3, logn, (const uint32_t *)dom.tw_i, 1);                                       // iFFT (x D) of a, b, c together, bit-reversed out
        ntt::launch_ntt_batch(s, arr, 3, logn, (const uint32_t *)dom.tw_f, 0, (const uint32_t *)dom.pwr_f);          // * g^k / D on the way in, coset FFT, natural out
        // (ab - c) / Z(g) -> coset iFFT -> * g^-k / D, un-reversed, canonical words: one set of passes
        ntt::launch_ntt_final(s, arr[0], arr[1], arr[2], logn, (const uint32_t *)dom.tw_i, (const uint32_t *)dom.zinv, (const uint32_t *)dom.pwr_i, h_words);
    }
    HIPCHK(hipGetLastError());
    if (out_len) *out_len = D;
    if (out_h) {
        const uint32_t *src = h_words;
        if (montgomery & DGPU_WM_H_MONTGOMERY) {          // the host copy as &[Fr] (what QAP::witness_map returns); a resident vector stays canonical
            if (kept) { if ((rc = hw.ensure(D * 32))) return rc; HIPCHK(hipMemcpyAsync(hw.p, h_words, D * 32, hipMemcpyDeviceToDevice, s)); }
            ntt::launch_fr_canonical_to_mont(s, hw.as<uint32_t>(), D);
            src = hw.as<uint32_t>();
        }
        if (hipMemcpyAsync(out_h, src, D * 32, hipMemcpyDeviceToHost, s) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_HIP; }
    }
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return DGPU_E_HIP; }
    if (out_handle) { *out_handle = register_handle(kept, D, 3); kept = nullptr; }     // registered last: no handle is left behind by a failing call
    if (gs.prof) prof_flush(sl);
    return DGPU_OK;
}

int32_t build_r1cs(Slot &sl, const Csr mats[3], size_t num_vars, size_t num_inputs, size_t num_constraints, int32_t montgomery, DevR1cs **out) {
    DevR1cs *r = new DevR1cs();
    r->num_vars = num_vars; r->num_inputs = num_inputs; r->num_constraints = num_constraints;
    for (int k = 0; k < 3; k++) { int32_t rc = upload_matrix(sl, mats[k], num_constraints, montgomery & 1, r->m[k]); if (rc) { free_r1cs(r); return rc; } }
    *out = r;
    return DGPU_OK;
}

}  // namespace
namespace dock { void free_r1cs_object(void *p) { free_r1cs((DevR1cs *)p); } }

extern "C" {

// a malformed matrix must not reach the device (k_csr_eval indexes cols / vals / z through it): row pointers start at 0, never
// decrease and end at nnz; every column names a variable.  O(rows + nnz) on the host, once per upload.
static bool check_csr(const uint64_t *rp, const uint32_t *cl, const uint64_t *vl, size_t nnz, size_t rows, size_t num_vars) {
    if (!rp || (nnz && (!cl || !vl))) return false;
    if (rp[0] != 0 || rp[rows] != nnz) return false;
    for (size_t i = 0; i < rows; i++) if (rp[i + 1] < rp[i]) return false;
    for (size_t k = 0; k < nnz; k++) if (cl[k] >= num_vars) return false;
    return true;
}

int32_t dgpu_r1cs_upload(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals, size_t a_nnz,
                         const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals, size_t b_nnz,
                         const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals, size_t c_nnz,
                         size_t num_vars, size_t num_inputs, size_t num_constraints, int32_t montgomery, uint64_t *handle) {
    if (!handle || num_inputs > num_vars || !check_csr(a_rowptr, a_cols, a_vals, a_nnz, num_constraints, num_vars) || !check_csr(b_rowptr, b_cols, b_vals, b_nnz, num_constraints, num_vars) || !check_csr(c_rowptr, c_cols, c_vals, c_nnz, num_constraints, num_vars)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    DevR1cs *r = nullptr;
    {
        SLOT_ACQUIRE(slot_lock, sl);
        HIPCHK(hipSetDevice(cur().device));
        Csr mats[3] = {{a_rowptr, a_cols, a_vals, a_nnz}, {b_rowptr, b_cols, b_vals, b_nnz}, {c_rowptr, c_cols, c_vals, c_nnz}};
        int32_t rc = build_r1cs(sl, mats, num_vars, num_inputs, num_constraints, montgomery, &r);
        if (rc) return rc;
    }
    *handle = register_handle(r, num_constraints, 4);
    return DGPU_OK;
}
// shape of a resident circuit (what a caller must agree with: the prover checks its n_inst / num_vars against it)
int32_t dgpu_r1cs_shape(uint64_t handle, size_t *num_vars, size_t *num_inputs, size_t *num_constraints) {
    HandleRef href(handle);
    if (!href.ok || href.h.kind != 4) return DGPU_E_BADARG;
    const DevR1cs *r = (const DevR1cs *)href.h.p;
    if (num_vars) *num_vars = r->num_vars;
    if (num_inputs) *num_inputs = r->num_inputs;
    if (num_constraints) *num_constraints = r->num_constraints;
    return DGPU_OK;
}
int32_t dgpu_r1cs_free(uint64_t handle) {
    Handle hd;
    if (!take_handle(handle, [](int k) { return k == 4; }, hd)) return DGPU_E_BADARG;     // waits for calls still using the circuit
    CtxScope on_owner(hd.ctx);
    if (cur().ready) (void)hipSetDevice(cur().device);
    free_r1cs((DevR1cs *)hd.p);
    return DGPU_OK;
}
int32_t dgpu_witness_map_r1cs(uint64_t r1cs, const uint64_t *assignment, size_t num_vars, int32_t montgomery, uint64_t *out_h, uint64_t *out_handle, size_t *out_len) {
    if (!assignment || (!out_h && !out_handle)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    HandleRef href(r1cs);
    if (!href.ok || href.h.kind != 4) return DGPU_E_BADARG;
    const DevR1cs *r = (const DevR1cs *)href.h.p;
    if (num_vars != r->num_vars) return DGPU_E_BADARG;
    CtxScope on_owner(href.h.ctx);
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    return witness_map_device(sl, *r, assignment, montgomery, out_h, out_handle, out_len);
}
// the assignment already resident (dgpu_scalars_upload: canonical after the upload): one upload of z serves the witness map and, at
// scalar offset 1, the prover's `assignment` = z[1..] (prover.rs:319-321)
int32_t dgpu_witness_map_r1cs_resident(uint64_t r1cs, uint64_t assignment, uint64_t *out_h, uint64_t *out_handle, size_t *out_len) {
    if (!out_h && !out_handle) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    HandleRef href(r1cs), hs(assignment);
    if (!href.ok || href.h.kind != 4 || !hs.ok || hs.h.kind != 3 || hs.h.ctx != href.h.ctx) return DGPU_E_BADARG;
    const DevR1cs *r = (const DevR1cs *)href.h.p;
    if (hs.h.n != r->num_vars) return DGPU_E_BADARG;
    CtxScope on_owner(href.h.ctx);
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    return witness_map_device(sl, *r, nullptr, 0, out_h, out_handle, out_len, (const uint32_t *)hs.h.p);
}
int32_t dgpu_witness_map(const uint64_t *a_rowptr, const uint32_t *a_cols, const uint64_t *a_vals, size_t a_nnz,
                         const uint64_t *b_rowptr, const uint32_t *b_cols, const uint64_t *b_vals, size_t b_nnz,
                         const uint64_t *c_rowptr, const uint32_t *c_cols, const uint64_t *c_vals, size_t c_nnz,
                         const uint64_t *assignment, size_t num_vars, size_t num_inputs, size_t num_constraints, int32_t montgomery,
                         uint64_t *out_h, uint64_t *out_handle, size_t *out_len) {
    if (!assignment || num_inputs > num_vars || (!out_h && !out_handle)) return DGPU_E_BADARG;
    if (!check_csr(a_rowptr, a_cols, a_vals, a_nnz, num_constraints, num_vars) || !check_csr(b_rowptr, b_cols, b_vals, b_nnz, num_constraints, num_vars) || !check_csr(c_rowptr, c_cols, c_vals, c_nnz, num_constraints, num_vars)) return DGPU_E_BADARG;
    if (!cur().ready) return DGPU_E_NODEVICE;
    SLOT_ACQUIRE(slot_lock, sl);
    HIPCHK(hipSetDevice(cur().device));
    Csr mats[3] = {{a_rowptr, a_cols, a_vals, a_nnz}, {b_rowptr, b_cols, b_vals, b_nnz}, {c_rowptr, c_cols, c_vals, c_nnz}};
    DevR1cs *r = nullptr;
    int32_t rc = build_r1cs(sl, mats, num_vars, num_inputs, num_constraints, montgomery, &r);
    if (rc) return rc;
    rc = witness_map_device(sl, *r, assignment, montgomery, out_h, out_handle, out_len);
    (void)hipStreamSynchronize(sl.stream);
    free_r1cs(r);
    return rc;
}

}  // extern "C"
