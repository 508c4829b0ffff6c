"""CPU: the compile-free checks this image allows on the Rust side of the boundary (no Rust toolchain here).

rust/dock_gpu/src/lib.rs declares the C ABI a second time (`extern "C"` block, `#[repr(C)]` structs): a drifted declaration there is a memory
error on the first call from Rust and nothing in this image would compile it.  So: every function of the `extern` block exists in
include/dock_gpu.h with the same arity and the same integer widths / pointer constness parameter by parameter, every `#[repr(C)]` struct has the
header's fields in the header's order and types, generic.rs only calls functions lib.rs defines, and the diffs of rust/patches still apply to the
reference tree (when /root/reference is there) and only name `dock_gpu::generic::` functions that exist, with the arity generic.rs gives them."""
import os
import re
import shutil
import subprocess
import tempfile
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "dock_gpu.h")).read()
LIB = open(os.path.join(ROOT, "rust", "dock_gpu", "src", "lib.rs")).read()
GEN = open(os.path.join(ROOT, "rust", "dock_gpu", "src", "generic.rs")).read()


import sys
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_parse as A                     # noqa: E402  (the header parser shared with tools/gen_rust_ffi.py)
import gen_rust_ffi                       # noqa: E402

FFI = open(os.path.join(ROOT, "rust", "dock_gpu", "src", "ffi.rs")).read()
HOST = open(os.path.join(ROOT, "rust", "dock_gpu", "src", "host.rs")).read()
PARITY = open(os.path.join(ROOT, "rust", "dock_gpu", "tests", "parity.rs")).read()
split_top, c_type_to_rust, header_functions, header_structs = A.split_top, A.c_type_to_rust, A.header_functions, A.header_structs


def rust_extern_functions():
    m = re.search(r'extern "C" \{(.*?)\n\}', FFI, flags=re.S)
    assert m, "no extern block in ffi.rs"
    fns = {}
    for f in re.finditer(r"pub fn (dgpu_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*([^;]+?))?\s*;", m.group(1), flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in split_top(" ".join(f.group(2).split())) if p]
        fns[f.group(1)] = (f.group(3) or "()", params)
    return fns


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\n?\}", FFI, flags=re.S):
        body = re.sub(r"//[^\n]*", "", m.group(2))
        fields = []
        for f in split_top(" ".join(body.split())):
            if not f:
                continue
            name, ty = f.split(":", 1)
            ty = ty.strip()
            fn = re.match(r"unsafe extern \"C\" fn\((.*)\)$", ty)
            if fn:
                ty = "fn(" + ", ".join(p.split(":", 1)[1].strip() for p in split_top(fn.group(1))) + ")"
            fields.append((name.replace("pub", "").strip(), ty))
        out[m.group(1)] = fields
    return out


def norm(t):
    return t.replace("core::ffi::c_void", "c_void").replace(" ", "")


def test_extern_block_matches_the_header_parameter_by_parameter():
    hdr, rs = header_functions(), rust_extern_functions()
    # EVERY entry point of the header is bound (round 6: the block is generated from the header, tools/gen_rust_ffi.py) ...
    assert set(rs) == set(hdr), "not bound in ffi.rs: %s; bound but not declared: %s" % (sorted(set(hdr) - set(rs)), sorted(set(rs) - set(hdr)))
    assert len(rs) >= 110
    ret_map = {"int32_t": "i32", "size_t": "usize", "uint64_t": "u64", "const char *": "*const core::ffi::c_char"}
    for name, (rret, rparams) in rs.items():
        cret, cparams = hdr[name]
        assert ret_map[cret] == rret, (name, cret, rret)
        assert len(cparams) == len(rparams), "%s: %d parameters in the header, %d in ffi.rs" % (name, len(cparams), len(rparams))
        for i, (c, r) in enumerate(zip(cparams, rparams)):
            want = c_type_to_rust(c)
            assert norm(want) == norm(r), "%s parameter %d: header `%s` is %s, ffi.rs has %s" % (name, i, " ".join(c.split()), want, r)


def test_ffi_rs_is_what_the_generator_writes_today():
    """ffi.rs is generated from include/dock_gpu.h: a header edit without `python tools/gen_rust_ffi.py` leaves the Rust side behind"""
    assert gen_rust_ffi.render() == FFI, "rust/dock_gpu/src/ffi.rs is stale: run python tools/gen_rust_ffi.py"
    # every integer constant of the header has its value on the Rust side (render() asserts the values while it runs)
    for name, _ in A.header_int_defines():
        assert re.search(r"pub const %s: \w+ = " % name, FFI), name


def test_repr_c_structs_have_the_headers_fields_in_order():
    hs, rs = header_structs(), rust_structs()
    assert set(hs) == {"dgpu_lego_pk", "dgpu_bases_view", "dgpu_lego_pk_host", "dgpu_transcript", "dgpu_snarkpack_prover_srs", "dgpu_snarkpack_verifier_srs", "dgpu_groth16_vk"}
    for c in hs:
        r = A.struct_rust_name(c)
        assert r in rs, (c, r)
        assert [n for n, _ in hs[c]] == [n for n, _ in rs[r]], "field order of %s / %s: %s vs %s" % (c, r, hs[c], rs[r])
        for (n, ct), (_, rt) in zip(hs[c], rs[r]):
            assert norm(ct) == norm(rt), "%s.%s: header %s, ffi.rs %s" % (c, n, ct, rt)


# entry points that stay raw FFI (ffi.rs) without a safe wrapper in lib.rs / host.rs, and why
RAW_ONLY = {
    "dgpu_shutdown": "process teardown: the host's own decision (nothing to wrap)",
    "dgpu_last_hip_error": "diagnostic integer",
    "dgpu_get_min_gpu_n": "diagnostic", "dgpu_set_min_gpu_n": "one integer: called directly (tests/parity.rs setup)",
    "dgpu_set_small_msm_max": "tuning switch kept for the comparison tests", "dgpu_device_alloc_count": "diagnostic counter (tests/test_gpu_reserve.py)",
    "dgpu_init_devices": "init_devices wraps dgpu_init_device_list (a device may repeat in a list, not in a mask)",
    "dgpu_msm_g1": "packed x|y arrays: a Rust host holds Affine structs and calls the _strided form", "dgpu_msm_g1_mont": "the same", "dgpu_msm_g2": "the same", "dgpu_msm_g2_mont": "the same",
    "dgpu_bases_upload_g1": "packed form of dgpu_bases_upload_g1_strided", "dgpu_bases_upload_g2": "the same",
    "dgpu_scalars_upload": "resident scalars are driven by dgpu_legogroth16_prove[_host] inside the library", "dgpu_scalars_upload_parts": "the same", "dgpu_scalars_copy_range": "the same (sharded prover)",
    "dgpu_bases_table_shape": "the shared sort of a proof is scheduled inside dgpu_legogroth16_prove", "dgpu_scalars_sort": "the same", "dgpu_msm_g1_sorted": "the same", "dgpu_msm_g2_sorted": "the same",
    "dgpu_msm_g1_resident": "both operands resident: the prover call's internals and bench.py's timed region", "dgpu_msm_g2_resident": "the same",
    "dgpu_msm_g2_sharded": "G2 one-shot sharded: ShardedG2 covers the resident form a prover uses", "dgpu_msm_g2_sharded_resident": "the same",
    "dgpu_fold_g2": "fold_g1's twin; the sharded G2 call folds inside the library", "dgpu_lincomb_g1": "O(1) group arithmetic the prover call does inside the library; a Rust host has arkworks for it", "dgpu_lincomb_g2": "the same",
    "dgpu_multi_miller_loop_prepared": "served by dgpu_multi_miller_loop_mixed with no affine pairs (multi_miller_loop_mixed)",
    "dgpu_multi_miller_loop_segments": "the aggregation's segmented loops run inside dgpu_snarkpack_aggregate / _verify", "dgpu_multi_pairing_segments": "the same",
    "dgpu_multi_miller_loop_sharded": "pairs over several GPUs: no reference call site holds enough pairs; kept raw",
    "dgpu_fp12_mul": "GT arithmetic of the checker: a Rust host keeps arkworks' PairingOutput for it", "dgpu_fp12_pow": "the same", "dgpu_fp12_multi_pow": "the same", "dgpu_gt_in_subgroup": "the same (Validate::Yes of a deserialised PairingOutput)",
    "dgpu_window_table_mul_to_bases_g2": "G2 twin of WindowTableG1::multiply_many_to_bases (CRS generator): raw",
    "dgpu_g1_mul_add_batch": "SnarkPack folding steps: inside dgpu_snarkpack_aggregate", "dgpu_g2_mul_add_batch": "the same", "dgpu_g1_fold_prepare": "the same", "dgpu_g2_fold_prepare": "the same",
    "dgpu_fold_prepare_pair": "the same", "dgpu_g1_fold_apply": "the same", "dgpu_g2_fold_apply": "the same", "dgpu_fold_free": "the same",
    "dgpu_witness_map": "one-shot form (matrices cross PCIe per call): host.rs keeps circuits resident by content hash and calls dgpu_witness_map_r1cs",
    "dgpu_witness_map_r1cs_resident": "z already resident: the prover call's internals", "dgpu_r1cs_shape": "R1cs keeps its shape on the Rust side",
    "dgpu_handle_len": "the wrappers keep lengths on the Rust side", "dgpu_handle_context": "diagnostic", "dgpu_shard_part": "layout of a sharded handle: diagnostic (ShardedG1::shards reads the count)",
}


def test_every_entry_point_has_a_safe_wrapper_or_a_reason():
    """include/dock_gpu.h -> ffi.rs (all of it) -> a safe wrapper in lib.rs / host.rs, or a line in RAW_ONLY saying why not"""
    wrapped = set(re.findall(r"\b(dgpu_[a-z0-9_]+)\s*\(", LIB + HOST))
    hdr = set(header_functions())
    assert wrapped <= hdr, "wrappers call entry points the header does not declare: %s" % sorted(wrapped - hdr)
    unexplained = hdr - wrapped - set(RAW_ONLY)
    assert not unexplained, "neither wrapped nor explained: %s" % sorted(unexplained)
    stale = set(RAW_ONLY) & wrapped
    assert not stale, "listed as raw-only but wrapped: %s" % sorted(stale)
    assert len(wrapped) >= 60


def test_parity_rs_has_a_case_for_every_wrapper():
    """tests/parity.rs (the cargo-side pin against arkworks) names every public function / type of lib.rs and host.rs that reaches the library"""
    names = set()
    for src in (LIB, HOST):
        for m in re.finditer(r"pub fn (\w+)\s*(?:<[^{;(]*?>)?\s*\(", src):
            names.add(m.group(1))
    # constructors / accessors are exercised through their types; helpers that never reach the library
    skip = {"new", "upload", "upload_csr", "handle", "raw", "shards", "fq12_from_words", "fq12_to_words", "pack_g1", "pack_g2", "flatten", "set_bytes", "set_min_n", "verify_samples",
            "verify_every_record", "invalidate", "clear", "stats", "multiply_many", "multiply_many_to_bases", "msm_bigint", "msm_resident", "upload_scalars", "witness_map",
            "device_count", "context_count", "set_device", "error_string"}
    assert "set_auto_shard_min_n" in PARITY
    types = ("ResidentG1", "GpuProvingKey", "GpuPreparedVerifyingKey", "GpuProverSrs", "ShardedG1", "ShardedG2", "WindowTableG1", "WindowTableG2", "R1cs", "HostProvingKey", "cache::")
    missing = sorted(n for n in names - skip if not re.search(r"\b%s\b" % n, PARITY))
    assert not missing, "tests/parity.rs has no case for: %s" % missing
    for t in types:
        assert t in PARITY, "tests/parity.rs never touches %s" % t
    for n in ("error_string", "device_count", "cache::stats", "cache::invalidate", "verify_every_record", "multiply_many_to_bases", "msm_resident"):
        assert n.split("::")[-1] in PARITY, n


def rust_fns(src):
    """name -> number of value parameters of every `pub fn` (free functions; `self` not counted)"""
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*(?:<[^{;]*?>)?\s*\(", src):
        i = m.end(); depth = 1; j = i
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0); j += 1
        params = [p for p in split_top(" ".join(src[i:j - 1].split())) if p and not re.match(r"&?(mut )?self$", p)]
        out[m.group(1)] = len(params)
    return out


def call_arity(src, start):
    depth, j = 1, start
    while depth:
        depth += {"(": 1, ")": -1}.get(src[j], 0); j += 1
    return len([p for p in split_top(" ".join(src[start:j - 1].split())) if p])


def test_generic_rs_calls_only_what_lib_rs_defines():
    assert "pub mod generic;" in LIB
    lib_fns = rust_fns(LIB)
    used = re.finditer(r"crate::(\w+)\s*\(", GEN)
    seen = 0
    for m in used:
        seen += 1
        assert m.group(1) in lib_fns, "generic.rs calls crate::%s, which lib.rs does not define" % m.group(1)
        assert call_arity(GEN, m.end()) == lib_fns[m.group(1)], "crate::%s called with %d arguments, defined with %d" % (m.group(1), call_arity(GEN, m.end()), lib_fns[m.group(1)])
    assert seen >= 8
    # the TypeId dispatch falls through to arkworks for every other curve: each generic entry point has such a tail
    for fn in ("msm_unchecked", "msm_bigint", "multi_miller_loop", "final_exponentiation", "g2_prepare"):
        body = GEN[GEN.index("pub fn %s<" % fn):]
        body = body[:body.index("\n}\n") + 3]
        assert re.search(r"\b(G::Group|E)::%s\(|E::G2Prepared::from" % fn, body), "generic::%s has no arkworks fall-through" % fn


PATCHES = sorted(f for f in os.listdir(os.path.join(ROOT, "rust", "patches")) if f.endswith(".diff"))


def test_patches_name_only_generic_functions_that_exist():
    gen_fns = rust_fns(GEN)
    assert len(PATCHES) >= 10
    n = 0
    for f in PATCHES:
        lines = open(os.path.join(ROOT, "rust", "patches", f)).read().split("\n")
        added = "\n".join(l[1:] for l in lines if l.startswith("+") and not l.startswith("+++"))
        for m in re.finditer(r"dock_gpu::generic::(\w+)(?:::<[^>]*>)?\s*\(", added):
            n += 1
            assert m.group(1) in gen_fns, "%s names generic::%s, which generic.rs does not define" % (f, m.group(1))
            assert call_arity(added, m.end()) == gen_fns[m.group(1)], (f, m.group(1), call_arity(added, m.end()), gen_fns[m.group(1)])
        if "cargo" in f:
            continue
        # the reference must compile unchanged without the feature: a patch REMOVES nothing, and every block of lines it adds starts with a
        # `#[cfg(feature = "gpu")]` / `#[cfg(not(feature = "gpu"))]` attribute, or is a new item that carries both arms inside
        assert '#[cfg(feature = "gpu")]' in added, f
        if [l for l in lines if l.startswith("-") and not l.startswith("---")]:
            # a patch that REPLACES a line carries the replaced text again under the other arm (0004: the line is re-indented into a cfg block)
            assert '#[cfg(not(feature = "gpu"))]' in added, "%s: the reference must compile unchanged without the feature" % f
            continue
        blocks, cur = [], []
        for l in lines:
            if l.startswith("+") and not l.startswith("+++"):
                cur.append(l[1:])
            elif cur:
                blocks.append(cur); cur = []
        if cur:
            blocks.append(cur)
        for b in blocks:
            code = [x.strip() for x in b if x.strip() and not x.strip().startswith("//")]
            if not code:
                continue
            both = any('#[cfg(feature = "gpu")]' in x for x in code) and any('#[cfg(not(feature = "gpu"))]' in x for x in code)
            assert code[0].startswith('#[cfg(feature = "gpu")]') or code[0].startswith('#[cfg(not(feature = "gpu"))]') or both, (f, code[:3])
    assert n >= 20


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils/src"), reason="the reference tree is only present in the build container")
def test_patches_apply_to_the_reference_tree():
    with tempfile.TemporaryDirectory() as tmp:
        for crate in ("utils", "legogroth16"):
            os.makedirs(os.path.join(tmp, crate, "src"))
            shutil.copy(os.path.join("/root/reference", crate, "Cargo.toml"), os.path.join(tmp, crate))
        for rel in ("utils/src/pairs.rs", "utils/src/owned_pairs.rs", "utils/src/randomized_mult_checker.rs", "utils/src/randomized_pairing_check.rs", "utils/src/msm.rs",
                    "legogroth16/src/prover.rs", "legogroth16/src/verifier.rs", "legogroth16/src/r1cs_to_qap.rs"):
            shutil.copy(os.path.join("/root/reference", rel), os.path.join(tmp, rel))
        for f in PATCHES:
            r = subprocess.run(["patch", "-p1", "-s", "--no-backup-if-mismatch", "-i", os.path.join(ROOT, "rust", "patches", f)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, (f, r.stdout, r.stderr)
        patched = open(os.path.join(tmp, "utils/src/pairs.rs")).read()
        assert "dock_gpu::generic::msm_unchecked(self.left, self.right)" in patched
        assert "dock_gpu::generic::witness_map_from_matrices::<F>(" in open(os.path.join(tmp, "legogroth16/src/r1cs_to_qap.rs")).read()
        assert "fn gpu_proof_with_reduction<E: Pairing>(" in open(os.path.join(tmp, "legogroth16/src/prover.rs")).read()
        assert "pub fn variable_base_msm<G: ark_ec::AffineRepr>(" in open(os.path.join(tmp, "utils/src/msm.rs")).read()
        # with the feature off nothing changed: stripping the gpu arms gives back the reference's text
        ref = open("/root/reference/utils/src/pairs.rs").read()
        stripped = re.sub(r'\s*#\[cfg\(feature = "gpu"\)\]\n[^\n]*\n', "\n", patched).replace('        #[cfg(not(feature = "gpu"))]\n', "")
        assert "".join(stripped.split()) == "".join(ref.split())
