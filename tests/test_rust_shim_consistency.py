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


def strip_c_comments(s):
    return re.sub(r"/\*.*?\*/", " ", s, flags=re.S)


def split_top(s, sep=","):
    """split at separators that are not inside parentheses / brackets / angle brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


C_INT = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "void": "core::ffi::c_void", "char": "u8"}


def c_type_to_rust(decl):
    """one C parameter / field declaration (name included) -> the Rust type that must stand in the binding"""
    d = decl.strip()
    is_array = bool(re.search(r"\[[^\]]*\]\s*$", d))
    d = re.sub(r"\[[^\]]*\]\s*$", "", d).strip()
    const = bool(re.search(r"\bconst\b", d))
    d = re.sub(r"\bconst\b", "", d)
    d = re.sub(r"\bstruct\b", "", d).strip()
    stars = d.count("*")
    d = d.replace("*", " ")
    toks = d.split()
    base = toks[0]
    if base in C_INT:
        rt = C_INT[base]
    else:
        assert base.startswith("dgpu_"), "unknown C type in %r" % decl
        rt = "".join(p.capitalize() for p in base.split("_"))          # dgpu_lego_pk -> DgpuLegoPk
    n_ptr = stars + (1 if is_array else 0)
    for _ in range(n_ptr):
        rt = ("*const " if const else "*mut ") + rt
    return rt


def header_functions():
    src = strip_c_comments(HDR)
    src = re.sub(r"typedef struct.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(int32_t|size_t|uint64_t|const char \*)\s*(dgpu_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [] if params.strip() in ("", "void") else split_top(params)
        fns[name] = (ret.strip(), ps)
    return fns


def header_structs():
    src = strip_c_comments(HDR)
    out = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            fp = re.match(r"(\w[\w\s]*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", stmt)       # function pointer member
            if fp:
                fields.append((fp.group(2), "fn(" + ", ".join(c_type_to_rust(p) for p in split_top(fp.group(3))) + ")"))
                continue
            # `const uint64_t *a, *b, *c` / `size_t n` / `uint64_t a, b`
            first = split_top(stmt)
            head = re.match(r"((?:const\s+)?\w+)\s*(.*)$", first[0])
            base = head.group(1)
            decls = [head.group(2)] + first[1:]
            for dcl in decls:
                name = re.sub(r"[\*\s]", "", dcl)
                fields.append((name, c_type_to_rust(base + " " + dcl)))
        out[m.group(1)] = fields
    return out


def rust_extern_functions():
    m = re.search(r'extern "C" \{(.*?)\n\}', LIB, flags=re.S)
    assert m, "no extern block in lib.rs"
    fns = {}
    for f in re.finditer(r"pub fn (dgpu_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([\w:]+))?\s*;", m.group(1), flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in split_top(" ".join(f.group(2).split())) if p]
        fns[f.group(1)] = (f.group(3) or "()", params)
    return fns


def rust_structs():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*pub struct (\w+)\s*\{(.*?)\n?\}", LIB, flags=re.S):
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
    assert len(rs) >= 25
    ret_map = {"int32_t": "i32", "size_t": "usize", "uint64_t": "u64"}
    for name, (rret, rparams) in rs.items():
        assert name in hdr, "lib.rs binds %s, which include/dock_gpu.h does not declare" % name
        cret, cparams = hdr[name]
        assert ret_map[cret] == rret, (name, cret, rret)
        assert len(cparams) == len(rparams), "%s: %d parameters in the header, %d in lib.rs" % (name, len(cparams), len(rparams))
        for i, (c, r) in enumerate(zip(cparams, rparams)):
            want = c_type_to_rust(c)
            # `const void *` carries the caller's own structs: lib.rs spells it *const core::ffi::c_void (or the `_` cast at the call site)
            assert norm(want) == norm(r), "%s parameter %d: header `%s` is %s, lib.rs has %s" % (name, i, " ".join(c.split()), want, r)


def test_repr_c_structs_have_the_headers_fields_in_order():
    hs, rs = header_structs(), rust_structs()
    pairs = {"dgpu_lego_pk": "DgpuLegoPk", "dgpu_transcript": "DgpuTranscript", "dgpu_snarkpack_prover_srs": "DgpuSnarkpackProverSrs",
             "dgpu_snarkpack_verifier_srs": "DgpuSnarkpackVerifierSrs", "dgpu_groth16_vk": "DgpuGroth16Vk"}
    for c, r in pairs.items():
        assert c in hs and r in rs, (c, r)
        assert [n for n, _ in hs[c]] == [n for n, _ in rs[r]], "field order of %s / %s: %s vs %s" % (c, r, hs[c], rs[r])
        for (n, ct), (_, rt) in zip(hs[c], rs[r]):
            assert norm(ct) == norm(rt), "%s.%s: header %s, lib.rs %s" % (c, n, ct, rt)


def test_every_struct_of_the_header_is_bound():
    assert set(header_structs()) == {"dgpu_lego_pk", "dgpu_transcript", "dgpu_snarkpack_prover_srs", "dgpu_snarkpack_verifier_srs", "dgpu_groth16_vk"}


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
    assert len(PATCHES) >= 8
    n = 0
    for f in PATCHES:
        added = "".join(l[1:] for l in open(os.path.join(ROOT, "rust", "patches", f)) if l.startswith("+") and not l.startswith("+++"))
        for m in re.finditer(r"dock_gpu::generic::(\w+)(?:::<[^>]*>)?\s*\(", added):
            n += 1
            assert m.group(1) in gen_fns, "%s names generic::%s, which generic.rs does not define" % (f, m.group(1))
            assert call_arity(added, m.end()) == gen_fns[m.group(1)], (f, m.group(1))
        if f.endswith(".rs.diff") or "cargo" not in f:
            assert '#[cfg(feature = "gpu")]' in added and '#[cfg(not(feature = "gpu"))]' in added, "%s: the reference must compile unchanged without the feature" % f
    assert n >= 15


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils/src"), reason="the reference tree is only present in the build container")
def test_patches_apply_to_the_reference_tree():
    with tempfile.TemporaryDirectory() as tmp:
        for crate in ("utils", "legogroth16"):
            os.makedirs(os.path.join(tmp, crate, "src"))
            shutil.copy(os.path.join("/root/reference", crate, "Cargo.toml"), os.path.join(tmp, crate))
        for rel in ("utils/src/pairs.rs", "utils/src/owned_pairs.rs", "utils/src/randomized_mult_checker.rs", "utils/src/randomized_pairing_check.rs",
                    "legogroth16/src/prover.rs", "legogroth16/src/verifier.rs"):
            shutil.copy(os.path.join("/root/reference", rel), os.path.join(tmp, rel))
        for f in PATCHES:
            r = subprocess.run(["patch", "-p1", "-s", "--no-backup-if-mismatch", "-i", os.path.join(ROOT, "rust", "patches", f)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 0, (f, r.stdout, r.stderr)
        patched = open(os.path.join(tmp, "utils/src/pairs.rs")).read()
        assert "dock_gpu::generic::msm_unchecked(self.left, self.right)" in patched
        # with the feature off nothing changed: stripping the gpu arms gives back the reference's text
        ref = open("/root/reference/utils/src/pairs.rs").read()
        stripped = re.sub(r'\s*#\[cfg\(feature = "gpu"\)\]\n[^\n]*\n', "\n", patched).replace('        #[cfg(not(feature = "gpu"))]\n', "")
        assert "".join(stripped.split()) == "".join(ref.split())
