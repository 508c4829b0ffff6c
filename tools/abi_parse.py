"""Parser of include/dock_gpu.h for the Rust side of the boundary: every function (return type, parameters) and every struct of the C ABI, and the
Rust type each C declaration must have in a binding.  Used by tools/gen_rust_ffi.py (which writes rust/dock_gpu/src/ffi.rs) and by
tests/test_rust_shim_consistency.py (which checks the committed ffi.rs against the header parameter by parameter)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR_PATH = os.path.join(ROOT, "include", "dock_gpu.h")

C_INT = {"int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "void": "core::ffi::c_void", "char": "core::ffi::c_char"}
RET = {"int32_t": "i32", "size_t": "usize", "uint64_t": "u64", "const char *": "*const core::ffi::c_char"}


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


def struct_rust_name(c_name):
    return "".join(p.capitalize() for p in c_name.split("_"))          # dgpu_lego_pk -> DgpuLegoPk


def c_type_to_rust(decl):
    """one C parameter / field declaration (name included) -> the Rust type that must stand in the binding"""
    d = decl.strip()
    is_array = bool(re.search(r"\[[^\]]*\]\s*$", d))
    d = re.sub(r"\[[^\]]*\]\s*$", "", d).strip()
    # `const T *const *p`: a pointer to const pointers to const T
    consts = len(re.findall(r"\bconst\b", d))
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
        rt = struct_rust_name(base)
    n_ptr = stars + (1 if is_array else 0)
    for _ in range(n_ptr):
        rt = ("*const " if consts else "*mut ") + rt
    return rt


def c_decl_name(decl):
    d = re.sub(r"\[[^\]]*\]\s*$", "", decl.strip())
    return re.findall(r"\w+", d)[-1]


def header_text():
    return open(HDR_PATH).read()


def header_functions(hdr=None):
    """name -> (C return type, [parameter declarations]) in the header's order"""
    src = strip_c_comments(hdr or header_text())
    src = re.sub(r"typedef struct.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(int32_t|size_t|uint64_t|const char \*)\s*(dgpu_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ps = [] if params.strip() in ("", "void") else split_top(params)
        fns[name] = (ret.strip(), ps)
    return fns


def header_structs(hdr=None, with_fn_params=False):
    """name -> [(field, rust type)]; a function-pointer member's type is `fn(T1, T2, ...)` (with_fn_params: `fn(name: T, ...)`)"""
    src = strip_c_comments(hdr or header_text())
    out = {}
    for m in re.finditer(r"typedef struct (\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            fp = re.match(r"(\w[\w\s]*?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", stmt)       # function pointer member
            if fp:
                ps = split_top(fp.group(3))
                inner = ", ".join((c_decl_name(p) + ": " if with_fn_params else "") + c_type_to_rust(p) for p in ps)
                fields.append((fp.group(2), "fn(" + inner + ")"))
                continue
            # `const uint64_t *a, *b, *c` / `size_t n` / `uint64_t a, b` / `dgpu_bases_view a, b`
            first = split_top(stmt)
            head = re.match(r"((?:const\s+)?\w+)\s*(.*)$", first[0])
            base = head.group(1)
            decls = [head.group(2)] + first[1:]
            for dcl in decls:
                name = re.sub(r"[\*\s]", "", dcl)
                fields.append((name, c_type_to_rust(base + " " + dcl)))
        out[m.group(1)] = fields
    return out


def header_int_defines(hdr=None):
    """#define NAME <integer expression of literals> -> (name, text) for the constants a binding repeats"""
    out = []
    for m in re.finditer(r"^#define\s+(DGPU_\w+)\s+(.+?)\s*(?:/\*.*)?$", hdr or header_text(), flags=re.M):
        out.append((m.group(1), m.group(2).strip()))
    return out
