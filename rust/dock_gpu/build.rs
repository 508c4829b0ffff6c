// rust/dock_gpu/build.rs — links libdock_gpu.so.  Same role as the reference's oblivious_transfer/build.rs:11-15 (the only FFI precedent in
// docknetwork/crypto: a `cc` build of c/transpose.c linked with #[link(name = "transpose")], src/util.rs:216-220); here the native code is a
// prebuilt shared library (hipcc is not something a cargo build should drive), found through DOCK_GPU_LIB_DIR.
fn main() {
    println!("cargo:rerun-if-env-changed=DOCK_GPU_LIB_DIR");
    let dir = std::env::var("DOCK_GPU_LIB_DIR").unwrap_or_else(|_| {
        // default: the in-tree build of this repository (crypto_amd/libdock_gpu.so, made by `python -c 'import __graft_entry__ as g; g.build()'`)
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../crypto_amd", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=dock_gpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
