// UNBUILT SOURCE.  Points the linker at the in-tree library (`make -C rustcrypto-elliptic-curves_amd`).
fn main() {
    let root = std::env::var("ECGPU_LIB_DIR").unwrap_or_else(|_| format!("{}/../../lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={root}");
    println!("cargo:rustc-link-lib=dylib=ecgpu");
    println!("cargo:rerun-if-env-changed=ECGPU_LIB_DIR");
}
