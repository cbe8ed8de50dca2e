// Points the linker at libtriple_accel_amd.so (built by `make -C triple_accel_amd/csrc` in the repository root).
fn main() {
    let dir = std::env::var("TRIPLE_ACCEL_AMD_LIB_DIR").unwrap_or_else(|_| "../../triple_accel_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=triple_accel_amd");
    println!("cargo:rerun-if-env-changed=TRIPLE_ACCEL_AMD_LIB_DIR");
}
