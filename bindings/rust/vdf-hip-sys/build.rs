// Where the two shared libraries live: VDF_AMD_LIB_DIR, or vdf_amd/ of a checkout three levels up.
fn main() {
    let dir = std::env::var("VDF_AMD_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../../vdf_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=VDF_AMD_LIB_DIR");
}
