// Links libcomms_hip.so (built by comms_rs_amd/csrc/Makefile).  UNTESTED: no rustc here.
fn main() {
    let dir = std::env::var("COMMS_HIP_LIB_DIR").unwrap_or_else(|_| "../lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=comms_hip");
}
