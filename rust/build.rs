// build.rs -- links the prebuilt C-ABI library (libksched_hip.so, `make lib` in the ksched repository) when the crate is
// built with `--features ksched`.  Without the feature nothing is linked: `cargo test parity_dump` (rust/pin_parity.sh)
// runs on any box with cargo, no GPU and no library needed.
fn main() {
    println!("cargo:rerun-if-env-changed=KSCHED_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_KSCHED").is_none() {
        return;
    }
    let dir = std::env::var("KSCHED_LIB_DIR").expect("--features ksched: set KSCHED_LIB_DIR to the directory holding libksched_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ksched_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
