// build.rs of image-rs/jpeg-decoder with the `hip` feature: where libjpgpu.so lives.
// JPGPU_LIB_DIR = directory holding libjpgpu.so (jpeg-decoder_amd/ in the jpeg-decoder_amd repository after
// `python -c "import __graft_entry__ as g; g.build()"` or `make -C jpeg-decoder_amd/csrc`).
fn main() {
    println!("cargo:rerun-if-env-changed=JPGPU_LIB_DIR");
    if std::env::var_os("CARGO_FEATURE_HIP").is_some() {
        let dir = std::env::var("JPGPU_LIB_DIR").unwrap_or_else(|_| String::from("/usr/local/lib"));
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-lib=dylib=jpgpu");
        // so that `cargo test --features hip` finds the library without LD_LIBRARY_PATH
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
}
