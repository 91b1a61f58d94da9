// Links libtinyrt.so.  TINYRT_LIB_DIR = directory holding it (default: the in-tree build location).
fn main() {
    let dir = std::env::var("TINYRT_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../../tiny-raytracer_amd", manifest)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=tinyrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=TINYRT_LIB_DIR");
}
