// Links librt_hip.so.  RT_HIP_LIB_DIR = the directory that holds it (rt_amd/ in this repository after
// `python -c "import __graft_entry__ as g; g.build()"`); the HIP runtime comes with it as a dependency.
fn main() {
    let dir = std::env::var("RT_HIP_LIB_DIR").expect("set RT_HIP_LIB_DIR to the directory of librt_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
}
