// build.rs of the `local_features` crate once it links the MI355X path (new file in the reference tree:
// local_features/build.rs).  liblf_mkd.so is built outside cargo (`make -C local-features_amd`, hipcc
// --offload-arch=gfx950); LF_MKD_LIB_DIR names the directory that holds it.
fn main() {
    println!("cargo:rerun-if-env-changed=LF_MKD_LIB_DIR");
    let dir = std::env::var("LF_MKD_LIB_DIR")
        .expect("set LF_MKD_LIB_DIR to the directory holding liblf_mkd.so (make -C local-features_amd)");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=lf_mkd");
    // src/hip/device.rs stages host arrays with hipMalloc / hipMemcpy (the runtime liblf_mkd.so itself links)
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    // the loader must find the library at run time as well
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
