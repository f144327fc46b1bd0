// No device-code build step: the gfx950 kernels ship inside libdiffsol_hip.so (built by `python -m diffsol_amd.build`, hipcc --offload-arch=gfx950).
// DIFFSOL_HIP_LIB_DIR points at diffsol_amd/lib; the host-side library is only needed for the DiffSL front end (src/diffsl.rs).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=DIFFSOL_HIP_LIB_DIR");
    let dir = env::var("DIFFSOL_HIP_LIB_DIR").unwrap_or_else(|_| {
        let manifest = env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{manifest}/../../diffsol_amd/lib")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=diffsol_hip");
    println!("cargo:rustc-link-lib=dylib=diffsol_hip_host");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
