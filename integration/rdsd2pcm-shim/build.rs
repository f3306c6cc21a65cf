// D2D_LIB_DIR = the directory that holds libdsd2dxd_amd.so (…/dsd2dxd_amd after `make -C dsd2dxd_amd/csrc`)
fn main() {
    let dir = std::env::var("D2D_LIB_DIR").expect("set D2D_LIB_DIR to the directory of libdsd2dxd_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=dsd2dxd_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=D2D_LIB_DIR");
}
