//! Tells rustc where `libbevy_mi355x.so` is: `BEVY_MI355X_LIB_DIR`, or `bevy_amd/` of this repository (where
//! `python -m bevy_amd.build` leaves it).
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var_os("BEVY_MI355X_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("../../bevy_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=BEVY_MI355X_LIB_DIR");
}
