// Links libq3tts.so; Q3TTS_LIB_DIR = directory holding it (default: the in-tree build output).
fn main() {
    let dir = std::env::var("Q3TTS_LIB_DIR").unwrap_or_else(|_| "../qwen3_tts_rs_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=q3tts");
    println!("cargo:rerun-if-env-changed=Q3TTS_LIB_DIR");
}
