//! Pins what this repository could not pin (SURVEY.md §8c "parity unpinned"): drop this file into getreu/stringsext as
//! `tests/pin_vectors.rs` (next to `pin_vectors.json`), add `serde_json = "1"` to `[dev-dependencies]` (serde is already a
//! dependency), and run `cargo test --test pin_vectors`.  It feeds every hand-derived sequence of stringsext-amd's
//! `tests/golden/decoder_vectors.py` to the REAL `encoding_rs` 0.8.34 the crate depends on (Cargo.toml:19) through the very call
//! the scanner makes (`decode_to_str_without_replacement`, src/finding_collection.rs:138-143) and asserts (result, read, written);
//! then it decodes the table cells the GPU library has from one source only.  A failure names the vector: it is a difference
//! between encoding_rs and the restatement in `oracle/sxo.c` / `csrc/sx_codec_core.hpp` (e.g. the +-2-byte question after an
//! unpaired high surrogate, the gb18030 pending digit, ISO-2022-JP's pending `$`), to be fixed there.  It could not be run where
//! it was written: no Rust toolchain in that image.
use encoding_rs::{DecoderResult, Encoding};
use serde_json::Value;

fn unhex(s: &str) -> Vec<u8> {
    (0..s.len()).step_by(2).map(|i| u8::from_str_radix(&s[i..i + 2], 16).unwrap()).collect()
}

#[test]
fn decoder_sequences_match_encoding_rs() {
    let v: Value = serde_json::from_str(include_str!("pin_vectors.json")).unwrap();
    let mut failures = Vec::new();
    for seq in v["decoder"].as_array().unwrap() {
        let (name, label) = (seq["name"].as_str().unwrap(), seq["encoding"].as_str().unwrap());
        let enc = Encoding::for_label(label.as_bytes()).unwrap_or_else(|| panic!("no encoding for label {label}"));
        let mut dec = enc.new_decoder_without_bom_handling(); // src/scanner.rs:76
        for (ci, call) in seq["calls"].as_array().unwrap().iter().enumerate() {
            let bytes = unhex(call["hex"].as_str().unwrap());
            let last = call["last"].as_bool().unwrap();
            let mut rest: &[u8] = &bytes;
            // the scanner's loop (src/finding_collection.rs:134-143,292-325): after Malformed the decoder is called again
            for (si, step) in call["steps"].as_array().unwrap().iter().enumerate() {
                let mut out = String::with_capacity(1024);
                let (res, read) = dec.decode_to_str_without_replacement(rest, &mut out, last);
                let r = match res { DecoderResult::InputEmpty => "E", DecoderResult::OutputFull => "F", DecoderResult::Malformed(_, _) => "M" };
                let want = (step[0].as_str().unwrap(), step[1].as_u64().unwrap() as usize, step[2].as_u64().unwrap() as usize);
                if (r, read, out.len()) != want {
                    failures.push(format!("{label}: {name}: call {ci} step {si}: encoding_rs gives ({r}, {read}, {}), the vector says {want:?}", out.len()));
                }
                rest = &rest[read..];
            }
        }
    }
    assert!(failures.is_empty(), "{} of the hand-derived sequences differ from encoding_rs:\n{}", failures.len(), failures.join("\n"));
}

#[test]
fn single_source_table_cells_match_encoding_rs() {
    let v: Value = serde_json::from_str(include_str!("pin_vectors.json")).unwrap();
    let mut failures = Vec::new();
    for cell in v["cells"].as_array().unwrap() {
        let (label, hex, text) = (cell[0].as_str().unwrap(), cell[1].as_str().unwrap(), cell[2].as_str().unwrap());
        let enc = Encoding::for_label(label.as_bytes()).unwrap();
        let (got, had_errors) = enc.decode_without_bom_handling(&unhex(hex));
        if had_errors || got != text {
            failures.push(format!("{label} {hex}: encoding_rs {:?}, the table {:?}", got, text));
        }
    }
    assert!(failures.is_empty(), "{} single-source cells differ:\n{}", failures.len(), failures[..failures.len().min(40)].join("\n"));
}
