/*
 * sxo.h — ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of the reference's per-Mission byte-stream scan
 * (getreu/stringsext v2.3.5).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (stringsext_amd/) never links, imports or calls it.
 *
 * What is restated (reference file:line):
 *   FindingCollection::from        src/finding_collection.rs:84-342
 *   ScannerState                   src/scanner.rs:40-89
 *   SplitStr::next                 src/helper.rs:206-433
 *   Utf8Filter bit tests           src/mission.rs:333-348
 *   Slicer grid                    src/input.rs:120-167
 *   merger loop / framing          src/main.rs:116-139
 *   Finding ordering / print       src/finding.rs:92-155
 * Third-party arithmetic restated from its published algorithm (the crate is
 * NOT in /root/reference): encoding_rs 0.8.34 (Cargo.toml:19,
 * Cargo.lock:147-150) — WHATWG Encoding Standard decoders for UTF-8,
 * UTF-16LE/BE, x-user-defined, single-byte tables, Big5 and EUC-JP, with the crate's
 * `decode_to_str_without_replacement` (read, written, Malformed) contract.
 *
 * Pin status: pinned against tests/functional/expected_output1/2/3 and the
 * in-file unit-test known answers of scanner.rs, finding_collection.rs,
 * helper.rs, main.rs (see tests/test_oracle_*.py).  UTF-16 surrogate
 * accounting, UTF-16BE output and the UTF-8 "un-read" rule have no golden
 * vector in the reference: those parts are "parity unpinned" (DESIGN.md §3),
 * and so is every legacy encoding (tables from ICU dumps, oracle/gen_tables.py).
 */
#ifndef SXO_H
#define SXO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SXO_ENC_X_USER_DEFINED = 0, SXO_ENC_UTF8 = 1, SXO_ENC_UTF16LE = 2, SXO_ENC_UTF16BE = 3,
       SXO_ENC_SINGLE_BYTE_BASE = 16 /* + table index, see sxo_single_byte_name */,
       SXO_ENC_BIG5 = 64, SXO_ENC_EUC_JP = 65, SXO_ENC_SHIFT_JIS = 66, SXO_ENC_EUC_KR = 67, SXO_ENC_GB18030 = 68, SXO_ENC_GBK = 69 /* the same decoder, another name */,
       SXO_ENC_REPLACEMENT = 70, SXO_ENC_ISO_2022_JP = 71 };

enum { SXO_BEFORE = 0, SXO_EXACT = 1, SXO_AFTER = 2 };

/* Fields of `Mission` read by the scan (src/mission.rs:382-421). */
typedef struct {
    uint8_t  mission_id;
    uint8_t  encoding;                 /* SXO_ENC_* */
    uint8_t  chars_min_nb;
    uint8_t  require_same_unicode_block;
    int16_t  grep_char;                /* -1 = None */
    uint8_t  print_encoding_as_ascii;
    uint8_t  _pad;
    uint32_t output_line_char_nb_max;
    uint64_t af_lo, af_hi;             /* Utf8Filter::af (u128) */
    uint64_t ubf;                      /* Utf8Filter::ubf */
    uint64_t counter_offset;
} sxo_mission;

typedef struct {
    uint64_t position;
    uint32_t s_off, s_len;             /* into sxo_fc.arena (UTF-8) */
    uint8_t  precision;                /* SXO_BEFORE / EXACT / AFTER */
    uint8_t  completes_previous;
    uint8_t  mission_id;
    int16_t  input_file_id;            /* -1 = None */
} sxo_finding;

/* One FindingCollection. Owned by the caller; release with sxo_fc_free. */
typedef struct {
    sxo_finding* v;
    size_t       n, cap;
    uint8_t*     arena;
    size_t       arena_len, arena_cap;
    uint64_t     first_byte_position;
    int          str_buf_overflow;
} sxo_fc;

typedef struct sxo_state sxo_state;

sxo_state* sxo_state_new(const sxo_mission* m);
void       sxo_state_free(sxo_state* s);
/* Carry accessors (scanner.rs:40-69), for the unit-test known answers. */
uint64_t       sxo_state_consumed_bytes(const sxo_state* s);
int            sxo_state_maybe_cut(const sxo_state* s);
const uint8_t* sxo_state_leftover(const sxo_state* s, size_t* len);

/* == FindingCollection::from(ss, input_file_id, input_buffer, is_last_input_buffer) */
void sxo_fc_init(sxo_fc* fc);
void sxo_fc_free(sxo_fc* fc);
int  sxo_scan_slice(sxo_state* s, int input_file_id, const uint8_t* buf, size_t len,
                    int is_last_input_buffer, sxo_fc* out);

/* SplitStr as a standalone iterator (helper.rs:169-433) for its own KATs. */
typedef struct {
    uint32_t s_off, s_len;
    uint8_t  completes_previous, is_maybe_cut, to_be_filtered_again, min_ok, grep_ok;
} sxo_split_result;
size_t sxo_split_str(const uint8_t* inp, size_t len, uint8_t chars_min_nb, int same_block,
                     int last_s_was_maybe_cut, int invalid_bytes_after_inp, uint64_t af_lo,
                     uint64_t af_hi, uint64_t ubf, int grep_char, size_t s_char_nb_max,
                     sxo_split_result* out, size_t out_cap);

/* == main::run(): whole CLI pass over in-memory "files"; returns the exact
 * bytes stdout would receive. radix: 0 = no -t, else 'x' | 'd' | 'o'.
 * flush_at_eof = 0 reproduces the CLI (input.rs:130-137 never reports the last
 * buffer); 1 passes is_last=true with the final non-empty slice. */
typedef struct { const uint8_t* data; size_t len; } sxo_file;
int sxo_run(const sxo_mission* ms, int n_missions, const sxo_file* files, int n_files, int radix,
            int no_metadata, int flush_at_eof, uint8_t** out, size_t* out_len);
void sxo_free(void* p);

/* Same pass, but findings are only counted (cpu_baseline timing leg). */
int sxo_run_count(const sxo_mission* ms, int n_missions, const sxo_file* files, int n_files,
                  uint64_t* n_findings, uint64_t* n_string_bytes);

/* Maximal runs of consecutive decoded characters that pass af/ubf, with no
 * malformed sequence in between, found by sequentially running the decoder
 * over buf (decoder fresh at buf[0]; `stream_parity` = (stream offset of
 * buf[0]) & 1 for UTF-16).  Ignores -g and -r.  Reports runs with
 * >= min_chars chars as (first byte, one-past-last byte, chars). */
typedef struct { uint64_t start, end; uint64_t chars; } sxo_run_rec;
size_t sxo_runs(const sxo_mission* m, const uint8_t* buf, size_t len, int stream_parity,
                uint64_t min_chars, sxo_run_rec* out, size_t out_cap);

const char* sxo_encoding_name(int encoding);

/* One decoder on its own: decode_to_str_without_replacement(src, dst, last) -> result (0 InputEmpty, 1 OutputFull,
 * 2 Malformed), bytes read, bytes written.  State persists between steps like encoding_rs' Decoder. */
void* sxo_decoder_new(int encoding);
int   sxo_decoder_step(void* d, const uint8_t* src, size_t n, uint8_t* dst, size_t cap, int last, size_t* read, size_t* written);
void  sxo_decoder_free(void* d);

/* Deterministic synthetic background. */
void sxo_fill_background(uint8_t* dst, uint64_t first_byte_index, size_t len, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
