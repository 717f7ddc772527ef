/*
 * stringsext_amd.h — C-ABI of the MI355X-native replacement for stringsext's
 * per-Mission byte-stream scan.
 *
 * The reference (getreu/stringsext v2.3.5, Rust) has no FFI layer; its seam is
 *
 *     FindingCollection::from(ss: &mut ScannerState, input_file_id: Option<u8>,
 *                             input_buffer: &[u8], is_last_input_buffer: bool)
 *         -> Pin<Box<FindingCollection>>                 src/finding_collection.rs:84-89
 *
 * called once per Mission per 4096-byte slice from the loop in
 * src/main.rs:153-168 and drained by the merger in src/main.rs:118-136.  A
 * 4 KiB call is useless for a GPU, so this library replaces that LOOP for one
 * large chunk of one input file: sx_scan() returns exactly the findings the
 * loop would have pushed to the merger for the chunk's slices, already in the
 * merger's order, and carries the same state between calls that ScannerState
 * carries between slices (src/scanner.rs:40-69).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success or a negative SX_E_* code, with text from sx_last_error(); the
 * caller owns inputs, the library owns sx_result until sx_result_free(); a
 * context is bound to ONE HIP device and is not thread-safe; internally the
 * Missions are scanned by ONE fused kernel that reads the buffer once where
 * their classifiers allow (round 6, csrc/sx_fused.hip; else their scan kernels
 * queue up in one HIP stream) and everything after them
 * (records -> runs, the exact replay, copies) runs in a second one
 * (SX_OPT_MISSION_STREAMS: a scan stream per Mission, the reference's one
 * thread per Mission, src/main.rs:97,151).  There is no CPU fallback: without
 * a HIP device sx_create() fails with SX_E_NO_DEVICE.
 */
#ifndef STRINGSEXT_AMD_H
#define STRINGSEXT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct of this header changes size or layout or an enum value changes meaning (2: sx_stats grew by the wave /
 * re-scan / piece fields in round 3, SX_ENC_ISO_2022_JP was added; packed findings, round 4; 4: sx_stats grew by the fused scan's fields,
 * round 6).  A consumer compares it with
 * sx_abi_version() before it hands the library a struct to fill. */
#define SX_ABI_VERSION 4

enum {
    SX_OK = 0,
    SX_E_INVALID = -1,     /* bad argument / unsupported mission */
    SX_E_NO_DEVICE = -2,   /* no usable HIP device: the product never runs on the CPU */
    SX_E_HIP = -3,         /* a HIP call failed */
    SX_E_NOMEM = -4,
    SX_E_STATE = -5,       /* call not valid for this context (e.g. device scan on a host-only ctx) */
    SX_E_HALO = -6         /* sx_scan_shard*: the buffer must begin further in front of own_lo (Big5 / EUC-JP: no byte outside
                              the lead range lies between the buffer start and own_lo, so its token grid is unknown) */
};

/* Encoding ids == `Encoding::name()` of encoding_rs as used at src/mission.rs:681,
 * src/finding.rs:147.  "ascii" is x-user-defined + print_encoding_as_ascii
 * (src/mission.rs:675-679). */
enum {
    SX_ENC_X_USER_DEFINED = 0, SX_ENC_UTF8 = 1, SX_ENC_UTF16LE = 2, SX_ENC_UTF16BE = 3,
    SX_ENC_KOI8_R = 16, SX_ENC_IBM866 = 17, SX_ENC_ISO_8859_2 = 18, SX_ENC_ISO_8859_5 = 19,
    SX_ENC_ISO_8859_15 = 20, SX_ENC_WINDOWS_1251 = 21, SX_ENC_WINDOWS_1252 = 22,
    /* the rest of the WHATWG single-byte set (tables: csrc/gen_tables.py; parity unpinned) */
    SX_ENC_ISO_8859_3 = 23, SX_ENC_ISO_8859_4 = 24, SX_ENC_ISO_8859_6 = 25, SX_ENC_ISO_8859_7 = 26,
    SX_ENC_ISO_8859_8 = 27, SX_ENC_ISO_8859_8_I = 28, SX_ENC_ISO_8859_10 = 29,
    SX_ENC_ISO_8859_13 = 30, SX_ENC_ISO_8859_14 = 31, SX_ENC_ISO_8859_16 = 32, SX_ENC_KOI8_U = 33,
    SX_ENC_MACINTOSH = 34, SX_ENC_WINDOWS_874 = 35, SX_ENC_WINDOWS_1250 = 36,
    SX_ENC_WINDOWS_1253 = 37, SX_ENC_WINDOWS_1254 = 38, SX_ENC_WINDOWS_1255 = 39,
    SX_ENC_WINDOWS_1256 = 40, SX_ENC_WINDOWS_1257 = 41, SX_ENC_WINDOWS_1258 = 42,
    SX_ENC_X_MAC_CYRILLIC = 43,
    /* legacy multi-byte (tables: csrc/gen_tables.py; parity unpinned): a pending lead byte is the decoder state */
    SX_ENC_BIG5 = 64, SX_ENC_EUC_JP = 65, SX_ENC_SHIFT_JIS = 66, SX_ENC_EUC_KR = 67,
    SX_ENC_GB18030 = 68, SX_ENC_GBK = 69,   /* one decoder (four-byte tokens too), two names */
    /* the "replacement" encoding (ISO-2022-KR, ISO-2022-CN, HZ-GB-2312): its decoder reports one error and nothing else */
    SX_ENC_REPLACEMENT = 70,
    /* ISO-2022-JP: escape sequences select the character set, so the meaning of a byte depends on unbounded history: a Mission
     * with it is ONE sequential pass on the host (never on the device, never sharded) */
    SX_ENC_ISO_2022_JP = 71
};

/* `Precision` — src/finding.rs:34-46 */
enum { SX_PRECISION_BEFORE = 0, SX_PRECISION_EXACT = 1, SX_PRECISION_AFTER = 2 };

/* The fields of `Mission` the scan reads — src/mission.rs:382-421; `filter` is
 * `Utf8Filter{af: u128, ubf: u64, grep_char: Option<u8>}` (src/mission.rs:307-327). */
typedef struct sx_mission {
    uint8_t  mission_id;
    uint8_t  encoding;                   /* SX_ENC_* */
    uint8_t  chars_min_nb;
    uint8_t  require_same_unicode_block;
    int16_t  grep_char;                  /* -1 = None */
    uint8_t  print_encoding_as_ascii;
    uint8_t  reserved;
    uint32_t output_line_char_nb_max;
    uint64_t af_lo, af_hi;
    uint64_t ubf;
    uint64_t counter_offset;
} sx_mission;

/* `Finding` — src/finding.rs:51-74 (`s` = arena[str_off .. str_off+str_len], UTF-8). */
typedef struct sx_finding {
    uint64_t position;
    uint32_t str_off, str_len;
    uint8_t  precision;
    uint8_t  completes_previous;         /* s_completes_previous_s */
    uint8_t  mission_id;
    uint8_t  reserved;
    int16_t  input_file_id;              /* Option<u8>: -1 = None */
    uint16_t reserved2;
    uint32_t slice_index;                /* 4 KiB slice of this chunk that produced it */
} sx_finding;

/* The same finding in 16 bytes, as string-dense results cross PCIe (round 4): `-e ascii -n 4` on a binary yields a finding per 85
 * bytes, BASELINE config 5 411 M of them per 64 GiB — such scans are bound by moving the records to the host, and half of sx_finding
 * is the same for every record of a segment (input_file_id), follows from the others (slice_index = slice_base + (position -
 * position of the segment's buffer byte 0 for that Mission) / 4096) or is padding.  Segments whose findings were written by the
 * device's dense paths (the wave-cooperative stage B of a single Mission, the device-side merger of several) are stored this way;
 * sx_result_segment_packed() hands them out as they are, sx_result_segment() expands them to sx_finding on first use, and
 * sx_print_findings() reads either. */
typedef struct sx_finding16 {
    uint64_t position;
    uint32_t str_off;
    uint16_t str_len;
    uint8_t  flags;                      /* bits 0-1: precision (SX_PRECISION_*), bit 2: completes_previous */
    uint8_t  mission_id;
} sx_finding16;
typedef struct sx_segment_info {         /* what the records of one packed segment share */
    int32_t  packed;                     /* 1: the segment's records are sx_finding16, 0: sx_finding */
    int32_t  input_file_id;
    uint32_t slice_base;                 /* slice_index of the segment's buffer byte 0 */
    uint32_t reserved;
    uint64_t position0[256];             /* by mission_id: `position` of the segment's buffer byte 0 (counter_offset + bytes consumed before it) */
} sx_segment_info;

/* Device run record: one maximal stretch of bytes belonging to valid,
 * filter-accepted characters (ignoring -g / -r), with its character count.
 * (Inside the library a run that crosses window starts may travel as several pieces, one per window: a piece
 * that begins at a window start has bit 63 of `chars` set and, in the low bits, its distance from the run's start.
 * sx_device_runs never returns pieces; sx_replay_runs accepts them.) */
typedef struct sx_run {
    uint64_t start;                      /* byte offset of the first byte, chunk relative */
    uint64_t end;                        /* one past the last byte */
    uint64_t chars;
} sx_run;

typedef struct sx_stats {
    uint64_t bytes_scanned;              /* input bytes x missions examined on the device */
    uint64_t run_records;                /* long-run records the device reported (all missions) */
    uint64_t replay_bytes;               /* input bytes the host replayed (all missions) */
    uint64_t findings;
    double   kernel_ms[16];              /* per mission: device scan kernel, HIP events on its stream */
    double   device_ms;                  /* all mission streams, first launch -> last completion */
    double   h2d_ms, d2h_ms, replay_ms, total_ms;
    uint64_t heavy_tiles;                /* 1 KiB tiles that needed the general cross-lane path (all missions) */
    uint64_t wave_windows;               /* decoder-input windows replayed by the wave-cooperative stage B (all missions) */
    double   wave_count_ms, wave_write_ms; /* ... its two passes, HIP events around their launches (all missions and slabs) */
    uint64_t rescans;                    /* scan kernels launched a second time (their records overflowed the regions / the pool) */
    double   rescan_ms;                  /* ... host time until their records were there */
    uint64_t wave_desc_overflows;        /* wave stage B: slabs written by the window-parallel writer because a wavefront found more than its descriptors hold */
    uint64_t seq_pieces;                 /* pieces a buffer with gigabytes of output was scanned in, one after the other (0: in one go) */
    uint64_t fast_regions;               /* (ABI 3) stage B, lane per region: regions settled by the fast pre-pass (one run inside one window) ... */
    uint64_t general_regions;            /* ... and regions it left to the general replay kernel */
    uint64_t wave_repairs;               /* (ABI 3) wave stage B: count launches repeated for wavefronts whose warm-up windows gave them a wrong entry state (-g) */
    double   fused_ms;                   /* (ABI 4) fused scan launches (one read of the buffer for several Missions), HIP events around them; kernel_ms[k] of
                                            every Mission such a launch scanned holds the same duration */
    uint64_t fused_launches;             /* (ABI 4) ... how many */
    uint64_t fused_mask;                 /* (ABI 4) ... bit k: Mission k's last buffer was scanned by a fused launch */
} sx_stats;

typedef struct sx_ctx sx_ctx;
typedef struct sx_result sx_result;

/* Tunables (0 = library default). */
typedef struct sx_options {
    uint32_t subchunk_bytes;             /* bytes one wavefront streams sequentially (multiple of 1024) */
    uint32_t record_capacity;            /* device run-record slots per mission */
    uint32_t replay_threads;             /* host threads for the exact replay */
    uint32_t flags;                      /* SX_OPT_* */
} sx_options;
enum {
    SX_OPT_GENERIC_KERNELS = 1u,  /* force the table-driven classifiers (testing) */
    SX_OPT_DEVICE_REPLAY = 2u,    /* run the exact replay (stage B) on the device even for small inputs */
    SX_OPT_HOST_REPLAY = 4u,      /* never run stage B on the device */
    SX_OPT_TILE_TRAVERSAL = 8u,   /* (rounds 1-5: scan kernels over independent overlapping tiles, grid-stride — measured slower in every round
                                     and removed in round 6; the flag is accepted and ignored) */
    SX_OPT_MISSION_STREAMS = 16u, /* a scan stream per mission (default: one scan stream + one for everything else) */
    SX_OPT_NO_FUSED_SCAN = 64u,   /* (round 6) one scan launch per Mission, each reading the whole buffer (rounds 1-5), instead of ONE launch that
                                     reads it once for all Missions whose classifiers the fused kernel holds (csrc/sx_fused.hip) */
    SX_OPT_RESULT_ON_DEVICE = 32u /* (round 5) a context with ONE Mission: a buffer's result that the device wrote in one block — a string-dense
                                     buffer (the wave path: text, `-e ascii -n 4` on binaries, where moving the findings to the host is what bounds
                                     the scan: sx_finding16 records) or a sparse one replayed on the device (sx_finding records) — stays in HBM:
                                     sx_result_segment_device() hands out device pointers to its records and strings, for hosts that go on
                                     working there.  Valid until the NEXT buffer is scanned on the context — every chunk of sx_scan_stream / sx_scan_file is one; a
                                     call that accumulates its chunks into one result keeps them in host memory — or the context is destroyed (the memory
                                     is the context's).  The
                                     host accessors (sx_result_segment, ..._packed, sx_print_findings, ...) still work: the first one copies the
                                     segment to the host (SX_E_STATE if a later scan has overwritten it).  Every other result is in host memory
                                     as without the flag; the sharded entry points ignore it. */
};

/* ---- Mission front end (src/mission.rs:448-749, src/options.rs:12-33) ------------------------------
 * From the reference's option strings to sx_mission[]: per-encoding overrides
 * (`-e ENC[,MIN[,AF[,UBF[,GREP]]]]`), global flags, defaults, alias prefix matching, WHATWG encoding
 * labels, the reference's error texts.  Pointers may be NULL (= flag not given). */
typedef struct sx_cli_flags {
    const char* counter_offset;          /* -s */
    const char* const* encodings;        /* -e, in command-line order */
    int n_encodings;
    const char* chars_min;               /* -n */
    int same_unicode_block;              /* -r */
    const char* ascii_filter;            /* -a */
    const char* unicode_block_filter;    /* -u */
    const char* grep_char;               /* -g */
    const char* output_line_len;         /* -q */
} sx_cli_flags;
typedef struct sx_enc_opt {              /* Missions::parse_enc_opt's tuple of Options */
    int has_name; char name[64];
    int has_chars_min; uint8_t chars_min;
    int has_af; uint64_t af_lo, af_hi;
    int has_ubf; uint64_t ubf;
    int has_grep_char; uint8_t grep_char;
} sx_enc_opt;
int sx_missions_from_flags(const sx_cli_flags* flags, sx_mission* out, int cap, int* n_out, char* err, size_t err_cap);
int sx_parse_enc_opt(const char* enc_opt, sx_enc_opt* out, char* err, size_t err_cap);
int sx_encoding_for_label(const char* label);      /* SX_ENC_*; -1 not a label; -2 a label of an encoding not built in */
const char* sx_encoding_name(uint32_t encoding);   /* Encoding::name(), e.g. "UTF-16LE" */
/* The decoder table of a legacy encoding as uint16_t words (single byte: 128 code points for 0x80..0xFF;
 * Big5 / EUC-JP: the index blob, layout in csrc/sx_codec_core.hpp); NULL, *n_words = 0 if the encoding has none. */
const uint16_t* sx_decoder_table(uint32_t encoding, uint64_t* n_words);

/* Lower level: does the wave-cooperative stage B (csrc/sx_wave_core.hpp) cover this Mission — no -g, no -r (unless at most one UTF-8
 * lead byte passes the filter: then -r never breaks a string),
 * 1 <= chars_min_nb <= output_line_char_nb_max <= 64, a single-byte encoding or UTF-8 (the reference's rules it
 * relies on: src/helper.rs:315-322, 349-421)?  Returns 0 if not covered, < 0 on error, else the class byte it keeps per
 * input byte in classes[256] and 1 + family: 1 for a single-byte encoding (bit 0 a character, bit 1 its UTF-8 lead byte passes
 * af / ubf — src/mission.rs:333-348 —, bit 2 / 3 its UTF-8 form has 2 / 3 bytes), 2 for UTF-8 (bits 0-2: 0 never valid,
 * 1 ASCII, 2 continuation byte, 3 / 4 / 5 lead byte of 2 / 3 / 4; bit 3 a character that starts with it passes), 3 for
 * UTF-16LE / BE — then classes[] must hold 512 bytes: [hb] per high byte of a unit: bits 0-3 the low byte's quadrants (lo >> 6)
 * whose characters pass, bit 4 hb == 0 (then [256 + lo] bit 0 says whether U+00lo passes), bit 5 / 6 a high / low surrogate
 * (bits 0-3 of a high surrogate: the astral character it begins passes). */
int sx_wave_classes(const sx_mission* mission, uint8_t* classes);
/* ... and for the two-byte family (sx_wave_classes returns 5: Big5 — only if the Mission rejects U+00C0.. and U+0300.. —,
 * Shift_JIS, EUC-KR; classes[]: as a single-byte encoding's for the bytes that are characters on their own, bit 4 = lead byte):
 * 4 bits per byte pair, index lead | trail << 8, eight per word: bit 0 the index maps the pair, bit 1 its character passes the
 * filter, bits 2-3: its UTF-8 form has 2 / 3 / 4 bytes, 3 = it yields two code points.  out8192 or NULL. */
const uint32_t* sx_wave_pair_codes(const sx_mission* mission, uint32_t* out8192);

/* ... and the same classes as SWAR ranges, if the Mission's can be put that way (csrc/sx_device.hpp WvSwar, 26 words): what the wave
 * kernels classify with then.  Returns 1 and fills out26, 0 if the Mission's classes stay a table, < 0 on error.  (Test harness.) */
int sx_wave_swar(const sx_mission* mission, uint32_t* out26);
/* ... two-byte family with sx_wave_swar() == 1: 2 bits per byte pair (bit 0 mapped, bit 1 accepted), index lead | trail << 8, sixteen per word.
 * EUC-JP (family 5): the first 1105 words, 2 bits per cell of index jis0208 (cells 0 .. 8835: (lead - A1) * 94 + trail - A1) and of index jis0212 (from cell 8836). */
const uint32_t* sx_wave_pair_codes2(const sx_mission* mission, uint32_t* out4096);

/* Lower level: which classifier stage A runs for this Mission (csrc/sx_device.hpp ClassifierKind: 0-2 the table kernels, 3 / 4 the
 * two-byte range kernels of the default filters, 5 / 8 single-byte ranges, 6 / 7 the double-byte token classifiers, 9 / 10 the range
 * kernels for filters with three-byte leads / surrogate pairs, csrc/sx_classify_ranges.hpp) and its range parameters: out20 =
 * a_lo, a_hi, u_lo, u_hi, l3_lo, l3_hi, n_ranges, rng_c1[6], rng_c2[6], 0.  generic != 0: as with SX_OPT_GENERIC_KERNELS.
 * Returns the kind (>= 0) or an error (< 0).  (Test harness: a filter that silently fell back to a table kernel would be a
 * performance regression no parity test sees.) */
int sx_scan_classifier(const sx_mission* mission, int generic, uint32_t* out20);

int  sx_abi_version(void);

/* hip_device >= 0: bind to that device.  hip_device == SX_HOST_ONLY: a context
 * without device that supports ONLY sx_replay_runs() (used when run records
 * come from elsewhere: other ranks, tests). */
#define SX_HOST_ONLY (-1)
int  sx_create(sx_ctx** out, const sx_mission* missions, int n_missions, int hip_device,
               const sx_options* opt);
void sx_destroy(sx_ctx* ctx);
const char* sx_last_error(const sx_ctx* ctx); /* ctx may be NULL: last sx_create failure */

/* Replaces the loop src/main.rs:153-168 for `len` bytes of ONE input file.
 * The chunk starts on the reference's slice grid (a multiple of 4096 bytes
 * into the file, src/input.rs:22,121-123); every chunk except the last one of
 * a file is a multiple of 4096 long.  `is_last_input_buffer` reaches the final
 * slice of the chunk exactly as the Slicer's third tuple member would
 * (src/input.rs:118,166) — the reference CLI always passes false.
 * sx_scan: bytes in host memory (copied to HBM first);
 * sx_scan_device: bytes already resident in HBM on this context's device. */
int sx_scan(sx_ctx* ctx, const uint8_t* bytes, uint64_t len, int input_file_id,
            int is_last_input_buffer, sx_result** out);
int sx_scan_device(sx_ctx* ctx, const void* device_bytes, uint64_t len, int input_file_id,
                   int is_last_input_buffer, sx_result** out);

/* Ingest pipeline (the Slicer's job, src/input.rs:57-167, for chunks instead of 4 KiB slices):
 * a reader thread fills pinned buffers from `read` (returns bytes read, 0 at the end of the
 * input, < 0 on error; short reads are fine) and copies them to HBM while the chunk before is
 * scanned.  Every chunk of `chunk_bytes` (rounded to the 4096-byte grid; 0 = 256 MiB) gives
 * one result, handed to `sink` in input order; the sink owns it (sx_result_free) and returns
 * 0 to go on.  sx_scan_file reads a file with read(2) ("-" = stdin). */
typedef int64_t (*sx_read_fn)(void* user, uint8_t* dst, uint64_t max_bytes);
typedef int (*sx_result_fn)(void* user, sx_result* result);
int sx_scan_stream(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                   sx_result_fn sink, void* sink_user);
int sx_scan_file(sx_ctx* ctx, const char* path, uint64_t chunk_bytes, int input_file_id,
                 sx_result_fn sink, void* sink_user);

/* Reset the carried ScannerState of every mission (scanner.rs:73-88). */
int sx_reset(sx_ctx* ctx);

/* Lower level, stage A: device long-run records of one mission for a
 * device-resident buffer (stream_parity = (stream offset of byte 0) & 1).
 * Returns all maximal runs with >= min_chars characters, sorted by start.
 * *runs is malloc'd; release with sx_free(). */
int sx_device_runs(sx_ctx* ctx, int mission_index, const void* device_bytes, uint64_t len,
                   int stream_parity, uint64_t min_chars, sx_run** runs, uint64_t* n_runs);

/* (ABI 4) The same for n Missions of the context in one call (mission_indices[i], min_chars[i] -> runs[i], n_runs[i]; every runs[i]
 * is malloc'd: sx_free()).  Missions whose classifiers the fused kernel holds (csrc/sx_fused.hip) share ONE launch that reads the
 * buffer once — what sx_scan* does for them; sx_get_stats().fused_mask says which did. */
int sx_device_runs_multi(sx_ctx* ctx, const int* mission_indices, int n, const void* device_bytes, uint64_t len,
                         int stream_parity, const uint64_t* min_chars, sx_run** runs, uint64_t* n_runs);

/* Lower level, stage B: exact replay on the host given run records per
 * mission (runs[m] sorted by start, chunk-relative).  Works on SX_HOST_ONLY
 * contexts; same carry semantics and result as sx_scan. */
int sx_replay_runs(sx_ctx* ctx, const uint8_t* bytes, uint64_t len, int input_file_id,
                   int is_last_input_buffer, const sx_run* const* runs, const uint64_t* n_runs,
                   sx_result** out);

/* Byte-range sharding of ONE input file over several contexts — one process per GPU
 * (multi-GPU row of the scope table).  The buffer holds file bytes
 * [buf_off, buf_off+buf_len), buf_off a multiple of 4096 (the slice grid is the file's);
 * it should reach `halo` bytes beyond [own_lo, own_hi) on both sides where the file does.
 * The call returns the findings of every replay region that BEGINS in
 * [max(own_lo, start_at[m]), own_hi) for mission m, following the last such region to its
 * own end even beyond own_hi, and reports in end_pos[m] (file offset) where mission m's
 * replay stopped: the next rank's start_at[m].  start_at == NULL means own_lo for every
 * mission (first attempt; a rank must repeat the call with reuse_runs=1 if the previous
 * rank's end_pos turns out to lie beyond own_lo).  If end_pos[m] == buf_off+buf_len although
 * the file goes on, the buffer was too short for a run that crosses it: repeat with a
 * larger halo.  A Big5 / EUC-JP mission needs a byte outside the lead range between the buffer start and own_lo
 * (buf_off > 0): without one the call fails with SX_E_HALO — repeat with a larger halo in front.  A mission with
 * chars_min_nb 0 cannot be sharded (SX_E_INVALID).  Positions are counter_offset + file_stream_off + file offset.  The context's
 * carried state is used only by the shard that starts the file (buf_off == own_lo == 0) and
 * updated only by the shard whose own_hi is the buffer end.
 *   sx_scan_shard_device: bytes resident in HBM;  sx_scan_shard: host bytes (uploaded);
 *   sx_replay_shard_runs: stage B only, run records (buffer relative) supplied by the caller. */
int sx_scan_shard_device(sx_ctx* ctx, const void* device_bytes, uint64_t buf_off, uint64_t buf_len,
                         uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at,
                         uint64_t file_stream_off, int input_file_id, int reuse_runs,
                         sx_result** out, uint64_t* end_pos);
int sx_scan_shard(sx_ctx* ctx, const uint8_t* bytes, uint64_t buf_off, uint64_t buf_len,
                  uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at,
                  uint64_t file_stream_off, int input_file_id, int reuse_runs,
                  sx_result** out, uint64_t* end_pos);
int sx_replay_shard_runs(sx_ctx* ctx, const uint8_t* bytes, uint64_t buf_off, uint64_t buf_len,
                         uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at,
                         uint64_t file_stream_off, int input_file_id,
                         const sx_run* const* runs, const uint64_t* n_runs,
                         sx_result** out, uint64_t* end_pos);

/* The whole sharded scan of one file for one rank of a job of `world` ranks (one process per GPU): own range + halo
 * (sx_shard_bounds), the "where did everybody stop" exchange, the repeat when the previous rank ran past this rank's
 * start, wider halos where a run (or, Big5 / EUC-JP, a stretch without token boundary) crosses them.  The transport is
 * the caller's: `allgather` must deliver every rank's `bytes` bytes to all ranks, in rank order (RCCL, MPI, ...), and
 * return 0.  `get_buffer` returns the file bytes [lo, hi) — in HBM of the context's device (*is_device = 1, 16-byte
 * aligned) or in host memory — valid until its next call.  `get_runs` is normally NULL; if given, stage A is skipped
 * and the runs of the buffer come from the caller (host bytes; CPU tests, runs from elsewhere).
 * *out = this rank's findings (segment `rank` of the file's findings, in order); counts[k] / overflow[k] (arrays of
 * `world`, may be NULL) = findings of rank k / how many of its last findings lie behind its range end.  Gathering the
 * Finding buffers is the caller's (e.g. one gather over RCCL); sx_shard_splice() puts gathered buffers in order.
 * Errors: what does not depend on the rank (a Mission with chars_min_nb == 0 or ISO-2022-JP and world > 1, ...) is refused on
 * every rank before anything is exchanged; a rank whose own work fails (memory, HIP, its buffer callback) still joins the
 * exchange with its error code in its row, and EVERY rank returns an error in that round (SX_E_STATE on the healthy ones).
 * A stream of several files: call once per file, in order, on every rank, with file_stream_off = the bytes of the files
 * before.  The state the reference carries from file to file (decoder, leftover, cut flag: src/main.rs:153-168, one
 * ScannerState per Mission for the whole stream) is known to the last rank at the end of a file; it travels to all ranks in
 * one more all-gather of the same callback (<= sizeof decoder + 4q + 40 bytes per Mission) and enters the next file's first
 * shard, so the result is what one process scanning the files in a row gives. */
typedef int (*sx_allgather_fn)(void* user, const void* send, uint64_t bytes, void* recv);
typedef int (*sx_shard_buffer_fn)(void* user, uint64_t lo, uint64_t hi, const void** ptr, int* is_device);
typedef int (*sx_shard_runs_fn)(void* user, const uint8_t* bytes, uint64_t buf_off, uint64_t buf_len,
                                const sx_run* const** runs, const uint64_t** n_runs);
void sx_shard_bounds(uint64_t file_len, int world, int rank, uint64_t* own_lo, uint64_t* own_hi);
int sx_scan_sharded(sx_ctx* ctx, int rank, int world, uint64_t file_len, uint64_t file_stream_off, int input_file_id,
                    uint64_t halo, sx_shard_buffer_fn get_buffer, void* buffer_user, sx_shard_runs_fn get_runs, void* runs_user,
                    sx_allgather_fn allgather, void* allgather_user, sx_result** out, uint64_t* counts, uint64_t* overflow);
/* The ranks' finding buffers -> ONE result in the reference's print order (rank k's findings behind its range end
 * merged into the head of rank k+1's: slice, position, Mission).  str_off of findings[k] is relative to arenas[k]. */
int sx_shard_splice(const sx_finding* const* findings, const uint64_t* n_findings, const uint8_t* const* arenas,
                    const uint64_t* arena_lens, int world, uint64_t file_len, sx_result** out);
/* (ABI 3) The same for ranks whose findings arrive in SEGMENTS (a rank with more than 4 GiB of strings ships its result segment by segment,
 * every segment with its own str_off space): seg s = (findings[s], n_findings[s], arenas[s], arena_lens[s]); rank 0's n_segs_of_rank[0]
 * segments come first, then rank 1's, ...  The result has as many segments as its strings need (< 2 GiB each, cut between findings). */
int sx_shard_splice_segs(const sx_finding* const* findings, const uint64_t* n_findings, const uint8_t* const* arenas,
                         const uint64_t* arena_lens, const uint32_t* n_segs_of_rank, int world, uint64_t file_len, sx_result** out);

/* ---- (round 6) The sharded scan's transport inside the library: RCCL over xGMI, loaded with dlopen ("librccl.so.1"; a host without
 * it still loads this library and scans one GPU).  One process per GPU:
 *     rank 0: sx_transport_rccl_id(id);  ship the 128 bytes to every rank (the launcher's job: a file, an env var, MPI, a socket);
 *     every rank: sx_transport_rccl_create(&t, hip_device, rank, world, id);              -- ncclCommInitRank, a stream of its own
 *     sx_scan_sharded(ctx, rank, world, ..., sx_transport_allgather, t, &mine, counts, overflow);   -- the transport is the callback's `user`
 *     sx_transport_gather(t, mine, 0, file_len, &all);    -- rank 0: ONE result in the reference's order (sx_shard_splice_segs inside); else NULL
 * The gather: an all-gather of the segment sizes, then grouped ncclSend / ncclRecv of exactly those sizes into one device buffer at
 * the root and one copy to the host — no collective on the data path.  Errors: SX_E_STATE without librccl,
 * SX_E_HIP for a failing HIP / RCCL call; text from sx_transport_last_error (NULL: the last failed id / create call). */
#define SX_TRANSPORT_ID_BYTES 128
typedef struct sx_transport sx_transport;
int  sx_transport_rccl_id(uint8_t* id128);
int  sx_transport_rccl_create(sx_transport** out, int hip_device, int rank, int world, const uint8_t* id128);
void sx_transport_destroy(sx_transport* t);
const char* sx_transport_last_error(const sx_transport* t);
int  sx_transport_allgather(void* transport, const void* send, uint64_t bytes, void* recv);   /* an sx_allgather_fn */
int  sx_transport_gather(sx_transport* t, const sx_result* mine, int root, uint64_t file_len, sx_result** out);

/* (round 5, SX_OPT_RESULT_ON_DEVICE) Segment i where it lies in HBM: *d_records = n records (sx_finding16 if *packed, else sx_finding;
 * what they share: *info), *d_arena = arena_len bytes of strings (str_off counts from there).  *d_records == NULL: the segment is in host
 * memory (read it with sx_result_segment / sx_result_segment_packed).  SX_E_STATE: a later scan has reused the memory.  One caller at a
 * time per result: the host accessors' first use of such a segment moves it to host memory. */
int               sx_result_segment_device(const sx_result* r, uint64_t i, const void** d_records, uint64_t* n_findings,
                                           const uint8_t** d_arena, uint64_t* arena_len, int* packed, sx_segment_info* info);
uint64_t          sx_result_count(const sx_result* r);
/* The findings come in one or more segments, in print order: a buffer scanned piece by piece adds a segment per
 * piece; a single Mission with millions of runs is replayed in slabs, one segment each (a slab travels to the host while
 * the next is replayed); several Missions with a large output are interleaved on the device in parts of at most 2 GiB of
 * strings, one segment each (str_off has 32 bits).  The segments' memory is pinned host memory the device wrote
 * directly.  Each segment has its own arena: str_off counts from that arena's start.  A segment stored as sx_finding16 records is
 * expanded to sx_finding on its first sx_result_segment() call (a copy in host memory; sx_result_segment_packed() avoids it). */
uint64_t          sx_result_segments(const sx_result* r);
int               sx_result_segment(const sx_result* r, uint64_t index, const sx_finding** findings,
                                    uint64_t* n_findings, const uint8_t** arena, uint64_t* arena_len);
/* A segment as it is stored: *packed = 1 -> `findings` points at n_findings sx_finding16 and *info says what they share (info may
 * be NULL if the caller only wants to know); *packed = 0 -> at sx_finding, as sx_result_segment() returns them.  No copy either way. */
int               sx_result_segment_packed(const sx_result* r, uint64_t index, const void** findings, uint64_t* n_findings,
                                           const uint8_t** arena, uint64_t* arena_len, int* packed, sx_segment_info* info);
/* Contiguous view of all segments (joined by a copy on first use if there are several;
 * NULL if the strings exceed 4 GiB — use the segments then). */
const sx_finding* sx_result_findings(const sx_result* r);
const uint8_t*    sx_result_arena(const sx_result* r, uint64_t* len);
void              sx_result_free(sx_result* r);

/* `Finding::print` for every finding of a result — src/finding.rs:112-155.
 * n_inputs = ARGS.inputs.len(); radix 0 (no -t) | 'x' | 'd' | 'o'.  The caller
 * frames the whole output with SX_OUTPUT_BOM and a final "\n"
 * (src/main.rs:116,138).  *out is malloc'd; release with sx_free(). */
#define SX_OUTPUT_BOM "\xEF\xBB\xBF"
int sx_print_findings(const sx_ctx* ctx, const sx_result* r, int n_inputs, int radix,
                      int no_metadata, uint8_t** out, uint64_t* out_len);

int  sx_get_stats(const sx_ctx* ctx, sx_stats* out); /* of the last scan call */
void sx_free(void* p);

/* Synthetic input: fills device memory with the background
 * byte stream, byte i = little-endian byte (i&7) of splitmix64-mix(seed + ((i>>3)+1)*phi). */
int sx_fill_background_device(sx_ctx* ctx, void* device_bytes, uint64_t first_byte_index,
                              uint64_t len, uint64_t seed);
/* Device memory helpers so that callers without a HIP binding can stage data. */
int sx_device_alloc(sx_ctx* ctx, uint64_t bytes, void** device_ptr);
int sx_device_free(sx_ctx* ctx, void* device_ptr);
int sx_device_upload(sx_ctx* ctx, void* device_dst, const void* host_src, uint64_t bytes);
int sx_device_download(sx_ctx* ctx, void* host_dst, const void* device_src, uint64_t bytes);
/* Read-only streaming pass over a device buffer (measured HBM read ceiling). */
int sx_device_read_bandwidth(sx_ctx* ctx, const void* device_bytes, uint64_t len, int repeats,
                             double* gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif
