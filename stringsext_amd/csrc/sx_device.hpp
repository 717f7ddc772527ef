// sx_device.hpp — interface between the host code and the HIP translation unit.
//
// Stage A of the scan: for one Mission, the device classifies every input byte
// ("does this byte belong to a valid character of the Mission's encoding that
// passes the af/ubf filter?", the work of encoding_rs' decoders plus
// Utf8Filter::pass_af_filter/pass_ubf_filter, reference src/mission.rs:333-348)
// and reports every maximal stretch of such bytes that holds enough characters
// to be able to produce a Finding (reference src/helper.rs:315-322: a string
// needs >= chars_min_nb chars, or output_line_char_nb_max to be cut).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/stringsext_amd.h"

namespace sx {

// One wavefront streams `subchunk` consecutive bytes in 1 KiB tiles.
constexpr uint32_t kTileBytes = 1024;

// Device run record (16 bytes).
struct DevRun {
    uint64_t start;        // chunk-relative offset of the first byte
    uint32_t len;          // bytes
    uint32_t chars_flags;  // low 30 bits: characters (saturating); flags below
};
constexpr uint32_t kRecStartOpen = 0x80000000u;  // stretch begins at a sub-chunk start (may continue a previous one)
constexpr uint32_t kRecEndOpen = 0x40000000u;    // stretch reaches the sub-chunk end (may be continued)
constexpr uint32_t kRecCharsMask = 0x3FFFFFFFu;
// a reserved slot that was never filled (see Emitter in sx_kernels.hip)
constexpr uint32_t kRecInvalidLen = 0xFFFFFFFFu, kRecInvalidFlags = 0xFFFFFFFFu;

// ScanParams::counters: [0] records appended to the pool / records that found no room in their region, [3] the fullest sub-chunk's
// records (persistent grid: the next sub-chunk); [1], [2] are the host's sums of the statistics' shards: word kStatBase + shard *
// kStatStride (+ 0: tiles on the general path, + 1: records of the launch), shard = sub-chunk number & (kStatShards - 1)
constexpr uint32_t kStatBase = 32, kStatShards = 16, kStatStride = 32, kCounterWords = kStatBase + kStatShards * kStatStride;

enum ClassifierKind : uint32_t {
    kClsSingleByteLut = 0,  // x-user-defined and WHATWG single-byte tables: 256-entry accept LUT
    kClsUtf8Lut = 1,        // UTF-8, any af/ubf: 256-entry class LUT + SWAR validity
    kClsUtf16Lut = 2,       // UTF-16LE/BE, any af/ubf: two 256-entry LUTs (high byte / low byte)
    kClsUtf8Range2 = 3,     // UTF-8, af = one range, ubf = one range of 2-byte leads: pure SWAR, no LUT
    kClsUtf16Range = 4,     // UTF-16, af = one range, ubf = one range below U+0800, no astral: pure SWAR
    kClsSingleByteRange = 5,// single byte, accept set = one range of bytes < 0x80 (+ all/none of >= 0x80)
    kClsBig5 = 6,           // Big5, Shift_JIS, EUC-KR: token classifier, 2-bit pair table (16 KB) in LDS
    kClsEucJp = 7,          // EUC-JP: the same with three-byte tokens and two pair tables (32 KB)
    kClsSingleByteRanges = 8,// single byte, accept set = up to 6 byte ranges: SWAR, no LUT
    kClsUtf8Range3 = 9,     // UTF-8, af = one range, 2-byte leads = one range, 3-byte leads = one range of E1..EF, no 4-byte leads (sx_classify_ranges.hpp)
    kClsUtf8Range2x2 = 11,  // UTF-8, af = one range, two ranges of 2-byte leads (-u Latin), nothing longer (sx_classify_ranges.hpp; second range in l3_lo / l3_hi)
    kClsUtf16Ranges = 10    // UTF-16, accepted units = up to 2 ranges below U+8000, 1 across it, 1 above; astral: one range of high surrogates (sx_classify_ranges.hpp)
};

struct ScanParams {
    const uint8_t* data;   // device pointer, chunk byte 0
    uint64_t len;          // chunk bytes
    uint32_t subchunk;     // bytes per wavefront (multiple of kTileBytes)
    uint32_t min_chars;    // report stretches with >= min_chars characters
    uint32_t cand_bytes;   // a stretch shorter than this many bytes can never qualify (1..14)
    uint32_t cand_sh[5];   // shift amounts of the "cand_bytes consecutive ones" test (0 = unused step)
    uint32_t parity;       // UTF-16: (stream offset of byte 0) & 1; Big5 / EUC-JP: bytes at the chunk start that finish the token pending on entry
    uint32_t big_endian;   // UTF-16BE
    uint32_t capacity;     // record slots
    uint32_t persistent;   // 0: one wavefront per sub-chunk; else: this many blocks, sub-chunks handed out by counters[3]
    DevRun* recs;
    uint32_t region_cap;   // 0: shared record pool; else: slots per sub-chunk (region mode, see Emitter)
    uint32_t* region_counts; // region mode: records of sub-chunk w
    uint32_t* grid_flags;    // double-byte kernels: per sub-chunk, bit 0 "its token grid is known", bits 1-2 the hang-over at its first byte (zeroed before every launch)
    uint32_t* counters;    // [0] records appended (may exceed capacity = overflow), [1] slow-path tiles, [3] next sub-chunk (persistent grid)
    // range classifiers
    uint32_t a_lo, a_hi;   // accepted ASCII / single-unit range (inclusive)
    uint32_t u_lo, u_hi;   // UTF-8: accepted 2-byte lead range; UTF-16: accepted unit range [u_lo,u_hi]
                           // (an EMPTY range is lo > hi with lo >= 0x81 — Mission::from_c writes 0x81 / 0x80 — and an empty unit slot is rng_c1 = 0,
                           // rng_c2 = 0x7FFF7FFF: launch_v2 always runs the most general instantiations, Utf8Range3T<true, 2> / Utf16RangesT<.., 2, 1, 1, 1>,
                           // which rely on exactly these encodings; tests/test_classify_ranges.py runs those instantiations with empty slots)
    uint32_t l3_lo, l3_hi; // kClsUtf8Range3: accepted 3-byte lead range (within E1..EF)
    uint32_t high_all;     // single-byte range: every byte >= 0x80 accepted
    uint32_t n_ranges;     // kClsSingleByteRanges: ranges in use (the others are empty); per range, replicated over the four bytes:
    uint32_t rng_c1[6], rng_c2[6], rng_hi[6];  // 0x80 - lo7, 0x7F - hi7, and 0 for a range of bytes >= 0x80 / ~0 for one below
                           // kClsUtf16Ranges: n_ranges = below | across << 4 | above << 8 | astral << 12; slots 0, 1 / 2 / 3, 4 / 5 (high surrogates, or a third range below); per unit, replicated over the
                           // two units: rng_c1 = 0x8000 - lo15, rng_c2 = 0x8000 + hi15 (an empty slot: 0, 0x7FFF)
    uint32_t wave_prio;    // 1: the scan wavefronts raise their issue priority (s_setprio)
    uint32_t lr_c1[2], lr_c2[2];  // Big5 / Shift_JIS / EUC-KR: the lead byte ranges (low 7 bits; 0x80 - lo, 0x7F - hi, replicated)
    uint32_t high1;        // Shift_JIS: bytes >= 0x80 outside the lead ranges can be characters (0x80, A1..DF): lut[] holds all 256
    uint32_t gb4;          // gb18030 / GBK: lead digit lead digit may be a four-byte character (the two-byte grammar reads it as error, digit, error,
                           // digit): exact since round 4 — every other candidate of a row is a token, good if its pointer is in range and its character passes (sx_kernels.hip)
    const uint16_t* gb_ranges; // ... device: index gb18030 ranges, [208 breakpoint pointers][208 code points]
    uint64_t ubf;          // ... the Mission's Unicode-block filter (bit = the UTF-8 lead byte & 0x3F)
    uint32_t af_is_range;  // Big5 / EUC-JP: the accepted ASCII bytes are [a_lo,a_hi] (else lut[0..255] holds them: 0x80 / 0)
    const uint32_t* pair_lut;  // Big5 / EUC-JP: device, 2 bits per byte pair (index = the pair as a little-endian u16; EUC-JP: + 65536
                               // for the last two bytes of 8F xx xx): 0 unmapped, 1 mapped, 3 accepted, 2 accepted and two characters
    // table classifiers
    uint8_t lut[512];
};

// ---- stage B on the device (sx_replay_dev.hip) ----
// Regions longer than ReplayParams::max_windows go back to the host.  Text-like input (lines of a few dozen
// chars) chains regions over ~60 windows on average, and giving thousands of them back costs far more than the
// lanes that work through them, hence the generous default.
constexpr uint32_t kMaxRegionWindowsDefault = 512;
enum : uint32_t { kRegionOk = 0, kRegionChained = 1, kRegionNotMine = 2, kRegionTooLong = 3 };
struct ReplayParams {
    const uint8_t* data;      // device: buffer byte 0 (on the slice grid)
    uint64_t len;
    const sx_run* runs;       // device: sorted long runs of this mission (buffer relative)
    uint64_t n_runs;
    uint64_t lo, hi;          // only regions that begin in [lo, hi)
    uint64_t consumed0, stream0;
    uint32_t slice_base;
    uint32_t encoding;        // SX_ENC_*
    const uint16_t* table;    // device: 128-entry single-byte table, Big5 / EUC-JP blob (sx_codec_core.hpp), or nullptr
    uint32_t chars_min_nb, same_block, q, W, long_run;
    uint32_t skip;            // 1: take the shortcuts of sx_replay_core.hpp (0: decode every byte, for comparison)
    int32_t grep_char, mission_id, file_id;
    uint64_t af_lo, af_hi, ubf;
    // pass-1 output cache (all nullptr/0: no cache): slot_of[i] = slot of run i if it replays,
    // *n_heads = how many runs replay; the slot geometry follows from arena_bytes / *n_heads
    const uint32_t* slot_of;
    const uint32_t* n_heads;
    uint8_t* cache_arena;
    uint64_t arena_bytes;
    const uint32_t* head_list;   // head_list[slot] = run index: the replay kernels take one lane per REPLAYING run
    uint32_t max_windows;        // a region that needs more windows is given back (kRegionTooLong)
    uint32_t str_off_base;       // pass 2 (flagged form): added to every str_off (strings of the host's entry part come first)
    uint32_t entry_skip;         // double-byte encodings: bytes at the buffer start that finish the token pending on entry
    // (round 5) the token grid of stage A (ScanParams::grid_flags: per sub-chunk of grid_sub bytes, bit 0 known, bits 1-2 the hang-over at its
    // first byte), or nullptr: dbcs_sync_before's walk back ends at a sub-chunk start
    const uint32_t* grid_flags; uint32_t grid_sub;
    uint64_t n_look;             // runs[] may be read up to here (0: n_runs) — a slab's kernels visit n_runs of them, its regions look on
    // the fast pre-pass of pass 1 (round 5, sx_replay_dev.hip replay_fast_kernel): the slots of the replaying runs it left to the
    // general kernel, and how many (device; nullptr: no pre-pass, the general kernel visits every replaying run)
    uint32_t* hard_list;
    uint32_t* n_hard;
};

struct ReplayRegionOut {
    uint64_t end;             // where the region's replay stopped (a window start, buffer relative)
    uint32_t n_find, n_bytes, status, pad;  // pad: 1 = pass 1 kept the whole output in the region's cache slot
};
// Fills slot_of / n_heads (device arrays of n_runs + 1 u32) for the cache: which runs replay at all.
size_t replay_heads_scratch_bytes(uint64_t n_runs);
hipError_t launch_replay_heads(const ReplayParams& P, uint32_t* slot_of, uint32_t* n_heads, uint32_t* head_list, ReplayRegionOut* ro,
                               void* scratch, size_t scratch_bytes,
                               hipStream_t stream);
hipError_t launch_replay_count(const ReplayParams& P, ReplayRegionOut* out, hipStream_t stream);
// Does the fast pre-pass cover this Mission?  (UTF-8, no -g / -r, 1 <= n <= q: a run is one stretch of accepted characters and
// SplitStr never abandons a call's text.)
bool replay_fast_covers(const ReplayParams& P);
hipError_t launch_replay_fast(const ReplayParams& P, ReplayRegionOut* out, hipStream_t stream);
hipError_t launch_replay_write(const ReplayParams& P, const uint64_t* region_index, const uint64_t* fbase,
                               const uint64_t* abase, uint64_t n_regions, sx_finding* findings, uint8_t* arena,
                               hipStream_t stream);

// long runs cut into pieces at the window starts they cross (sx_replay_core.hpp kPieceCont): P.runs = the joined runs
size_t split_scratch_bytes(uint64_t n_runs);
hipError_t launch_split_count(const ReplayParams& P, void* scratch, size_t scratch_bytes, uint64_t* d_total, hipStream_t stream);
hipError_t launch_split_write(const ReplayParams& P, const void* scratch, uint64_t n_pieces, sx_run* out, hipStream_t stream);

// "which regions stand" + output offsets on the device (sx_replay_dev.hip)
// runs resolved per lane in the first stage of the stitch: the second stage is one wavefront walking the block
// summaries, so with many runs (string-dense input) larger blocks keep that walk short
inline uint32_t stitch_block_runs(uint64_t n_runs) {
    if (const char* e = getenv("SX_STITCH_BLOCK")) { const int v = atoi(e); if (v > 0) return (uint32_t)v; }  // tests
    return n_runs > (1ull << 20) ? 512u : 128u;
}
enum : uint32_t { kTotEnd = 0, kTotLast, kTotFindings, kTotBytes, kTotStanding, kTotReplayBytes, kTotTooLong, kTotLastStart, kTotCount };
size_t stitch_scratch_bytes(uint64_t n_runs);
size_t stitch_blocks_bytes(uint64_t n_runs);
hipError_t launch_stitch_blocks(const ReplayParams& P, const ReplayRegionOut* ro, uint8_t* stands, void* blocks,
                                uint64_t* totals, hipStream_t stream);
hipError_t launch_stitch_finish(const ReplayParams& P, const ReplayRegionOut* ro, uint8_t* stands, const void* blocks,
                                uint64_t E0, uint64_t* fpos, uint64_t* apos, uint64_t* totals, void* scratch,
                                size_t scratch_bytes, hipStream_t stream);
hipError_t launch_replay_write_flagged(const ReplayParams& P, const ReplayRegionOut* ro, const uint8_t* stands,
                                       const uint64_t* fpos, const uint64_t* apos, sx_finding* findings,
                                       uint8_t* arena, uint64_t avg_out_bytes, hipStream_t stream);

// ---- stage B for string-dense Missions, wave-cooperative (sx_wave_dev.hip, sx_wave_core.hpp) ----
// One lane per decoder-input window, 64 consecutive windows per wavefront batch; two passes (count, write).
// byte classes of a wave-path Mission as SWAR ranges instead of the 256-entry table (sx_wave_core.hpp wv_classify16_single_swar)
struct WvSwar {
    uint32_t cls;                    // 0: the class table; 1: ranges
    uint32_t n;                      // ranges in use
    uint32_t c1[6], c2[6], hi[6];    // per range, replicated over the four bytes: 0x80 - lo7, 0x7F - hi7, 0 for bytes >= 0x80 / ~0 for bytes below
    uint32_t hi_len;                 // single byte: UTF-8 bytes of an accepted byte >= 0x80 (2 / 3); two-byte family: of an accepted pair
    uint32_t lr_c1[2], lr_c2[2];     // two-byte family: the lead byte ranges (low 7 bits), as ScanParams::lr_c1
    uint32_t kana;                   // EUC-JP: 8E + A1..DF (half-width katakana, U+FF61..) passes the filter
};
struct WaveParams {
    const uint8_t* data;      // device: buffer byte 0 (on the slice grid)
    uint64_t len;
    uint64_t consumed0;
    uint32_t slice_base;
    uint32_t W, wps, q, n_min;   // window bytes, windows per slice, output_line_char_nb_max, chars_min_nb
    uint64_t g_lo, g_hi;      // windows [g_lo, g_hi) (numbered through the buffer) are replayed here
    uint32_t nwin;            // windows a wavefront owns
    uint32_t inject;          // the exact state at window g_lo (wv_pack), from the host
    int32_t mission_id, file_id;
    uint32_t family;          // 0: single-byte decoders, 1: UTF-8, 4: the two-byte family (Big5, Shift_JIS, EUC-KR), 5: EUC-JP
    const uint8_t* lut;       // device: 256 class bytes (single byte: WVC_*; UTF-8: WVU_*)
    const uint16_t* table;    // device: the decoder table (single byte: 128 entries; nullptr = x-user-defined; two-byte family: its blob)
    const uint32_t* pairs;    // device, two-byte family: 4 bits per byte pair (lead | trail << 8), sx_wave_core.hpp wv_classify16_dbcs
    uint32_t encoding;        // SX_ENC_* (the two-byte family's decoders are picked at run time)
    uint32_t entry_skip;      // two-byte family / EUC-JP: bytes at the buffer's start that finish the token pending on entry (0 / 1; EUC-JP: 0 .. 2)
    // pass 1 out, per wavefront: findings, string bytes, the entry state it assumed for its first window, the state after its last
    uint32_t *wave_nf, *wave_nb, *wave_in, *wave_out;
    // two-byte family: per wavefront, bit 0 "the hang-over at my first tile is known", bit 1 that hang-over — a wavefront whose way back to a
    // token boundary crosses its predecessor's whole range without meeting a byte outside the lead range takes it from there (zeroed per launch)
    uint32_t* wave_grid;
    // pass 2 in: exclusive sums of the above; the launch's output segment starts at (f_sub, a_sub)
    const uint64_t *wave_fbase, *wave_abase;
    uint64_t f_sub, a_sub;
    sx_finding* findings;     // (packed: sx_finding16 records)
    uint8_t* arena;
    uint32_t packed;          // the records are written as sx_finding16 (include/stringsext_amd.h)
    uint32_t str_off_base;    // added to every str_off
    uint64_t v0, v1;          // the wavefronts of this launch: [v0, v1)
    // descriptors (sx_wave_core.hpp WvDesc, three words each): the count pass leaves one per finding, desc_cap per wavefront, for the
    // writer that works a lane per finding (launch_wave_emit); nullptr: the window-parallel writer (launch_wave_write) is the only one
    uint32_t* desc;
    uint32_t desc_cap;
    WvSwar swar;              // cls != 0: the classes come from ranges, `lut` is not read
    const uint32_t* pairs2;   // ... two-byte family then: 2 bits per byte pair (bit 0 mapped, bit 1 accepted)
    // -r on a UTF-8 / UTF-16 Mission (helper.rs:279-296): the wave path does not know it; it is right as long as -r cannot break a string in this
    // buffer — at most ONE lead byte that passes ubf occurs in it.  The count pass ORs the lead bytes it meets into *lead_set (bit = lead &
    // 0x3F); a wavefront that meets two gives up at once, and the host gives the buffer back if the set (with the leftover's) holds two
    uint64_t* lead_set;       // device, or nullptr: no -r
    uint64_t ubf;
    int32_t grep_char;        // -g (round 5): the ASCII code every string must hold, -1: none (sx_wave_core.hpp WvWin::GC)
    // Round 5, repairs.  With -g the state a window hands on can depend on how far back a stretch began (a stretch of accepted chars without
    // the grep char hands on "q chars carried" and "nothing" in turns, window by window), so a wavefront's warm-up windows may leave it with
    // a wrong entry state — before: the whole buffer went back to the lane-per-region path.  redo: a count launch in which only the
    // wavefronts whose assumption was wrong (wave_in[v] != wave_out[v - 1]) run again, from wave_out[v - 1] at their first own window and
    // without warm-up; the host repeats it until the verification passes (a chain of wrong wavefronts needs a launch per link).
    // use_entry: the window-parallel writer takes wave_out[v - 1] the same way (set after repairs; families 0-2).
    uint32_t redo, use_entry;
    // -r on the wave path itself (round 5; sx_wave_core.hpp wv_stretch_same, WvWin::D): families 0 - 2 without -g; lead_set is nullptr then
    uint32_t same;
};
size_t wave_scratch_bytes(uint64_t n_waves);
// pass 1 of wavefronts [v0, v1) + exclusive sums from v0 on + verification; totals (device, 4 x u64): findings, string bytes,
// wavefronts whose assumed entry state was wrong (then nothing of this replay may be used), packed state after the last window
hipError_t launch_wave_count(const WaveParams& P, uint64_t v0, uint64_t v1, uint64_t* fbase, uint64_t* abase, uint64_t* totals,
                             void* scratch, size_t scratch_bytes, hipStream_t stream);
hipError_t launch_wave_write(const WaveParams& P, uint64_t v0, uint64_t v1, hipStream_t stream);
// the same output from the count pass' descriptors (P.desc; totals[2] >> 32 says how many wavefronts found more than desc_cap: then not this one)
hipError_t launch_wave_emit(const WaveParams& P, uint64_t v0, uint64_t v1, hipStream_t stream);

// interleave several missions' findings on the device (sx_sort.hip); every src and out = [findings][string bytes]
hipError_t merge_findings_device_part(const sx_finding* const* f, const uint8_t* const* a, const uint64_t* nf, const uint64_t* nb,
                                      const uint32_t* off0, int n_missions, void* out, void* scratch, size_t scratch_bytes,
                                      hipStream_t stream, int packed = 0);
bool merge_part_can_pack(uint64_t n, int n_missions);   // the one-pass merger takes the part (else the radix sort, which writes sx_finding only)
hipError_t launch_slab_cuts(const ReplayParams& P, uint32_t n_slabs, uint64_t* idx, uint64_t* hi, hipStream_t stream);
hipError_t launch_merge_cuts(const sx_finding* f, uint64_t n, uint64_t nb, const uint64_t* cuts, uint32_t n_cuts, uint64_t* idx,
                             uint64_t* off, hipStream_t stream);
size_t merge_findings_scratch_bytes(uint64_t n_findings, int n_missions);
hipError_t launch_copy_bytes(void* dst, const void* src, uint64_t bytes, uint32_t workgroups, hipStream_t stream);
// a few words (4-aligned, a multiple of 4 bytes) into pinned host memory by a one-wavefront kernel instead of the runtime's blit
hipError_t launch_small_copy(void* pinned_dst, const void* dev_src, size_t bytes, hipStream_t stream);

// order run records by start on the device (sx_sort.hip); unused slots end up last with start = ~0
size_t sort_scratch_bytes(uint32_t n);
hipError_t sort_records(DevRun* recs, uint32_t n, uint64_t max_key, void* scratch, size_t scratch_bytes, hipStream_t stream);

// region mode: records of all sub-chunks, in order, packed into `out`; *total = their number
size_t compact_scratch_bytes(uint64_t n_regions);
hipError_t invalidate_region_slack(DevRun* recs, const uint32_t* counts, uint64_t n_regions, uint32_t region_cap, hipStream_t stream);
hipError_t compact_regions(const DevRun* recs, const uint32_t* counts, uint64_t n_regions, uint32_t region_cap, DevRun* out,
                           uint32_t* total, void* scratch, size_t scratch_bytes, hipStream_t stream);

// join sorted records into runs (>= min_chars chars) on the device; out holds up to n runs
size_t merge_scratch_bytes(uint32_t n);
hipError_t merge_sorted_records(const DevRun* recs, uint32_t n, uint64_t min_chars, void* scratch, size_t scratch_bytes,
                                sx_run* out, uint32_t* out_count, hipStream_t stream);

hipError_t launch_scan(ClassifierKind kind, const ScanParams& p, hipStream_t stream);
// One launch that reads the buffer once for up to kFusedMax Missions (sx_fused.hip): slot s of FusedParams holds the Mission whose
// classifier fused_slot_of() puts there (-1: the Mission cannot be fused and keeps its own launch); used = bit mask of the slots filled.
constexpr int kFusedMax = 3;
struct FusedParams {
    ScanParams m[kFusedMax];          // (first: the kernel reads the slow paths' parameters out of the kernel-argument segment by slot number)
    const uint8_t* data;              // what all slots share (set by launch_scan_fused from the used slots)
    uint64_t len;
    uint32_t subchunk;
    uint32_t fsh[kFusedMax][4];       // per slot: the shifts of the fast loop's candidate test (reach cand_f = min(cand_bytes, 12) bytes)
    uint32_t cand_f[kFusedMax];
    uint32_t pf_zero[kFusedMax];      // prefilter slots: the bits that are clear in the high byte of every accepted unit
};
int fused_slot_of(ClassifierKind kind, const ScanParams& p);
hipError_t launch_scan_fused(const FusedParams& fp, uint32_t used, hipStream_t stream);
hipError_t launch_fill_background(uint8_t* dst, uint64_t first_index, uint64_t len, uint64_t seed,
                                  hipStream_t stream);
hipError_t launch_read_sum(const uint8_t* src, uint64_t len, uint64_t* out, hipStream_t stream, uint32_t subchunk = 0);
// dst[seg_dst[i] .. ] = src[seg_src[i] .. seg_src[i]+seg_len[i]) for i < n
hipError_t launch_gather(const uint8_t* src, uint8_t* dst, const uint64_t* seg_src, const uint64_t* seg_dst,
                         const uint32_t* seg_len, uint32_t n, hipStream_t stream);

}  // namespace sx
