// sx_host.hpp — host side of the scan (C++17).  Mirrors the reference's operator
// interface for the path: Mission (src/mission.rs:382-421), Utf8Filter (:307-349),
// Decoder (encoding_rs 0.8.34 as used at src/finding_collection.rs:138-143),
// SplitStr (src/helper.rs:58-433), ScannerState (src/scanner.rs:40-89),
// Finding / Precision (src/finding.rs:34-74), FindingCollection::from
// (src/finding_collection.rs:84-342).
//
// Stage B: the device reports where a Finding can arise (long runs); the code here
// re-runs the reference's exact sequential semantics over just those windows, so that
// positions, precision marks, line cuts and `+` continuations are bit-exact.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "sx_device.hpp"
#ifndef SXD
#define SXD inline
#define SXD_NOINLINE inline
#endif
#include "sx_codec_core.hpp"

namespace sx {

constexpr size_t kInputBufLen = 4096;  // INPUT_BUF_LEN, src/input.rs:22

// ---------------------------------------------------------------------------------------
// Utf8Filter — src/mission.rs:307-349
// ---------------------------------------------------------------------------------------
struct Utf8Filter {
    uint64_t af_lo = 0, af_hi = 0, ubf = 0;
    int grep_char = -1;
    bool pass_af_filter(uint8_t b) const {
        b &= 127;
        return ((b < 64 ? af_lo >> b : af_hi >> (b - 64)) & 1) != 0;
    }
    bool pass_ubf_filter(uint8_t b) const { return ((ubf >> (b & 0x3F)) & 1) != 0; }
    // the filter applied to a whole character, via its UTF-8 lead byte
    bool pass_lead(uint8_t lead) const { return (lead & 0x80) ? pass_ubf_filter(lead) : pass_af_filter(lead); }
};

// ---------------------------------------------------------------------------------------
// Mission — src/mission.rs:382-421, plus what the device needs for it
// ---------------------------------------------------------------------------------------
struct Mission {
    sx_mission c{};
    Utf8Filter filter;
    size_t q = 64;          // output_line_char_nb_max
    size_t window = 128;    // decoder_input_window = 2*q, src/finding_collection.rs:120
    uint32_t long_run = 4;  // min(chars_min_nb, q): fewer chars can never yield a Finding
    bool is_utf16() const { return c.encoding == SX_ENC_UTF16LE || c.encoding == SX_ENC_UTF16BE; }
    bool is_dbcs() const { return c.encoding >= SX_ENC_BIG5 && c.encoding <= SX_ENC_GBK; }   // a pending lead byte is the decoder state
    // the decoder's state at a byte is not derivable from the bytes near it (ISO-2022-JP: the set an escape sequence selected):
    // no stage A, no device, no shards — FindingCollection::from over every window, in order, on the host
    bool host_sequential() const { return c.encoding == SX_ENC_ISO_2022_JP; }
    const char* encoding_name() const;

    // device classifier for this mission
    ClassifierKind kind = kClsSingleByteLut;
    ScanParams proto{};  // a_lo.., lut filled in; data/len/recs set per launch
    std::vector<uint32_t> pair_lut;  // Big5 / EUC-JP: the pair codes the kernel keeps in LDS (ScanParams::pair_lut)
    // wave-cooperative stage B (sx_wave_core.hpp): the Mission is one it covers, and its class byte per input byte
    bool wave_ok = false;
    bool wave_lead_check = false;   // -r on a UTF-8 / UTF-16 Mission: the wave path only for buffers in which at most one lead byte passes ubf (WaveParams::lead_set)
    bool wave_same = false;         // -r without -g, families 0 - 2 (round 5): the wave kernels apply it themselves (WaveParams::same); no lead check then
    uint32_t wave_family = 0;   // 0: single-byte decoders, 1: UTF-8, 4: the two-byte family (Big5, Shift_JIS, EUC-KR)
    std::vector<uint8_t> wave_lut;
    std::vector<uint32_t> wave_pairs;   // two-byte family: 4 bits per byte pair (sx_wave_core.hpp wv_classify16_dbcs)
    std::vector<uint32_t> wave_pairs2;  // ... and 2 bits per pair (bit 0 mapped, bit 1 accepted) when wave_swar says the lengths need no table
    WvSwar wave_swar{};                 // cls != 0: the same classes as SWAR ranges — what the kernels use then (SX_WAVE_LUT=1: the table anyway)
    // Big5 / EUC-JP, per buffer (set by the schedule before stage A/B of a buffer; the replay only reads it):
    // how many bytes at the buffer start finish the token that was pending on entry — where its token grid begins
    mutable uint32_t buf_entry_skip = 0;
    static int from_c(const sx_mission& in, bool force_generic, Mission* out, std::string* err);
};

// ---------------------------------------------------------------------------------------
// Decoder — encoding_rs `Decoder` restricted to decode_to_str_without_replacement
// ---------------------------------------------------------------------------------------
enum class DecoderResult { InputEmpty, OutputFull, Malformed };
struct DecodeStep {
    DecoderResult result;
    size_t read, written;
};

class Decoder {
public:
    explicit Decoder(int encoding = SX_ENC_UTF8) { reset(encoding); }
    void reset(int encoding);
    int encoding() const { return d_.enc; }
    Decoder new_decoder_without_bom_handling() const { return Decoder(d_.enc); }
    DecodeStep decode_to_str_without_replacement(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, bool last);
    // nothing pending: no partial UTF-8 sequence, no half UTF-16 unit, no surrogate waiting for its pair, no lead byte
    bool idle() const;
    // double-byte encodings: how many of the next bytes finish the token that is pending now (0: none pending,
    // or its last byte will be given back) — where the token grid of what follows begins
    uint32_t entry_skip(const uint8_t* next_bytes, uint64_t avail) const;
    // the same in the grammar of the scan kernel, which differs for gb18030 only: the kernel reads a four-byte token as
    // lead(error) digit lead(error) digit — aligned with the true tokens, pending is only ever one lead byte
    uint32_t entry_skip_scan(const uint8_t* next_bytes, uint64_t avail) const;
    DDecoder& raw() { return d_; }

private:
    DDecoder d_;  // the one decoder implementation, sx_codec_core.hpp
};
const uint16_t* single_byte_table(int encoding);  // nullptr for x-user-defined / non-table encodings
const uint16_t* decoder_table(int encoding, size_t* n_words);  // single-byte table or Big5 / EUC-JP blob (sx_codec_core.hpp)
bool encoding_is_known(int encoding);
const char* encoding_name(int encoding);

// ---------------------------------------------------------------------------------------
// SplitStr — src/helper.rs:58-433
// ---------------------------------------------------------------------------------------
struct SplitStrResult {
    const uint8_t* s = nullptr;
    size_t len = 0;
    bool s_completes_previous_s = false, s_is_maybe_cut = false, s_is_to_be_filtered_again = false;
    bool s_satisfies_min_char_rule = false, s_satisfies_grep_char_rule = false;
};
class SplitStr {
public:
    SplitStr(const uint8_t* inp, size_t len, uint8_t chars_min_nb, bool require_same_unicode_block,
             bool last_s_was_maybe_cut, bool invalid_bytes_after_inp, const Utf8Filter& f, size_t s_char_nb_max)
        : pm_{ f.af_lo, f.af_hi, f.ubf, f.grep_char, (uint32_t)s_char_nb_max, chars_min_nb, require_same_unicode_block ? 1u : 0u },
          it_{ inp, inp + len, inp, last_s_was_maybe_cut, invalid_bytes_after_inp } {}
    bool next(SplitStrResult* out);

private:
    SplitParams pm_;
    DSplit it_;  // the one SplitStr implementation, sx_codec_core.hpp
};

// ---------------------------------------------------------------------------------------
// ScannerState — src/scanner.rs:40-89
// ---------------------------------------------------------------------------------------
struct ScannerState {
    Decoder decoder;
    std::string last_scan_run_leftover;
    bool last_run_str_was_printed_and_is_maybe_cut_str = false;
    uint64_t consumed_bytes = 0;  // starts at counter_offset
    uint64_t stream_bytes = 0;    // bytes fed to the decoder since it was created (UTF-16 unit parity)
    void reset(const Mission& m) {
        decoder.reset(m.c.encoding);
        last_scan_run_leftover.clear();
        last_run_str_was_printed_and_is_maybe_cut_str = false;
        consumed_bytes = m.c.counter_offset;
        stream_bytes = 0;
    }
    // Nothing carried that a later window could need: no leftover, no pending cut — and no bytes
    // pending in the decoder: a character that straddles the chunk boundary belongs to a run
    // that the device scan of neither chunk sees whole, so the next chunk's first windows must
    // be replayed from this exact state (found by tools/gpu_fuzz.py).
    bool clean() const {
        return last_scan_run_leftover.empty() && !last_run_str_was_printed_and_is_maybe_cut_str && decoder.idle();
    }
};

// ---------------------------------------------------------------------------------------
// Bytes of the current chunk, possibly only partly present on the host
// ---------------------------------------------------------------------------------------
class ByteView {
public:
    virtual ~ByteView() {}
    // pointer to bytes [off, off+n) of the chunk; n <= 4096; stays valid for the lifetime of
    // the view.  `hint` is a caller-owned cursor (callers walk the chunk in ascending order).
    virtual const uint8_t* span(uint64_t off, size_t n, size_t* hint) = 0;
    virtual bool all_on_host() const { return false; }   // every byte is here: reading the chunk front to back costs nothing extra
};
class HostBytes : public ByteView {
public:
    explicit HostBytes(const uint8_t* p) : p_(p) {}
    const uint8_t* span(uint64_t off, size_t, size_t*) override { return p_ + off; }
    bool all_on_host() const override { return true; }
private:
    const uint8_t* p_;
};

// ---------------------------------------------------------------------------------------
// Findings of one mission for one chunk
// ---------------------------------------------------------------------------------------
// Pinned host blocks that travel from the context (D2H target) to a result (zero copy) and
// back when the result is freed.  Shared so that a result may outlive its context.
struct PinnedPool {
    struct Block { void* p = nullptr; size_t cap = 0; };
    std::mutex mu;
    std::vector<Block> free_blocks;
    Block take(size_t bytes);   // a pooled block of >= bytes, or a new allocation (sx_api.cpp: hipHostMalloc)
    void give(Block b);
    ~PinnedPool();
};

// What the records of a packed segment share (include/stringsext_amd.h sx_finding16, sx_segment_info)
struct SegInfo {
    int file_id = -1;
    uint32_t slice_base = 0;
    uint64_t pos0[256];   // by mission_id
};
inline sx_finding expand_finding(const sx_finding16& p, const SegInfo& si) {
    sx_finding f;
    f.position = p.position; f.str_off = p.str_off; f.str_len = p.str_len;
    f.precision = (uint8_t)(p.flags & 3u); f.completes_previous = (uint8_t)((p.flags >> 2) & 1u);
    f.mission_id = p.mission_id; f.reserved = 0; f.input_file_id = (int16_t)si.file_id; f.reserved2 = 0;
    f.slice_index = si.slice_base + (uint32_t)((p.position - si.pos0[p.mission_id]) / 4096u);
    return f;
}
inline sx_finding16 pack_finding(const sx_finding& f) {
    sx_finding16 p;
    p.position = f.position; p.str_off = f.str_off; p.str_len = (uint16_t)f.str_len;
    p.flags = (uint8_t)((f.precision & 3u) | (f.completes_previous ? 4u : 0u)); p.mission_id = f.mission_id;
    return p;
}

struct MissionFindings {
    std::vector<sx_finding> v;  // str_off relative to `arena`
    std::string arena;
    uint64_t replay_bytes = 0;
    // alternatively the findings live in a pinned block (device replay output, no copy):
    // [ext_nf x sx_finding][ext_na bytes of strings]
    PinnedPool::Block ext{};
    size_t ext_nf = 0, ext_na = 0;
    // ... and, until the next scan of the mission, also still on the device (same layout):
    // several missions' findings are interleaved there instead of on the host
    const void* dev_copy = nullptr;
    // dev_only: the copy to the host was put off (several missions are interleaved on the device first); ext_nf / ext_na
    // count what dev_copy holds, data() / strings() are not valid until it is fetched
    bool dev_only = false;
    // round 5, SX_OPT_RESULT_ON_DEVICE: the caller wants it there (dev_only stays up in the finished result; the host accessors fetch it on
    // first use).  dev_copy is the context's memory: valid while *dev_epoch_ref == dev_epoch (every scan call advances the context's epoch)
    bool keep_on_device = false;
    std::shared_ptr<std::atomic<uint64_t>> dev_epoch_ref;
    uint64_t dev_epoch = 0;
    // further segments of the same mission, in order (a mission replayed in slabs; only with a single mission)
    std::vector<MissionFindings> more;
    // packed: the pinned block holds [ext_nf x sx_finding16][strings] — segments of a result that the device's dense writers
    // produced; `info` says what the records share.  data() expands them into `expanded` on first use (get(i): one record, no copy).
    bool packed = false;
    std::shared_ptr<SegInfo> info;
    mutable std::vector<sx_finding> expanded;
    size_t rec_size() const { return packed ? sizeof(sx_finding16) : sizeof(sx_finding); }
    size_t count() const { return (ext.p || dev_only) ? ext_nf : v.size(); }
    void expand() const;   // sx_replay.cpp
    const sx_finding* data() const { if (packed) { expand(); return expanded.data(); } return ext.p ? (const sx_finding*)ext.p : v.data(); }
    const sx_finding16* data16() const { return packed ? (const sx_finding16*)ext.p : nullptr; }
    sx_finding get(size_t i) const { return packed ? expand_finding(data16()[i], *info) : data()[i]; }
    const char* strings() const { return ext.p ? (const char*)ext.p + ext_nf * rec_size() : arena.data(); }
    size_t strings_len() const { return (ext.p || dev_only) ? ext_na : arena.size(); }
};

// Exact replay of FindingCollection::from over the windows that matter.
// `st` is the carried ScannerState (exact on entry, exact on return).
void replay_chunk(const Mission& m, ScannerState& st, ByteView& bytes, uint64_t len, int input_file_id,
                  bool is_last_input_buffer, const sx_run* runs, uint64_t n_runs, MissionFindings* out);

// The same, split into parts that can run on different threads: part 0 starts from the exact
// carried state, the others speculate that nothing is carried where they start;
// replay_stitch() verifies that against the exact part before and repairs where it was wrong.
struct ReplayPart {
    MissionFindings findings;
    struct Region { uint64_t start, end; size_t f0, f1; };
    std::vector<Region> regions;
    uint64_t end_pos = 0;
    ScannerState state;
};
void replay_plan(uint64_t len, unsigned max_parts, std::vector<uint64_t>* bounds);  // bounds[k]..bounds[k+1]
void replay_plan_range(uint64_t lo, uint64_t hi, unsigned max_parts, std::vector<uint64_t>* bounds);
void replay_part(const Mission& m, const ScannerState& entry, uint64_t consumed0, uint64_t stream0, ByteView& bytes,
                 uint64_t len, int file_id, bool is_last, const sx_run* runs, uint64_t n_runs, uint64_t lo, uint64_t hi,
                 bool entry_exact, ReplayPart* part);
// FindingCollection::from over every window of [lo, hi) (window starts; hi may be len), one after the other from the exact
// state `st` at lo; on return `st` is the exact state at hi, whatever is pending there (no run list needed).
void replay_exact_windows(const Mission& m, ScannerState& st, uint64_t consumed0, uint64_t stream0, ByteView& bytes, uint64_t len,
                          int file_id, uint64_t lo, uint64_t hi, MissionFindings* out, bool is_last = false);
void replay_stitch(const Mission& m, ScannerState& st, uint64_t consumed0, uint64_t stream0, ByteView& bytes,
                   uint64_t len, int file_id, bool is_last, const sx_run* runs, uint64_t n_runs,
                   std::vector<ReplayPart>& parts, MissionFindings* out, unsigned copy_threads,
                   uint64_t* end_pos);

// Byte ranges of the chunk that the replay will (very likely) touch, for sparse download of
// device-resident input.  Appends [lo,hi) pairs (unsorted, may overlap).
void replay_ranges(const Mission& m, const ScannerState& st, uint64_t len, const sx_run* runs, uint64_t n_runs,
                   unsigned parts, std::vector<std::pair<uint64_t, uint64_t>>* ranges);

// Turn raw device records (any order, sub-chunk pieces flagged open) into maximal runs
// with >= min_chars characters, sorted by start.
void merge_device_runs(const DevRun* recs, size_t n, uint64_t min_chars, uint64_t subchunk, std::vector<sx_run>* out);
// the same for records already sorted by start (unused slots last, start == ~0)
void merge_sorted_device_runs(const DevRun* recs, size_t n, uint64_t min_chars, std::vector<sx_run>* out);

// k-way merge in the reference's order: slice by slice, then (position, mission_id)
// — src/main.rs:118-136, src/finding.rs:92-109.
// What one call returns: the findings in the reference's print order, as one or more segments
// (a large device-resident buffer is scanned piece by piece; every piece adds a segment).
struct Result {
    std::vector<MissionFindings> segs;
    std::shared_ptr<PinnedPool> pool;  // where the segments' pinned blocks go back to
    ~Result() { release(); }
    void release() { for (auto& s : segs) if (s.ext.p && pool) { pool->give(s.ext); s.ext = {}; } }
    size_t count() const { size_t n = 0; for (auto& s : segs) n += s.count(); return n; }
    // one contiguous findings array + arena (copies if there are several segments)
    bool flatten(std::string* err);
};
void merge_findings(std::vector<MissionFindings>& per_mission, const std::shared_ptr<PinnedPool>& pool, Result* out);

// Finding::print — src/finding.rs:112-155
void print_findings(const std::vector<Mission>& missions, const Result& r, int n_inputs, int radix, bool no_metadata,
                    std::string* out);

}  // namespace sx
