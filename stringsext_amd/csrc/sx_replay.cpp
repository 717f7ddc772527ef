// sx_replay.cpp — stage B: exact host replay of the reference scan around long runs.
//
// The device tells us every maximal run of accepted characters that is long enough to
// matter (>= min(chars_min_nb, output_line_char_nb_max) chars).  Everything the
// reference prints comes from the neighbourhood of such a run:
//   * a Finding needs >= chars_min_nb chars, or continues a cut string
//     (src/helper.rs:315-322, 353-355, 410-415);
//   * between long runs the carried state is only ever a short `leftover`
//     (src/finding_collection.rs:269-285) that dies with the run it belongs to.
// So we run FindingCollection::from (src/finding_collection.rs:84-342) verbatim, but only
// over windows from three windows before a long run until the carried state is clean
// again (leftover empty, `maybe_cut` false).  Three windows = 6q bytes > 4(q-1) bytes, the
// longest a run of < q chars can be, so whatever short leftover existed where we start has
// ended before the first window that can print.  The decoder's private state at the start
// is re-derived from the 8 bytes before it.  The chunk's first windows (exact carried
// state) and last windows (exact state for the next chunk) are always replayed.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "sx_host.hpp"

namespace sx {

// ---------------------------------------------------------------------------------------
// SplitStr::next — src/helper.rs:206-433
// ---------------------------------------------------------------------------------------
bool SplitStr::next(SplitStrResult* out) {
    const bool grep_needed = f_.grep_char >= 0;
    bool grep_char_ok = !grep_needed;
    const uint8_t* ok_s_p = p_;
    size_t ok_s_len = 0, ok_char_nb = 0;
    uint8_t last_multi_char_leading_byte = 0;

    while (p_ < inp_end_ && ok_char_nb < max_) {  // exits 1 and 2, :237
        const uint8_t lead = *p_;
        size_t char_len = 1;
        if ((lead & 0x80) == 0) {
            if (!grep_char_ok && f_.grep_char == lead) grep_char_ok = true;  // :252
        } else if ((lead & 0xE0) == 0xC0) char_len = 2;
        else if ((lead & 0xF0) == 0xE0) char_len = 3;
        else if ((lead & 0xF8) == 0xF0) char_len = 4;

        bool char_is_ok, goto_next_char = true;
        if (char_len == 1) char_is_ok = f_.pass_af_filter(lead);  // :276
        else if (f_.pass_ubf_filter(lead)) {                     // :279
            char_is_ok = !same_block_ || lead == last_multi_char_leading_byte || last_multi_char_leading_byte == 0;
            if (!char_is_ok) goto_next_char = false;  // same char is scanned again as a string start, :289-291
            last_multi_char_leading_byte = lead;
        } else { char_is_ok = false; last_multi_char_leading_byte = 0; }

        if (char_is_ok) { ok_s_len += char_len; ok_char_nb++; p_ += char_len; continue; }
        if (goto_next_char) p_ += char_len;
        const bool exit3 = last_cut_ && ok_char_nb > 0 && ok_s_p == inp_start_;  // :315
        const bool exit4 = ok_char_nb >= chars_min_nb_ && grep_char_ok;          // :317
        if (exit3 || exit4) break;
        ok_s_len = 0; ok_char_nb = 0; ok_s_p = p_; grep_char_ok = !grep_needed;  // :327-330
    }
    if (ok_s_len == 0) return false;  // :343

    const bool touches_left = ok_s_p == inp_start_;
    const bool touches_right = ok_s_p + ok_s_len >= inp_end_;
    const bool is_maybe_cut = ok_char_nb >= max_ || (touches_right && !invalid_after_);
    const bool completes = touches_left && last_cut_;
    const bool again = !completes && touches_right && !invalid_after_ && (ok_char_nb < max_ || !grep_char_ok);
    const bool min_rule = ok_char_nb >= chars_min_nb_;
    if (!completes && !again && (!grep_char_ok || !min_rule)) return false;  // :410-415
    if (ok_char_nb >= max_) inp_start_ = p_;                                  // :418-420
    last_cut_ = is_maybe_cut;                                                 // :421

    out->s = ok_s_p; out->len = ok_s_len;
    out->s_completes_previous_s = completes; out->s_is_maybe_cut = is_maybe_cut;
    out->s_is_to_be_filtered_again = again; out->s_satisfies_min_char_rule = min_rule;
    out->s_satisfies_grep_char_rule = grep_char_ok;
    return true;
}

// ---------------------------------------------------------------------------------------
// The window grid: slices of 4096 bytes from the chunk start (src/input.rs:22,121-123),
// windows of 2q bytes inside a slice (src/finding_collection.rs:120,124-131).
// ---------------------------------------------------------------------------------------
static inline uint64_t window_start(uint64_t p, size_t W) {
    const uint64_t s = p / kInputBufLen * kInputBufLen;
    return s + (p - s) / W * W;
}
static inline uint64_t back_windows(uint64_t p, size_t W, int k) {
    uint64_t w = window_start(p, W);
    for (int i = 0; i < k && w > 0; i++) w = window_start(w - 1, W);
    return w;
}
static inline uint64_t window_end(uint64_t p, size_t W, uint64_t len) {  // end of the window containing p
    const uint64_t s = p / kInputBufLen * kInputBufLen;
    uint64_t e = window_start(p, W) + W;
    if (e > s + kInputBufLen) e = s + kInputBufLen;
    return e < len ? e : len;
}
constexpr int kLeadWindows = 3;

namespace {

class ChunkReplay {
public:
    ChunkReplay(const Mission& m, ScannerState& st, ByteView& bytes, uint64_t len, int file_id, bool is_last,
                const sx_run* runs, uint64_t n_runs, MissionFindings* out)
        : m_(m), st_(st), bytes_(bytes), len_(len), file_id_(file_id), is_last_(is_last), runs_(runs),
          n_runs_(n_runs), out_(out), W_(m.window), consumed0_(st.consumed_bytes), stream0_(st.stream_bytes) {
        size_t cap = 0x9192;  // OUTPUT_BUF_LEN, src/finding.rs:23
        const size_t need = 4 * m.q + 3 * kInputBufLen + 64;
        ob_.resize(cap < need ? need : cap);
        tail_start_ = len ? back_windows(len - 1, W_, kLeadWindows) : 0;
    }

    void run() {
        uint64_t pos = 0;
        while (pos < len_) {
            if (st_.clean()) {
                const uint64_t t = next_trigger(pos);
                if (t > pos) { prime(pos, t); pos = t; continue; }
            }
            pos = scan_from(pos);
        }
        st_.consumed_bytes = consumed0_ + len_;
        st_.stream_bytes = stream0_ + len_;
    }

private:
    // first window start at or after which the replay must be running
    uint64_t next_trigger(uint64_t pos) {
        while (ri_ < n_runs_ && runs_[ri_].end <= pos) ri_++;
        uint64_t t = tail_start_;
        if (ri_ < n_runs_) {
            const uint64_t r = back_windows(runs_[ri_].start, W_, kLeadWindows);
            if (r < t) t = r;
        }
        return t;
    }

    // Bring the decoder from its exact state at `from` to its exact state at `to` without
    // looking at what it decodes.  Far jumps restart it 8 bytes (whole units) before `to`:
    // UTF-8 state depends on <= 3 bytes, UTF-16 on the last unit and the stream parity.
    void prime(uint64_t from, uint64_t to) {
        uint64_t p = from;
        if (to - from > 16) {
            p = to - 8;
            if (m_.is_utf16() && ((stream0_ + p) & 1)) p -= 1;
            st_.decoder.reset(m_.c.encoding);
        }
        uint8_t sink[96];
        while (p < to) {
            const size_t n = (size_t)std::min<uint64_t>(to - p, 16);
            const uint8_t* s = bytes_.span(p, n);
            size_t i = 0;
            for (;;) {
                const DecodeStep r = st_.decoder.decode_to_str_without_replacement(s + i, n - i, sink, sizeof sink, false);
                i += r.read;
                if (r.result == DecoderResult::InputEmpty) break;
            }
            p += n;
        }
    }

    // FindingCollection::from over consecutive windows beginning at window start `pos`;
    // returns the first window start at which the carried state is clean and nothing
    // nearby needs replaying (or len).
    uint64_t scan_from(uint64_t pos) {
        uint8_t* ob = ob_.data();
        const size_t cap = ob_.size();
        while (pos < len_) {
            const uint64_t soff = pos / kInputBufLen * kInputBufLen;
            const size_t slen = (size_t)std::min<uint64_t>(kInputBufLen, len_ - soff);
            const bool is_last_input_buffer = is_last_ && soff + slen == len_;
            const uint64_t consumed = consumed0_ + soff;
            const uint32_t slice_index = (uint32_t)(soff / kInputBufLen);
            const size_t first_finding = out_->v.size();
            size_t din = (size_t)(pos - soff), dend = 0, dout = 0;

            size_t leftover_len = 0;  // :101-114
            if (!st_.last_scan_run_leftover.empty()) {
                leftover_len = st_.last_scan_run_leftover.size();
                memcpy(ob, st_.last_scan_run_leftover.data(), leftover_len);
                st_.last_scan_run_leftover.clear();
                dout = leftover_len;
            }
            bool maybe_cut = st_.last_run_str_was_printed_and_is_maybe_cut_str;  // :115
            bool extra_round = false, is_last_window = false, stopped = false;

            while (din < slen) {  // :124
                if (din + W_ < slen) dend = din + W_;
                else { is_last_window = true; dend = slen; }
                const size_t wbase = din;
                const uint8_t* wp = bytes_.span(soff + wbase, dend - wbase);
                out_->replay_bytes += dend - wbase;

                for (;;) {  // 'decoder, :134
                    const DecodeStep r = st_.decoder.decode_to_str_without_replacement(
                        wp + (din - wbase), dend - din, ob + dout, cap - dout, extra_round);
                    uint8_t precision = SX_PRECISION_EXACT;  // :146

                    if (r.written > 0 && din == 0 && (ob[dout] & 0x80)) {  // :153,176
                        Decoder fresh = st_.decoder.new_decoder_without_bom_handling();
                        uint8_t probe[8] = { 0 }, have[8] = { 0 };
                        const size_t pn = std::min<size_t>(slen, 32);
                        const DecodeStep pr = fresh.decode_to_str_without_replacement(bytes_.span(soff, pn), pn, probe,
                                                                                     sizeof probe, true);
                        const size_t filled = std::min<size_t>(8, dout + r.written);
                        memcpy(have, ob, filled);  // beyond what was written the arena is zero, :55
                        if (pr.written == 0 || memcmp(have, probe, pr.written) != 0) precision = SX_PRECISION_BEFORE;
                    }

                    size_t split_start = dout;  // :211-221
                    const size_t split_end = dout + r.written;
                    if (leftover_len > 0) { split_start -= leftover_len; leftover_len = 0; precision = SX_PRECISION_BEFORE; }

                    const bool invalid_bytes_after = r.result == DecoderResult::Malformed
                                                     || (is_last_window && is_last_input_buffer);  // :234
                    const bool continue_str_if_possible = maybe_cut;  // :240-241
                    maybe_cut = false;

                    SplitStr it(ob + split_start, split_end - split_start, m_.c.chars_min_nb,
                                m_.c.require_same_unicode_block != 0, continue_str_if_possible, invalid_bytes_after,
                                m_.filter, m_.q);
                    SplitStrResult ch;
                    while (it.next(&ch)) {  // :246
                        if (!ch.s_is_to_be_filtered_again) {
                            sx_finding f{};
                            f.position = consumed + din;  // :260
                            f.str_off = (uint32_t)out_->arena.size();
                            f.str_len = (uint32_t)ch.len;
                            f.precision = precision;
                            f.completes_previous = ch.s_completes_previous_s;
                            f.mission_id = m_.c.mission_id;
                            f.input_file_id = (int16_t)file_id_;
                            f.slice_index = slice_index;
                            out_->arena.append((const char*)ch.s, ch.len);
                            out_->v.push_back(f);
                            leftover_len = 0;
                            maybe_cut = ch.s_is_maybe_cut;  // :268
                        } else {
                            leftover_len = ch.len;  // :281
                            maybe_cut = false;
                        }
                        precision = SX_PRECISION_AFTER;  // :289
                    }
                    dout += r.written;  // :292
                    din += r.read;      // :294

                    if (r.result == DecoderResult::InputEmpty) {
                        if (is_last_window && is_last_input_buffer && !extra_round) extra_round = true;
                        else break;
                    } else if (r.result == DecoderResult::OutputFull) {  // :306-323
                        out_->v.resize(first_finding);
                        dout = 0;
                    }
                }
                // a window boundary inside the slice: may we stop here?
                if (din < slen && leftover_len == 0 && !maybe_cut && next_trigger(soff + din) > soff + din) {
                    stopped = true;
                    break;
                }
            }
            // :330-338
            st_.last_scan_run_leftover.assign((const char*)ob + dout - leftover_len, leftover_len);
            st_.last_run_str_was_printed_and_is_maybe_cut_str = maybe_cut;
            pos = soff + din;
            if (stopped) return pos;
            if (st_.clean() && next_trigger(pos) > pos) return pos;
        }
        return pos;
    }

    const Mission& m_;
    ScannerState& st_;
    ByteView& bytes_;
    const uint64_t len_;
    const int file_id_;
    const bool is_last_;
    const sx_run* runs_;
    const uint64_t n_runs_;
    MissionFindings* out_;
    const size_t W_;
    const uint64_t consumed0_, stream0_;
    std::vector<uint8_t> ob_;
    uint64_t tail_start_ = 0;
    uint64_t ri_ = 0;
};

}  // namespace

void replay_chunk(const Mission& m, ScannerState& st, ByteView& bytes, uint64_t len, int input_file_id,
                  bool is_last_input_buffer, const sx_run* runs, uint64_t n_runs, MissionFindings* out) {
    ChunkReplay(m, st, bytes, len, input_file_id, is_last_input_buffer, runs, n_runs, out).run();
}

void replay_ranges(const Mission& m, const ScannerState& st, uint64_t len, const sx_run* runs, uint64_t n_runs,
                   std::vector<std::pair<uint64_t, uint64_t>>* ranges) {
    if (len == 0) return;
    const size_t W = m.window;
    auto add = [&](uint64_t lo, uint64_t hi) {
        lo = lo > 16 ? lo - 16 : 0;  // decoder priming
        if (hi > len) hi = len;
        if (lo < hi) ranges->emplace_back(lo, hi);
    };
    if (!st.clean()) add(0, 4 * W);
    for (uint64_t i = 0; i < n_runs; i++) {
        const uint64_t lo = back_windows(runs[i].start, W, kLeadWindows);
        const uint64_t last = runs[i].end ? runs[i].end - 1 : 0;
        add(lo, window_end(last < len ? last : len - 1, W, len) + 4 * W);
    }
    add(back_windows(len - 1, W, kLeadWindows), len);
}

void merge_device_runs(const DevRun* recs, size_t n, uint64_t min_chars, std::vector<sx_run>* out) {
    std::vector<DevRun> v(recs, recs + n);
    std::sort(v.begin(), v.end(), [](const DevRun& a, const DevRun& b) { return a.start < b.start; });
    out->clear();
    for (size_t i = 0; i < v.size();) {
        sx_run r{ v[i].start, v[i].start + v[i].len, v[i].chars_flags & kRecCharsMask };
        bool open_end = (v[i].chars_flags & kRecEndOpen) != 0;
        size_t k = i + 1;
        while (open_end && k < v.size() && (v[k].chars_flags & kRecStartOpen) && v[k].start == r.end) {
            r.end = v[k].start + v[k].len;
            r.chars += v[k].chars_flags & kRecCharsMask;
            open_end = (v[k].chars_flags & kRecEndOpen) != 0;
            k++;
        }
        if (r.chars >= min_chars) out->push_back(r);
        i = k;
    }
}

void merge_findings(std::vector<MissionFindings>& per, Result* out) {
    out->findings.clear();
    out->arena.clear();
    size_t total = 0, bytes = 0;
    for (auto& mf : per) { total += mf.v.size(); bytes += mf.arena.size(); }
    out->findings.reserve(total);
    out->arena.reserve(bytes);
    std::vector<size_t> idx(per.size(), 0), base(per.size(), 0);
    for (size_t k = 0; k < per.size(); k++) { base[k] = out->arena.size(); out->arena += per[k].arena; }
    for (;;) {
        int best = -1;
        for (size_t k = 0; k < per.size(); k++) {
            if (idx[k] >= per[k].v.size()) continue;
            if (best < 0) { best = (int)k; continue; }
            const sx_finding& a = per[k].v[idx[k]];
            const sx_finding& b = per[best].v[idx[best]];
            if (a.slice_index < b.slice_index || (a.slice_index == b.slice_index && a.position < b.position)) best = (int)k;
        }
        if (best < 0) break;
        sx_finding f = per[best].v[idx[best]++];
        f.str_off += (uint32_t)base[best];
        out->findings.push_back(f);
    }
}

void print_findings(const std::vector<Mission>& missions, const Result& r, int n_inputs, int radix, bool no_metadata,
                    std::string* out) {
    char num[40];
    for (const sx_finding& f : r.findings) {
        const Mission* m = nullptr;
        for (const Mission& c : missions) if (c.c.mission_id == f.mission_id) { m = &c; break; }
        out->push_back('\n');  // src/finding.rs:113
        if (!no_metadata) {
            if (n_inputs > 1 && f.input_file_id >= 0) { out->push_back((char)(f.input_file_id + 64)); out->push_back(' '); }
            if (radix) {
                out->push_back(f.precision == SX_PRECISION_AFTER ? '>' : f.precision == SX_PRECISION_EXACT ? ' ' : '<');
                int k;
                if (radix == 'x') k = snprintf(num, sizeof num, "%llx", (unsigned long long)f.position);
                else if (radix == 'o') k = snprintf(num, sizeof num, "%llo", (unsigned long long)f.position);
                else k = snprintf(num, sizeof num, "%llu", (unsigned long long)f.position);
                out->append(num, (size_t)k);
                out->append(f.completes_previous ? "+\t" : " \t");
            }
            if (missions.size() > 1 && m) {
                out->push_back('(');
                out->push_back((char)(m->c.mission_id + 97));
                out->push_back(' ');
                out->append(m->c.print_encoding_as_ascii ? "ascii" : m->encoding_name());
                out->append(")\t");
            }
        }
        out->append(r.arena.data() + f.str_off, f.str_len);
    }
}

}  // namespace sx
