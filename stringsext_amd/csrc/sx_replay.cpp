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
// from the window in which a long run begins until the run is over and no cut string is
// pending (`maybe_cut` false).  What the reference carries into that first window is
// re-derived from the few bytes before it:
//   * the decoder's private state (UTF-8: <= 3 bytes; UTF-16: stream parity + last unit);
//   * whether a `leftover` exists.  Between long runs a leftover is a run of < min(n,q)
//     accepted chars hanging over the window edge; it can never be printed or reach q, and
//     its only observable effect is `Precision::Before` on the first string of the next
//     decoder call (src/finding_collection.rs:214-221).  So its content is irrelevant: we
//     carry the last accepted char before the edge if there is one, nothing otherwise.
//     (If it joined a run of >= min(n,q) chars, that run would begin before the edge and
//     the device would have reported it from there.)
// The chunk's first windows start from the exact carried ScannerState and are followed
// strictly (until leftover AND maybe_cut are clear: a run split over two chunks is seen by
// neither chunk's device scan); the last window is always replayed so that the state
// handed to the next chunk is exact.
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <mutex>
#include <thread>

#include "sx_host.hpp"
#include "sx_replay_core.hpp"   // the rules shared with the device replay: pieces of long runs (kPieceCont), derive_in_run

namespace sx {

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

struct RegionLog {
    uint64_t start, end;  // window starts: replayed [start, end)
    size_t f0, f1;        // findings [f0, f1) of the worker's list
};

// One worker: replays the regions that begin in [lo, hi) of the chunk.
class RangeReplay {
public:
    RangeReplay(const Mission& m, ByteView& bytes, uint64_t len, int file_id, bool is_last, const sx_run* runs,
                uint64_t n_runs, uint64_t consumed0, uint64_t stream0)
        : m_(m), bytes_(bytes), len_(len), file_id_(file_id), is_last_(is_last), runs_(runs), n_runs_(n_runs),
          W_(m.window), consumed0_(consumed0), stream0_(stream0) {
        const size_t cap = 0x9192;  // OUTPUT_BUF_LEN, src/finding.rs:23
        const size_t need = 4 * m.q + 3 * kInputBufLen + 64;
        ob_.resize(cap < need ? need : cap);
    }

    // `st` describes position `lo`: the exact carried state (entry_exact) or "nothing carried"
    // (then the decoder is re-derived from the bytes before lo).  Regions that begin in
    // [lo, hi) are replayed, the last one to its own end.  owns_tail: also bring the state to
    // the chunk end exactly.  Returns the position up to which the state is known clean/exact.
    uint64_t run(ScannerState& st, uint64_t lo, uint64_t hi, bool entry_exact, bool owns_tail,
                 MissionFindings* out, std::vector<RegionLog>* log) {
        st_ = &st; out_ = out; hi_ = hi; owns_tail_ = owns_tail;
        // first run that ends after lo (runs are disjoint and sorted: so are their ends)
        ri_ = (uint64_t)(std::partition_point(runs_, runs_ + n_runs_, [lo](const sx_run& r) { return r.end <= lo; }) - runs_);
        strict_ = entry_exact && !st.clean();  // exact carried state: follow it until it is clean
        // no exact state here: re-derive decoder state AND leftover presence from the bytes before lo
        if (!entry_exact) derive_state(lo, ~0ull, st.decoder);  // no exact state anywhere near: pos out of reach
        uint64_t pos = lo;
        while (pos < len_) {
            if (!strict_ && !st.last_run_str_was_printed_and_is_maybe_cut_str) {
                const uint64_t r = next_region_start(pos);
                // nothing more to replay: the tail's owner brings the state to the buffer end,
                // anybody else stops at the end of its own range (the halo behind it is not its business)
                if (r >= len_) { pos = owns_tail_ ? len_ : std::max(pos, std::min(hi_, len_)); break; }
                if (r >= hi_ && !owns_tail_) { pos = std::max(pos, hi_); break; }
                pos = r;
            }
            RegionLog rg{ pos, 0, out->v.size(), 0 };
            pos = scan_from(pos);
            strict_ = false;
            rg.end = pos; rg.f1 = out->v.size();
            if (log) log->push_back(rg);
        }
        st.consumed_bytes = consumed0_ + pos;
        st.stream_bytes = stream0_ + pos;
        return pos;
    }

private:
    // Re-derive what the reference carries into window start B: run a decoder over the few
    // bytes before B (from `pos` with its exact decoder `d_pos` if that is nearer than 8
    // bytes).  Sets st_->decoder to the decoder state at B and st_->leftover to the last
    // accepted character if it is the last thing delivered before B (see file comment).
    void derive_state(uint64_t B, uint64_t pos, const Decoder& d_pos) {
        // B is where a continuation piece of a long run begins (sx_replay_core.hpp kPieceCont): the state there is a
        // function of the run — unless the exact state is at hand anyway
        if (!(pos == B)) {
            const sx_run* pc = std::lower_bound(runs_, runs_ + n_runs_, B, [](const sx_run& r, uint64_t b) { return r.start < b; });
            if (pc != runs_ + n_runs_ && pc->start == B && (pc->chars & kPieceCont)) {
                const uint64_t delta = pc->chars & ~kPieceCont;
                if (delta < 4ull * m_.q) {
                    uint8_t ob[kObCapBig];   // (<= 4q bytes, q <= 255)
                    bool cut = false;
                    const uint8_t* from = bytes_.span(B - delta, (size_t)delta, &hint_);
                    DDecoder& dd = st_->decoder.raw();
                    const uint16_t* table = decoder_table(m_.c.encoding, nullptr);
                    uint32_t n = 0;
                    switch (enc_family(m_.c.encoding)) {
                    case 1: n = derive_in_run<1>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    case 2: n = derive_in_run<2>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    case 3: n = derive_in_run<3>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    case 4: n = derive_in_run<4>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    case 5: n = derive_in_run<5>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    default: n = derive_in_run<0>((uint32_t)m_.q, m_.c.encoding, table, from, (uint32_t)delta, dd, ob, kObCapBig, &cut); break;
                    }
                    st_->last_scan_run_leftover.assign((const char*)ob, n);
                    st_->last_run_str_was_printed_and_is_maybe_cut_str = cut;
                    return;
                }
                derive_plain(B, pos, d_pos);   // the decoder's pending bytes
                st_->last_scan_run_leftover.clear();
                st_->last_run_str_was_printed_and_is_maybe_cut_str = true;
                return;
            }
        }
        derive_plain(B, pos, d_pos);
    }
    void derive_plain(uint64_t B, uint64_t pos, const Decoder& d_pos) {
        Decoder d(m_.c.encoding);
        // With --same-unicode-block (-r) the leftover's content matters after all: SplitStr re-scans it
        // and remembers the lead byte of its last multi-byte char, which decides where the NEXT stretch
        // is cut (src/helper.rs:279-292).  The stretch is < min(n,q) chars, so look back that far.
        const uint64_t back = m_.c.require_same_unicode_block ? 4ull * m_.long_run + 8 : 8;
        uint64_t p = B >= back ? B - back : 0;
        if (m_.is_utf16() && ((stream0_ + p) & 1)) p = p ? p - 1 : p + 1;
        if (p <= pos && pos <= B) { p = pos; d = d_pos; }
        else if (m_.is_dbcs()) p = dbcs_sync(B, p);
        if (p > B) p = B;
        uint8_t sink[96], last[4], mb[4];
        size_t last_len = 0, mb_len = 0;   // last accepted char; last accepted multi-byte char of the same stretch
        if (p < B) {
            uint64_t at = p;
            while (at < B) {  // in pieces: the sink is small
                const size_t n = (size_t)std::min<uint64_t>(B - at, 24);
                const uint8_t* s = bytes_.span(at, n, &hint_);
                size_t i = 0;
                for (;;) {
                    const DecodeStep r = d.decode_to_str_without_replacement(s + i, n - i, sink, sizeof sink, false);
                    i += r.read;
                    for (size_t w = 0; w < r.written;) {
                        const uint8_t lead = sink[w];
                        const size_t cl = lead < 0x80 ? 1 : lead < 0xE0 ? 2 : lead < 0xF0 ? 3 : 4;
                        if (m_.filter.pass_lead(lead)) {
                            memcpy(last, sink + w, cl); last_len = cl;
                            if (cl > 1) { memcpy(mb, sink + w, cl); mb_len = cl; }
                        } else { last_len = 0; mb_len = 0; }
                        w += cl;
                    }
                    if (r.result == DecoderResult::InputEmpty) break;
                    if (r.result == DecoderResult::Malformed) { last_len = 0; mb_len = 0; }
                }
                at += n;
            }
        }
        st_->decoder = d;
        st_->last_scan_run_leftover.clear();
        if (last_len) {
            if (m_.c.require_same_unicode_block && mb_len && last_len == 1) st_->last_scan_run_leftover.assign((const char*)mb, mb_len);
            st_->last_scan_run_leftover.append((const char*)last, last_len);
        }
        st_->last_run_str_was_printed_and_is_maybe_cut_str = false;
    }

    // Double-byte encodings: a token boundary in [lim, lim + 2], not beyond B (sx_replay_core.hpp dbcs_sync_before,
    // over a ByteView): from the nearest byte outside the lead range in front of lim — the decoder is neutral right
    // after it — or from the buffer start, where the token pending on entry ends after Mission::buf_entry_skip bytes.
    uint64_t dbcs_sync(uint64_t B, uint64_t lim) {
        const int enc = m_.c.encoding;
        const bool big5 = enc_family((uint32_t)enc) == 4;   // the two-byte family
        auto lead_range = [&](uint8_t b) { return big5 ? dbcs_may_be_pending_after<4>(b, enc) : dbcs_may_be_pending_after<5>(b, enc); };   // (gb18030: digits too)
        uint64_t r = lim;
        while (r > 0) {
            const size_t n = (size_t)std::min<uint64_t>(r, 64);
            const uint8_t* s = bytes_.span(r - n, n, &hint_);
            size_t k = n;
            while (k > 0 && lead_range(s[k - 1])) k--;
            r = r - n + k;
            if (k > 0) break;
        }
        if (r == 0) r = m_.buf_entry_skip;
        while (r < lim) {
            const size_t n = (size_t)std::min<uint64_t>(len_ - r, 4);
            const uint8_t* s = bytes_.span(r, n, &hint_);
            r += big5 ? dbcs_token_len<4>(s, n, enc) : dbcs_token_len<5>(s, n, enc);
        }
        return r < B ? r : B;
    }

    // First window start >= pos at which the replay must be running (len_ if there is none).
    // No cut string is pending at pos and the decoder state at pos is exact.
    uint64_t next_region_start(uint64_t pos) {
        while (ri_ < n_runs_ && runs_[ri_].end <= pos) ri_++;
        uint64_t want = len_;
        if (ri_ < n_runs_) want = window_start(runs_[ri_].start, W_);
        if (owns_tail_ && len_) want = std::min(want, tail_start());
        if (want >= len_) return len_;
        if (want <= pos) return pos;                   // state at pos is exact: just go on
        if (want >= hi_ && !owns_tail_) return want;  // somebody else's
        derive_state(want, pos, st_->decoder);
        return want;
    }

    // FindingCollection::from over consecutive windows beginning at window start `pos`;
    // returns the first window start at which the carried state is clean and the next
    // region does not begin right there (or len).
    uint64_t scan_from(uint64_t pos) {
        ScannerState& st = *st_;
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
            if (!st.last_scan_run_leftover.empty()) {
                leftover_len = st.last_scan_run_leftover.size();
                memcpy(ob, st.last_scan_run_leftover.data(), leftover_len);
                st.last_scan_run_leftover.clear();
                dout = leftover_len;
            }
            bool maybe_cut = st.last_run_str_was_printed_and_is_maybe_cut_str;  // :115
            bool extra_round = false, is_last_window = false, stopped = false;

            while (din < slen) {  // :124
                if (din + W_ < slen) dend = din + W_;
                else { is_last_window = true; dend = slen; }
                const size_t wbase = din;
                const uint8_t* wp = bytes_.span(soff + wbase, dend - wbase, &hint_);
                out_->replay_bytes += dend - wbase;

                for (;;) {  // 'decoder, :134
                    const DecodeStep r = st.decoder.decode_to_str_without_replacement(
                        wp + (din - wbase), dend - din, ob + dout, cap - dout, extra_round);
                    uint8_t precision = SX_PRECISION_EXACT;  // :146

                    if (r.written > 0 && din == 0 && (ob[dout] & 0x80)) {  // :153,176
                        Decoder fresh = st.decoder.new_decoder_without_bom_handling();
                        uint8_t probe[8] = { 0 }, have[8] = { 0 };
                        const size_t pn = std::min<size_t>(slen, 32);
                        const DecodeStep pr = fresh.decode_to_str_without_replacement(bytes_.span(soff, pn, &hint_), pn, probe,
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

                    // A chunk is only ever returned if it continues a cut string, may be carried
                    // over the window edge, or has >= chars_min_nb chars (helper.rs:410-415):
                    // a few bytes in front of a malformed sequence cannot be any of these.
                    const bool may_yield = continue_str_if_possible || !invalid_bytes_after
                                           || split_end - split_start >= m_.c.chars_min_nb;
                    if (split_end > split_start && may_yield) {
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
                if (din < slen && soff + din >= hard_stop_) { stopped = true; break; }   // replay_exact_windows: the state as it is
                // a window boundary inside the slice: may we stop here?
                if (din < slen && !maybe_cut && may_drop(ob + dout - leftover_len, leftover_len) && !digit_pending()
                    && region_over(soff + din)) {
                    stopped = true;
                    break;
                }
            }
            // :330-338
            st.last_scan_run_leftover.assign((const char*)ob + dout - leftover_len, leftover_len);
            st.last_run_str_was_printed_and_is_maybe_cut_str = maybe_cut;
            pos = soff + din;
            if (stopped || pos >= hard_stop_) return pos;
            if (!maybe_cut && may_drop(ob + dout - leftover_len, leftover_len) && !digit_pending() && region_over(pos)) return pos;
        }
        return pos;
    }

    // gb18030: lead + digit (+ lead) pending.  If the token ends in an error the digit is read again and is a character of
    // the NEXT window although its byte lies in this one (its run ends at the window edge: no run crosses it) — the region
    // goes on.  (A plain pending lead byte is different: the character it begins has its last byte in the next window.)
    bool digit_pending() { return enc_is_gb(m_.c.encoding) && st_->decoder.raw().gb2 != 0; }

    // May the replay stop although this leftover is carried?  Only a leftover of fewer than
    // min(n,q) chars is inert (file comment).  A longer one is a long run that reached the
    // window edge (possibly followed by an incomplete character) and is still waiting to be
    // printed by the next decoder call (src/helper.rs:389-392).
    bool may_drop(const uint8_t* lo, size_t n) const {
        if (n == 0) return true;
        if (strict_) return false;
        if (n < m_.long_run) return true;
        size_t chars = 0;
        for (size_t i = 0; i < n; i++) chars += (lo[i] & 0xC0) != 0x80;
        return chars < m_.long_run;
    }

    // no cut string pending at window start p: is there nothing that forces the replay to continue right here?
    bool region_over(uint64_t p) {
        if (p < hard_stop_ && hard_stop_ != ~0ull) return false;   // replay_exact_windows: every window up to there
        while (ri_ < n_runs_ && runs_[ri_].end <= p) ri_++;
        // only a long run across p keeps the region going; one that begins at or behind p starts its own
        // (with -g also one that begins in the window at p: sx_replay_core.hpp regions_may_touch)
        if (ri_ < n_runs_ && (m_.c.grep_char < 0 ? runs_[ri_].start < p : window_start(runs_[ri_].start, W_) <= p)) return false;
        if (ri_ < n_runs_ && runs_[ri_].start == p && (runs_[ri_].chars & kPieceCont)) return false;  // a piece of the run that lies across p
        if (owns_tail_ && len_ && tail_start() <= p) return false;
        return true;
    }

    // The state handed to the next chunk must be exact, leftover content included (the next
    // chunk may complete it to a long run that neither chunk's device scan sees as one).  A
    // leftover not covered by a long run is < min(n,q) chars <= 4(q-1) bytes < 2 windows, so
    // starting three windows early makes it exact whatever was derived at the start.
    uint64_t tail_start() const { return back_windows(len_ - 1, W_, kLeadWindows); }

    const Mission& m_;
    ByteView& bytes_;
    const uint64_t len_;
    const int file_id_;
    const bool is_last_;
    const sx_run* runs_;
    const uint64_t n_runs_;
    const size_t W_;
    const uint64_t consumed0_, stream0_;
    std::vector<uint8_t> ob_;
    ScannerState* st_ = nullptr;
    MissionFindings* out_ = nullptr;
    uint64_t hi_ = 0, ri_ = 0;
    size_t hint_ = 0;
    bool owns_tail_ = false, strict_ = false;
    uint64_t hard_stop_ = ~0ull;   // replay_exact_windows: stop at this window start whatever is pending

public:
    // every window of [lo, hi) in turn from the exact state `st` at lo; `st` becomes the exact state at hi
    uint64_t run_exact(ScannerState& st, uint64_t lo, uint64_t hi, MissionFindings* out) {
        st_ = &st; out_ = out; hi_ = hi; owns_tail_ = false; strict_ = true; ri_ = n_runs_;
        hard_stop_ = hi;
        const uint64_t pos = lo < hi ? scan_from(lo) : lo;
        st.consumed_bytes = consumed0_ + pos;
        st.stream_bytes = stream0_ + pos;
        return pos;
    }
};

}  // namespace

void replay_plan_range(uint64_t lo, uint64_t hi, unsigned max_parts, std::vector<uint64_t>* bounds) {
    // partition boundaries on the slice grid; a partition is worth a thread from ~4 MiB on
    bounds->clear();
    const uint64_t len = hi > lo ? hi - lo : 0;
    uint64_t parts = std::min<uint64_t>(max_parts ? max_parts : 1, len / (4u << 20));
    if (parts < 1) parts = 1;
    const uint64_t per = (len / parts + kInputBufLen - 1) / kInputBufLen * kInputBufLen;
    for (uint64_t k = 0; k < parts; k++) bounds->push_back(std::min(hi, lo + k * per));
    bounds->push_back(hi);
}
void replay_plan(uint64_t len, unsigned max_parts, std::vector<uint64_t>* bounds) { replay_plan_range(0, len, max_parts, bounds); }

void replay_part(const Mission& m, const ScannerState& entry, uint64_t consumed0, uint64_t stream0, ByteView& bytes,
                 uint64_t len, int file_id, bool is_last, const sx_run* runs, uint64_t n_runs, uint64_t lo, uint64_t hi,
                 bool entry_exact, ReplayPart* part) {
    part->state = entry;
    if (!entry_exact) part->state.decoder.reset(m.c.encoding);
    RangeReplay rr(m, bytes, len, file_id, is_last, runs, n_runs, consumed0, stream0);
    std::vector<RegionLog> log;
    part->end_pos = rr.run(part->state, lo, hi, entry_exact, hi >= len, &part->findings, &log);
    part->regions.clear();
    for (const RegionLog& r : log) part->regions.push_back({ r.start, r.end, r.f0, r.f1 });
}

void replay_exact_windows(const Mission& m, ScannerState& st, uint64_t consumed0, uint64_t stream0, ByteView& bytes, uint64_t len,
                          int file_id, uint64_t lo, uint64_t hi, MissionFindings* out, bool is_last) {
    RangeReplay rr(m, bytes, len, file_id, is_last, nullptr, 0, consumed0, stream0);
    (void)rr.run_exact(st, lo, hi < len ? hi : len, out);
}

// Serial verification of speculative parts (part k assumed "nothing carried" at its start):
// keep a part's regions only from where the exact replay before it is clean and idle;
// where a speculative region straddles that point, replay exactly from there (rare).
void replay_stitch(const Mission& m, ScannerState& st, uint64_t consumed0, uint64_t stream0, ByteView& bytes,
                   uint64_t len, int file_id, bool is_last, const sx_run* runs, uint64_t n_runs,
                   std::vector<ReplayPart>& parts, MissionFindings* out, unsigned copy_threads, uint64_t* end_pos) {
    // 1. decide (serially, cheap) which finding ranges survive, repairing where needed
    struct Seg { const MissionFindings* src; size_t f0, f1; };
    std::vector<Seg> segs;
    std::deque<ReplayPart> fixes;
    uint64_t E = 0;
    ScannerState cur = st;
    for (size_t k = 0; k < parts.size(); k++) {
        ReplayPart& p = parts[k];
        out->replay_bytes += p.findings.replay_bytes;
        size_t i = 0;
        if (k > 0) {
            for (;;) {
                while (i < p.regions.size() && p.regions[i].start < E) i++;
                if (i == 0 || p.regions[i - 1].end <= E) break;
                // region i-1 began under a wrong assumption and reaches beyond E: redo it exactly
                fixes.emplace_back();
                ReplayPart& fix = fixes.back();
                replay_part(m, cur, consumed0, stream0, bytes, len, file_id, is_last, runs, n_runs, E,
                            std::min(len, p.regions[i - 1].end), false, &fix);
                segs.push_back({ &fix.findings, 0, fix.findings.v.size() });
                out->replay_bytes += fix.findings.replay_bytes;
                E = std::max(E, fix.end_pos);
                cur = fix.state;
            }
        }
        const bool kept = i < p.regions.size();
        if (kept) segs.push_back({ &p.findings, p.regions[i].f0, p.regions.back().f1 });
        if (k == 0 || kept || p.end_pos > E) {
            if (k == 0 || kept) cur = p.state;
            E = std::max(E, p.end_pos);
        }
    }
    cur.consumed_bytes = consumed0 + E;
    cur.stream_bytes = stream0 + E;
    st = cur;
    if (end_pos) *end_pos = E;

    // 2. copy the surviving findings and their strings (parallel: the bulk of the work)
    std::vector<size_t> fbase(segs.size() + 1, 0), abase(segs.size() + 1, 0);
    for (size_t i = 0; i < segs.size(); i++) {
        const Seg& g = segs[i];
        size_t ab = 0;
        if (g.f1 > g.f0) {
            const sx_finding& a = g.src->v[g.f0];
            const sx_finding& b = g.src->v[g.f1 - 1];
            ab = (size_t)(b.str_off + b.str_len) - a.str_off;  // a worker appends strings in finding order
        }
        fbase[i + 1] = fbase[i] + (g.f1 - g.f0);
        abase[i + 1] = abase[i] + ab;
    }
    out->v.resize(fbase.back());
    out->arena.resize(abase.back());
    auto copy_seg = [&](size_t i) {
        const Seg& g = segs[i];
        if (g.f1 <= g.f0) return;
        const uint32_t src0 = g.src->v[g.f0].str_off;
        memcpy(&out->arena[abase[i]], g.src->arena.data() + src0, abase[i + 1] - abase[i]);
        for (size_t j = g.f0; j < g.f1; j++) {
            sx_finding f = g.src->v[j];
            f.str_off = (uint32_t)(abase[i] + (f.str_off - src0));
            out->v[fbase[i] + (j - g.f0)] = f;
        }
    };
    const size_t nt = std::min<size_t>(copy_threads ? copy_threads : 1, segs.size());
    if (nt <= 1 || fbase.back() < 50000) { for (size_t i = 0; i < segs.size(); i++) copy_seg(i); }
    else {
        std::atomic<size_t> next{ 0 };
        std::vector<std::thread> th;
        for (size_t t = 0; t < nt; t++)
            th.emplace_back([&]() { for (size_t i; (i = next.fetch_add(1)) < segs.size();) copy_seg(i); });
        for (auto& t : th) t.join();
    }
}

void replay_chunk(const Mission& m, ScannerState& st, ByteView& bytes, uint64_t len, int input_file_id,
                  bool is_last_input_buffer, const sx_run* runs, uint64_t n_runs, MissionFindings* out) {
    const uint64_t consumed0 = st.consumed_bytes, stream0 = st.stream_bytes;
    std::vector<ReplayPart> parts(1);
    replay_part(m, st, consumed0, stream0, bytes, len, input_file_id, is_last_input_buffer, runs, n_runs, 0, len, true,
                &parts[0]);
    replay_stitch(m, st, consumed0, stream0, bytes, len, input_file_id, is_last_input_buffer, runs, n_runs, parts, out, 1, nullptr);
}

void replay_ranges(const Mission& m, const ScannerState& st, uint64_t len, const sx_run* runs, uint64_t n_runs,
                   unsigned parts, std::vector<std::pair<uint64_t, uint64_t>>* ranges) {
    if (len == 0) return;
    const size_t W = m.window;
    const size_t first = ranges->size();
    const uint64_t prime = m.is_dbcs() ? 96 : 16;  // double-byte: back to a byte outside the lead range (longer: fetched on demand)
    auto add = [&](uint64_t lo, uint64_t hi) {
        lo = lo > prime ? lo - prime : 0;  // decoder priming
        if (hi > len) hi = len;
        if (lo < hi) ranges->emplace_back(lo, hi);
    };
    for (uint64_t i = 0; i < n_runs; i++) {
        // a region starts in the window of the run's first byte and usually ends with the
        // window of its last byte; anything beyond is fetched on demand
        const uint64_t last = runs[i].end ? runs[i].end - 1 : 0;
        add(window_start(runs[i].start, W), window_end(last < len ? last : len - 1, W, len) + W);
    }
    // a handful of extras: carried state at the chunk start, exact tail, decoder priming at part starts
    const size_t mid = ranges->size();
    if (!st.clean()) add(0, 4 * W);
    std::vector<uint64_t> bounds;
    replay_plan(len, parts, &bounds);
    for (size_t k = 1; k + 1 < bounds.size(); k++) add(bounds[k], bounds[k] + 1);
    add(back_windows(len - 1, W, kLeadWindows), len);
    std::inplace_merge(ranges->begin() + first, ranges->begin() + mid, ranges->end());
}

void merge_device_runs(const DevRun* recs, size_t n, uint64_t min_chars, uint64_t subchunk, std::vector<sx_run>* out) {
    // A record starts inside the sub-chunk of the wavefront that wrote it, and a wavefront
    // writes in (almost) ascending order: bucket by sub-chunk, then order each small bucket.
    auto valid = [](const DevRun& r) { return !(r.len == kRecInvalidLen && r.chars_flags == kRecInvalidFlags); };
    if (subchunk == 0) subchunk = 1;
    size_t m = 0;
    uint64_t max_b = 0;
    for (size_t i = 0; i < n; i++)
        if (valid(recs[i])) { max_b = std::max(max_b, recs[i].start / subchunk); m++; }
    std::vector<uint32_t> head(max_b + 2, 0);
    for (size_t i = 0; i < n; i++)
        if (valid(recs[i])) head[recs[i].start / subchunk + 1]++;
    for (size_t b = 1; b < head.size(); b++) head[b] += head[b - 1];
    std::vector<DevRun> v(m);
    {
        std::vector<uint32_t> cur(head.begin(), head.end() - 1);
        for (size_t i = 0; i < n; i++)
            if (valid(recs[i])) v[cur[recs[i].start / subchunk]++] = recs[i];
    }
    auto less = [](const DevRun& a, const DevRun& b) { return a.start < b.start || (a.start == b.start && a.len < b.len); };
    for (size_t b = 0; b + 1 < head.size(); b++)
        if (head[b + 1] - head[b] > 1 && !std::is_sorted(v.begin() + head[b], v.begin() + head[b + 1], less))
            std::sort(v.begin() + head[b], v.begin() + head[b + 1], less);
    out->clear();
    out->reserve(m);
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

void merge_sorted_device_runs(const DevRun* v, size_t n, uint64_t min_chars, std::vector<sx_run>* out) {
    out->clear();
    out->reserve(n / 2);
    for (size_t i = 0; i < n && v[i].start != ~0ull;) {
        sx_run r{ v[i].start, v[i].start + v[i].len, v[i].chars_flags & kRecCharsMask };
        bool open_end = (v[i].chars_flags & kRecEndOpen) != 0;
        size_t k = i + 1;
        while (open_end && k < n && v[k].start != ~0ull && (v[k].chars_flags & kRecStartOpen) && v[k].start == r.end) {
            r.end = v[k].start + v[k].len;
            r.chars += v[k].chars_flags & kRecCharsMask;
            open_end = (v[k].chars_flags & kRecEndOpen) != 0;
            k++;
        }
        if (r.chars >= min_chars) out->push_back(r);
        i = k;
    }
}

// Interleave the missions' findings of one buffer (src/main.rs k-merge order: slice by slice,
// position, then mission) and append them to `out` as one more segment.
void merge_findings(std::vector<MissionFindings>& per, const std::shared_ptr<PinnedPool>& pool, Result* out) {
    out->pool = pool;
    size_t nonempty = 0, which = 0;
    for (size_t k = 0; k < per.size(); k++) if (per[k].count()) { nonempty++; which = k; }
    auto release = [&](MissionFindings& mf) { if (mf.ext.p && pool) pool->give(mf.ext); mf.ext = {}; };
    if (nonempty <= 1) {  // nothing to interleave: hand the storage over as it is
        for (size_t k = 0; k < per.size(); k++) if (!(nonempty == 1 && k == which)) release(per[k]);
        if (nonempty == 1) {
            std::vector<MissionFindings> more = std::move(per[which].more);
            per[which].more.clear();
            out->segs.push_back(std::move(per[which])); per[which].ext = {};
            for (auto& m : more) { out->segs.push_back(std::move(m)); m.ext = {}; }
        }
        return;
    }
    // One segment normally.  str_off is 32 bits: when the Missions' strings together exceed 4 GiB the findings go out as
    // several segments (in print order, each with its own arena), cut where the running string size would pass 4 GiB
    // (ADVICE, round 1: the offsets used to wrap silently).
    size_t total = 0, bytes = 0;
    for (auto& mf : per) { total += mf.count(); bytes += mf.strings_len(); }
    std::vector<size_t> idx(per.size(), 0);
    uint64_t seg_cap = 0xFFFFFFF0ull;
    if (const char* e = getenv("SX_HOST_MERGE_SEG_BYTES")) seg_cap = std::max<uint64_t>(1, (uint64_t)atoll(e));   // (tests: results of several segments without gigabytes of strings)
    if (bytes <= seg_cap) {   // the usual case: the arenas are copied whole, the offsets rebased
        out->segs.emplace_back();
        MissionFindings& m = out->segs.back();
        m.v.reserve(total);
        m.arena.reserve(bytes);
        std::vector<size_t> base(per.size(), 0);
        for (size_t k = 0; k < per.size(); k++) { base[k] = m.arena.size(); m.arena.append(per[k].strings(), per[k].strings_len()); }
        for (;;) {
            int best = -1;
            for (size_t k = 0; k < per.size(); k++) {
                if (idx[k] >= per[k].count()) continue;
                if (best < 0) { best = (int)k; continue; }
                const sx_finding& a = per[k].data()[idx[k]];
                const sx_finding& b = per[best].data()[idx[best]];
                if (a.slice_index < b.slice_index || (a.slice_index == b.slice_index && a.position < b.position)) best = (int)k;
            }
            if (best < 0) break;
            sx_finding f = per[best].data()[idx[best]++];
            f.str_off += (uint32_t)base[best];
            m.v.push_back(f);
        }
        total = 0;   // done
    }
    size_t done = 0;
    while (done < total) {
        out->segs.emplace_back();
        MissionFindings& m = out->segs.back();
        m.v.reserve(std::min<size_t>(total - done, 1u << 24));
        for (;;) {
            int best = -1;
            for (size_t k = 0; k < per.size(); k++) {
                if (idx[k] >= per[k].count()) continue;
                if (best < 0) { best = (int)k; continue; }
                const sx_finding& a = per[k].data()[idx[k]];
                const sx_finding& b = per[best].data()[idx[best]];
                if (a.slice_index < b.slice_index || (a.slice_index == b.slice_index && a.position < b.position)) best = (int)k;
            }
            if (best < 0) break;
            const sx_finding& src = per[best].data()[idx[best]];
            if (!m.v.empty() && m.arena.size() + src.str_len > seg_cap) break;   // next segment
            sx_finding f = src;
            f.str_off = (uint32_t)m.arena.size();
            m.arena.append(per[best].strings() + src.str_off, src.str_len);
            m.v.push_back(f);
            idx[best]++; done++;
        }
    }
    for (auto& mf : per) release(mf);
}

// a packed segment as sx_finding records (sx_result_segment, the host-side merges and splices): once, by a few threads.  Two callers at once
// (ADVICE r4): one lock for all results — the records are filled into a vector of their own and swapped in when they are complete, so no
// reader ever sees a vector of the right size with half of its records.  (32 bytes per finding on top of the packed 16: callers that
// mind read sx_result_segment_packed.)
void MissionFindings::expand() const {
    static std::mutex mu;
    if (!packed) return;
    std::lock_guard<std::mutex> g(mu);
    if (expanded.size() == ext_nf) return;
    std::vector<sx_finding> out(ext_nf);
    const sx_finding16* src = data16();
    const SegInfo& si = *info;
    const size_t n = ext_nf;
    const unsigned nt = n < (1u << 20) ? 1u : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    auto work = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) out[i] = expand_finding(src[i], si); };
    if (nt <= 1) work(0, n);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, n * t / nt, n * (t + 1) / nt);
        for (auto& t : th) t.join();
    }
    expanded.swap(out);
}

bool Result::flatten(std::string* err) {
    if (segs.size() <= 1) return true;
    size_t total = 0, bytes = 0;
    for (auto& s : segs) { total += s.count(); bytes += s.strings_len(); }
    if (bytes > 0xFFFFFFFFull) { if (err) *err = "more than 4 GiB of strings: read the result segment by segment"; return false; }
    MissionFindings m;
    m.v.reserve(total);
    m.arena.reserve(bytes);
    for (auto& s : segs) {
        const uint32_t base = (uint32_t)m.arena.size();
        m.arena.append(s.strings(), s.strings_len());
        const sx_finding* f = s.data();
        for (size_t i = 0, n = s.count(); i < n; i++) { m.v.push_back(f[i]); m.v.back().str_off += base; }
    }
    release();
    segs.clear();
    segs.push_back(std::move(m));
    return true;
}

void print_findings(const std::vector<Mission>& missions, const Result& r, int n_inputs, int radix, bool no_metadata,
                    std::string* out) {
    char num[40];
    for (const MissionFindings& seg : r.segs)
    for (size_t fi = 0; fi < seg.count(); fi++) {
        const sx_finding f = seg.get(fi);   // (a packed segment is read as it is)
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
        out->append(seg.strings() + f.str_off, f.str_len);
    }
}

}  // namespace sx
