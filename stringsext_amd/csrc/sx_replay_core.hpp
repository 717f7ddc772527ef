// sx_replay_core.hpp — the body of the device-side exact replay (stage B); decoders and SplitStr
// come from sx_codec_core.hpp.  Included by
// sx_replay_dev.hip with SXD = `__device__ __forceinline__`; the test-only harness
// tests/native/replay_core_host.cpp includes it with SXD = `inline` so that the very same
// code can be compared region by region with the host replayer on a machine without GPU.
#pragma once
#include <stdint.h>

#include "sx_device.hpp"
#include "sx_codec_core.hpp"

namespace sx {


constexpr u32 kSliceLen = 4096;  // INPUT_BUF_LEN, src/input.rs:22

// ------------------------------------------------------------------------------------------
// Window grid (twin of sx_replay.cpp)
// ------------------------------------------------------------------------------------------
SXD u64 win_start(u64 p, u32 W) { const u64 s = p / kSliceLen * kSliceLen; return s + (p - s) / W * W; }
SXD u64 next_win_start(u64 p, u32 W) {  // start of the window after the one that holds p
    const u64 s = p / kSliceLen * kSliceLen;
    const u64 e = s + (p - s) / W * W + W;
    return e < s + kSliceLen ? e : s + kSliceLen;
}

// May a region end at a window start where a run begins, and the next one start right there?
SXD bool regions_may_touch(const ReplayParams& P) { return P.grep_char < 0; }
// A run whose window start lies inside or in front of the run before it is inside that run's region.
SXD bool run_is_chained(const ReplayParams& P, u64 i, u64 want) {
    if (i == 0) return false;
    return regions_may_touch(P) ? want < P.runs[i - 1].end : want <= next_win_start(P.runs[i - 1].end - 1, P.W);
}

// ------------------------------------------------------------------------------------------
// Shortcuts through bytes that cannot matter (device replay only; the host replayer decodes
// everything, and the two must agree — tests/test_replay_core.py, tests/test_gpu_parity.py).
//
// Between decoder calls the reference's loop state is "clean" when nothing is carried:
// no leftover, no pending cut, decoder idle.  From a clean call start p:
//  (A) if the next long run starts at rs > p inside this window, every decoder call that ends
//      before the run's own call yields nothing and leaves the state clean again (a call that
//      yields needs >= min(chars_min_nb, q) accepted chars in a row = a long run, and stage A
//      reported none there), so the replay may jump to the start of the run's call: the first
//      byte vs such that [vs, rs) decodes without error — found by walking BACK from rs;
//  (B) if no long run starts in the rest of this window, nothing is emitted up to its end and
//      the state there is what a region start derives (derive_at), so the replay jumps there.
// ------------------------------------------------------------------------------------------
// A copy of the buffer bytes [lo, hi) that is cheaper to read than the buffer (the kernels: the window's staging row in
// LDS, with the 16 bytes in front of the region's first window): the look-back walks below read single bytes, and from
// global memory every one of them is a round trip the lane waits for.
struct NearBytes {
    const u8* p = nullptr;
    u64 lo = 0, hi = 0;
};
struct ByteReader {
    const u8* far;
    NearBytes nr;
    SXD u8 operator[](u64 pos) const { return pos - nr.lo < nr.hi - nr.lo ? nr.p[pos - nr.lo] : far[pos]; }
    // a pointer to [pos, pos + n): the near copy if it holds all of it
    SXD const u8* span(u64 pos, u64 n) const { return (pos >= nr.lo && pos + n <= nr.hi) ? nr.p + (pos - nr.lo) : far + pos; }
};

// Start of the decoder call that contains the char boundary rs, not before the call start p.
// Returns p if everything in [p, rs) is valid (the call at p is the one).
template <int ENC>
SXD u64 call_start_before(const ReplayParams& P, u64 p, u64 rs, NearBytes nr = NearBytes{}) {
    const ByteReader bytes{ P.data, nr };
    u64 b = rs;
    if (ENC == 1) {
        while (b > p) {
            const u8 x = bytes[b - 1];
            if (x < 0x80) { b--; continue; }
            if (x >= 0xC0) break;  // a lead right before a boundary: truncated sequence
            // continuation bytes: find the lead
            u32 k = 1;
            u64 j = b - 1;
            bool found = false;
            while (j > p && k <= 3) {
                j--;
                const u8 y = bytes[j];
                if ((y & 0xC0) == 0x80) { k++; continue; }
                found = true;
                break;
            }
            if (!found) break;  // ran into p (or too many continuation bytes): stray continuation bytes after p
            const u8 lead = bytes[j];
            u32 need = 0;
            u8 lo = 0x80, hi = 0xBF;
            if (lead >= 0xC2 && lead <= 0xDF) need = 1;
            else if (lead >= 0xE0 && lead <= 0xEF) { need = 2; if (lead == 0xE0) lo = 0xA0; if (lead == 0xED) hi = 0x9F; }
            else if (lead >= 0xF0 && lead <= 0xF4) { need = 3; if (lead == 0xF0) lo = 0x90; if (lead == 0xF4) hi = 0x8F; }
            if (need != k) break;
            const u8 second = bytes[j + 1];
            if (second < lo || second > hi) break;
            b = j;
        }
        return b;
    }
    if (ENC == 2 || ENC == 3) {
        constexpr bool be = ENC == 3;
        if ((rs - p) & 1) return p;  // not on the unit grid of the call at p: no shortcut
        while (b >= p + 2) {
            const u32 u = be ? ((u32)bytes[b - 2] << 8) | bytes[b - 1] : ((u32)bytes[b - 1] << 8) | bytes[b - 2];
            if ((u & 0xF800) != 0xD800) { b -= 2; continue; }
            if ((u & 0xFC00) == 0xDC00) {  // low surrogate: valid only right after a high one
                if (b >= p + 4) {
                    const u32 h = be ? ((u32)bytes[b - 4] << 8) | bytes[b - 3] : ((u32)bytes[b - 3] << 8) | bytes[b - 4];
                    if ((h & 0xFC00) == 0xD800) { b -= 4; continue; }
                }
                break;
            }
            // an unpaired high surrogate: the decoder reads on into the unit after it before it
            // reports the error, so the replay starts at the surrogate itself
            b -= 2;
            break;
        }
        return b;
    }
    if (ENC == 4 || ENC == 5) {
        // double-byte: tokens cannot be told apart walking backwards; p is a clean call start (decoder
        // neutral), so walk forward and remember where the last malformed token ended
        DDecoder dd;
        ddec_reset(dd, (int)P.encoding, P.table);
        u8 sink[48];
        u64 at = p, vs = p;
        while (at < rs) {
            const u32 n = (u32)(rs - at < 12 ? rs - at : 12);
            const DStep r = ddecode<ENC>(dd, P.data + at, n, sink, sizeof sink, false);
            at += r.read;
            // a call starts here with an idle decoder (gb18030: not while bytes of a broken four-byte token wait in the queue —
            // their characters belong to the call that starts here, whose position this is: an earlier start stays the answer)
            if (r.result == RES_MALFORMED && dd.rq_n == 0) vs = at;
        }
        return vs;
    }
    if (!P.table) return p;  // x-user-defined: every byte is a character, the call at p is the one
    while (b > p) {
        const u8 x = bytes[b - 1];
        if (x >= 0x80 && P.table[x - 0x80] == 0) break;
        b--;
    }
    return b;
}

// ------------------------------------------------------------------------------------------
// Long runs are cut at the window starts they cross.
//
// Inside a stretch of accepted characters the reference's state at a window start B is a function of the
// stretch alone, PROVIDED its first character follows a rejected (valid) character decoded in the same window
// (then SplitStr has just reset: ok_s_p != inp_start_p, so nothing "completes" a previous string,
// helper.rs:327-330,353-355) and the Mission has no -g, no -r and n <= q:
//   * the chars of the run accumulate as leftover from window to window (helper.rs:389-392: a stretch that
//     touches the right edge and is shorter than q goes back to be filtered again) until a decoder call's
//     text reaches q chars; there the first piece is cut (maybe_cut), and every later piece of the call
//     touches inp_start_p with last_s_was_maybe_cut set, so it "completes" and is printed whatever its length
//     (helper.rs:349-421): from that window on the state at every window start is (no leftover, maybe_cut);
//   * so with C = chars of the run that complete in front of B:  C < q -> leftover = those C chars, no cut
//     pending;  C >= q -> no leftover, cut pending.  (A char is at most 4 bytes: B - run start >= 4q bytes
//     means C >= q without looking.)  The decoder's pending bytes at B follow from the bytes in front of B as ever.
// A run with this property is cut into PIECES at the window starts inside it; a piece that begins at such a
// window start carries kPieceCont | (its start - the run's start) in sx_run::chars.  Every piece is a region of
// its own (one window), so a long line of text no longer chains its windows into one serial replay.
// tests/test_replay_core.py checks the rule against the oracle on the CPU (pieces driven like the device drives them).
// ------------------------------------------------------------------------------------------
constexpr u64 kPieceCont = 1ull << 63;

SXD bool mission_splittable(const ReplayParams& P) {
    return P.grep_char < 0 && !P.same_block && P.chars_min_nb >= 1 && P.chars_min_nb <= P.q;
}
// window starts s with s < x, and the position of window start number idx (slices of 4096 bytes, windows of W inside)
SXD u64 win_starts_below(u64 x, u32 W) { const u64 wps = (kSliceLen + W - 1) / W; return x / kSliceLen * wps + (x % kSliceLen + W - 1) / W; }
SXD u64 win_start_no(u64 idx, u32 W) { const u64 wps = (kSliceLen + W - 1) / W; return idx / wps * kSliceLen + idx % wps * W; }

// Do the bytes right in front of rs form a valid character (rejected, or it would belong to the run)?
template <int ENC>
SXD bool valid_char_before(const ReplayParams& P, u64 rs) {
    const u8* bytes = P.data;
    if (rs == 0) return false;
    const u8 x = bytes[rs - 1];
    if (ENC == 1) {
        if (x < 0x80) return true;
        if (x >= 0xC0) return false;
        u32 k = 1;
        u64 j = rs - 1;
        bool found = false;
        while (j > 0 && k <= 3) {
            j--;
            if ((bytes[j] & 0xC0) == 0x80) { k++; continue; }
            found = true;
            break;
        }
        if (!found) return false;
        const u8 lead = bytes[j];
        u32 need = 0;
        u8 lo = 0x80, hi = 0xBF;
        if (lead >= 0xC2 && lead <= 0xDF) need = 1;
        else if (lead >= 0xE0 && lead <= 0xEF) { need = 2; if (lead == 0xE0) lo = 0xA0; if (lead == 0xED) hi = 0x9F; }
        else if (lead >= 0xF0 && lead <= 0xF4) { need = 3; if (lead == 0xF0) lo = 0x90; if (lead == 0xF4) hi = 0x8F; }
        if (need != k) return false;
        const u8 second = bytes[j + 1];
        return second >= lo && second <= hi;
    }
    if (ENC == 2 || ENC == 3) {
        constexpr bool be = ENC == 3;
        if (rs < 2) return false;
        const u32 u = be ? ((u32)bytes[rs - 2] << 8) | bytes[rs - 1] : ((u32)bytes[rs - 1] << 8) | bytes[rs - 2];
        if ((u & 0xF800) != 0xD800) return true;
        if ((u & 0xFC00) != 0xDC00 || rs < 4) return false;
        const u32 h = be ? ((u32)bytes[rs - 4] << 8) | bytes[rs - 3] : ((u32)bytes[rs - 3] << 8) | bytes[rs - 4];
        return (h & 0xFC00) == 0xD800;
    }
    if (ENC == 4 || ENC == 5) return x < 0x80;   // an ASCII byte is a character whatever stands in front of it (own, trail, or given back)
    return x < 0x80 || !P.table || P.table[x - 0x80] != 0;
}

// bytes of the character that starts at rs (a valid one: the first of a run)
template <int ENC>
SXD u32 char_len_at(const ReplayParams& P, u64 rs) {
    const u8* b = P.data + rs;
    if (ENC == 1) return b[0] < 0x80 ? 1u : b[0] < 0xE0 ? 2u : b[0] < 0xF0 ? 3u : 4u;
    if (ENC == 2) return (b[1] & 0xFC) == 0xD8 ? 4u : 2u;
    if (ENC == 3) return (b[0] & 0xFC) == 0xD8 ? 4u : 2u;
    if (ENC == 4 || ENC == 5) return dbcs_token_len<(ENC == 4 || ENC == 5) ? ENC : 4>(b, P.len - rs, (int)P.encoding);
    return 1;
}

// gb18030 / GBK: stage A reports a SUPERSET of the runs there (ScanParams::gb4), and the pieces' premise is a run of accepted
// characters that begins where the decoder is neutral.  So a run is only cut if that can be seen: the byte in front of it is
// ASCII and no digit (a character of its own, rejected or it would belong to the run; the decoder is neutral behind it), and
// the run — at most kGbVerifyMax bytes, a lane decodes it alone — decodes without error into accepted characters only and ends
// with the decoder neutral.  Anything else is replayed as one run, as in round 1.
constexpr u64 kGbVerifyMax = 2048;
SXD bool gb_run_is_exact(const ReplayParams& P, u64 rs, u64 re) {
    if (rs == 0 || re - rs > kGbVerifyMax) return false;
    const u8 x = P.data[rs - 1];
    if (x >= 0x80 || gb_digit(x)) return false;
    DDecoder d;
    ddec_reset(d, (int)P.encoding, P.table);
    u8 sink[48];
    u64 at = rs;
    while (at < re) {
        const u32 n = (u32)(re - at < 12 ? re - at : 12);
        const DStep r = ddec_gb18030(d, P.data + at, n, sink, sizeof sink, false);
        if (r.result != RES_INPUT_EMPTY) return false;
        for (u32 w = 0; w < r.written;) {
            const u8 lead = sink[w];
            if (!pass_lead(P, lead)) return false;
            w += lead < 0x80 ? 1 : lead < 0xE0 ? 2 : lead < 0xF0 ? 3 : 4;
        }
        at += r.read;
    }
    return d.dlead == 0 && d.gb2 == 0 && d.rq_n == 0;
}

// How many pieces run i of the UNSPLIT list P.runs is cut into (1: not cut).  The run's first character must be
// delivered in the same window as the rejected character in front of it (a character belongs to the window that
// holds its LAST byte): no window start in [rs, last byte of the first character].
template <int ENC>
SXD u64 split_count(const ReplayParams& P, u64 i) {
    if (!mission_splittable(P)) return 1;
    const u64 rs = P.runs[i].start, re = P.runs[i].end;
    const u64 inside = win_starts_below(re, P.W) - win_starts_below(rs + 1, P.W);   // window starts in (rs, re)
    if (inside == 0) return 1;
    const u64 fe = rs + char_len_at<ENC>(P, rs) - 1;
    if (fe >= re || win_start(fe, P.W) >= rs) return 1;
    if (ENC == 4 && enc_is_gb((int)P.encoding)) return gb_run_is_exact(P, rs, re) ? inside + 1 : 1;
    return valid_char_before<ENC>(P, rs) ? inside + 1 : 1;
}
// piece k (0 .. count-1) of run i
SXD sx_run split_piece(const ReplayParams& P, u64 i, u64 k, u64 count) {
    const sx_run r = P.runs[i];
    if (count <= 1) return r;
    const u64 first = win_starts_below(r.start + 1, P.W);   // number of the first window start behind the run's start
    sx_run o;
    o.start = k == 0 ? r.start : win_start_no(first + k - 1, P.W);
    o.end = k + 1 == count ? r.end : win_start_no(first + k, P.W);
    o.chars = k == 0 ? r.chars : (kPieceCont | (o.start - r.start));
    return o;
}
template <int ENC> SXD bool run_is_piece(const sx_run& r) { return (r.chars & kPieceCont) != 0; }

// The state at the start of a continuation piece (see above).  `from` = the bytes from the run's start up to the
// window start (delta bytes, delta < 4q: the caller handles the other case).  Returns the leftover's length in ob.
template <int ENC>
SXD u32 derive_in_run(u32 q, u32 encoding, const uint16_t* table, const u8* from, u32 delta, DDecoder& dec, u8* ob, u32 ob_cap,
                      bool* cut_pending) {
    ddec_reset(dec, (int)encoding, table);
    u32 w = 0, k = 0;
    while (k < delta) {   // no error can occur: the run is made of valid characters
        const DStep r = ddecode<ENC>(dec, from + k, delta - k, ob + w, ob_cap - w, false);
        k += r.read; w += r.written;
        if (r.result != RES_MALFORMED) break;
    }
    u32 chars = 0;
    for (u32 t = 0; t < w; t++) chars += (ob[t] & 0xC0) != 0x80;
    *cut_pending = chars >= q;
    return chars >= q ? 0u : w;
}

// Double-byte encodings: a token boundary in [lim, lim + 2] (never beyond `at`), so that decoding from it sees
// everything from lim on.  Found from the nearest byte outside the lead range in front of lim — the decoder
// is neutral right after such a byte — or from `floor`, where it is neutral too (a clean call start; at the
// buffer start after `skip0` bytes that finish the token pending on entry), jumping token by token from there.
// The walk back is as long as the stretch of lead-range bytes in front of lim.
// Round 5: the walk back is bounded — where stage A has published its token grid (grid: per sub-chunk of grid_sub bytes the hang-over at its
// first byte, ScanParams::grid_flags) it ends at the start of lim's sub-chunk, where the tokens are known to begin after that many bytes.
// (Before: to the nearest byte outside the lead range however far — inside a fill of lead-range bytes every region walked to its beginning.)
template <int ENC>
SXD u64 dbcs_sync_before(const u8* bytes, u64 len, u64 at, u64 floor, u32 skip0, u64 lim, int enc, const u32* grid = nullptr, u32 grid_sub = 0) {
    if (grid && grid_sub) {
        const u64 sub = lim / grid_sub, s0 = sub * grid_sub;
        if (s0 > floor) { const u32 v = grid[sub]; if (v & 1u) { floor = s0; skip0 = (v >> 1) & 3u; } }
    }
    u64 r = lim;
    // (four bytes per load while at least four are left: a lane's loads wait for one another — a byte per load was 70 ms for regions 150 KB
    // into a fill)
    while (r >= floor + 4) {
        u32 x;
        __builtin_memcpy(&x, bytes + r - 4, 4);
        if (!dbcs_may_be_pending_after<ENC>((u8)(x >> 24), enc)) break;
        if (!dbcs_may_be_pending_after<ENC>((u8)(x >> 16), enc)) { r -= 1; break; }
        if (!dbcs_may_be_pending_after<ENC>((u8)(x >> 8), enc)) { r -= 2; break; }
        if (!dbcs_may_be_pending_after<ENC>((u8)x, enc)) { r -= 3; break; }
        r -= 4;
    }
    while (r > floor && dbcs_may_be_pending_after<ENC>(bytes[r - 1], enc)) r--;
    if (r == floor) r += skip0;
    while (r < lim) r += dbcs_token_len<ENC>(bytes + r, len - r, enc);
    return r < at ? r : at;
}

// The state the reference carries into the window that starts at `at`, when nothing long
// crosses `at` (RangeReplay::derive_state): the decoder's pending bytes, and one accepted char
// as leftover if it is the last thing delivered before `at`.  Decodes from max(at - 8, floor).
// Not inlined: it runs once or twice per region, and as a call it keeps the replay loop's register
// pressure down.  (Inlined twice into the 64-VGPR replay kernels, one variant — UTF-16BE, count only —
// came out of the compiler wrong: results changed with the occupancy attribute alone.  See DESIGN.md §9.)
template <int ENC>
SXD_NOINLINE u32 derive_at(const ReplayParams& P, u64 at, u64 floor, DDecoder& dec, u8* ob, NearBytes nr = NearBytes{}) {
    const u8* bytes = P.data;
    const ByteReader rd{ P.data, nr };
    ddec_reset(dec, (int)P.encoding, P.table);
    // with -r the lead byte of the leftover's last multi-byte char matters (see RangeReplay::derive_state)
    const u64 back = P.same_block ? 4ull * P.long_run + 8 : 8;
    u64 p = at >= back ? at - back : 0;
    if ((ENC == 2 || ENC == 3) && ((P.stream0 + p) & 1)) p = p ? p - 1 : p + 1;
    if (p < floor) p = floor;
    if (p > at) p = at;
    if (ENC == 4 || ENC == 5) p = dbcs_sync_before<ENC>(bytes, P.len, at, floor, floor == 0 ? P.entry_skip : 0u, p, (int)P.encoding, P.grid_flags, P.grid_sub);
    u8 sink[40], last[4], mb[4];
    u32 last_len = 0, mb_len = 0;
    while (p < at) {   // in pieces: the sink is small
        const u32 n = (u32)(at - p < 12 ? at - p : 12);
        const u8* src = rd.span(p, n);
        u32 k = 0;
        for (;;) {
            const DStep r = ddecode<ENC>(dec, src + k, n - k, sink, sizeof sink, false);
            k += r.read;
            for (u32 w = 0; w < r.written;) {
                const u8 lead = sink[w];
                const u32 cl = lead < 0x80 ? 1 : lead < 0xE0 ? 2 : lead < 0xF0 ? 3 : 4;
                if (pass_lead(P, lead)) {
                    for (u32 t = 0; t < cl; t++) last[t] = sink[w + t];
                    last_len = cl;
                    if (cl > 1) { for (u32 t = 0; t < cl; t++) mb[t] = sink[w + t]; mb_len = cl; }
                } else { last_len = 0; mb_len = 0; }
                w += cl;
            }
            if (r.result == RES_INPUT_EMPTY) break;
            if (r.result == RES_MALFORMED) { last_len = 0; mb_len = 0; }
        }
        p += n;
    }
    u32 out = 0;
    if (last_len) {
        if (P.same_block && mb_len && last_len == 1) { for (u32 t = 0; t < mb_len; t++) ob[out++] = mb[t]; }
        for (u32 t = 0; t < last_len; t++) ob[out++] = last[t];
    }
    return out;
}

constexpr u32 kObCap = 4 * 64 + 3 * 128 + 16;  // leftover (<= 4q bytes) + one window's output; q <= 64
constexpr u32 kMaxWindow = 128;                 // W = 2q
// ... and for 64 < q <= 255 (round 4: the device replays those too, in instantiations of their own — QBIG —; rounds 1-3: the host, ten
// times slower on sparse input)
constexpr u32 kObCapBig = 4 * 255 + 3 * 510 + 16, kMaxWindowBig = 512;
constexpr u32 kBackBytes = 16;                  // of the buffer in front of a region's first window, staged with it (derive_at looks back 8)

// Copy of one window into private memory: 16 bytes per load where the source is aligned (it is, in the
// default geometry: buffers are 256-byte aligned and W = 128), bytes otherwise.
SXD void stage_window(const u8* src, u32 n, u8* win) {
    u32 k = 0;
    if (((uintptr_t)src & 15) == 0) {
        for (; k + 16 <= n; k += 16) *(uint4*)(win + k) = *(const uint4*)(src + k);
    } else if (((uintptr_t)src & 3) == 0) {
        for (; k + 4 <= n; k += 4) *(u32*)(win + k) = *(const u32*)(src + k);
    }
    for (; k < n; k++) win[k] = src[k];
}

// the same for a destination that is only 4-aligned (an LDS row, sx_replay_dev.hip win_row_bytes)
SXD void stage_window_words(const u8* src, u32 n, u8* win) {
    u32 k = 0;
    if (((uintptr_t)src & 15) == 0) {
        for (; k + 16 <= n; k += 16) {
            const uint4 v = *(const uint4*)(src + k);
            u32* w = (u32*)(win + k);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        }
    } else if (((uintptr_t)src & 3) == 0) {
        for (; k + 4 <= n; k += 4) *(u32*)(win + k) = *(const u32*)(src + k);
    }
    for (; k < n; k++) win[k] = src[k];
}

// One region.  MODE 0: count only; 1: write findings and strings at fout/aout; 2: count, and
// keep the output in the region's small cache slot (fout/aout) as long as it fits — o.pad says
// whether it did, so that the second pass is a copy for almost every region.
// The cache slot of a region: [cap_f findings][cap_b string bytes].  All replaying regions share one
// arena; the slot size is the arena divided by their number (between 160 bytes and 4 KiB), so that
// sparse input gets by with little memory and string-dense input (long regions, many findings
// each) still finds room for nearly every region's output.
struct CacheGeom { u32 slot_bytes, cap_f, cap_b; };
SXD CacheGeom cache_geom(u64 arena_bytes, u64 n_heads) {
    u64 sb = n_heads ? arena_bytes / n_heads : 4096;
    if (sb > 4096) sb = 4096;
    if (sb < 160) sb = 160;
    sb &= ~15ull;
    CacheGeom g;
    g.slot_bytes = (u32)sb;
    g.cap_f = (u32)(sb / 96);
    if (g.cap_f < 2) g.cap_f = 2;
    g.cap_b = g.slot_bytes - g.cap_f * (u32)sizeof(sx_finding);
    return g;
}
// EXT_WIN: the window's staging copy lies where the caller says (the kernels: a row in LDS) instead of
// in a private array.
template <int MODE, int ENC, bool EXT_WIN = false, bool QBIG = false>
SXD void replay_region(const ReplayParams& P, u64 i, ReplayRegionOut& o, sx_finding* fout, u8* aout, u64 abase,
                       u32 cap_f = 0, u32 cap_b = 0, u8* win_ext = nullptr) {
    const u8* bytes = P.data;
    const u64 len = P.len;
    const u32 W = P.W;
    const u64 want = win_start(P.runs[i].start, W);
    constexpr u32 kOb = QBIG ? kObCapBig : kObCap, kWinMax = QBIG ? kMaxWindowBig : kMaxWindow;
    u8 ob[kOb];
    // the staging row: [kBackBytes in front of the region's first window][the window]
    alignas(16) u8 win_own[EXT_WIN ? 16 : kWinMax + kBackBytes];
    u8* const row = EXT_WIN ? win_ext : win_own;
    u8* const win = row + kBackBytes;
    u64 staged = ~0ull;
    NearBytes near;   // what of the buffer the row holds right now
    DDecoder dec;

    // ---- derive the state the reference would carry into `want` (RangeReplay::derive_state)
    if (want >= kBackBytes) {   // the look-back of derive_at reads the row, not the buffer
        if (EXT_WIN) stage_window_words(bytes + want - kBackBytes, kBackBytes, row);
        else stage_window(bytes + want - kBackBytes, kBackBytes, row);
        near.p = row; near.lo = want - kBackBytes; near.hi = want;
    }
    u32 leftover_len;
    bool maybe_cut = false;
    if (P.runs[i].chars & kPieceCont) {   // a window start inside a run: the state is a function of the run (see kPieceCont)
        const u64 delta = P.runs[i].chars & ~kPieceCont;
        if (delta < 4ull * P.q) leftover_len = derive_in_run<ENC>(P.q, P.encoding, P.table, bytes + (want - delta), (u32)delta, dec, ob, kOb, &maybe_cut);
        else { (void)derive_at<ENC>(P, want, 0, dec, ob, near); leftover_len = 0; maybe_cut = true; }
    } else leftover_len = derive_at<ENC>(P, want, 0, dec, ob, near);

    u64 ri = i;  // first run not yet behind us
    u32 n_find = 0, n_bytes = 0, windows = 0;
    u32 status = kRegionOk;
    u64 pos = want;
    bool cached = MODE == 2;

    // region_over(p): nothing forces the replay to go on at window start p
    // A run that begins at or behind p is the business of the region that starts there: with nothing
    // pending at p, that region derives exactly this state — only a long run ACROSS p keeps this one
    // going.  Not with -g: a stretch of q chars without the grep char ends SplitStr's iteration for the
    // whole decoder call (helper.rs:410-415), so what follows it in the window is never carried, which
    // the bytes in front of p cannot tell; there a region only ends in front of a window without runs.
    // a continuation piece begins at p: the region that starts there derives the very state this one has reached
    const u64 n_look = P.n_look ? P.n_look : P.n_runs;
    auto piece_at = [&](u64 p) -> bool {
        while (ri < n_look && P.runs[ri].end <= p) ri++;
        return ri < n_look && P.runs[ri].start == p && (P.runs[ri].chars & kPieceCont);
    };
    auto region_over = [&](u64 p) -> bool {
        while (ri < n_look && P.runs[ri].end <= p) ri++;
        if (ri < n_look && (regions_may_touch(P) ? P.runs[ri].start < p : win_start(P.runs[ri].start, W) <= p)) return false;
        return true;
    };
    auto may_drop = [&](const u8* lo, u32 n) -> bool {
        if (n == 0) return true;
        if (n < P.long_run) return true;
        u32 chars = 0;
        for (u32 k = 0; k < n; k++) chars += (lo[k] & 0xC0) != 0x80;
        return chars < P.long_run;
    };

    bool done = false;
    while (!done && pos < len) {
        const u64 soff = pos / kSliceLen * kSliceLen;
        const u32 slen = (u32)(len - soff < kSliceLen ? len - soff : kSliceLen);
        const u64 consumed = P.consumed0 + soff;
        const u32 slice_index = (u32)(soff / kSliceLen) + P.slice_base;
        u32 din = (u32)(pos - soff);
        bool is_last_window = false;
        // the leftover sits at the front of ob (the reference copies it there per slice, :101-114;
        // here it is moved there per window — strings are copied out at once, so only ob[0..] matters)
        while (din < slen) {
            u32 dend;
            if (din + W < slen) dend = din + W; else { is_last_window = true; dend = slen; }
            if (++windows > P.max_windows) { status = kRegionTooLong; done = true; break; }
            const u32 wb = din;  // window start (din moves on inside the window)
            u32 dout = leftover_len;
            for (;;) {  // 'decoder
                // the window's bytes go through a copy made with 16-byte loads (the kernels: a row in LDS): the decoders
                // and the look-back walks of the shortcuts read byte by byte, and a byte load per lane from 64 different
                // cache lines is the slowest way to read
                if (staged != soff + wb) {
                    if (EXT_WIN) stage_window_words(bytes + soff + wb, dend - wb, win);
                    else stage_window(bytes + soff + wb, dend - wb, win);
                    staged = soff + wb;
                    if (near.p == row && staged == want) near.hi = soff + dend;   // the first window: the row holds the bytes in front of it too
                    else { near.p = win; near.lo = staged; near.hi = soff + dend; }
                }
                if (P.skip && leftover_len == 0 && !maybe_cut && din < dend && ddec_idle<ENC>(dec)) {
                    const u64 p = soff + din, wend = soff + dend;
                    while (ri < n_look && P.runs[ri].end <= p) ri++;
                    const u64 rs = ri < n_look ? P.runs[ri].start : ~0ull;
                    if (rs >= wend) {  // (B) nothing long starts in the rest of this window
                        // Nothing is pending there and no long run lies across wend (it would begin before
                        // wend): the region ends at wend, the state is never read.  (With -g only if no run
                        // begins in the next window either, and the derived leftover — one char — is droppable.)
                        if (regions_may_touch(P) || (P.long_run > 1 && (ri >= n_look || win_start(rs, W) > wend))) {
                            leftover_len = 0; dout = 0; din = dend;
                            ddec_reset(dec, (int)P.encoding, P.table);
                            break;
                        }
                        leftover_len = derive_at<ENC>(P, wend, p, dec, ob, near);
                        dout = leftover_len;
                        din = dend;
                        break;
                    }
                    if (rs > p) {      // (A) jump to the call that holds the next long run
                        const u64 vs = call_start_before<ENC>(P, p, rs, near);
                        if (vs > p) din = (u32)(vs - soff);
                    }
                }
                const DStep r = ddecode<ENC>(dec, win + (din - wb), dend - din, ob + dout, kOb - dout, false);
                if (r.result == RES_OUTPUT_FULL) { status = kRegionTooLong; done = true; break; }
                u8 precision = SX_PRECISION_EXACT;
                if (r.written > 0 && din == 0 && (ob[dout] & 0x80)) {  // slice-start probe, :176-207
                    DDecoder fresh;
                    ddec_reset(fresh, (int)P.encoding, P.table);
                    u8 probe[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
                    const u32 pn = slen < 32 ? slen : 32;
                    const DStep pr = ddecode<ENC>(fresh, bytes + soff, pn, probe, 8, true);
                    const u32 filled = dout + r.written < 8 ? dout + r.written : 8;
                    bool same = pr.written != 0;
                    for (u32 t = 0; t < pr.written && same; t++) same = (t < filled ? ob[t] : 0) == probe[t];
                    if (!same) precision = SX_PRECISION_BEFORE;
                }
                u32 split_start = dout;
                const u32 split_end = dout + r.written;
                if (leftover_len > 0) { split_start -= leftover_len; leftover_len = 0; precision = SX_PRECISION_BEFORE; }
                const bool invalid_after = r.result == RES_MALFORMED;  // is_last_input_buffer is never set here
                const bool continue_str = maybe_cut;
                maybe_cut = false;
                const bool may_yield = continue_str || !invalid_after || split_end - split_start >= P.chars_min_nb;
                if (split_end > split_start && may_yield) {
                    DSplit it{ ob + split_start, ob + split_end, ob + split_start, continue_str, invalid_after };
                    DChunk ch;
                    while (dsplit_next(P, it, ch)) {
                        if (!ch.again) {
                            bool put = MODE != 0;
                            if (MODE == 2) { put = n_find < cap_f && n_bytes + ch.len <= cap_b; cached = cached && put; }
                            if (put) {
                                sx_finding f;
                                f.position = consumed + din;
                                f.str_off = (u32)(abase + n_bytes);
                                f.str_len = ch.len;
                                f.precision = precision;
                                f.completes_previous = ch.completes ? 1 : 0;
                                f.mission_id = (u8)P.mission_id;
                                f.reserved = 0;
                                f.input_file_id = (int16_t)P.file_id;
                                f.reserved2 = 0;
                                f.slice_index = slice_index;
                                fout[n_find] = f;
                                for (u32 t = 0; t < ch.len; t++) aout[n_bytes + t] = ch.s[t];
                            }
                            n_find++; n_bytes += ch.len;
                            leftover_len = 0;
                            maybe_cut = ch.maybe_cut;
                        } else {
                            leftover_len = ch.len;
                            maybe_cut = false;
                        }
                        precision = SX_PRECISION_AFTER;
                    }
                }
                dout += r.written;
                din += r.read;
                if (r.result == RES_INPUT_EMPTY) break;
            }
            if (done) break;
            // move the leftover to the front for the next window / slice
            if (leftover_len) {
                const u32 from = dout - leftover_len;
                if (from) for (u32 t = 0; t < leftover_len; t++) ob[t] = ob[from + t];
            }
            // (gb18030 with lead + digit pending: if the token ends in an error the digit is a character of the NEXT window — go on)
            const bool digit_pending = (ENC == 4) && enc_is_gb((int)P.encoding) && dec.gb2 != 0;
            if (piece_at(soff + din) || (!maybe_cut && !digit_pending && may_drop(ob, leftover_len) && region_over(soff + din))) { done = true; break; }
        }
        pos = soff + din;
        (void)is_last_window;
    }
    o.end = pos;
    o.n_find = n_find;
    o.n_bytes = n_bytes;
    o.status = status;
    o.pad = cached ? 1u : 0u;
}

// the encoding family picked at run time (host-side test harness)
template <int MODE>
SXD void replay_region_any(const ReplayParams& P, u64 i, ReplayRegionOut& o, sx_finding* fout, u8* aout, u64 abase,
                           u32 cap_f = 0, u32 cap_b = 0) {
#define SX_RR(E)                                                                                  \
    do {                                                                                          \
        if (P.q > 64) replay_region<MODE, E, false, true>(P, i, o, fout, aout, abase, cap_f, cap_b); \
        else replay_region<MODE, E>(P, i, o, fout, aout, abase, cap_f, cap_b);                     \
    } while (0)
    switch (enc_family(P.encoding)) {
        case 1: SX_RR(1); break;
        case 2: SX_RR(2); break;
        case 3: SX_RR(3); break;
        case 4: SX_RR(4); break;
        case 5: SX_RR(5); break;
        default: SX_RR(0); break;
    }
#undef SX_RR
}


}  // namespace sx
