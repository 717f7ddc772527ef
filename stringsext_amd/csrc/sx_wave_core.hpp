// sx_wave_core.hpp — stage B as bit arithmetic: FindingCollection::from (reference
// src/finding_collection.rs:84-342) and SplitStr::next (src/helper.rs:206-433) restated over BIT MASKS of one
// decoder-input window, so that one lane replays one window in a few dozen instructions per string
// instead of decoding it byte by byte (sx_replay_core.hpp, which stays the path for everything this one
// does not cover).  Included by sx_wave_dev.hip as device code and by tests/native/wave_core_host.cpp
// as host code: the same source is compared with the oracle on the CPU before it runs on the GPU.
//
// What the reference does per window (W = 2q bytes, :120-131) is a function of
//   * which bytes END a character the decoder delivers in this window (E), which of those characters pass
//     the filter on their UTF-8 lead byte (A; src/mission.rs:333-348),
//   * where decoder calls start (CS): a call ends at a malformed sequence (:298-325) and the next one starts
//     behind it — `Finding::position` is the start of the CALL (:260),
//   * the state carried in: leftover chars (:101-116, 269-285) and the "maybe cut" flag (:240-241, 266-268).
// The classification that produces E / A / CS is stage A's, per byte and data-parallel; the state machine
// below is sequential per window but tiny.  Covered: Missions without -g and -r and with 1 <= n <= q <= 64
// (SplitStr then never abandons a call's text half way, helper.rs:410-415) — `wv_mission_ok`.
//
// State between windows: (lc, lb, lback, cut) = leftover chars / their UTF-8 bytes / source bytes from the
// leftover's first byte to the window start, and the cut flag.  It is a function of at most the three windows
// in front (a window's first stretch is the only thing that depends on what is carried in, and W = 2q bytes hold
// at least q/2 chars), which is what the kernels' warm-up and their verification rely on.
#pragma once
#include <stdint.h>

#include "sx_codec_core.hpp"

namespace sx {

typedef int32_t i32;

// ------------------------------------------------------------------------------------------
// 128-bit masks: bit i = byte i of the window
// ------------------------------------------------------------------------------------------
struct WvMask { u64 lo, hi; };

SXD u32 wv_ctz64(u64 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_ctzll(v);
#else
    return (u32)__builtin_ctzll(v);
#endif
}
SXD u32 wv_popc64(u64 v) { return (u32)__builtin_popcountll(v); }
SXD WvMask wm_zero() { return WvMask{ 0, 0 }; }
SXD WvMask wm_and(WvMask a, WvMask b) { return WvMask{ a.lo & b.lo, a.hi & b.hi }; }
SXD WvMask wm_andn(WvMask a, WvMask b) { return WvMask{ a.lo & ~b.lo, a.hi & ~b.hi }; }   // a & ~b
SXD WvMask wm_or(WvMask a, WvMask b) { return WvMask{ a.lo | b.lo, a.hi | b.hi }; }
SXD bool wm_any(WvMask a) { return (a.lo | a.hi) != 0; }
SXD u32 wm_popc(WvMask a) { return wv_popc64(a.lo) + wv_popc64(a.hi); }
SXD WvMask wm_bit(u32 i) { return i < 64 ? WvMask{ 1ull << i, 0 } : WvMask{ 0, 1ull << (i - 64) }; }
SXD bool wm_test(WvMask a, u32 i) { return i < 64 ? ((a.lo >> i) & 1) != 0 : ((a.hi >> (i - 64)) & 1) != 0; }
SXD u64 wv_low64(u32 n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // the n lowest bits
// bits [0, n)
SXD WvMask wm_below(u32 n) { return n <= 64 ? WvMask{ wv_low64(n), 0 } : WvMask{ ~0ull, wv_low64(n - 64) }; }
// bits [a, b), a <= b <= 128
SXD WvMask wm_range(u32 a, u32 b) { return wm_andn(wm_below(b), wm_below(a)); }
// lowest set bit at or above `from`; 128 if there is none
SXD u32 wm_next(WvMask m, u32 from) {
    if (from < 64) {
        const u64 l = m.lo & ~wv_low64(from);
        if (l) return wv_ctz64(l);
        return m.hi ? 64 + wv_ctz64(m.hi) : 128u;
    }
    if (from >= 128) return 128u;
    const u64 h = m.hi & ~wv_low64(from - 64);
    return h ? 64 + wv_ctz64(h) : 128u;
}
// (highest set bit at or below `at`) or -1
SXD i32 wm_prev(WvMask m, u32 at) {
    const WvMask b = wm_and(m, wm_below(at + 1));
    if (b.hi) return 127 - (i32)__builtin_clzll(b.hi);
    if (b.lo) return 63 - (i32)__builtin_clzll(b.lo);
    return -1;
}
SXD WvMask wm_shl1(WvMask m) { return WvMask{ m.lo << 1, (m.hi << 1) | (m.lo >> 63) }; }
// the k-th (1-based) set bit of m at or above `from`; 128 if there are fewer (halving by popcounts: a line of q chars is cut out of a
// stretch of multi-byte characters this way, helper.rs:237)
SXD u32 wm_select(WvMask m, u32 from, u32 k) {
    const WvMask r = wm_andn(m, wm_below(from));
    const u32 pl = wv_popc64(r.lo);
    u64 v;
    u32 base;
    if (k <= pl) { v = r.lo; base = 0; } else { v = r.hi; base = 64; k -= pl; }
    if (k == 0 || k > wv_popc64(v)) return 128u;
    u32 x = (u32)v;
    const u32 p32 = (u32)__builtin_popcount(x);
    if (k > p32) { k -= p32; x = (u32)(v >> 32); base += 32; }
#pragma unroll
    for (u32 h = 16; h >= 1; h >>= 1) {   // the k-th set bit of x lies in its low h bits, or behind them
        const u32 lowbits = x & ((1u << h) - 1u);
        const u32 c = (u32)__builtin_popcount(lowbits);
        if (k > c) { k -= c; x >>= h; base += h; } else x = lowbits;
    }
    return base;
}

// ------------------------------------------------------------------------------------------
// State carried from window to window, packed into 32 bits for the lane-to-lane exchange
// ------------------------------------------------------------------------------------------
// (round 5, -g: lg = the leftover holds the grep char — SplitStr walks the leftover's chars again when it is prepended, helper.rs:252-254)
// (-r: lm = the code — 1 .. 63, 0: none — of the lead byte of the leftover's last multi-byte character: where SplitStr's
// last_multi_char_leading_byte stands when it has walked the prepended leftover again, helper.rs:279-296.  Six bits where room is: 25, 27-28,
// 30-31 and 29 — lg's: with -g AND -r the code has five bits (Missions with more than 31 accepted multi-byte lead bytes keep the check per
// buffer) and the readers mask it; lback <= 4q + 3 needs nine.)
struct WvState { u32 lc, lb, lback, cut, lg, lm; };
SXD u32 wv_pack(const WvState& s) {
    return s.lc | (s.lb << 7) | (s.lback << 16) | (s.cut << 26) | (s.lg << 29) | ((s.lm & 1u) << 25) | (((s.lm >> 1) & 3u) << 27) | (((s.lm >> 3) & 3u) << 30) |
           (((s.lm >> 5) & 1u) << 29);
}
SXD WvState wv_unpack(u32 v) {
    return WvState{ v & 127u, (v >> 7) & 511u, (v >> 16) & 511u, (v >> 26) & 1u, (v >> 29) & 1u,
                    ((v >> 25) & 1u) | (((v >> 27) & 3u) << 1) | (((v >> 30) & 3u) << 3) | (((v >> 29) & 1u) << 5) };
}
constexpr u32 kWvPendBit = 1u << 27;   // on the state after a buffer's LAST window only, bits 27-28: bytes of the token it ends inside (two-byte family: 1; EUC-JP: 1 / 2)

struct WvParams { u32 q, n_min, grep, same; };   // grep: the Mission has -g (WvWin::GC says where its char stands); same: -r (WvWin::D, MBA)

// One window as the state machine sees it.
struct WvWin {
    WvMask E, A;     // char ends delivered in this window; those whose char passes the filter
    WvMask F;        // first bytes of the characters (multi-byte encodings; single byte: unused)
    WvMask CS;       // bit i (1 <= i < n): a decoder call starts at byte i
    WvMask LS;       // first bytes of the stretches of accepted chars that have >= n BYTES (so every stretch of >= n chars is among them)
    WvMask G;        // bytes of accepted characters (single byte: == A): LS's stretches; a candidate with too few CHARS is dropped by its popcount
    WvMask O2, O3;   // single byte: bytes whose UTF-8 form has 2 / 3 bytes (str_len = source bytes + popc(O2) + 2 popc(O3))
    WvMask O4;       // double byte: O2 / O3 / O4 = last bytes of the chars whose UTF-8 form has >= 2 / >= 3 / 4 bytes
    u32 n;           // bytes in the window
    u32 pre_empty;   // an empty decoder call at byte 0 precedes the first one (UTF-8: the byte a pending sequence rejects is read again)
    u32 tail_empty;  // a decoder call starts exactly at the window end: one more (empty) call in this window
    u32 head_back;   // bytes of the first delivered character that lie in front of the window (0..3)
    u32 probe_before;// the slice-start probe (:176-207) marks the first call's first chunk `Before`
    u32 slice_start; // the window is the first of its slice
    u32 probe_hb;    // double byte, first window of a slice: bytes of the first char in front of the slice (the probe is settled by decoding)
    u32 head_pend;   // double byte: bytes in front of the window that belong to a token still incomplete there (a pending lead byte: 0 / 1)
    u32 tail_pend;   // double byte: the window ends inside a token (its last byte is a lead byte waiting for its trail)
    WvMask PB;       // UTF-16: a call that starts here (in the masks) really starts two bytes later — its first character was kept from the call before
    WvMask GC;       // -g (WvParams::grep): the E bits of the characters that ARE the grep char (an ASCII character, accepted or not: helper.rs:252-254
                     // looks at it before the filter does)
    // -r (WvParams::same, round 5; helper.rs:279-296).  MBA: the E bits of the accepted multi-byte characters; D: those whose lead byte differs
    // from the one of the multi-byte character in front of them in the same call's text, that one accepted too (a rejected one, or the
    // call's start, clears last_multi_char_leading_byte) — a BREAK in front of them unless SplitStr began a new walk in between (wv_stretch_same).
    // mb0_e / mb0_code: the first call's first multi-byte character, if accepted (128: none): what stands in front of it is the leftover;
    // mbl_code: the lead code of the window's last accepted multi-byte character
    WvMask MBA, D;
    u32 mb0_e, mb0_code, mbl_code;
};

enum { WV_BEFORE = 0, WV_EXACT = 1, WV_AFTER = 2 };   // == SX_PRECISION_*
// A fourth value only between wv_call and the writer: "Exact unless the slice-start probe says Before", with the bytes / source
// bytes of the leftover the window began with in bits 8.. / 17..  (finding_collection.rs:176-207 compares the first 8 bytes of the
// slice's OUTPUT BUFFER with a fresh decoder's; after an empty first call — UTF-8: the byte a pending sequence rejects is read
// again — the leftover that call consumed still lies at the front of that buffer.)  wv_resolve_probe settles it from the bytes.
enum { WV_PROBE = 3 };
SXD u32 wv_probe_pack(u32 lb, u32 lback) { return WV_PROBE | (lb << 8) | (lback << 17); }
// (bits 27-28: bytes in front of the slice the decoder of the first call holds — `hb` —; bits 29-30, with a leftover at a second
// call at byte 0: bytes in front of the slice that were the pending token's, not the leftover's)
SXD u32 wv_probe_hb(u32 prec) { return (prec >> 27) & 3u; }
SXD u32 wv_probe_pend(u32 prec) { return (prec >> 29) & 3u; }

// (same_block: -r as far as it can matter — csrc/sx_mission.cpp drops it where at most one UTF-8 lead byte passes the filter, helper.rs:279-296)
// (round 5: -g is covered — a stretch, or a line of q chars cut out of one, counts only if it holds the grep char, and a line of q chars without
// it that neither completes the string before nor is carried on ends SplitStr's iteration for the rest of the call's text, helper.rs:410-415)
SXD bool wv_mission_ok(int grep_char, u32 same_block, u32 n_min, u32 q) {
    return grep_char < 128 && !same_block && n_min >= 1 && n_min <= q && q <= 64;
}

// ------------------------------------------------------------------------------------------
// -r (require_same_unicode_block, helper.rs:279-296; round 5).  SplitStr keeps the lead byte of the last multi-byte character that passed
// (last_multi_char_leading_byte: cleared when a walk — a next() call — begins and by a multi-byte character that does not pass; an ASCII
// character leaves it alone); a multi-byte character that passes but has ANOTHER lead byte is "rejected, and looked at again": the string in
// hand ends in front of it as if a rejected character stood there, and it begins the next one.  In masks: w.D holds such characters as if the
// variable were never cleared by a new walk (a property of the call's text alone), and a D bit is a break only if a multi-byte accepted
// character lies between the walk's beginning `r` (the call's start, or where the last chunk that was handed out ended) and it — the first
// multi-byte character of a walk never breaks.  What the leftover contributes is in the state (lm: the code of its last multi-byte
// character's lead byte) and meets the first call's first multi-byte character (w.mb0_e, w.mb0_code) here.
// One stretch of accepted chars [a, er) of the call in hand, sub-stretch by sub-stretch; otherwise wv_call's `stretch` (no -g: Missions with
// both stay on the other path).
// ------------------------------------------------------------------------------------------
template <int KIND, class EMIT>
struct WvSameCtx {
    const WvParams& P; const WvWin& w; WvState& st; EMIT& emit;
    WvMask Ec; u32 din, cend, n; bool inv_after;
    u32 lbytes, lback, lsrc, lm_in;
    bool grep, lg_in;           // -g as well (round 5, last): the stretch's chunks count only with the grep char (wv_call's rules)
    u32* prec; u32* cut_cend;   // (cut_cend: the stretch-by-stretch driver's; else nullptr)
    u32* r; bool* rv;           // the walk's beginning (a bound on E bits) / the leftover's characters still belong to the walk
};
// Returns true if SplitStr's iteration ends here for the whole call (-g: a line of q chars without the grep char that neither completes the
// string before nor is carried on, helper.rs:410-415).
template <int KIND, class EMIT>
SXD bool wv_stretch_same(const WvSameCtx<KIND, EMIT>& c, u32 a, u32 er, u32 pre, bool comp0) {
    constexpr bool BYTES = KIND == 0;
    const WvParams& P = c.P; const WvWin& w = c.w; WvState& st = c.st;
    const WvMask av = wm_and(c.Ec, wm_range(a, er));
    auto first_src = [&](u32 e) -> i32 { if (BYTES) return (i32)e; const i32 f = wm_prev(w.F, e); return f < 0 ? -(i32)w.head_back : f; };
    u32 at = a, carried = pre, carried_b = pre ? c.lbytes : 0u;
    bool comp = comp0;
    // the BREAKS of this stretch as the walk in hand sees them: D bits with a multi-byte accepted character between the walk's beginning and
    // them (again after every chunk that is handed out: a new walk begins there)
    auto breaks = [&]() -> WvMask {
        if (*c.rv && c.lm_in) {   // the leftover holds one: every D bit counts, and so does the first call's first multi-byte character if its lead byte is another
            WvMask Dx = w.D;
            if (w.mb0_e < 128 && w.mb0_code != c.lm_in) Dx = wm_or(Dx, wm_bit(w.mb0_e));
            return wm_and(Dx, av);
        }
        const u32 m0 = wm_next(w.MBA, *c.r);
        return m0 < 127 ? wm_and(wm_andn(w.D, wm_below(m0 + 1)), av) : wm_zero();
    };
    WvMask B = breaks();
    const bool rej_is_grep = c.grep && er < 128 && wm_test(w.GC, er);   // the rejected char behind the stretch is the grep char: it counts for the chunk it ends
    for (;;) {
        const WvMask left = wm_andn(av, wm_below(at));
        const u32 e_first = wm_next(left, 0);
        if (!carried && e_first >= 128) return false;
        // the sub-stretch in hand ends in front of the first break (its own first character never ends it; a leftover in front: it may)
        const u32 lower = carried ? e_first : e_first + 1;
        const u32 be = lower < 128 ? wm_next(B, lower) : 128u;
        const bool by_break = be < 128;
        const WvMask sub = wm_and(left, wm_below(be));
        const u32 cnt = carried + wm_popc(sub);
        const bool ends_by_rej = by_break || er < 128;
        const u32 pn = cnt < P.q ? cnt : P.q;
        const bool is_q = pn == P.q;
        const u32 rem = cnt - pn;
        const bool tr = rem == 0 && !ends_by_rej;
        const u32 inw = pn - carried;
        // the chunk's last character in the window (-g needs it before it knows whether the chunk counts)
        u32 last_e = at;
        if (inw) last_e = BYTES ? e_first + inw - 1 : (rem == 0 ? (u32)wm_prev(sub, 127) : wm_select(sub, at, inw));
        bool gok = true;
        if (c.grep) {   // (a break is a multi-byte character: it never is the grep char — only the rejected char at the stretch's end can count, wv_call)
            gok = (carried && c.lg_in) || (!is_q && rem == 0 && !by_break && rej_is_grep);
            if (!gok && inw) gok = wm_any(wm_and(w.GC, wm_range(e_first, last_e + 1)));
        }
        if (!is_q && !tr && !comp && (pn < P.n_min || !gok)) {   // helper.rs:315-330: dropped; the walk goes on (no new next(): its beginning stays)
            if (!by_break) return false;
            at = be; carried = 0; carried_b = 0;
            continue;
        }
        const bool maybe_cut = is_q || (tr && !c.inv_after);
        const bool again = !comp && tr && !c.inv_after && (!is_q || !gok);
        if (!comp && !again && (pn < P.n_min || !gok)) return is_q;   // helper.rs:410-415: None (what could follow stands behind a line of q chars: never looked at)
        const i32 src = carried ? -(i32)c.lback : first_src(e_first);   // (the chunk's first source byte: looked up only for chunks that count)
        u32 out_b = carried_b;
        i32 src_end = src + (i32)(carried ? (KIND == 1 ? c.lbytes : c.lsrc) : 0u);
        if (inw) {
            src_end = (i32)last_e + 1;
            if (KIND == 1) out_b = (u32)(src_end - src);
            else {
                const WvMask tr_ = wm_range(e_first, last_e + 1);
                out_b += inw + wm_popc(wm_and(w.O2, tr_)) + (BYTES ? 2 * wm_popc(wm_and(w.O3, tr_)) : wm_popc(wm_and(w.O3, tr_)) + wm_popc(wm_and(w.O4, tr_)));
            }
        }
        if (again) {   // carried (it touches the text's end): nothing follows
            const bool mb_here = inw && wm_any(wm_and(w.MBA, wm_range(e_first, last_e + 1)));
            st.lc = pn; st.lb = out_b; st.lback = (u32)((i32)c.n - src); st.cut = 0; st.lg = c.grep && gok ? 1u : 0u;
            st.lm = mb_here ? w.mbl_code : (carried ? c.lm_in : 0u);
            return false;
        }
        c.emit(c.din + (KIND == 3 && c.cend > c.din && wm_test(w.PB, c.din) ? 2u : 0u), *c.prec, comp, src, (u32)(src_end - src), out_b);
        st.lc = 0; st.lb = 0; st.lback = 0; st.lm = 0; st.lg = 0; st.cut = maybe_cut ? 1u : 0u;
        if (c.cut_cend) *c.cut_cend = c.cend;
        *c.prec = WV_AFTER;
        *c.r = inw ? last_e + 1 : at;   // a new walk begins behind what was handed out (the leftover alone: where the stretch stood)
        *c.rv = false;
        B = breaks();
        comp = is_q;   // helper.rs:418-421: what follows a full line touches inp_start_p with the cut flag up; what follows a break does not
        carried = 0; carried_b = 0;
        if (inw) at = last_e + 1;
    }
}

// One decoder call [din, cend) of the window: finding_collection.rs:146-290 with SplitStr::next (helper.rs:206-433) restated per
// STRETCH of accepted chars instead of per char.  With n <= q and no -g / -r, SplitStr's walk over a call's text comes to this:
//   * the text is a row of stretches of accepted chars separated by rejected ones; a stretch is cut into lines of q chars
//     (helper.rs:237) and a rest; every line is pushed; a rest that follows a line "completes" it (:365, :418-421) and is pushed
//     whatever its length;
//   * only the stretch at the very start of the text (the leftover belongs to it, finding_collection.rs:211-227) can complete a
//     string of the call before (`cont`, :240-241); a stretch behind a rejected char never touches inp_start_p (helper.rs:327-330);
//   * any other stretch counts if it has >= n chars (:317), or if it ends the text of a call that did not end in an error — then
//     it is carried as the leftover (:389-392) however short; everything else is dropped without a trace (:327-330).
// So a call costs: its first stretch, its LONG stretches (found through w.LS), its last stretch.
// EMIT(din, precision, completes, src_rel, src_len, out_len): src_rel = first source byte relative to the window start
// (negative: in front of it), out_len = bytes of the string.
// KIND 0: single-byte decoders (a char per byte; string bytes from O2 / O3); 1: UTF-8 (the string is the source bytes);
// 2: double-byte decoders (chars from E / F; string bytes from O2 / O3 / O4 at the chars' last bytes); 3: UTF-16 — as 2, no slice-start
// probe left open, and a call whose first character was kept from the call before (w.PB) reports its position two bytes later.
template <int KIND, class EMIT>
SXD void wv_call(const WvParams& P, const WvWin& w, WvState& st, u32 din, u32 cend, bool invalid_after, bool first_call, EMIT& emit,
                 u32 probe = 0) {
    const bool cont = st.cut != 0;   // :240-241: consumed by this call whatever it yields
    st.cut = 0;
    const u32 lrem = st.lc, lbytes = st.lb, lback = st.lback, lm_in = P.grep ? st.lm & 31u : st.lm;   // (-g AND -r: the state's bit 29 is lg's — five bits of lead code, sx_mission.cpp sees to it)
    const bool lg_in = st.lg != 0;
    const u32 lsrc = KIND >= 2 ? lback - (first_call || din == 0 ? w.head_pend : 0u) : lback;   // the leftover's own source bytes
    const bool has_left = lrem > 0;
    st.lc = 0; st.lb = 0; st.lback = 0; st.lg = 0; st.lm = 0;   // :211-227: the leftover is prepended, then gone
    const WvMask rng = wm_range(din, cend);
    const WvMask Ec = wm_and(w.E, rng);
    if (!has_left && !wm_any(Ec)) return;
    constexpr bool BYTES = KIND == 0;
    const WvMask Rc = wm_andn(Ec, w.A);   // rejected (valid) chars of this call
    u32 prec = (has_left || (first_call && w.probe_before)) ? WV_BEFORE : WV_EXACT;   // :146, 214-221, 176-207
    if (!BYTES && !has_left && (probe || (first_call && w.probe_hb))) {
        // a call at byte 0 of a slice whose first char is not ASCII: the probe runs (:176).  UTF-8 / a second call: against the
        // leftover the empty first call consumed; double-byte, first call: against a decoder that holds the lead byte of the slice before
        const u32 fe0 = wm_next(Ec, 0);
        const bool non_ascii = fe0 < 128 && (KIND == 1 ? !wm_test(w.F, fe0) : wm_test(w.O2, fe0));
        if (non_ascii) prec = probe ? probe : (wv_probe_pack(0, 0) | (w.probe_hb << 27));
    }

    // One stretch: `pre` chars carried in front of it (the leftover), its accepted chars = the E bits in [a, er).
    // comp0: its first piece completes the string before.  Returns true if SplitStr's iteration ends here for the whole call (-g: a line
    // of q chars without the grep char that neither completes the string before nor is carried on, helper.rs:410-415).
    const bool GREP = P.grep != 0;
    u32 walk_r = din;          // -r: where SplitStr's walk in hand began
    bool walk_rv = has_left;   // ... with the leftover's characters in it
    auto stretch = [&](u32 a, u32 er, u32 pre, bool comp0) -> bool {
        if (P.same) {   // -r: sub-stretch by sub-stretch (wv_stretch_same; with -g too)
            const WvSameCtx<KIND, EMIT> sc{ P, w, st, emit, Ec, din, cend, w.n, invalid_after, lbytes, lback, lsrc, lm_in, GREP, lg_in, &prec, nullptr, &walk_r, &walk_rv };
            return wv_stretch_same<KIND, EMIT>(sc, a, er, pre, comp0);
        }
        const WvMask av = wm_and(Ec, wm_range(a, er));
        const bool ends_by_rej = er < 128;          // a rejected char follows (else the call's text ends with it)
        const bool rej_is_grep = GREP && ends_by_rej && wm_test(w.GC, er);   // ... and it is the grep char: it counts for the stretch it ends (helper.rs:252-254)
        u32 rem = pre + wm_popc(av);
        u32 at = a;                                  // E bit of the next char to hand out
        i32 src = 0;                                 // first source byte of the next piece
        if (pre) src = -(i32)lback;
        else if (BYTES) src = (i32)a;
        else { const i32 f = wm_prev(w.F, a); src = f < 0 ? -(i32)w.head_back : f; }
        bool comp = comp0;
        u32 carried = pre, carried_b = pre ? lbytes : 0u;
        while (rem) {
            const u32 pn = rem < P.q ? rem : P.q;
            const bool is_q = pn == P.q;
            rem -= pn;
            const bool tr = rem == 0 && !ends_by_rej;                       // touches the end of the call's text
            if (!GREP && !is_q && !tr && !comp && pn < P.n_min) return false;   // helper.rs:315-330: dropped
            // the piece's chars inside the window: pn - carried of them, from E bit `at` on
            const u32 inwin = pn - carried;
            bool gok = true;
            if (GREP) {   // does the piece hold the grep char?  (the leftover's chars are walked again: lg; a piece of fewer than q chars that a rejected char ends: that char too)
                gok = (carried && lg_in) || (!is_q && rem == 0 && rej_is_grep);
                if (!gok && inwin) {
                    const u32 le = BYTES ? at + inwin - 1 : wm_select(av, at, inwin);
                    gok = wm_any(wm_and(w.GC, wm_range(at, le + 1)));
                }
                if (!is_q && !tr && !comp && (pn < P.n_min || !gok)) return false;   // helper.rs:315-330: dropped, the walk goes on behind the rejected char
            }
            const bool maybe_cut = is_q || (tr && !invalid_after);          // :353-355
            const bool again = !comp && tr && !invalid_after && (!is_q || !gok);   // :389-392 (without -g a line of q chars is never carried)
            if (!comp && !again && (pn < P.n_min || !gok)) return is_q;     // :410-415: None — the rest of the call's text is never looked at (it only has a rest behind a line of q chars)
            u32 last_e = at, out_b = carried_b;
            // (a leftover on its own: its source bytes; UTF-8: lback also counts the bytes of a character that was pending behind it)
            i32 src_end = src + (i32)(carried ? (KIND == 1 ? lbytes : lsrc) : 0u);
            if (inwin) {
                last_e = BYTES ? at + inwin - 1 : wm_select(av, at, inwin);
                src_end = (i32)last_e + 1;
                if (BYTES) {
                    const WvMask tr_ = wm_range(at, last_e + 1);
                    out_b += inwin + wm_popc(wm_and(w.O2, tr_)) + 2 * wm_popc(wm_and(w.O3, tr_));
                } else if (KIND >= 2) {
                    const WvMask tr_ = wm_range(at, last_e + 1);   // (the length bits sit on the chars' last bytes)
                    out_b += inwin + wm_popc(wm_and(w.O2, tr_)) + wm_popc(wm_and(w.O3, tr_)) + wm_popc(wm_and(w.O4, tr_));
                } else out_b = (u32)(src_end - src);    // UTF-8 in, UTF-8 out: the string is the source bytes
            }
            if (again) { st.lc = pn; st.lb = out_b; st.lback = (u32)((i32)w.n - src); st.cut = 0; st.lg = GREP && gok ? 1u : 0u; }   // finding_collection.rs:269-285
            else {                                                                                     // :255-268
                // (UTF-16: the empty call in front of byte 0 is the real call [0, 2) — it starts where it says)
                emit(din + (KIND == 3 && cend > din && wm_test(w.PB, din) ? 2u : 0u), prec, comp, src, (u32)(src_end - src), out_b);
                st.lc = 0; st.lb = 0; st.lback = 0; st.cut = maybe_cut ? 1u : 0u;
            }
            prec = WV_AFTER;   // :289
            comp = true;       // helper.rs:418-421: what follows a full line touches inp_start_p with the cut flag set
            carried = 0; carried_b = 0;
            src = src_end;
            if (inwin) at = last_e + 1;
        }
        return false;
    };

    // ---- the stretch at the start of the text
    const u32 fe = wm_next(Ec, 0);                                   // the call's first char
    const bool first_acc = fe < 128 && wm_test(w.A, fe);
    u32 pos = din;                                                   // stretches that begin below pos are done
    if (has_left || first_acc) {
        const u32 er = first_acc ? wm_next(Rc, fe) : (fe < 128 ? fe : 128u);   // leftover alone: the rejected char right behind it, or nothing
        if (stretch(first_acc ? fe : (fe < 128 ? fe : cend), er, lrem, cont)) return;
        pos = er < 128 ? er + 1 : 128u;
    }
    // ---- long stretches behind it
    const WvMask LSc = wm_and(w.LS, rng);
    // ---- the last stretch, if the text ends with accepted chars and the call did not end in an error: carried however short
    u32 tail_a = 128;
    if (!invalid_after) {
        const i32 el = wm_prev(Ec, 127);
        if (el >= 0 && wm_test(w.A, (u32)el)) {
            const i32 r = wm_prev(Rc, (u32)el);
            tail_a = wm_next(Ec, r < 0 ? 0u : (u32)r + 1);
        }
    }
    while (pos < 128) {
        u32 sb = wm_next(LSc, pos);                                  // first byte of the next stretch of >= n bytes
        u32 a = sb < 128 ? (BYTES ? sb : wm_next(Ec, sb)) : 128u;     // ... its first char
        if (tail_a >= pos && tail_a < a) a = tail_a;
        if (a >= 128) break;
        const u32 er = wm_next(Rc, a);
        if (stretch(a, er, 0u, false)) return;
        pos = er < 128 ? er + 1 : 128u;
    }
}

// One window: its decoder calls in order (finding_collection.rs:134-325).  Calls that hold no accepted char and
// meet no leftover only clear the cut flag: they are skipped in bulk.
template <int KIND, class EMIT>
SXD void wv_window_calls(const WvParams& P, const WvWin& w, WvState& st, EMIT& emit, bool skip_idle_calls = true) {
    u32 probe = 0;
    if (w.pre_empty) {
        if (KIND != 3 && w.slice_start && st.lb) probe = wv_probe_pack(st.lb, st.lback) | (KIND == 2 ? w.head_pend << 29 : 0u);
        wv_call<KIND>(P, w, st, 0u, 0u, true, false, emit);
    }
    u32 din = 0;
    bool first = true;
    for (;;) {
        if (skip_idle_calls && st.lc == 0 && !(first && (w.probe_before || probe))) {
            const u32 a = wm_next(w.A, din);
            if (a >= w.n) { st.cut = 0; return; }   // (at least the call at din is still to come, and none of them yields)
            const i32 cs = wm_prev(w.CS, a);        // start of the call that delivers the next accepted char
            if (cs > (i32)din) { st.cut = 0; din = (u32)cs; first = false; }
            // Nothing carried in at all (no leftover, no cut pending): a call that holds no stretch of >= n bytes and is not the
            // window's last one hands SplitStr a text of short stretches only — each is dropped (helper.rs:315-330), the last of
            // them too, because the call ended in an error (finding_collection.rs:269: only a call that ran into the window end
            // leaves a leftover).  On binary data that is nearly every call: go on at the next call that can matter.
            if (st.cut == 0) {
                const u32 ls = wm_next(w.LS, din);
                if (ls >= w.n && w.tail_empty) return;           // (the last real call ends in an error as well)
                i32 t = wm_prev(w.CS, 127);                      // the window's last call
                if (ls < w.n) { const i32 c = wm_prev(w.CS, ls); if (c < t) t = c; }
                if (t > (i32)din) { din = (u32)t; first = false; }
            }
        }
        u32 cend = wm_next(w.CS, din + 1);
        if (cend > w.n) cend = w.n;
        const bool last = cend >= w.n;
        wv_call<KIND>(P, w, st, din, cend, !last || w.tail_empty != 0, first, emit, din == 0 ? probe : 0u);
        first = false;
        if (last) break;
        din = cend;
    }
    if (w.tail_empty) wv_call<KIND>(P, w, st, w.n, w.n, false, false, emit);
}

// ------------------------------------------------------------------------------------------
// The same window, STRETCH BY STRETCH instead of call by call (round 4; what the kernels run).  wv_window_calls above walks the
// window's decoder calls — on binary data a call starts every three or four bytes and nearly none of them yields anything; the
// counters said 600 (single byte) to 1 800 (two-byte family) vector instructions per KiB for it.  What can yield is known from
// the masks without visiting the calls:
//   * the first call's text-start stretch (the leftover joins it, it may complete the string before):   once per window;
//   * stretches of >= n accepted characters:   the LS bits, wherever they stand;
//   * the stretch at the start of the call that follows an emission with the cut flag up (helper.rs:418-421):   `forced`, rare;
//   * the last call's last stretch, which becomes the leftover however short (helper.rs:389-392):   `tail_a`.
// The call a stretch belongs to is looked up when the stretch is visited (`din` = the CS bit in front of it: Finding::position,
// finding_collection.rs:260; Exact for a call's first finding, After for the others, :289), and what the calls in between
// would have done — clear the cut flag (:240-241) — is done when the walk crosses them.  wv_call's `stretch` is unchanged.
// tests/native/wave_core_host.cpp runs both drivers on every window it sees and requires the same emissions and the same exit state.
// ------------------------------------------------------------------------------------------
// The last call's last stretch, looked at once per window: where it starts (the end bit of its first char; 128: the text does not end
// with accepted chars, or the call ends in an error) and what it leaves behind when nothing is carried into it — fewer than q chars:
// they are the leftover (helper.rs:389-392); more: lines of q chars, the last piece touches the text end with the flag up.
struct WvTail { u32 a, state; };
// (GREPT: -g as a compile-time constant — the kernels are instantiated with and without it: WvWin::GC costs four registers that a Mission
// without -g must not pay for, and did, in spills: `-e ascii -n 4` 192 -> 146 GiB/s with -g decided at run time)
template <int KIND, bool GREPT, bool SAMET = false>
SXD WvTail wv_tail_g(const WvParams& P, const WvWin& w) {
    constexpr bool BYTES = KIND == 0;
    if (w.tail_empty || w.n == 0) return WvTail{ 128u, 0u };
    const i32 tcs = wm_prev(w.CS, 127);
    const WvMask El = wm_andn(w.E, wm_below(tcs < 0 ? 0u : (u32)tcs));
    const i32 el = wm_prev(El, 127);
    if (el < 0 || !wm_test(w.A, (u32)el)) return WvTail{ 128u, 0u };
    const i32 r = wm_prev(wm_andn(El, w.A), (u32)el);
    const u32 a = wm_next(El, r < 0 ? 0u : (u32)r + 1);
    // -r: a tail with a lead byte change inside is no plain leftover (wv_window walks it: tail_simple there is false); what it most likely
    // leaves is what stands behind the LAST change — the exchange of the entry states starts from that guess
    u32 a2 = a;
    if (SAMET) { const i32 d = wm_prev(wm_and(w.D, wm_range(a + 1, (u32)el + 1)), 127); if (d >= 0) a2 = (u32)d; }
    const WvMask rng = wm_range(a2, (u32)el + 1);
    const u32 c = wm_popc(wm_and(El, rng));
    // (with -g this is only a guess: a line of q chars without the grep char is no string, and may end the walk — the state is then
    // settled by wv_window like any other stretch's, tail_simple is false for it)
    if (c >= P.q) return WvTail{ a, wv_pack(WvState{ 0, 0, 0, 1, 0, 0 }) };
    i32 src;
    if (BYTES) src = (i32)a2;
    else { const i32 f = wm_prev(w.F, a2); src = f < 0 ? -(i32)w.head_back : f; }
    u32 out_b;
    if (KIND == 1) out_b = (u32)(el + 1 - src);
    else out_b = c + wm_popc(wm_and(w.O2, rng)) + (BYTES ? 2 * wm_popc(wm_and(w.O3, rng)) : wm_popc(wm_and(w.O3, rng)) + wm_popc(wm_and(w.O4, rng)));
    u32 lg = 0, lm = 0;
    if (GREPT) lg = wm_any(wm_and(w.GC, rng)) ? 1u : 0u;
    if (SAMET) lm = wm_any(wm_and(w.MBA, rng)) ? w.mbl_code : 0u;
    return WvTail{ a, wv_pack(WvState{ c, out_b, (u32)((i32)w.n - src), 0, lg, lm }) };
}
template <int KIND>
SXD WvTail wv_tail(const WvParams& P, const WvWin& w) {
    return P.grep ? (P.same ? wv_tail_g<KIND, true, true>(P, w) : wv_tail_g<KIND, true>(P, w)) : P.same ? wv_tail_g<KIND, false, true>(P, w) : wv_tail_g<KIND, false>(P, w);
}
// What the window hands on if what it was handed does not matter: every lane starts the exchange of the entry states from its
// predecessor's guess instead of from "nothing carried" (which is wrong behind every window that ends inside a line of text or in
// a stretch of accepted bytes: a third to all of them).  It is wrong when the tail is the text-start stretch of its call and something
// is carried into it (a leftover, the cut flag) — the loop that follows compares and repeats, so a wrong guess only costs a round
// (measured in the host harness: 1.00 rounds per batch on binary data, 1.01 on text).
template <int KIND>
SXD u32 wv_exit_guess(const WvParams& P, const WvWin& w) { return wv_tail<KIND>(P, w).state; }

template <int KIND, bool GREPT, bool SAMET, class EMIT>
SXD void wv_window_g(const WvParams& P, const WvWin& w, WvState& st, EMIT& emit, const WvTail& tail) {
    constexpr bool BYTES = KIND == 0;
    const u32 n = w.n;
    u32 probe = 0;
    // ---- an empty call in front of byte 0 (the byte a pending sequence rejected is read again): it takes the leftover and the cut flag
    if (w.pre_empty) {
        if (KIND != 3 && w.slice_start && st.lb) probe = wv_probe_pack(st.lb, st.lback) | (KIND == 2 ? w.head_pend << 29 : 0u);
        const bool cont = st.cut != 0;
        const u32 lc = st.lc, lb = st.lb, lback = st.lback;
        const bool lgp = st.lg != 0;
        st.lc = 0; st.lb = 0; st.lback = 0; st.cut = 0; st.lg = 0; st.lm = 0;
        if (lc && (cont || (lc >= P.n_min && (!GREPT || lgp))))   // (its text ends with the call, the call in an error: helper.rs:410-415)
            emit(0u, (u32)WV_BEFORE, cont, -(i32)lback, KIND == 1 ? lb : (KIND >= 2 ? lback - w.head_pend : lback), lb);
    }
    // ---- the call in hand
    u32 din = 0, cend = wm_next(w.CS, 1);
    if (cend > n) cend = n;
    bool inv_after = cend < n || w.tail_empty != 0;
    WvMask Ec = wm_and(w.E, wm_below(cend));
    const bool cont0 = st.cut != 0;
    st.cut = 0;
    const u32 lrem = st.lc, lbytes = st.lb, lback = st.lback, lm_in = GREPT ? st.lm & 31u : st.lm;   // (-g AND -r: bit 29 is lg's)
    const bool lg_in = st.lg != 0;
    const u32 lsrc = KIND >= 2 ? lback - w.head_pend : lback;
    const bool has_left = lrem > 0;
    st.lc = 0; st.lb = 0; st.lback = 0; st.lg = 0; st.lm = 0;
    constexpr bool GREP = GREPT;
    u32 walk_r = 0;            // -r: where SplitStr's walk in hand began (wv_stretch_same)
    bool walk_rv = has_left;   // ... with the leftover's characters in it
    u32 prec = (has_left || w.probe_before) ? WV_BEFORE : WV_EXACT;
    u32 cut_cend = 0;   // end of the call whose emission left st.cut up
    // the tail, taken alone, is just the leftover (-r: not if the lead byte changes inside it — tail.state is only a guess then)
    const bool tail_simple = wv_unpack(tail.state).lc != 0 && !(SAMET && tail.a < 127 && wm_any(wm_andn(w.D, wm_below(tail.a + 1))));

    // wv_call's stretch, for the call in hand: `pre` chars carried in front (the leftover), accepted chars = the E bits in [a, er).
    // It runs ONCE per trip of the loop below, and only for stretches that yield or carry (a wavefront pays for it whenever one lane needs it)
    // (returns true if SplitStr's iteration ends here for the whole call: -g, wv_call)
    auto stretch = [&](u32 a, u32 er, u32 pre, bool comp0) -> bool {
        if (SAMET) {   // -r: sub-stretch by sub-stretch (with -g too)
            const WvSameCtx<KIND, EMIT> sc{ P, w, st, emit, Ec, din, cend, n, inv_after, lbytes, lback, lsrc, lm_in, GREP, lg_in, &prec, &cut_cend, &walk_r, &walk_rv };
            return wv_stretch_same<KIND, EMIT>(sc, a, er, pre, comp0);
        }
        const WvMask av = wm_and(Ec, wm_range(a, er));
        const bool ends_by_rej = er < 128;
        const bool rej_is_grep = GREP && ends_by_rej && wm_test(w.GC, er);
        u32 rem = pre + wm_popc(av);
        u32 at = a;
        i32 src = 0;
        if (pre) src = -(i32)lback;
        else if (BYTES) src = (i32)a;
        else { const i32 f = wm_prev(w.F, a); src = f < 0 ? -(i32)w.head_back : f; }
        bool comp = comp0;
        u32 carried = pre, carried_b = pre ? lbytes : 0u;
        while (rem) {
            const u32 pn = rem < P.q ? rem : P.q;
            const bool is_q = pn == P.q;
            rem -= pn;
            const bool tr = rem == 0 && !ends_by_rej;
            if (!GREP && !is_q && !tr && !comp && pn < P.n_min) return false;   // helper.rs:315-330
            const u32 inw = pn - carried;
            bool gok = true;
            if (GREP) {
                gok = (carried && lg_in) || (!is_q && rem == 0 && rej_is_grep);
                if (!gok && inw) {
                    const u32 le = BYTES ? at + inw - 1 : (rem == 0 ? (u32)wm_prev(av, 127) : wm_select(av, at, inw));
                    gok = wm_any(wm_and(w.GC, wm_range(at, le + 1)));
                }
                if (!is_q && !tr && !comp && (pn < P.n_min || !gok)) return false;
            }
            const bool maybe_cut = is_q || (tr && !inv_after);              // :353-355
            const bool again = !comp && tr && !inv_after && (!is_q || !gok);   // :389-392
            if (!comp && !again && (pn < P.n_min || !gok)) return is_q;     // :410-415
            u32 last_e = at, out_b = carried_b;
            i32 src_end = src + (i32)(carried ? (KIND == 1 ? lbytes : lsrc) : 0u);
            if (inw) {
                // (a stretch that is one piece ends at its last char: no rank-select)
                last_e = BYTES ? at + inw - 1 : (rem == 0 ? (u32)wm_prev(av, 127) : wm_select(av, at, inw));
                src_end = (i32)last_e + 1;
                if (KIND == 1) out_b = (u32)(src_end - src);
                else {
                    const WvMask tr_ = wm_range(at, last_e + 1);
                    out_b += inw + wm_popc(wm_and(w.O2, tr_)) + (BYTES ? 2 * wm_popc(wm_and(w.O3, tr_)) : wm_popc(wm_and(w.O3, tr_)) + wm_popc(wm_and(w.O4, tr_)));
                }
            }
            if (again) { st.lc = pn; st.lb = out_b; st.lback = (u32)((i32)n - src); st.cut = 0; st.lg = GREP && gok ? 1u : 0u; }   // finding_collection.rs:269-285
            else {                                                                                   // :255-268
                emit(din + (KIND == 3 && wm_test(w.PB, din) ? 2u : 0u), prec, comp, src, (u32)(src_end - src), out_b);
                st.lc = 0; st.lb = 0; st.lback = 0; st.cut = maybe_cut ? 1u : 0u;
                cut_cend = cend;
            }
            prec = WV_AFTER;
            comp = true;
            carried = 0; carried_b = 0;
            src = src_end;
            if (inw) at = last_e + 1;
        }
        return false;
    };

    // ---- the first call's text-start stretch: the next trip's work (`have`), or settled here
    u32 pos = 0;   // stretches whose first char ends below pos are done
    bool have = false, comp0 = false;
    u32 a = 128, er = 128, pre = 0;
    if (has_left || wm_any(Ec)) {
        if (!BYTES && !has_left && (probe || w.probe_hb)) {   // wv_call: the slice-start probe stays open until the writer decodes
            const u32 fe0 = wm_next(Ec, 0);
            const bool non_ascii = fe0 < 128 && (KIND == 1 ? !wm_test(w.F, fe0) : wm_test(w.O2, fe0));
            if (non_ascii) prec = probe ? probe : (wv_probe_pack(0, 0) | (w.probe_hb << 27));
        }
        const u32 fe = wm_next(Ec, 0);
        const bool first_acc = fe < 128 && wm_test(w.A, fe);
        if (has_left || first_acc) {
            er = first_acc ? wm_next(wm_andn(Ec, w.A), fe) : (fe < 128 ? fe : 128u);
            a = first_acc ? fe : (fe < 128 ? fe : cend);
            const u32 total = lrem + (first_acc ? wm_popc(wm_and(Ec, wm_range(fe, er))) : 0u);
            if (!cont0 && total < P.n_min && (er < 128 || inv_after)) pos = er < 128 ? er + 1 : cend;   // dropped (helper.rs:315-330, 410-415)
            else if (!has_left && !cont0 && a == tail.a && tail_simple) { st = wv_unpack(tail.state); return; }   // the window is one call and ends inside its first stretch
            else { have = true; pre = lrem; comp0 = cont0; }
        }
    } else pos = cend;

    // ---- every stretch that can yield, in order.  The call behind an emission with the flag up, if it starts with an accepted char:
    // its first stretch completes the string before however short it is (helper.rs:418-421) -> the walk must stop there (`forced`)
    u32 forced = 128;
    for (;;) {
        if (!have) {
            // the next stretch worth a visit: a tight loop of its own — stretches of >= n BYTES with fewer than n CHARS (two-byte
            // characters on binary data: two in three) are dropped by their popcount before anything else is looked up
            for (;;) {
                const u32 sb = pos < 128 ? wm_next(w.LS, pos) : 128u;
                u32 m = 128;
                if (forced >= pos) m = forced;
                if (tail.a >= pos && tail.a < m) m = tail.a;
                if (!BYTES && sb < 128 && m >= sb) {
                    const u32 e = wm_next(WvMask{ ~w.G.lo, ~w.G.hi }, sb);   // first byte behind the stretch
                    if (m >= e && wm_popc(wm_and(w.A, wm_range(sb, e))) < P.n_min) { pos = e; continue; }
                }
                a = sb < 128 ? (BYTES ? sb : wm_next(w.E, sb)) : 128u;
                if (m < a) a = m;
                break;
            }
            if (a >= 128) break;
            // (-r: unless the leftover's multi-byte character still stands in the walk — its stretch was dropped — and the first one of the tail has another lead byte)
            const bool lead_meets = SAMET && walk_rv && lm_in && w.mb0_e < 128 && w.mb0_e > tail.a && w.mb0_code != lm_in;
            if (a == tail.a && tail_simple && st.cut == 0 && !lead_meets) { st = wv_unpack(tail.state); break; }   // the tail, nothing carried into it: the leftover
            comp0 = false;
            if (a >= cend) {   // another call: the ones walked over cleared the cut flag, the one right behind the emission takes it
                const i32 c = wm_prev(w.CS, a);
                const u32 d = c < 0 ? 0u : (u32)c;
                const bool cont = st.cut != 0 && d == cut_cend;
                st.cut = 0;
                din = d;
                cend = wm_next(w.CS, a + 1);
                if (cend > n) cend = n;
                inv_after = cend < n || w.tail_empty != 0;
                Ec = wm_and(w.E, wm_range(din, cend));
                prec = WV_EXACT;
                walk_r = din; walk_rv = false;
                comp0 = cont && wm_next(Ec, 0) == a;   // (only the stretch at the very start of the text: helper.rs:327-330)
            }
            er = wm_next(wm_andn(Ec, w.A), a);
            pre = 0;
        }
        have = false;
        const bool abandoned = stretch(a, er, pre, comp0);
        pos = abandoned ? cend : (er < 128 ? er + 1 : cend);   // (-g, a line of q chars without the grep char: nothing more of this call's text is looked at)
        forced = 128;
        if (st.cut && cend < n) {
            const u32 fe2 = wm_next(w.E, cend);
            if (fe2 < 128 && wm_test(w.A, fe2) && fe2 < wm_next(w.CS, cend + 1)) forced = fe2;
        }
    }
    // the cut flag outlives the window only if the window's last call raised it; an empty call at the window end takes it too
    if (st.cut && (cut_cend < n || w.tail_empty)) st.cut = 0;
}
template <int KIND, class EMIT>
SXD void wv_window(const WvParams& P, const WvWin& w, WvState& st, EMIT& emit, const WvTail& tail) {
    if (P.grep && P.same) wv_window_g<KIND, true, true>(P, w, st, emit, tail);
    else if (P.grep) wv_window_g<KIND, true, false>(P, w, st, emit, tail);
    else if (P.same) wv_window_g<KIND, false, true>(P, w, st, emit, tail);
    else wv_window_g<KIND, false, false>(P, w, st, emit, tail);
}
template <int KIND, class EMIT>
SXD void wv_window(const WvParams& P, const WvWin& w, WvState& st, EMIT& emit, bool = true) {
    const WvTail tail = wv_tail<KIND>(P, w);
    wv_window<KIND>(P, w, st, emit, tail);
}

// ------------------------------------------------------------------------------------------
// Classification, single-byte encodings (x-user-defined and the WHATWG tables): a class byte per input byte,
// bit 0 valid, bit 1 accepted, bit 2 / 3 the UTF-8 form has 2 / 3 bytes.  16 bytes -> four 16-bit masks.
// ------------------------------------------------------------------------------------------
enum { WVC_VALID = 1, WVC_ACC = 2, WVC_O2 = 4, WVC_O3 = 8 };
struct WvMasks16 { u32 v, a, o2, o3; };

template <class LUT>
SXD WvMasks16 wv_classify16_single(const LUT& lut, u32 x0, u32 x1, u32 x2, u32 x3, u32 avail) {
    const u32 xs[4] = { x0, x1, x2, x3 };
    WvMasks16 m{ 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const u32 c = lut[(xs[k >> 2] >> (8 * (k & 3))) & 0xFFu];
        m.v |= (c & 1u) << k; m.a |= ((c >> 1) & 1u) << k; m.o2 |= ((c >> 2) & 1u) << k; m.o3 |= ((c >> 3) & 1u) << k;
    }
    const u32 keep = avail >= 16 ? 0xFFFFu : ((1u << avail) - 1u);
    m.v &= keep; m.a &= keep; m.o2 &= keep; m.o3 &= keep;
    return m;
}

// ---- the same classes WITHOUT the table (round 4): a Mission whose 256 bytes are all characters, whose accepted bytes are at most six
// ranges (each on one side of 0x80: KOI8-R + Cyrillic is 20..7E, A3, B3, C0..FF; `ascii` one range) and whose accepted bytes >= 0x80 all
// have UTF-8 forms of the same length needs two masks per 16 bytes — accepted, >= 0x80 — and three SWAR operations per range and
// dword for them, no LDS lookups (16 per lane and tile, on random bytes four to five of them to the same bank).  The valid mask is
// all ones, O2 / O3 = accepted & high by the length.  sx_mission.cpp decides (Mission::wave_swar); the table version above stays as
// the statement the host harness compares this with, and as the path of every other Mission.
// flags at bit 7 of every byte of four dwords -> 16 bits
SXD u32 wv_movemask16_b7(u32 f0, u32 f1, u32 f2, u32 f3) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, 0x80402010u, lo, false);
    u32 hi = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(f3, 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);   // (each flag byte is 0x80: the sums are 128 x mask)
#else
    const u32 f[4] = { f0, f1, f2, f3 };
    u32 m = 0;
    for (int k = 0; k < 16; k++) m |= ((f[k >> 2] >> (8 * (k & 3) + 7)) & 1u) << k;
    return m;
#endif
}
// accepted bytes of one dword: flags at bit 7.  K = the ranges computed (the Mission's, rounded up to 1 / 3 / 6: the count is the
// same for every lane, so the kernels pick the instantiation with a scalar branch)
template <int K>
SXD u32 wv_swar_accepted(const WvSwar& R, u32 x) {
    const u32 t = x & 0x7F7F7F7Fu;
    u32 f = 0;
#pragma unroll
    for (int k = 0; k < K; k++) f |= (t + R.c1[k]) & ~(t + R.c2[k]) & (x ^ R.hi[k]);   // (ranges not in use are empty: lo 1, hi 0)
    return f & 0x80808080u;
}
struct WvMasks16R { u32 a, hi; };
template <int K = 6>
SXD WvMasks16R wv_classify16_single_swar(const WvSwar& R, u32 x0, u32 x1, u32 x2, u32 x3, u32 avail) {
    const u32 keep = avail >= 16 ? 0xFFFFu : ((1u << avail) - 1u);
    WvMasks16R m;
    m.a = wv_movemask16_b7(wv_swar_accepted<K>(R, x0), wv_swar_accepted<K>(R, x1), wv_swar_accepted<K>(R, x2), wv_swar_accepted<K>(R, x3)) & keep;
    m.hi = wv_movemask16_b7(x0 & 0x80808080u, x1 & 0x80808080u, x2 & 0x80808080u, x3 & 0x80808080u) & keep;
    return m;
}

// 128 bits at bit offset `o` of a bit array held as dwords (the kernels: the masks of a batch of windows in LDS)
template <class WORDS>
SXD WvMask wv_extract(const WORDS& words, u32 o, u32 n) {
    const u32 di = o >> 5, sh = o & 31u;
    u32 w[5];
#pragma unroll
    for (int k = 0; k < 5; k++) w[k] = words[di + k];
    u32 m[4];
#pragma unroll
    for (int k = 0; k < 4; k++) m[k] = sh ? (w[k] >> sh) | (w[k + 1] << (32 - sh)) : w[k];
    WvMask r{ (u64)m[0] | ((u64)m[1] << 32), (u64)m[2] | ((u64)m[3] << 32) };
    return wm_and(r, wm_below(n));
}

SXD WvMask wm_shr(WvMask m, u32 k) {   // 0 < k < 64
    return WvMask{ (m.lo >> k) | (m.hi << (64 - k)), m.hi >> k };
}
// first bits of the runs of >= n set bits in G (n >= 1)
SXD WvMask wv_long_starts(WvMask G, u32 n) {
    WvMask r = G;   // bit i: bits i .. i + have - 1 are all set
    u32 have = 1;
    while (have < n) {
        const u32 sh = have < n - have ? have : n - have;
        r = wm_and(r, sh < 64 ? wm_shr(r, sh) : WvMask{ r.hi, 0 });
        have += sh;
    }
    const WvMask starts = wm_andn(G, wm_shl1(G));
    return wm_and(starts, r);
}

// the window of a single-byte Mission from its valid / accepted / length masks
SXD WvWin wv_win_single(WvMask V, WvMask A, WvMask O2, WvMask O3, u32 n, u32 n_min) {
    WvWin w;
    w.E = V; w.A = A; w.F = V; w.O2 = O2; w.O3 = O3; w.n = n;
    w.LS = wv_long_starts(A, n_min); w.G = A;
    const WvMask bad = wm_andn(wm_below(n), V);        // a byte without a character: Malformed(1, 0), the call ends behind it
    w.CS = wm_and(wm_shl1(bad), wm_below(n));
    w.tail_empty = n && wm_test(bad, n - 1) ? 1u : 0u;
    w.pre_empty = 0; w.head_back = 0; w.probe_before = 0; w.slice_start = 0; w.probe_hb = 0; w.head_pend = 0; w.tail_pend = 0;
    w.O4 = wm_zero();
    return w;
}

// the window of a single-byte Mission whose classes came as ranges (wv_classify16_single_swar)
SXD WvWin wv_win_single_swar(WvMask A, WvMask HI, u32 hi_len, u32 n, u32 n_min) {
    const WvMask ah = wm_and(A, HI);
    return wv_win_single(wm_below(n), A, hi_len == 2 ? ah : wm_zero(), hi_len == 3 ? ah : wm_zero(), n, n_min);
}

// ------------------------------------------------------------------------------------------
// Classification, UTF-8 (WHATWG "utf-8 decoder" as encoding_rs runs it, sx_codec_core.hpp ddec_utf8): per byte, from the
// 4 bytes in front of the lane's 16 and the 4 behind them (a lane loads 24 bytes; nothing crosses lanes).
//   * every lead byte C2..F4 starts a parse where it stands (mid-sequence it is the byte the decoder rejects and reads
//     again); it consumes the continuation bytes that are in range (the first one narrowed for E0 / ED / F0 / F4);
//     complete -> a character (F on its first byte, E on its last); short -> Malformed(bytes, 0) and the call ends IN FRONT
//     of the byte that did not fit (MB on that byte);
//   * a continuation byte nobody consumed, C0, C1, F5..FF -> Malformed(1, 0), the call ends BEHIND it (MA).
// class byte per input byte: bits 0-2 kind (0 bad, 1 ASCII, 2 continuation, 3 / 4 / 5 lead of 2 / 3 / 4), bit 3 accepted.
// ------------------------------------------------------------------------------------------
enum { WVU_BAD = 0, WVU_ASCII = 1, WVU_CONT = 2, WVU_LEAD2 = 3, WVU_LEAD3 = 4, WVU_LEAD4 = 5, WVU_ACC = 8 };
struct WvMasks16U { u32 e, a, f, g, ma, mb; };

// b[0..23] = the bytes at lane offset -4 .. +19; have_lo / have_hi: which of them exist (indices [have_lo, have_hi))
template <class LUT>
SXD WvMasks16U wv_classify16_utf8(const LUT& lut, const u8* b, u32 have_lo, u32 have_hi) {
    WvMasks16U m{ 0, 0, 0, 0, 0, 0 };
    u32 covered = 0;   // bit i: byte i (0..23) is consumed by a lead in front of it
    u32 kind[24];
#pragma unroll
    for (int i = 0; i < 24; i++) kind[i] = ((u32)i >= have_lo && (u32)i < have_hi) ? (u32)lut[b[i]] : 0xFFu;   // 0xFF: no such byte
#pragma unroll
    for (int i = 1; i < 20; i++) {   // leads at lane offsets -3 .. 15
        const u32 k = kind[i];
        if (k == 0xFFu || (k & 7u) < WVU_LEAD2) continue;
        const u32 need = (k & 7u) - 2u;   // continuation bytes
        const u8 lead = b[i];
        u32 lo = 0x80, hi = 0xBF;
        if (lead == 0xE0) lo = 0xA0; else if (lead == 0xED) hi = 0x9F; else if (lead == 0xF0) lo = 0x90; else if (lead == 0xF4) hi = 0x8F;
        u32 got = 0;
        bool open = false, stop = false;   // open: ran into the end of what exists: still pending there
#pragma unroll
        for (int t = 1; t <= 3; t++) {   // (constant indices: the arrays stay in registers)
            if ((u32)t > need || stop) continue;
            if (i + t >= 24 || kind[i + t >= 24 ? 23 : i + t] == 0xFFu) { open = true; stop = true; continue; }
            const u8 c = b[i + t >= 24 ? 23 : i + t];
            if (c < lo || c > hi) { stop = true; continue; }
            lo = 0x80; hi = 0xBF;
            got++;
        }
        covered |= ((1u << got) - 1u) << (i + 1);
        const int j = i - 4;   // lane offset of the lead
        if (got == need) {
            const int last = j + (int)need;
            if (j >= 0 && j < 16) m.f |= 1u << j;
            if (last >= 0 && last < 16) { m.e |= 1u << last; if (k & WVU_ACC) m.a |= 1u << last; }
            if (k & WVU_ACC) {   // bits j .. last, clipped to the lane's 16
                const u32 span = ((2u << (need + 0)) - 1u);   // need + 1 ones
                m.g |= (j >= 0 ? span << j : span >> (-j)) & 0xFFFFu;
            }
        } else if (!open) {
            const int off = j + (int)got + 1;   // the byte that did not fit: the call ends in front of it
            if (off >= 0 && off < 16) m.mb |= 1u << off;
        }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const u32 k = kind[j + 4];
        if (k == 0xFFu) continue;
        if ((k & 7u) == WVU_ASCII) { m.f |= 1u << j; m.e |= 1u << j; if (k & WVU_ACC) { m.a |= 1u << j; m.g |= 1u << j; } }
        else if ((k & 7u) == WVU_BAD || ((k & 7u) == WVU_CONT && !((covered >> (j + 4)) & 1u))) m.ma |= 1u << j;
    }
    return m;
}

// ---- the same classification as SWAR (round 4; WvSwar::cls == 1: the accepted first bytes — ASCII bytes and lead bytes — are at most six
// ranges).  Everything about byte p follows from the bytes p-3 .. p: the kind of the lead byte one, two and three bytes in front of it travels in
// one flag byte per input byte (K: E0 / ED / F0 / F4 — the leads whose first continuation is narrowed —, lead of 2 / 3 / 4, accepted), shifted by
// one, two and three bytes with v_alignbyte.  Only F looks ahead (a lead byte begins a character if the character completes).  Five masks leave
// the lane (E, A, F, MA, MB); G follows from A and F where the window is built.
SXD u32 wv_alignbyte(u32 hi, u32 lo, u32 k) {   // bytes k .. k + 3 of lo:hi (k = 1 .. 3)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbyte(hi, lo, k);
#else
    return (u32)((((u64)hi << 32) | lo) >> (8 * k));
#endif
}
struct WvU8Dword { u32 k, c1, c2, e2, e3, e4, e, a, ma, mb, asc; };
// one dword: v its bytes, p the dword in front of it (its k / c1 / c2)
template <int KR>
SXD WvU8Dword wv_utf8_dword_swar(const WvSwar& R, u32 v, const WvU8Dword& p) {
    const u32 M = 0x80808080u, t = v & 0x7F7F7F7Fu;
    const u32 asc = ~v & M, cont = v & ~(v << 1) & M;
    const u32 l2 = (t + 0x3E3E3E3Eu) & ~(t + 0x20202020u) & v & M;     // C2..DF
    const u32 l3 = (t + 0x20202020u) & ~(t + 0x10101010u) & v & M;     // E0..EF
    const u32 l4 = (t + 0x10101010u) & ~(t + 0x0B0B0B0Bu) & v & M;     // F0..F4
    const u32 bad = v & M & ~(cont | l2 | l3 | l4);                    // C0, C1, F5..FF
    const u32 w = v & 0x0F0F0F0Fu;
    const u32 z0 = ~(w + 0x7F7F7F7Fu) & M, zd = ~((w ^ 0x0D0D0D0Du) + 0x7F7F7F7Fu) & M, z4 = ~((w ^ 0x04040404u) + 0x7F7F7F7Fu) & M;
    const u32 acc = wv_swar_accepted<KR>(R, v);
    WvU8Dword d;
    d.asc = asc;
    d.k = (l3 & z0) | ((l3 & zd) >> 1) | ((l4 & z0) >> 2) | ((l4 & z4) >> 3) | (l2 >> 4) | (l3 >> 5) | (l4 >> 6) | (acc >> 7);
    const u32 d1 = wv_alignbyte(d.k, p.k, 3), d2 = wv_alignbyte(d.k, p.k, 2), d3 = wv_alignbyte(d.k, p.k, 1);   // the flags of the byte 1 / 2 / 3 in front
    const u32 b5 = (v << 2) & M, b4 = (v << 3) & M, b54 = b5 | b4;
    const u32 lead1 = ((d1 << 4) | (d1 << 5) | (d1 << 6)) & M;
    const u32 narrow = (d1 & ~b5) | ((d1 << 1) & b5) | ((d1 << 2) & ~b54) | ((d1 << 3) & b54);   // E0 wants A0..BF, ED 80..9F, F0 90..BF, F4 80..8F
    d.c1 = cont & lead1 & ~narrow;
    const u32 c1sh = wv_alignbyte(d.c1, p.c1, 3);
    const u32 l34_2 = ((d2 << 5) | (d2 << 6)) & M;
    d.c2 = cont & c1sh & l34_2;
    const u32 c2sh = wv_alignbyte(d.c2, p.c2, 3);
    const u32 l4_3 = (d3 << 6) & M;
    const u32 c3 = cont & c2sh & l4_3;
    d.e2 = d.c1 & (d1 << 4) & M; d.e3 = d.c2 & (d2 << 5) & M; d.e4 = c3;
    d.e = asc | d.e2 | d.e3 | d.e4;
    d.a = ((asc & acc) | (d.e2 & (d1 << 7)) | (d.e3 & (d2 << 7)) | (d.e4 & (d3 << 7))) & M;
    d.ma = bad | (cont & ~(d.c1 | d.c2 | c3));
    d.mb = ((lead1 & ~d.c1) | (l34_2 & c1sh & ~d.c2) | (l4_3 & c2sh & ~c3)) & M;
    return d;
}
struct WvMasks16V { u32 e, a, f, ma, mb; };
// ws6: the dwords at lane offset -4 .. +19 (bytes that do not exist: zero); n_own: how many of the lane's 16 bytes exist
template <int KR>
SXD WvMasks16V wv_classify16_utf8_swar(const WvSwar& R, const u32* ws6, u32 n_own) {
    WvU8Dword z{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    WvU8Dword d[6];
    d[0] = wv_utf8_dword_swar<KR>(R, ws6[0], z);   // (what lies in front of these four bytes only matters to bytes in front of the lane)
#pragma unroll
    for (int k = 1; k < 6; k++) d[k] = wv_utf8_dword_swar<KR>(R, ws6[k], d[k - 1]);
    u32 f[4];
#pragma unroll
    for (int k = 1; k <= 4; k++)   // a lead byte begins a character if the character's last byte is there
        f[k - 1] = d[k].asc | wv_alignbyte(d[k + 1].e2, d[k].e2, 1) | wv_alignbyte(d[k + 1].e3, d[k].e3, 2) | wv_alignbyte(d[k + 1].e4, d[k].e4, 3);
    const u32 ex = n_own >= 16 ? 0xFFFFu : ((1u << n_own) - 1u);
    WvMasks16V m;
    m.e = wv_movemask16_b7(d[1].e, d[2].e, d[3].e, d[4].e) & ex;
    m.a = wv_movemask16_b7(d[1].a, d[2].a, d[3].a, d[4].a) & ex;
    m.f = wv_movemask16_b7(f[0], f[1], f[2], f[3]) & ex;
    m.ma = wv_movemask16_b7(d[1].ma, d[2].ma, d[3].ma, d[4].ma) & ex;
    m.mb = wv_movemask16_b7(d[1].mb, d[2].mb, d[3].mb, d[4].mb) & ex;
    return m;
}
// G from A and F: a byte belongs to an accepted character if it is its last byte, or the byte behind it does and is not its first
SXD WvMask wv_utf8_good_from(WvMask A, WvMask F) {
    WvMask g = A, x = A;
#pragma unroll
    for (int k = 0; k < 3; k++) { x = wm_shr(wm_andn(x, F), 1); g = wm_or(g, x); }
    return g;
}

// the window of a UTF-8 Mission.  f_back: the F bits of the three bytes in front of the window (bit 2 = the byte right before
// it); slice_start: the window is the first of its slice (the probe of finding_collection.rs:176-207 then runs: a fresh decoder
// cannot reproduce a first character that began in the slice before -> `Before`).
SXD WvWin wv_win_utf8(WvMask E, WvMask A, WvMask F, WvMask G, WvMask MA, WvMask MB, u32 f_back, bool slice_start, u32 n, u32 n_min) {
    WvWin w;
    w.E = E; w.A = A; w.F = F; w.O2 = wm_zero(); w.O3 = wm_zero(); w.n = n;
    w.LS = wv_long_starts(G, n_min); w.G = G;
    w.CS = wm_and(wm_or(wm_shl1(MA), MB), wm_andn(wm_below(n), wm_below(1)));
    w.tail_empty = n && wm_test(MA, n - 1) ? 1u : 0u;
    w.pre_empty = n && wm_test(MB, 0) ? 1u : 0u;
    w.head_back = 0;
    const u32 e0 = wm_next(E, 0);
    if (e0 < 128 && wm_prev(F, e0) < 0) w.head_back = (f_back & 4u) ? 1u : (f_back & 2u) ? 2u : (f_back & 1u) ? 3u : 0u;
    w.probe_before = slice_start && w.head_back ? 1u : 0u;
    w.slice_start = slice_start ? 1u : 0u;
    w.probe_hb = 0; w.head_pend = 0; w.tail_pend = 0; w.O4 = wm_zero();
    return w;
}

// ------------------------------------------------------------------------------------------
// Classification, the two-byte family (Big5, Shift_JIS, EUC-KR; sx_codec_core.hpp ddec_big5).  The decoder consumes TOKENS: one byte
// outside the lead range, or a lead byte with the byte behind it, whatever that is.  Where tokens start follows from where the
// one before started (the scan kernel's token classifier, sx_kernels.hip scan_kernel_dbcs, says the same in bit operations):
//   * per lane (16 bytes) the walk is done for both cases "my byte 0 starts a token" / "is the trail of the lane before" (wv_dbcs_walk);
//     the lanes' in -> out functions are composed along the wavefront, which gives every lane its case;
//   * a lead + trail that the index maps: a character (E on the trail, F on the lead); Big5's four pointers that yield TWO code
//     points: two characters, one per byte (D marks the second); else Malformed: an ASCII trail is read again (MB on it) and is a
//     character of its own, any other trail is consumed (MA on it);
//   * a byte on its own: a character (ASCII; Shift_JIS 0x80, A1..DF), or Malformed(1, 0) (MA).
// lut1[b]: WVC_* of the byte as a character on its own | WVC_LEAD; pair codes: 4 bits per (lead | trail << 8): bit 0 mapped, bit 1
// accepted (a pair of two: its first), bits 2-3: 0 / 1 / 2 = the UTF-8 form has 2 / 3 / 4 bytes, 3 = two characters of 2 bytes each.
// ------------------------------------------------------------------------------------------
enum { WVC_LEAD = 16 };
struct WvMasks16D { u32 e, a, f, g, ma, mb, o2, o3, o4; };

// lead-range mask of the lane's 16 bytes (bit 16 + k: the k-th byte behind them), existing bytes only
template <class LUT>
SXD u32 wv_dbcs_lead_mask(const LUT& lut1, const u8* b, u32 have_hi) {
    u32 lr = 0;
#pragma unroll
    for (int j = 0; j < 20; j++) if ((u32)(j + 4) < have_hi && (lut1[b[j + 4]] & WVC_LEAD)) lr |= 1u << j;
    return lr;
}
// token starts among the lane's bytes 0..15 when the first token starts at byte `first` (0 or 1); *over = how far the last token
// reaches into the next lane (0 / 1)
SXD u32 wv_dbcs_walk(u32 lr, u32 first, u32* over) {
    u32 s = 0, pos = first;
    while (pos < 16) { s |= 1u << pos; pos += ((lr >> pos) & 1u) ? 2u : 1u; }
    *over = pos - 16;
    return s;
}

template <class LUT, class PAIRS>
SXD WvMasks16D wv_classify16_dbcs(const LUT& lut1, const PAIRS& pairs, const u8* b, u32 have_lo, u32 have_hi, u32 lr, u32 starts, u32 cov_in) {
    WvMasks16D m{ 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    // tokens of two bytes that start at lane offsets -1 .. 15 (their trails at 0 .. 16): never at adjacent offsets
    const u32 two = (((starts & lr) << 1) | (cov_in ? 1u : 0u)) & 0x1FFFFu;   // bit p + 1: a two-byte token starts at p
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        const int p = ((two >> (2 * k)) & 1u) ? 2 * k - 1 : (((two >> (2 * k + 1)) & 1u) ? 2 * k : -2);
        if (p < -1 || p > 15) continue;
        if ((u32)(p + 5) >= have_hi || (u32)(p + 4) < have_lo) continue;   // the trail does not exist (yet): the token is pending (or its lead lies in front of the buffer)
        const u32 lead = b[p + 4], trail = b[p + 5];
        const u32 idx = lead | (trail << 8);
        const u32 code = (pairs[idx >> 3] >> ((idx & 7u) * 4)) & 15u;
        const u32 at_l = p >= 0 ? 1u << p : 0u, at_t = p + 1 <= 15 ? 1u << (p + 1) : 0u;
        if (code & 1u) {
            const u32 acc = (code >> 1) & 1u, len = code >> 2;
            if (len == 3) {
                // Big5's four pointers that yield TWO code points (U+00CA / U+00EA + a combining mark), both delivered when the trail
                // is read.  The wave path only takes Missions that reject both (sx_mission.cpp): for SplitStr two rejected chars in
                // a row are what one is — a break — so the token stands as ONE rejected char.
                m.f |= at_l; m.e |= at_t; m.o2 |= at_t;
            } else {
                m.f |= at_l; m.e |= at_t; m.o2 |= at_t;
                if (len >= 1) m.o3 |= at_t;
                if (len == 2) m.o4 |= at_t;
                if (acc) { m.a |= at_t; m.g |= at_l | at_t; }
            }
        } else if (trail < 0x80) {                          // the trail is read again: a character of its own
            m.mb |= at_t;
            const u32 c = lut1[trail];
            if (c & WVC_VALID) { m.f |= at_t; m.e |= at_t; if (c & WVC_ACC) { m.a |= at_t; m.g |= at_t; } }
            else m.ma |= at_t;
        } else m.ma |= at_t;
    }
    // tokens of one byte
    const u32 one = starts & ~lr & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (!((one >> j) & 1u) || (u32)(j + 4) >= have_hi) continue;
        const u32 c = lut1[b[j + 4]];
        if (c & WVC_VALID) {
            m.f |= 1u << j; m.e |= 1u << j;
            if (c & (WVC_O2 | WVC_O3)) m.o2 |= 1u << j;
            if (c & WVC_O3) m.o3 |= 1u << j;
            if (c & WVC_ACC) { m.a |= 1u << j; m.g |= 1u << j; }
        } else m.ma |= 1u << j;
    }
    return m;
}

// ---- the same classification as bit arithmetic (the kernels' path; the functions above stay as its statement byte by byte and as
// what tests/native/wave_core_host.cpp compares it with).  Masks of the lane's 16 bytes: bit j = byte j.
struct WvDbcsPre { u32 v1, a1, o2, o3, lr, asc; };   // a character on its own / accepted / UTF-8 form of >= 2 / 3 bytes / lead range / < 0x80

// flags at bit 0 of every byte of four dwords -> 16 bits
SXD u32 wv_movemask16(u32 f0, u32 f1, u32 f2, u32 f3) {
#if defined(__HIP_DEVICE_COMPILE__)
    u32 lo = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, 0x80402010u, lo, false);
    u32 hi = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(f3, 0x80402010u, hi, false);
    return lo | (hi << 8);
#else
    const u32 f[4] = { f0, f1, f2, f3 };
    u32 m = 0;
    for (int k = 0; k < 16; k++) m |= ((f[k >> 2] >> (8 * (k & 3))) & 1u) << k;
    return m;
#endif
}

// x: the lane's four dwords, avail: how many of its 16 bytes exist
template <class LUT>
SXD WvDbcsPre wv_dbcs_classes(const LUT& lut1, const u32* x, u32 avail) {
    u32 cw[4];
#pragma unroll
    for (int k = 0; k < 4; k++)
        cw[k] = (u32)lut1[x[k] & 0xFFu] | ((u32)lut1[(x[k] >> 8) & 0xFFu] << 8) | ((u32)lut1[(x[k] >> 16) & 0xFFu] << 16) | ((u32)lut1[x[k] >> 24] << 24);
    const u32 k1 = 0x01010101u, ex = avail >= 16 ? 0xFFFFu : ((1u << avail) - 1u);
    WvDbcsPre c;
    c.v1 = wv_movemask16(cw[0] & k1, cw[1] & k1, cw[2] & k1, cw[3] & k1) & ex;                                       // WVC_VALID
    c.a1 = wv_movemask16((cw[0] >> 1) & k1, (cw[1] >> 1) & k1, (cw[2] >> 1) & k1, (cw[3] >> 1) & k1) & ex;           // WVC_ACC
    c.o2 = wv_movemask16((cw[0] >> 2) & k1, (cw[1] >> 2) & k1, (cw[2] >> 2) & k1, (cw[3] >> 2) & k1) & ex;           // WVC_O2
    c.o3 = wv_movemask16((cw[0] >> 3) & k1, (cw[1] >> 3) & k1, (cw[2] >> 3) & k1, (cw[3] >> 3) & k1) & ex;           // WVC_O3
    c.lr = wv_movemask16((cw[0] >> 4) & k1, (cw[1] >> 4) & k1, (cw[2] >> 4) & k1, (cw[3] >> 4) & k1) & ex;           // WVC_LEAD
    c.asc = wv_movemask16((~x[0] >> 7) & k1, (~x[1] >> 7) & k1, (~x[2] >> 7) & k1, (~x[3] >> 7) & k1) & ex;
    return c;
}

// The token walk of wv_dbcs_walk without the walk.  Bit i of the result: byte i is the TRAIL of a two-byte token (bit 16: the lane's
// last token reaches into the next lane); cov: byte 0 is one.  A lead byte "escapes" the byte behind it unless it is escaped itself —
// runs of lead-range bytes alternate, and whether a run's first byte starts a token is the parity of where it stands: the carry of
// one addition tells the runs that start on odd positions from those on even ones.
SXD u32 wv_dbcs_trails(u32 lr, u32 cov) {
    const u32 even = 0x55555555u;
    const u32 b = lr & ~cov;                       // (byte 0 as a trail is no lead)
    const u32 follows = (b << 1) | cov;            // bytes behind a lead-range byte
    const u32 odd_starts = b & ~even & ~follows;   // runs that begin on an odd position
    const u32 inv = (odd_starts + b) << 1;         // ... flip the parity of everything up to the byte behind their end
    return (even ^ inv) & follows & 0x1FFFFu;
}

// ws6: the dwords at lane offset -4 .. +19; c: the classes of the lane's own bytes; tr = wv_dbcs_trails(c.lr, cov_in);
// back_exists: the byte in front of the lane's first lies inside the buffer; n_exist: existing bytes from the lane's first on (<= 20)
template <class PAIRS>
SXD WvMasks16D wv_classify16_dbcs_bits(const PAIRS& pairs, const u32* ws6, const WvDbcsPre& c, u32 tr, u32 cov_in, bool back_exists,
                                       u32 n_exist) {
    // "shifted" masks: bit p + 1 = lane offset p, so that the token whose lead lies in front of the lane (p = -1) has a bit
    const u32 starts_s = ((~tr & 0xFFFFu) << 1) | cov_in;
    const u32 lr_s = (c.lr << 1) | cov_in;
    const u32 ex_s = ((n_exist >= 20 ? 0xFFFFFu : ((1u << n_exist) - 1u)) << 1) | (back_exists ? 1u : 0u);
    const u32 two = starts_s & lr_s & ex_s & (ex_s >> 1);   // two-byte tokens whose both bytes exist, at their leads
    const u32 T = two << 1;                                 // ... at their trails
    u32 Mp = 0, Ac = 0, L0 = 0, L1 = 0;                     // the pair code's bits, at the trails
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        // the slot's token, if any, starts at lane offset 2k - 1 (its lead is byte 2k + 3 of ws6) or 2k (byte 2k + 4)
        const u32 has = (two >> (2 * k)) & 3u;
        if (!has) continue;
        const u32 odd = has >> 1;
        const int ie = 2 * k + 3, io = 2 * k + 4;
        const u32 u_even = (ie & 3) == 3 ? ((ws6[ie >> 2] >> 24) | ((ws6[(ie >> 2) + 1] & 0xFFu) << 8)) : ((ws6[ie >> 2] >> (8 * (ie & 3))) & 0xFFFFu);
        const u32 u_odd = (ws6[io >> 2 > 5 ? 5 : io >> 2] >> (8 * (io & 3))) & 0xFFFFu;   // (k = 8: no token starts at offset 16)
        const u32 idx = odd ? u_odd : u_even;               // lead | trail << 8
        const u32 code = (pairs[idx >> 3] >> ((idx & 7u) * 4)) & 15u;
        const u32 sh = 2 * k + 1 + odd;
        Mp |= (code & 1u) << sh; Ac |= ((code >> 1) & 1u) << sh; L0 |= ((code >> 2) & 1u) << sh; L1 |= (code >> 3) << sh;
    }
    const u32 v_s = c.v1 << 1, a_s = c.a1 << 1, asc_s = c.asc << 1;
    const u32 dbl = Mp & L0 & L1;                           // Big5's tokens of two code points: one rejected char (see above)
    const u32 acc2 = Ac & Mp & ~dbl;
    const u32 un = T & ~Mp;                                 // unmapped: an ASCII trail is read again as a character of its own
    const u32 un_a = un & asc_s;
    const u32 one = starts_s & ~lr_s & ex_s;                // tokens of one byte
    const u32 own = (un_a | one) & v_s;                     // characters of one byte
    u32 f = (Mp >> 1) | own;
    u32 e = Mp | own;
    u32 a = acc2 | (own & a_s);
    u32 g = acc2 | (acc2 >> 1) | (own & a_s);
    u32 ma = (un & ~asc_s) | ((un_a | one) & ~v_s);
    u32 o2 = Mp | (one & v_s & ((c.o2 | c.o3) << 1));
    u32 o3 = (Mp & (L0 ^ L1)) | (one & v_s & (c.o3 << 1));
    u32 o4 = Mp & L1 & ~L0;
    WvMasks16D m;
    m.e = (e >> 1) & 0xFFFFu; m.a = (a >> 1) & 0xFFFFu; m.f = (f >> 1) & 0xFFFFu; m.g = (g >> 1) & 0xFFFFu;
    m.ma = (ma >> 1) & 0xFFFFu; m.mb = (un_a >> 1) & 0xFFFFu;
    m.o2 = (o2 >> 1) & 0xFFFFu; m.o3 = (o3 >> 1) & 0xFFFFu; m.o4 = (o4 >> 1) & 0xFFFFu;
    return m;
}

// ---- the two-byte family without the byte table and with 2 bits per pair (round 4; WvSwar::cls == 1: Big5 / EUC-KR Missions whose
// accepted pairs all have UTF-8 forms of one length — Cjk, Asian, Kana, Hangul).  Byte classes as SWAR: lead range (two ranges of the
// low seven bits), < 0x80, accepted < 0x80 (the af filter as ranges); a byte below 0x80 is always a character on its own, one above never.
// Five masks leave the lane (E, A, F, MA, MB) — G and the length masks follow from them where the window is built (wv_win_dbcs_swar).
struct WvDbcsPreS { u32 lr, asc, a1; };
template <int K>
SXD WvDbcsPreS wv_dbcs_classes_swar(const WvSwar& R, const u32* x, u32 avail) {
    const u32 ex = avail >= 16 ? 0xFFFFu : ((1u << avail) - 1u), kM = 0x80808080u;
    u32 fl[4], fa[4], fs[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const u32 v = x[k], t = v & 0x7F7F7F7Fu;
        fl[k] = (((t + R.lr_c1[0]) & ~(t + R.lr_c2[0])) | ((t + R.lr_c1[1]) & ~(t + R.lr_c2[1]))) & v & kM;
        fs[k] = ~v & kM;
        fa[k] = wv_swar_accepted<K>(R, v);
    }
    WvDbcsPreS c;
    c.lr = wv_movemask16_b7(fl[0], fl[1], fl[2], fl[3]) & ex;
    c.asc = wv_movemask16_b7(fs[0], fs[1], fs[2], fs[3]) & ex;
    c.a1 = wv_movemask16_b7(fa[0], fa[1], fa[2], fa[3]) & ex;
    return c;
}
struct WvMasks16E { u32 e, a, f, ma, mb; };
// as wv_classify16_dbcs_bits; pairs2: 2 bits per (lead | trail << 8), sixteen per word: bit 0 mapped, bit 1 accepted
template <class PAIRS>
SXD WvMasks16E wv_classify16_dbcs_swar(const PAIRS& pairs2, const u32* ws6, const WvDbcsPreS& c, u32 tr, u32 cov_in, bool back_exists, u32 n_exist) {
    const u32 starts_s = ((~tr & 0xFFFFu) << 1) | cov_in;
    const u32 lr_s = (c.lr << 1) | cov_in;
    const u32 ex_s = ((n_exist >= 20 ? 0xFFFFFu : ((1u << n_exist) - 1u)) << 1) | (back_exists ? 1u : 0u);
    const u32 two = starts_s & lr_s & ex_s & (ex_s >> 1);   // two-byte tokens whose both bytes exist, at their leads
    const u32 T = two << 1;                                 // ... at their trails
    u32 Mp = 0, Ac = 0;
#pragma unroll
    for (int k = 0; k <= 8; k++) {
        const u32 has = (two >> (2 * k)) & 3u;
        if (!has) continue;
        const u32 odd = has >> 1;
        const int ie = 2 * k + 3, io = 2 * k + 4;
        const u32 u_even = (ie & 3) == 3 ? ((ws6[ie >> 2] >> 24) | ((ws6[(ie >> 2) + 1] & 0xFFu) << 8)) : ((ws6[ie >> 2] >> (8 * (ie & 3))) & 0xFFFFu);
        const u32 u_odd = (ws6[io >> 2 > 5 ? 5 : io >> 2] >> (8 * (io & 3))) & 0xFFFFu;
        const u32 idx = odd ? u_odd : u_even;
        const u32 code = (pairs2[idx >> 4] >> ((idx & 15u) * 2)) & 3u;
        const u32 sh = 2 * k + 1 + odd;
        Mp |= (code & 1u) << sh; Ac |= (code >> 1) << sh;
    }
    const u32 v_s = c.asc << 1, a_s = c.a1 << 1;
    const u32 acc2 = Ac & Mp;
    const u32 un = T & ~Mp;                                 // unmapped: a trail below 0x80 is read again as a character of its own
    const u32 un_a = un & v_s;
    const u32 one = starts_s & ~lr_s & ex_s;                // tokens of one byte
    const u32 own = (un_a | one) & v_s;                     // characters of one byte
    WvMasks16E m;
    m.f = (((Mp >> 1) | own) >> 1) & 0xFFFFu;
    m.e = ((Mp | own) >> 1) & 0xFFFFu;
    m.a = ((acc2 | (own & a_s)) >> 1) & 0xFFFFu;
    m.ma = (((un & ~v_s) | (one & ~v_s)) >> 1) & 0xFFFFu;
    m.mb = (un_a >> 1) & 0xFFFFu;
    return m;
}

// ------------------------------------------------------------------------------------------
// UTF-16LE / BE (round 4; sx_codec_core.hpp ddec_utf16) for buffers that begin on the unit grid (stream parity 0) and have an even length.
// A unit is a character (B), a high (H) or a low surrogate (L); H L is one character of four bytes.  Whole units are decoded in a fast loop;
// a unit that is the LAST of the decoder's input (the window) is read byte by byte, and so is whatever follows while a high surrogate is
// pending.  The two modes differ where a pending H is not followed by an L:
//   fast:  H x   -> Malformed, the call ends behind H, x is read again                                   (MA on H's last byte)
//   slow:  H H'  -> Malformed, the call ends behind H', H' is pending                                    (MA on H''s last byte)
//          H B   -> Malformed, the call ends behind B, and B is kept: it is the first character of the NEXT call (`pending_bmp`) — a
//                   character whose bytes lie in the call before.  In the masks the call boundary is put IN FRONT of B (MB on its first byte:
//                   the text of both calls is what it was) and PB says that calls starting there report their position two bytes later.
// Slow mode begins with a window's first unit if the unit in front of it (the window before's last, which is always left pending) is H, and goes
// on while the unit in front is H ("chain").  What the masks cannot say makes the wavefront give the buffer back (`exotic`): a kept B that is
// its window's last unit (it would be delivered in the next window with no byte there), and chains that cross a whole lane (seven high
// surrogates in a row).  lut: 512 bytes — [hb] for the unit's high byte: bits 0-3 "accepted" for the low byte's quadrant (lo >> 6), bit 4 hb == 0
// (then [256 + lo] bit 0 says accepted), bit 5 high surrogate (bits 0-3: the astral character it begins), bit 6 low surrogate.
// ------------------------------------------------------------------------------------------
enum { WVW_ZERO = 16, WVW_HIGH = 32, WVW_LOW = 64 };
struct WvMasks16W { u32 e, a, f, ma, mb, pb, o2, o3, o4, exotic; };
struct WvU16Unit { u32 kind /*0 B, 1 H, 2 L*/, acc, len; };
template <class LUT>
SXD WvU16Unit wv_utf16_unit(const LUT& lut, u32 u) {
    const u32 hb = u >> 8, lo = u & 0xFFu, t = lut[hb];
    WvU16Unit r;
    r.kind = (t & WVW_HIGH) ? 1u : (t & WVW_LOW) ? 2u : 0u;
    r.acc = (t & WVW_ZERO) ? (u32)(lut[256 + lo] & 1u) : ((t >> (lo >> 6)) & 1u);
    r.len = u < 0x80u ? 1u : u < 0x800u ? 2u : 3u;
    return r;
}
// (The decoder's state machine byte by byte over one window — the statement these marks are compared with, window by window — is in
// tests/native/wave_core_host.cpp.)  A lane's 16 bytes = 8 units, whatever windows they lie in.  x4: the lane's dwords; n_units: whole units that exist; hm / accm
// etc. come out for the neighbours.  wbm: bit j = unit j is the first of a window (bit 8: the unit behind the lane is).  prev_h / prev_acc: the
// unit in front of the lane is H / begins an accepted astral character; c0: the lane's first unit is read in slow mode; next_l: the unit
// behind the lane is L.
struct WvU16Lane { u32 hm, lm, accm, len2, len3; };   // bit j = unit j
template <class LUT>
SXD WvU16Lane wv_utf16_lane_units(const LUT& lut, bool be, const u32* x4, u32 n_units) {
    WvU16Lane r{ 0, 0, 0, 0, 0 };
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if ((u32)j >= n_units) continue;
        const u32 raw = (x4[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        const u32 u = be ? ((raw & 0xFFu) << 8) | (raw >> 8) : raw;
        const WvU16Unit x = wv_utf16_unit(lut, u);
        r.hm |= (x.kind == 1 ? 1u : 0u) << j; r.lm |= (x.kind == 2 ? 1u : 0u) << j; r.accm |= x.acc << j;
        r.len2 |= (x.len >= 2 ? 1u : 0u) << j; r.len3 |= (x.len >= 3 ? 1u : 0u) << j;
    }
    return r;
}
// chain bits: C(j) = H(j-1) && (wb(j) || C(j-1)) for j = 1..8 (bit 8: the unit behind the lane), bit 0 = c0 as handed in by the lane in front
SXD u32 wv_utf16_chain(u32 hm, u32 wbm, u32 c0) {
    u32 c = c0 & 1u, pc = c0 & 1u;
#pragma unroll
    for (int j = 1; j <= 8; j++) {
        const u32 cj = ((hm >> (j - 1)) & 1u) & (((wbm >> j) & 1u) | pc);
        c |= cj << j;
        pc = cj;
    }
    return c;
}
// what a lane hands on (bit 8 of the chain) depends on what it was handed only if all of its units are high surrogates and no window starts
// among units 1..8: the kernels take the value for c0 = 0 and give the buffer back where the two differ
SXD bool wv_utf16_transparent(u32 hm, u32 wbm) { return ((wv_utf16_chain(hm, wbm, 0u) ^ wv_utf16_chain(hm, wbm, 1u)) >> 8) != 0; }
SXD WvMasks16W wv_classify16_utf16(const WvU16Lane& L, u32 n_units, u32 wbm, u32 prev_h, u32 prev_acc, u32 c0, u32 next_l) {
    WvMasks16W m{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    const u32 c = wv_utf16_chain(L.hm, wbm, c0);
    const u32 lm9 = L.lm | (next_l << 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if ((u32)j >= n_units) continue;
        const u32 s = 2u * (u32)j, h = (L.hm >> j) & 1u, l = (L.lm >> j) & 1u, cj = (c >> j) & 1u;
        const u32 ph = j ? (L.hm >> (j - 1)) & 1u : prev_h, pacc = j ? (L.accm >> (j - 1)) & 1u : prev_acc;
        const u32 last_of_window = (wbm >> (j + 1)) & 1u;
        if (h) {
            if ((lm9 >> (j + 1)) & 1u) m.f |= 1u << s;                      // H L: the character begins here (wherever the call does)
            if (cj) m.ma |= 2u << s;                                       // slow: Malformed behind it, whatever follows
            else if (!last_of_window && !((lm9 >> (j + 1)) & 1u)) m.ma |= 2u << s;
        } else if (l) {
            if (ph) { m.e |= 2u << s; m.a |= (pacc * 2u) << s; m.o2 |= 2u << s; m.o3 |= 2u << s; m.o4 |= 2u << s; }
            else m.ma |= 2u << s;
        } else {
            const u32 acc = (L.accm >> j) & 1u;
            m.f |= 1u << s; m.e |= 2u << s; m.a |= (acc * 2u) << s;
            m.o2 |= (((L.len2 >> j) & 1u) * 2u) << s; m.o3 |= (((L.len3 >> j) & 1u) * 2u) << s;
            if (cj) { m.mb |= 1u << s; m.pb |= 1u << s; if (last_of_window) m.exotic = 1; }   // kept for the next call
        }
    }
    return m;
}
// What a wavefront keeps per lane and tile: four words of 16 bits.  The marks of a unit sit on its first byte (even bits: F, MB, "is a high
// surrogate", O3) or on its last (odd bits: E, A, MA, O2) — two masks share a word.
struct WvU16Packed { u32 m0, m1, m2, m3; };   // E | F,  A | MB,  MA | H,  O2 | O3 >> 1
SXD WvU16Packed wv_utf16_pack(const WvMasks16W& m, u32 hm) {
    u32 h = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) h |= ((hm >> j) & 1u) << (2 * j);
    return WvU16Packed{ m.e | m.f, m.a | m.mb, m.ma | h, m.o2 | (m.o3 >> 1) };
}
// the window of a UTF-16 Mission from those words; h_before: the unit in front of the window is a high surrogate (it is pending there: a
// window's last unit, if H, always is)
SXD WvWin wv_win_utf16(WvMask M0, WvMask M1, WvMask M2, WvMask M3, bool h_before, bool slice_start, u32 n, u32 n_min) {
    const WvMask odd{ 0xAAAAAAAAAAAAAAAAull, 0xAAAAAAAAAAAAAAAAull }, even{ 0x5555555555555555ull, 0x5555555555555555ull };
    WvWin w;
    w.E = wm_and(M0, odd); w.F = wm_and(M0, even); w.A = wm_and(M1, odd);
    const WvMask MB = wm_and(M1, even), MA = wm_and(M2, odd);
    w.O2 = wm_and(M3, odd); w.O3 = wm_shl1(wm_and(M3, even));
    w.O4 = wm_andn(w.E, wm_shl1(w.F));        // four bytes: the byte in front of the character's last one is not its first
    w.PB = MB; w.n = n;
    // (a high surrogate read in slow mode ends its call AND begins the character the low surrogate behind it completes, in the next call:
    // its bytes do not count as that character's, so that a stretch of accepted bytes never begins in front of its call)
    w.G = wm_andn(wv_utf8_good_from(w.A, w.F), wm_or(MA, wm_shr(MA, 1)));
    w.LS = wv_long_starts(w.G, n_min);
    w.CS = wm_and(wm_or(wm_shl1(MA), MB), wm_andn(wm_below(n), wm_below(1)));
    w.tail_empty = n && wm_test(MA, n - 1) ? 1u : 0u;
    w.pre_empty = n && wm_test(MB, 0) ? 1u : 0u;
    w.head_back = 0;
    const u32 e0 = wm_next(w.E, 0);
    if (e0 < 128 && wm_prev(w.F, e0) < 0) w.head_back = 2;   // H in front of the window, L here
    w.probe_before = slice_start && w.head_back ? 1u : 0u;   // (a fresh decoder meets the low surrogate alone: finding_collection.rs:176-207)
    w.slice_start = slice_start ? 1u : 0u;
    w.probe_hb = 0; w.head_pend = h_before ? 2u : 0u; w.tail_pend = 0;
    return w;
}
// the string of a UTF-16 finding: its source units [s, s + n)
SXD u32 wv_transcode_utf16(bool be, const u8* s, u32 n, u8* dst) {
    u32 w = 0;
    auto unit = [&](u32 at) -> u32 { uint16_t r; __builtin_memcpy(&r, s + at, 2); return be ? (u32)(((r & 0xFFu) << 8) | (r >> 8)) : (u32)r; };   // (one load per unit)
    for (u32 p = 0; p + 2 <= n; p += 2) {
        u32 u = unit(p);
        if ((u & 0xFC00u) == 0xD800u && p + 4 <= n) {
            const u32 v = unit(p + 2);
            u = 0x10000u + ((u & 0x3FFu) << 10) + (v & 0x3FFu);
            p += 2;
        }
        w += dput_cp(dst + w, u);
    }
    return w;
}

// ------------------------------------------------------------------------------------------
// EUC-JP (round 4; sx_codec_core.hpp ddec_eucjp): tokens of one byte, of two (a lead A1..FE or 8E with the byte behind it, whatever
// that is; also 8F with a byte outside A1..FE) and of three (8F, A1..FE, any byte: index jis0212).  A mapped token is a character (F on
// its first byte, E on its last); an unmapped one is Malformed and the decoder call ends: behind its last byte if that is >= 0x80
// (MA), in front of it if it is ASCII — the byte is read again as a character of its own (MB, and its own F / E).  The token that
// STARTS among a lane's 16 bytes is the lane's: its marks may lie one or two bytes beyond the lane (bits 16, 17 of the masks) and
// are handed to the next lane.  The UTF-8 form of a character has 2 or 3 bytes; the wave path takes Missions whose accepted ones
// all have the same length (Asian, Cjk, Kana: three), so the lengths need no mask (as wv_win_dbcs_swar).
// t2: 2 bits per cell (bit 0 mapped, bit 1 accepted), sixteen per word: cells 0 .. 8835 index jis0208, from 8836 on index jis0212.
// ------------------------------------------------------------------------------------------
constexpr u32 kWvJisCells = 94 * 94, kWvJisWords = (2 * kWvJisCells + 15) / 16;
struct WvMasks18 { u32 e, a, f, ma, mb; };
template <class T2>
SXD u32 wv_eucjp_cell(const T2& t2, u32 x, u32 y, u32 second_table) {
    const u32 r = x - 0xA1u, c = y - 0xA1u;
    if (r >= 94u || c >= 94u) return 0u;
    const u32 idx = second_table * kWvJisCells + r * 94u + c;
    return (t2[idx >> 4] >> ((idx & 15u) * 2)) & 3u;
}
// the code of the token that starts with `lead`: b1 / b2 = the bytes behind it, three = it has three bytes
template <class T2>
SXD u32 wv_eucjp_token_code(const T2& t2, u32 kana, u32 lead, u32 b1, u32 b2, bool three) {
    if (three) return wv_eucjp_cell(t2, b1, b2, 1u);
    if (lead == 0x8Eu) return (b1 - 0xA1u) < 0x3Fu ? (1u | (kana ? 2u : 0u)) : 0u;
    if (lead == 0x8Fu) return 0u;
    return wv_eucjp_cell(t2, lead, b1, 0u);
}
// The statement, byte by byte (the host harness compares the kernels' bit arithmetic with it): b[0..23] = the bytes at lane offset
// -4 .. +19, n_exist = how many exist from the lane's first byte on; first = bytes at the lane's start that belong to a token begun
// before (0 .. 2); lut1[b]: WVC_VALID / WVC_ACC of a byte below 0x80, WVC_LEAD for A1..FE, 8E, 8F.  *over = the same for the next lane.
template <class LUT, class T2>
SXD WvMasks18 wv_eucjp_walk(const LUT& lut1, const T2& t2, u32 kana, const u8* b, u32 n_exist, u32 first, u32* over) {
    WvMasks18 m{ 0, 0, 0, 0, 0 };
    u32 pos = first;
    while (pos < 16 && pos < n_exist) {
        const u32 c = b[4 + pos], cls = lut1[c];
        if (!(cls & WVC_LEAD)) {
            if (cls & WVC_VALID) { m.f |= 1u << pos; m.e |= 1u << pos; if (cls & WVC_ACC) m.a |= 1u << pos; }
            else m.ma |= 1u << pos;
            pos++;
            continue;
        }
        const u32 b1 = pos + 1 < n_exist ? b[4 + pos + 1] : 0u;
        const bool three = c == 0x8Fu && pos + 1 < n_exist && b1 >= 0xA1u && b1 <= 0xFEu;
        const u32 len = three ? 3u : 2u;
        if (pos + len > n_exist) { pos += len; break; }   // the buffer ends inside the token: it stays pending
        const u32 b2 = three ? b[4 + pos + 2] : 0u, last = three ? b2 : b1, end = pos + len - 1;
        const u32 code = wv_eucjp_token_code(t2, kana, c, b1, b2, three);
        if (code & 1u) { m.f |= 1u << pos; m.e |= 1u << end; if (code & 2u) m.a |= 1u << end; }
        else if (last < 0x80u) {
            m.mb |= 1u << end; m.f |= 1u << end; m.e |= 1u << end;
            if (lut1[last] & WVC_ACC) m.a |= 1u << end;
        } else m.ma |= 1u << end;
        pos += len;
    }
    *over = pos > 16 ? pos - 16 : 0u;
    return m;
}
// ---- the same as bit arithmetic.  Byte classes (bit j = the lane's byte j; bits 16 .. 19 the four bytes behind the lane)
struct WvEucPre { u32 hi, lr, c8f, asc, a1; };   // A1..FE / A1..FE, 8E, 8F / 8F / < 0x80 / accepted and < 0x80
SXD u32 wv_swar_eq(u32 v, u32 pat) { const u32 y = v ^ pat; return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u; }
template <int K>
SXD WvEucPre wv_eucjp_classes_swar(const WvSwar& R, const u32* x5, u32 n_exist) {
    const u32 kM = 0x80808080u;
    u32 fh[5], f8[5], ff[5], fs[5], fa[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const u32 v = x5[k], t = v & 0x7F7F7F7Fu;
        fh[k] = (t + 0x5F5F5F5Fu) & ~(t + 0x01010101u) & v & kM;      // low seven bits in 21 .. 7E, bit 7 set: A1..FE
        f8[k] = wv_swar_eq(v & 0xFEFEFEFEu, 0x8E8E8E8Eu);              // 8E, 8F
        ff[k] = f8[k] & (v << 7);                                      // ... 8F
        fs[k] = ~v & kM;
        fa[k] = wv_swar_accepted<K>(R, v);
    }
    const u32 ex = n_exist >= 20 ? 0xFFFFFu : ((1u << n_exist) - 1u);
    WvEucPre c;
    c.hi = (wv_movemask16_b7(fh[0], fh[1], fh[2], fh[3]) | (wv_movemask16_b7(fh[4], 0, 0, 0) << 16)) & ex;
    c.lr = (wv_movemask16_b7(fh[0] | f8[0], fh[1] | f8[1], fh[2] | f8[2], fh[3] | f8[3]) | (wv_movemask16_b7(fh[4] | f8[4], 0, 0, 0) << 16)) & ex;
    c.c8f = wv_movemask16_b7(ff[0], ff[1], ff[2], ff[3]) & ex;
    c.asc = (wv_movemask16_b7(fs[0], fs[1], fs[2], fs[3]) | (wv_movemask16_b7(fs[4], 0, 0, 0) << 16)) & ex;
    c.a1 = (wv_movemask16_b7(fa[0], fa[1], fa[2], fa[3]) | (wv_movemask16_b7(fa[4], 0, 0, 0) << 16)) & ex;
    return c;
}
// token starts for a hang-over of 0, 1 and 2 bytes: St[s] holds bit p if a token starts at byte p (bits >= 16: in the next lane);
// the byte behind a byte outside the lead range always starts one.  One round of wv_eucjp_orbit_step per token length in a row of
// lead-range bytes; the kernels iterate while any lane changes.
struct WvEucOrbit { u32 st[3], l2, l3; };
SXD WvEucOrbit wv_eucjp_orbit_init(const WvEucPre& c) {
    WvEucOrbit o;
    o.l3 = c.c8f & (c.hi >> 1) & 0xFFFFu;
    o.l2 = c.lr & ~o.l3 & 0xFFFFu;
    const u32 known = (~c.lr & 0xFFFFu) << 1;
    o.st[0] = known | 1u; o.st[1] = known | 2u; o.st[2] = known | 4u;
    return o;
}
SXD bool wv_eucjp_orbit_step(WvEucOrbit& o) {
    bool changed = false;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const u32 nw = o.st[s] | ((o.st[s] & o.l2) << 2) | ((o.st[s] & o.l3) << 3);
        changed = changed || nw != o.st[s];
        o.st[s] = nw;
    }
    return changed;
}
SXD u32 wv_eucjp_over(const WvEucOrbit& o, u32 s) { return wv_ctz64((u64)(o.st[s] >> 16) | 8ull); }   // bytes of the next lane the last token covers
// the lane's masks from its token starts.  ws6: the dwords at lane offset -4 .. +19; cov_in: bytes at the lane's start that belong to a
// token begun before; n_exist: existing bytes from the lane's first on (<= 20)
template <class T2>
SXD WvMasks18 wv_classify16_eucjp_swar(const T2& t2, u32 kana, const u32* ws6, const WvEucPre& c, const WvEucOrbit& o, u32 cov_in, u32 n_exist) {
    const u32 ex = n_exist >= 20 ? 0xFFFFFu : ((1u << n_exist) - 1u);
    const u32 s0 = o.st[cov_in] & ~((1u << cov_in) - 1u) & 0xFFFFu & ex;   // token starts among the lane's bytes
    const u32 multi = s0 & c.lr;
    WvMasks18 m{ 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u32 has = (multi >> (2 * k)) & 3u;
        if (!has) continue;
        const u32 odd = has >> 1, p = 2u * (u32)k + odd;
        const int bi = 4 + 2 * k;                                  // byte index of lane offset 2k in ws6
        const u32 w32 = (bi & 3) == 0 ? ws6[bi >> 2] : ((ws6[bi >> 2] >> 16) | (ws6[(bi >> 2) + 1 > 5 ? 5 : (bi >> 2) + 1] << 16));
        const u32 t = odd ? (w32 >> 8) | ((k == 7 ? ws6[5] >> 16 : 0u) << 24) : w32;   // the token's bytes (an odd start in the last slot: byte 18 too)
        const bool three = ((o.l3 >> p) & 1u) != 0;
        const u32 len = three ? 3u : 2u, end = p + len - 1;
        if (p + len > n_exist) continue;                          // pending at the buffer's end
        const u32 lead = t & 0xFFu, b1 = (t >> 8) & 0xFFu, b2 = (t >> 16) & 0xFFu, last = three ? b2 : b1;
        const u32 code = wv_eucjp_token_code(t2, kana, lead, b1, b2, three);
        if (code & 1u) { m.f |= 1u << p; m.e |= 1u << end; m.a |= (code >> 1) << end; }
        else if (last < 0x80u) { m.mb |= 1u << end; m.f |= 1u << end; m.e |= 1u << end; m.a |= ((c.a1 >> end) & 1u) << end; }
        else m.ma |= 1u << end;
    }
    const u32 one = s0 & ~c.lr;                                     // tokens of one byte: a character if below 0x80
    m.f |= one & c.asc; m.e |= one & c.asc; m.a |= one & c.a1; m.ma |= one & ~c.asc;
    return m;
}

// the window of a two-byte Mission.  back: the E | MA bits and the F bits of the byte right in front of the window.
SXD WvWin wv_win_dbcs(WvMask E, WvMask A, WvMask F, WvMask G, WvMask MA, WvMask MB, WvMask O2, WvMask O3, WvMask O4,
                      bool back_done, bool back_f, bool has_back, bool slice_start, u32 n, u32 n_min) {
    WvWin w;
    w.E = E; w.A = A; w.F = F; w.O2 = O2; w.O3 = O3; w.O4 = O4; w.n = n;
    w.LS = wv_long_starts(G, n_min); w.G = G;
    w.CS = wm_and(wm_or(wm_shl1(MA), MB), wm_andn(wm_below(n), wm_below(1)));
    w.tail_empty = n && wm_test(MA, n - 1) ? 1u : 0u;
    w.pre_empty = n && wm_test(MB, 0) ? 1u : 0u;
    w.head_pend = has_back && !back_done ? 1u : 0u;       // the byte in front is a lead byte still waiting for its trail
    w.head_back = 0;
    const u32 e0 = wm_next(E, 0);
    if (e0 == 0 && !wm_test(F, 0) && back_f) w.head_back = 1;   // the first char's lead byte lies in front of the window
    w.probe_before = 0;
    w.slice_start = slice_start ? 1u : 0u;
    w.probe_hb = slice_start ? w.head_back : 0u;
    w.tail_pend = n && !wm_test(E, n - 1) && !wm_test(MA, n - 1) ? 1u : 0u;
    return w;
}

// ... from the five masks of wv_classify16_dbcs_swar: a pair ends where a char ends that did not begin there; its UTF-8 form has
// pair_len bytes if it is accepted (the lengths of rejected chars are never asked for), at least two in any case (the slice-start
// probe's "not ASCII", finding_collection.rs:176); G = the accepted chars' bytes.  (The lead byte of a pair that ends in the NEXT window is
// not in G here: it is no character of this window.)
SXD WvWin wv_win_dbcs_swar(WvMask E, WvMask A, WvMask F, WvMask MA, WvMask MB, u32 pair_len, bool back_done, bool back_f, bool has_back,
                           bool slice_start, u32 n, u32 n_min) {
    const WvMask PE = wm_andn(E, F), APE = wm_and(A, PE);
    const WvMask G = wm_or(A, wm_shr(APE, 1));
    return wv_win_dbcs(E, A, F, G, MA, MB, PE, pair_len >= 3 ? APE : wm_zero(), pair_len >= 4 ? APE : wm_zero(), back_done, back_f, has_back,
                       slice_start, n, n_min);
}

// The window of an EUC-JP Mission from the five masks of wv_classify16_eucjp_swar.  done1 / done2: the byte one / two in front of the
// window ends a token (E | MA); f1 / f2: it begins a character; has1 / has2: it exists.  A token may have two bytes in front of the
// window (8F, A1..FE): head_pend / head_back / probe_hb go up to 2, tail_pend counts the bytes of the token the window ends inside.
SXD WvWin wv_win_eucjp_swar(WvMask E, WvMask A, WvMask F, WvMask MA, WvMask MB, u32 char_len, bool done1, bool done2, bool f1, bool f2,
                            bool has1, bool has2, bool slice_start, u32 n, u32 n_min) {
    const WvMask PE = wm_andn(E, F), APE = wm_and(A, PE);
    const WvMask g1 = wm_shr(APE, 1), g2 = wm_shr(wm_andn(g1, F), 1);        // the bytes in front of an accepted character's last one
    const WvMask G = wm_or(A, wm_or(g1, g2));
    WvWin w = wv_win_dbcs(E, A, F, G, MA, MB, PE, char_len >= 3 ? APE : wm_zero(), wm_zero(), done1, f1, has1, slice_start, n, n_min);
    const bool open1 = has1 && !done1, open2 = open1 && has2 && !done2;
    w.head_pend = open2 ? 2u : open1 ? 1u : 0u;
    w.head_back = (open1 && f1) ? 1u : (open2 && f2) ? 2u : 0u;
    w.probe_hb = slice_start ? w.head_back : 0u;
    const WvMask dn = wm_or(E, MA);
    const bool last_open = n && !wm_test(dn, n - 1);
    const bool prev_open = n >= 2 ? !wm_test(dn, n - 2) : open1;
    w.tail_pend = last_open ? (prev_open ? 2u : 1u) : 0u;
    return w;
}

// WV_PROBE settled (UTF-8): `slice` = the slice's first bytes (n of them, n <= 32 is enough), `left` = the lb bytes of the
// leftover.  The fresh decoder's output (at most 8 bytes, whole chars) must equal the first bytes of [leftover][the call's output];
// the call's output begins like the fresh decoder's (same bytes, same neutral decoder).
SXD u32 wv_resolve_probe(const u8* slice, u32 n, const u8* left, u32 lb) {
    DDecoder fresh;
    ddec_reset(fresh, 1, nullptr);
    u8 probe[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const DStep pr = ddec_utf8(fresh, slice, n < 32 ? n : 32, probe, 8, true);
    bool same = pr.written != 0;
    for (u32 t = 0; t < pr.written && same; t++) same = (t < lb ? left[t] : probe[t - lb]) == probe[t];
    return same ? WV_EXACT : WV_BEFORE;
}

// WV_PROBE settled for the two-byte family by decoding: the call's output comes from a decoder that holds the `hb` bytes in front of
// the slice (its pending lead byte; 0: it is neutral), the probe's from a fresh one (finding_collection.rs:176-207).  left_src: the
// source bytes of the leftover that lies at the front of the output buffer (lsrc of them, lb bytes once decoded).
SXD u32 wv_transcode_dbcs(int enc, const uint16_t* table, const u8* s, u32 n, u8* dst);
SXD u32 wv_resolve_probe_dbcs(int enc, const uint16_t* table, const u8* slice, u32 n, const u8* left_src, u32 lsrc, u32 lb, u32 hb) {
    DDecoder real, fresh;
    ddec_reset(real, enc, table); ddec_reset(fresh, enc, table);
    u8 sink[8], out[16], probe[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, left[64 * 4 + 8];
    if (lb) (void)wv_transcode_dbcs(enc, table, left_src, lsrc, left);
    const bool euc = enc == kEncEucJp;
    if (hb) (void)(euc ? ddec_eucjp(real, slice - hb, hb, sink, sizeof sink, false) : ddec_big5(real, slice - hb, hb, sink, sizeof sink, false));
    const u32 m = n < 32 ? n : 32;
    const DStep ro = euc ? ddec_eucjp(real, slice, m < 12 ? m : 12, out, sizeof out, false)
                         : ddec_big5(real, slice, m < 12 ? m : 12, out, sizeof out, false);   // (its first 8 bytes are all that is compared)
    const DStep pr = euc ? ddec_eucjp(fresh, slice, m, probe, 8, true) : ddec_big5(fresh, slice, m, probe, 8, true);
    bool same = pr.written != 0;
    for (u32 t = 0; t < pr.written && same; t++) {
        const u8 have = t < lb ? left[t] : (t - lb < ro.written ? out[t - lb] : (u8)0);
        same = have == probe[t];
    }
    return same ? WV_EXACT : WV_BEFORE;
}

// the string of a finding of the two-byte family: its source bytes [s, s + n) token by token (it begins and ends on token
// boundaries; tokens of two code points never occur in one: the Missions that take this path reject both)
SXD u32 wv_transcode_dbcs(int enc, const uint16_t* table, const u8* s, u32 n, u8* dst) {
    u32 w = 0, p = 0;
    while (p < n) {
        // the token's bytes in one load where four bytes are left (an unaligned dword: one trip to memory per token instead of one per byte —
        // the writer waits for these loads most of its time)
        u32 t4;
        if (p + 4 <= n) __builtin_memcpy(&t4, s + p, 4);
        else { t4 = s[p]; if (p + 1 < n) t4 |= (u32)s[p + 1] << 8; if (p + 2 < n) t4 |= (u32)s[p + 2] << 16; }
        const u8 b = (u8)t4, b1 = (u8)(t4 >> 8), b2 = (u8)(t4 >> 16);
        if (b < 0x80) { dst[w++] = b; p++; continue; }
        if (enc == kEncEucJp) {   // 8E + katakana, 8F + a cell of index jis0212, else a cell of index jis0208 (sx_codec_core.hpp ddec_eucjp)
            if (b == 0x8E) { w += dput_cp(dst + w, 0xFF61u - 0xA1u + b1); p += 2; }
            else if (b == 0x8F) { w += dput_cp(dst + w, table[kJisN + (b1 - 0xA1u) * 94u + (b2 - 0xA1u)]); p += 3; }
            else { w += dput_cp(dst + w, table[(b - 0xA1u) * 94u + (b1 - 0xA1u)]); p += 2; }
            continue;
        }
        if (two_byte_lead(enc, b)) {
            u32 second = 0;
            w += dput_cp(dst + w, two_byte_lookup(enc, table, b, b1, &second));
            p += 2;
            continue;
        }
        w += dput_cp(dst + w, two_byte_single(enc, b));
        p++;
    }
    return w;
}

// ------------------------------------------------------------------------------------------
// -g (round 5): WvWin::GC.  The bytes of the window that are the grep char, read by the window's own lane (eight loads of 16 bytes —
// the lines are in the cache: the batch's classification has just read them; only Missions with -g do this) and compared as SWAR;
// UTF-16 (KIND 3): the UNITS that are the grep char, marked on their last byte like E.  Restricted to characters where the window is
// complete: wv_set_grep.
// ------------------------------------------------------------------------------------------
template <int KIND>
SXD WvMask wv_grep_bytes(const u8* p, u32 n, u32 g, bool be) {
    WvMask r{ 0, 0 };
    const u32 pat = KIND == 3 ? (be ? (g << 8) : g) * 0x00010001u : g * 0x01010101u;
#pragma unroll
    for (u32 k = 0; k < 8; k++) {
        if (16 * k >= n) break;
        const u32 nb = n - 16 * k < 16 ? n - 16 * k : 16u;
        u32 x[4] = { 0, 0, 0, 0 };
        if (nb == 16) __builtin_memcpy(x, p + 16 * k, 16);
        else for (u32 t = 0; t < nb; t++) x[t >> 2] |= (u32)p[16 * k + t] << (8 * (t & 3));
        u32 f[4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const u32 y = x[d] ^ pat;
            if (KIND == 3) f[d] = ~(((y & 0x7FFF7FFFu) + 0x7FFF7FFFu) | y) & 0x80008000u;   // units that are zero: a flag on their last byte
            else f[d] = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u;
        }
        u64 m = wv_movemask16_b7(f[0], f[1], f[2], f[3]);
        if (nb < 16) m &= (1ull << nb) - 1ull;
        if (k < 4) r.lo |= m << (16 * k); else r.hi |= m << (16 * (k - 4));
    }
    return r;
}
// KIND 0 / 1: every ASCII byte that is a character (E); 2: a byte that is a character of its own (its first and last byte); 3: a unit (E)
template <int KIND>
SXD void wv_set_grep(WvWin& w, const WvParams& P, const u8* win_bytes, u32 g, bool be) {
    if (!P.grep) { w.GC = wm_zero(); return; }
    const WvMask raw = wv_grep_bytes<KIND>(win_bytes, w.n, g, be);
    w.GC = wm_and(raw, KIND == 2 ? wm_and(w.E, w.F) : w.E);
}

// ------------------------------------------------------------------------------------------
// -r (round 5): WvWin::MBA / D / mb0 / mbl, by the window's own lane from its bytes — one trip per multi-byte character, in order (text in a
// script beyond ASCII: up to W / 2 of them; only Missions with -r that can matter do this; W = 128 bytes at most: 8 chunks).  The lead byte of a character's UTF-8 form:
// UTF-8: its first byte; single byte: from the decoder's table (x-user-defined: EF); UTF-16: from the unit (a pair: F0 | plane bits).
// Codes: 1 + the rank of (lead & 0x3F) among the bits of ubf — five bits in the state (sx_mission.cpp takes Missions with <= 31 such leads).
// ------------------------------------------------------------------------------------------
SXD u32 wv_lead_code(u64 ubf, u32 lead) { return 1u + wv_popc64(ubf & 0x001FFFFFFFFFFFFCull & ((1ull << (lead & 0x3Fu)) - 1ull)); }   // (bits 2 .. 52: the lead bytes C2 .. F4)
SXD u32 wv_lead_of_cp(u32 cp) { return cp < 0x800u ? 0xC0u | (cp >> 6) : cp < 0x10000u ? 0xE0u | (cp >> 12) : 0xF0u | (cp >> 18); }
// The window's bytes are read 16 at a time into registers and the characters' bytes are taken from there (a load per character, each
// waiting for the one before, was a third of the -r kernels' time); LEADF: the lead byte of the UTF-8 form of a single-byte decoder's
// character b >= 0x80 (the kernels: from the decoder table).
template <int KIND, class LEADF>
SXD void wv_set_same(WvWin& w, const u8* win, u64 ubf, bool be, LEADF lead_of_high) {
    w.MBA = wm_zero(); w.D = wm_zero(); w.mb0_e = 128; w.mb0_code = 0; w.mbl_code = 0;
    // where a multi-byte character is looked up — KIND 0: at its byte (found chunk by chunk: the bytes >= 0x80 that are characters);
    // 1 (UTF-8): at its lead byte; 3 (UTF-16): at its unit's last byte (a pair: the low surrogate's)
    WvMask vis = wm_zero();
    if (KIND == 1) vis = wm_andn(w.F, w.E);
    else if (KIND == 3) vis = wm_and(w.E, w.O2);
    const bool any_cs = wm_any(wm_andn(w.CS, wm_bit(0)));
    u32 lm = 0;          // last_multi_char_leading_byte as the call's text alone determines it
    i32 cur_cs = -2;     // the call in hand (its CS bit; 0: the window's first call)
    bool first_seen = false;
    u32 mb0_lead = 0, mbl_lead = 0;
    // one character: its last byte e, the lead byte of its UTF-8 form
    auto visit = [&](u32 e, u32 lead) {
        i32 cs = 0;
        if (any_cs) { cs = wm_prev(w.CS, e); if (cs < 0) cs = 0; }   // (the window's first call, with or without a CS bit at 0)
        if (cs != cur_cs) { cur_cs = cs; lm = 0; }
        if (wm_test(w.A, e)) {
            w.MBA = wm_or(w.MBA, wm_bit(e));
            if (lm && lm != lead) w.D = wm_or(w.D, wm_bit(e));
            if (cs == 0 && !first_seen) { w.mb0_e = e; mb0_lead = lead; }
            mbl_lead = lead;
            lm = lead;
        } else lm = 0;
        if (cs == 0) first_seen = true;
    };
    if (KIND == 1 && w.head_back) {   // a character that began in front of the window and ends in it: its lead byte stands there
        const u32 e0 = wm_next(w.E, 0);
        if (e0 < 128 && wm_prev(w.F, e0) < 0) visit(e0, win[-(i32)w.head_back]);
    }
    u32 prev3 = 0;   // UTF-16: the last dword of the chunk before (a high surrogate may stand there)
    for (u32 k = 0; k < 8 && 16 * k < w.n; k++) {
        const u32 nb = w.n - 16 * k < 16 ? w.n - 16 * k : 16u;
        const u8* p = win + 16 * k;
        u32 x0 = 0, x1 = 0, x2 = 0, x3 = 0;   // (scalars, no array: nothing here may end up in scratch)
        if (nb == 16) { __builtin_memcpy(&x0, p, 4); __builtin_memcpy(&x1, p + 4, 4); __builtin_memcpy(&x2, p + 8, 4); __builtin_memcpy(&x3, p + 12, 4); }
        else for (u32 t = 0; t < nb; t++) { const u32 v = (u32)p[t] << (8 * (t & 3)); if (t < 4) x0 |= v; else if (t < 8) x1 |= v; else if (t < 12) x2 |= v; else x3 |= v; }
        u32 m;
        if (KIND == 0) m = (u32)wv_movemask16_b7(x0 & 0x80808080u, x1 & 0x80808080u, x2 & 0x80808080u, x3 & 0x80808080u) & (u32)((k < 4 ? w.E.lo >> (16 * k) : w.E.hi >> (16 * (k - 4))) & 0xFFFFu);
        else m = (u32)((k < 4 ? vis.lo >> (16 * k) : vis.hi >> (16 * (k - 4))) & 0xFFFFu);
        while (m) {
            const u32 jb = (u32)__builtin_ctz(m);
            m &= m - 1;
            const u32 pos = 16 * k + jb, sel = jb >> 2;
            const u32 xx = sel == 0 ? x0 : sel == 1 ? x1 : sel == 2 ? x2 : x3;
            if (KIND == 0) visit(pos, lead_of_high((xx >> (8 * (jb & 3))) & 0xFFu));
            else if (KIND == 1) {
                const u32 e = wm_next(w.E, pos);   // (its last byte; a character that ends in the next window is that window's)
                if (e < 128) visit(e, (xx >> (8 * (jb & 3))) & 0xFFu);
            } else {   // the unit's two bytes stand in one dword (jb is odd)
                const u32 raw = (xx >> (8 * ((jb & 3) - 1))) & 0xFFFFu, u = be ? ((raw & 0xFFu) << 8) | (raw >> 8) : raw;
                if ((u & 0xFC00u) == 0xDC00u) {   // the pair's low surrogate: its high one is the unit in front (in the dword in front: the chunk's, the chunk before's, or in front of the window)
                    u32 hraw;
                    if ((jb & 3) == 3) hraw = xx & 0xFFFFu;
                    else if (jb >= 4) { const u32 xp = sel == 1 ? x0 : sel == 2 ? x1 : x2; hraw = xp >> 16; }
                    else if (k > 0) hraw = prev3 >> 16;
                    else hraw = (u32)win[-2] | ((u32)win[-1] << 8);
                    const u32 h = be ? ((hraw & 0xFFu) << 8) | (hraw >> 8) : hraw;
                    visit(pos, wv_lead_of_cp(0x10000u + ((h & 0x3FFu) << 10) + (u & 0x3FFu)));
                } else visit(pos, wv_lead_of_cp(u));
            }
        }
        prev3 = x3;
    }
    if (mb0_lead) w.mb0_code = wv_lead_code(ubf, mb0_lead);
    if (mbl_lead) w.mbl_code = wv_lead_code(ubf, mbl_lead);
}
// (single-byte decoders: the lead byte from the decoder's table; nullptr: x-user-defined, U+F780 + b - 0x80)
struct WvLeadOfTable {
    const uint16_t* table;
    SXD u32 operator()(u32 b) const { return wv_lead_of_cp(table ? (u32)table[b - 0x80u] : 0xF780u + (b - 0x80u)); }
};

// ------------------------------------------------------------------------------------------
// Geometry: windows numbered through the buffer (slices of 4096 bytes, windows of W inside; the last window of a
// slice and of the buffer may be short)
// ------------------------------------------------------------------------------------------
constexpr u32 kWvSlice = 4096;
SXD u32 wv_wps(u32 W) { return (kWvSlice + W - 1) / W; }
SXD u64 wv_window_count(u64 len, u32 W) { return len / kWvSlice * wv_wps(W) + (len % kWvSlice + W - 1) / W; }
SXD void wv_window_at(u64 g, u32 W, u32 wps, u64 len, u64* start, u32* n) {
    const u64 s = g / wps, j = g - s * wps;
    const u64 ws = s * kWvSlice + j * W;
    u64 we = ws + W;
    if (we > (s + 1) * kWvSlice) we = (s + 1) * kWvSlice;
    if (we > len) we = len;
    *start = ws;
    *n = we > ws ? (u32)(we - ws) : 0u;
}
// number of the window that starts at byte position p (a window start)
SXD u64 wv_window_no(u64 p, u32 W, u32 wps) { return p / kWvSlice * wps + (p % kWvSlice) / W; }

// first byte of the 1 KiB tiles that cover a batch whose first window starts at span_lo: 16-byte aligned and at least three
// bytes in front of it (UTF-8: a character delivered in the first window may begin there)
SXD u64 wv_tile0(u64 span_lo) { const u64 a = span_lo & ~15ull; return a >= 16 ? a - 16 : a; }

// ------------------------------------------------------------------------------------------
// Descriptors: what the count pass leaves behind for a writer that works a LANE PER FINDING.  The window-parallel writer (the count
// pass run again with an emitter that writes) keeps the lanes of windows without findings idle and classifies every byte a second
// time; with descriptors the second pass reads 12 bytes per finding and nothing else of the count pass' work.
//   w0: string offset inside the wavefront's output (19 bits) | window index among the wavefront's own (10) | precision (2) | completes (1)
//   w1: din (8) | src_rel + 2048 (12) | src_len (12)        w2: out_len (9) | the WV_PROBE payload (lb 9, lback 10, hb 1)
// A wavefront has room for `cap` of them; one that finds more says so through its count (wave_nf > cap) and the launch falls back to the
// window-parallel writer.
// ------------------------------------------------------------------------------------------
struct WvDesc { u32 w0, w1, w2; };
constexpr u32 kWvDescMaxWin = 800;   // windows a wavefront may own with descriptors: 800 x 640 string bytes < 2^19
SXD WvDesc wv_desc_pack(u32 a_local, u32 widx, u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
    WvDesc d;
    d.w0 = (a_local & 0x7FFFFu) | ((widx & 1023u) << 19) | ((prec & 3u) << 29) | (completes ? 0x80000000u : 0u);
    d.w1 = (din & 255u) | (((u32)(src_rel + 2048) & 4095u) << 8) | ((src_len & 4095u) << 20);
    d.w2 = (out_len & 511u) | ((prec >> 8) << 9);
    return d;
}
SXD u32 wv_desc_a_local(const WvDesc& d) { return d.w0 & 0x7FFFFu; }
SXD u32 wv_desc_widx(const WvDesc& d) { return (d.w0 >> 19) & 1023u; }
SXD u32 wv_desc_prec(const WvDesc& d) { return ((d.w0 >> 29) & 3u) | ((d.w2 >> 9) << 8); }   // as wv_call handed it to the emitter
SXD bool wv_desc_completes(const WvDesc& d) { return (d.w0 >> 31) != 0; }
SXD u32 wv_desc_din(const WvDesc& d) { return d.w1 & 255u; }
SXD i32 wv_desc_src_rel(const WvDesc& d) { return (i32)((d.w1 >> 8) & 4095u) - 2048; }
SXD u32 wv_desc_src_len(const WvDesc& d) { return d.w1 >> 20; }
SXD u32 wv_desc_out_len(const WvDesc& d) { return d.w2 & 511u; }

// the count pass' second emitter: descriptors at slot[0 ..], string offsets from a_local on; `room` slots are left (beyond them: counted only)
struct WvDescEmit {
    WvDesc* slot;
    u32 room, a_local, widx;
    SXD void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
        if (room) { *slot++ = wv_desc_pack(a_local, widx, din, prec, completes, src_rel, src_len, out_len); room--; }
        a_local += out_len;
    }
};

// the count pass' emitter: counts, and keeps the first K findings of the window as descriptors (string offset: inside the window's own
// output; the wavefront's prefix sums are added when they are known) — most windows hold no more, and the others are replayed once
// more with WvDescEmit
template <int K> struct WvCountEmit {
    u32 nf = 0, nb = 0, widx = 0;
    WvDesc d0{ 0, 0, 0 }, d1{ 0, 0, 0 }, d2{ 0, 0, 0 };
    SXD void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
        if (K > 0 && nf < (u32)K) {
            const WvDesc x = wv_desc_pack(nb, widx, din, prec, completes, src_rel, src_len, out_len);
            if (nf == 0) d0 = x; else if (K > 1 && nf == 1) d1 = x; else if (K > 2) d2 = x;
        }
        nf++; nb += out_len;
    }
};

// the count pass' emitter since round 4: counts, and STAGES the window's first kWvStage findings as descriptors — in LDS, where the
// batch's masks lay (every lane has pulled its window's masks into registers by then): [finding j][word][lane], so that a
// wavefront's stores never meet in a bank whatever j the lanes are at.  When the batch's prefix sums are known every lane moves its
// descriptors to the wavefront's list.  Before, two were kept in registers and any window with more was replayed once more — on
// `-e ascii -n 4` (1.5 findings per window) nearly every batch paid that second replay: a quarter of the count pass.
// (-r kernels, round 5: twice as many — text whose lead bytes change every few characters leaves 5 to 10 findings per window, and their
// wavefronts have LDS to spare)
constexpr u32 kWvStage = 6, kWvStageSame = 12;
template <class PTR> struct WvStageEmit {
    PTR stage;
    u32 lane = 0, widx = 0, nf = 0, nb = 0, cap = kWvStage;
    SXD void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
        if (nf < cap) {
            const WvDesc x = wv_desc_pack(nb, widx, din, prec, completes, src_rel, src_len, out_len);
            PTR p = stage + nf * 192u + lane;
            p[0] = x.w0; p[64] = x.w1; p[128] = x.w2;
        }
        nf++; nb += out_len;
    }
};

constexpr u32 kWvWarm = 4;         // windows a wavefront replays in front of its own, only for their state
constexpr u32 kWvBatch = 64;       // windows per batch: one per lane
constexpr u32 kWvMaxTiles = 9;     // 1 KiB tiles that cover a batch of 64 windows of <= 128 bytes: 8 KiB + the 16..31 bytes in front of it (wv_tile0)

}  // namespace sx
