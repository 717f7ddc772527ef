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
// the k-th (1-based) set bit of m at or above `from`; 128 if there are fewer
SXD u32 wm_select(WvMask m, u32 from, u32 k) {
    WvMask r = wm_andn(m, wm_below(from));
    const u32 pl = wv_popc64(r.lo);
    u64 v;
    u32 base;
    if (k <= pl) { v = r.lo; base = 0; } else { v = r.hi; base = 64; k -= pl; }
    if (k > wv_popc64(v)) return 128u;
    for (u32 t = 1; t < k; t++) v &= v - 1;
    return base + wv_ctz64(v);
}

// ------------------------------------------------------------------------------------------
// State carried from window to window, packed into 32 bits for the lane-to-lane exchange
// ------------------------------------------------------------------------------------------
struct WvState { u32 lc, lb, lback, cut; };
SXD u32 wv_pack(const WvState& s) { return s.lc | (s.lb << 7) | (s.lback << 16) | (s.cut << 26); }
SXD WvState wv_unpack(u32 v) { return WvState{ v & 127u, (v >> 7) & 511u, (v >> 16) & 1023u, (v >> 26) & 1u }; }

struct WvParams { u32 q, n_min; };

// One window as the state machine sees it.
struct WvWin {
    WvMask E, A;     // char ends delivered in this window; those whose char passes the filter
    WvMask F;        // first bytes of the characters (multi-byte encodings; single byte: unused)
    WvMask CS;       // bit i (1 <= i < n): a decoder call starts at byte i
    WvMask O2, O3;   // single byte: bytes whose UTF-8 form has 2 / 3 bytes (str_len = source bytes + popc(O2) + 2 popc(O3))
    u32 n;           // bytes in the window
    u32 pre_empty;   // an empty decoder call at byte 0 precedes the first one (UTF-8: the byte a pending sequence rejects is read again)
    u32 tail_empty;  // a decoder call starts exactly at the window end: one more (empty) call in this window
    u32 head_back;   // bytes of the first delivered character that lie in front of the window (0..3)
    u32 probe_before;// the slice-start probe (:176-207) marks the first call's first chunk `Before`
};

enum { WV_BEFORE = 0, WV_EXACT = 1, WV_AFTER = 2 };   // == SX_PRECISION_*

SXD bool wv_mission_ok(int grep_char, u32 same_block, u32 n_min, u32 q) {
    return grep_char < 0 && !same_block && n_min >= 1 && n_min <= q && q <= 64;
}

// One decoder call [din, cend) of the window: finding_collection.rs:146-290 with SplitStr::next inlined.
// EMIT(din, precision, completes, src_rel, src_len, out_len): src_rel = first source byte relative to the window start
// (negative: in front of it), out_len = bytes of the string.
template <bool BYTES, class EMIT>
SXD void wv_call(const WvParams& P, const WvWin& w, WvState& st, u32 din, u32 cend, bool invalid_after, bool first_call, EMIT& emit) {
    const bool cont = st.cut != 0;   // :240-241: consumed by this call whatever it yields
    st.cut = 0;
    u32 lrem = st.lc;
    const u32 lbytes = st.lb, lback = st.lback;
    const bool has_left = lrem > 0;
    st.lc = 0; st.lb = 0; st.lback = 0;   // :211-227: the leftover is prepended, then gone
    const WvMask rng = wm_range(din, cend);
    const WvMask Ec = wm_and(w.E, rng);
    if (!has_left && !wm_any(Ec)) return;
    const WvMask Rc = wm_andn(Ec, w.A);   // rejected (valid) chars of this call
    u32 prec = (has_left || (first_call && w.probe_before)) ? WV_BEFORE : WV_EXACT;   // :146, 214-221, 176-207
    bool last_cut = cont, tl = true;      // SplitStr: last_s_was_maybe_cut; "the scan position is inp_start_p"
    u32 p = din;                          // chars whose last byte lies below p are consumed
    for (;;) {                            // one SplitStr::next per round
        u32 ok_n = 0, out_b = 0;
        i32 src0 = 0;
        u32 src_end = p;
        bool started = false, piece_tl = tl;
        if (lrem) { ok_n = lrem; out_b = lbytes; src0 = -(i32)lback; started = true; lrem = 0; }
        int reason = 0;   // 0: the call's text ended, 1: a rejected char, 2: q chars collected (helper.rs:237)
        for (;;) {
            const u32 e = wm_next(Ec, p);
            if (e >= 128) { reason = 0; break; }
            if (!wm_test(w.A, e)) {
                p = e + 1;
                if (ok_n == 0) { piece_tl = false; tl = false; continue; }   // helper.rs:327-330: the string would start behind it
                reason = 1;
                break;
            }
            const u32 er = wm_next(Rc, e);                       // accepted chars follow one another up to here
            const WvMask av = wm_and(Ec, wm_range(e, er));
            const u32 navail = wm_popc(av);
            const u32 take = P.q - ok_n < navail ? P.q - ok_n : navail;
            const u32 last = BYTES ? e + take - 1 : wm_select(av, e, take);
            i32 c0;   // first source byte of the char that ends at e
            if (BYTES) c0 = (i32)e;
            else { const i32 f = wm_prev(w.F, e); c0 = f < 0 ? -(i32)w.head_back : f; }
            if (!started) { src0 = c0; started = true; }
            if (BYTES) {
                const WvMask tr = wm_range(e, last + 1);
                out_b += take + wm_popc(wm_and(w.O2, tr)) + 2 * wm_popc(wm_and(w.O3, tr));
            } else out_b += (u32)((i32)(last + 1) - c0);
            ok_n += take;
            src_end = last + 1;
            p = last + 1;
            if (ok_n >= P.q) { reason = 2; break; }
        }
        if (ok_n == 0) break;   // helper.rs:343
        if (reason == 1) {
            const bool exit3 = last_cut && piece_tl, exit4 = ok_n >= P.n_min;   // helper.rs:315-317
            if (!exit3 && !exit4) { tl = false; continue; }                    // :327-330
        }
        const bool touches_right = reason == 0 || (reason == 2 && wm_next(Ec, p) >= 128);
        const bool maybe_cut = ok_n >= P.q || (touches_right && !invalid_after);     // helper.rs:353-355
        const bool completes = piece_tl && last_cut;                                // :365
        const bool again = !completes && touches_right && !invalid_after && ok_n < P.q;   // :389-392
        if (!completes && !again && ok_n < P.n_min) break;                          // :410-415
        tl = ok_n >= P.q;         // :418-420: inp_start_p moves behind a full line; else nothing later touches it
        last_cut = maybe_cut;     // :421
        if (again) {              // finding_collection.rs:269-285
            st.lc = ok_n; st.lb = out_b; st.lback = (u32)((i32)w.n - src0); st.cut = 0;
        } else {                  // :255-268
            emit(din, prec, completes, src0, (u32)((i32)src_end - src0), out_b);
            st.lc = 0; st.lb = 0; st.lback = 0; st.cut = maybe_cut ? 1u : 0u;
        }
        prec = WV_AFTER;          // :289
    }
}

// One window: its decoder calls in order (finding_collection.rs:134-325).  Calls that hold no accepted char and
// meet no leftover only clear the cut flag: they are skipped in bulk.
template <bool BYTES, class EMIT>
SXD void wv_window(const WvParams& P, const WvWin& w, WvState& st, EMIT& emit, bool skip_idle_calls = true) {
    if (w.pre_empty) wv_call<BYTES>(P, w, st, 0u, 0u, true, false, emit);
    u32 din = 0;
    bool first = true;
    for (;;) {
        if (skip_idle_calls && st.lc == 0 && !(first && w.probe_before)) {
            const u32 a = wm_next(w.A, din);
            if (a >= w.n) { st.cut = 0; return; }   // (at least the call at din is still to come, and none of them yields)
            const i32 cs = wm_prev(w.CS, a);        // start of the call that delivers the next accepted char
            if (cs > (i32)din) { st.cut = 0; din = (u32)cs; first = false; }
        }
        u32 cend = wm_next(w.CS, din + 1);
        if (cend > w.n) cend = w.n;
        const bool last = cend >= w.n;
        wv_call<BYTES>(P, w, st, din, cend, !last || w.tail_empty != 0, first, emit);
        first = false;
        if (last) break;
        din = cend;
    }
    if (w.tail_empty) wv_call<BYTES>(P, w, st, w.n, w.n, false, false, emit);
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

// the window of a single-byte Mission from its valid / accepted / length masks
SXD WvWin wv_win_single(WvMask V, WvMask A, WvMask O2, WvMask O3, u32 n) {
    WvWin w;
    w.E = V; w.A = A; w.F = V; w.O2 = O2; w.O3 = O3; w.n = n;
    const WvMask bad = wm_andn(wm_below(n), V);        // a byte without a character: Malformed(1, 0), the call ends behind it
    w.CS = wm_and(wm_shl1(bad), wm_below(n));
    w.tail_empty = n && wm_test(bad, n - 1) ? 1u : 0u;
    w.pre_empty = 0; w.head_back = 0; w.probe_before = 0;
    return w;
}

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

constexpr u32 kWvWarm = 4;         // windows a wavefront replays in front of its own, only for their state
constexpr u32 kWvBatch = 64;       // windows per batch: one per lane
constexpr u32 kWvMaxTiles = 10;    // 1 KiB tiles that cover a batch of 64 windows of <= 128 bytes (+ alignment slack)

}  // namespace sx
