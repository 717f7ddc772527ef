// sx_codec_core.hpp — the product's ONE implementation of the decoders and of SplitStr.
//
// Compiled three ways from this single source: as device code (sx_replay_dev.hip, SXD =
// `__device__ __forceinline__`), as host code inside the library (sx_decoder.cpp wraps it as the
// `Decoder` / `SplitStr` classes of sx_host.hpp, SXD = `inline`), and by the test harness
// tests/native/replay_core_host.cpp.  The oracle (oracle/sxo.c) is written independently.
//
// Decoders: the call contract of encoding_rs 0.8.34 `Decoder::decode_to_str_without_replacement`
// (a Cargo dependency of the reference, Cargo.toml:19; call sites src/finding_collection.rs:138-143,
// 180-194): what matters to the caller is (result, read, written) — a malformed sequence ends the
// call, `read` says where the next call starts.  Algorithms: WHATWG Encoding Standard "utf-8
// decoder", "utf-16 decoder" (with the crate's streaming treatment of unpaired surrogates),
// "x-user-defined decoder", "single-byte decoder", "Big5 decoder", "EUC-JP decoder".
// SplitStr::next: reference src/helper.rs:206-433; filter bit tests src/mission.rs:333-348.
#pragma once
#include <stdint.h>

namespace sx {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;

enum { RES_INPUT_EMPTY = 0, RES_OUTPUT_FULL = 1, RES_MALFORMED = 2 };

struct DStep { int result; u32 read, written; };

// Encoding families the code is compiled for (template parameter ENC): 1 UTF-8, 2 UTF-16LE,
// 3 UTF-16BE, 4 Big5, 5 EUC-JP, 0 every single-byte encoding (SX_ENC_* 16.., x-user-defined).
// Family 4 = the two-byte encodings (Big5, Shift_JIS, EUC-KR: a lead byte and one more; which one is a run-time
// value), family 5 = EUC-JP (three-byte tokens too).
constexpr int kEncUtf16le = 2, kEncUtf16be = 3;
constexpr int kEncBig5 = 64, kEncEucJp = 65, kEncShiftJis = 66, kEncEucKr = 67, kEncGb18030 = 68, kEncGbk = 69;  // == SX_ENC_* (include/stringsext_amd.h)
constexpr bool enc_is_gb(int e) { return e == kEncGb18030 || e == kEncGbk; }   // GBK decodes as gb18030
constexpr int enc_family(u32 encoding) {
    return encoding == 1 ? 1 : encoding == 2 ? 2 : encoding == 3 ? 3
         : (encoding == (u32)kEncBig5 || encoding == (u32)kEncShiftJis || encoding == (u32)kEncEucKr || enc_is_gb((int)encoding)) ? 4 : encoding == (u32)kEncEucJp ? 5 : 0;
}
// Layout of the double-byte tables (csrc/gen_tables.py): one uint16_t blob per encoding.
//   Big5: [kBig5N low halves][kBig5P2Words plane-2 bitmap];  EUC-JP: [kJisN jis0208][kJisN jis0212];
//   Shift_JIS: [kSjisN: index jis0208 up to the IBM extension rows];  EUC-KR: [kEucKrN].
//   gb18030 / GBK: [kGbN two-byte cells][kGbRanges breakpoint pointers][kGbRanges code points] (index gb18030 ranges).
constexpr u32 kBig5N = 126 * 157, kBig5P2Words = (kBig5N + 15) / 16, kJisN = 94 * 94, kSjisN = 11280, kEucKrN = 126 * 190;
constexpr u32 kGbN = 126 * 190, kGbRanges = 208;

struct DDecoder {
    int enc;
    u32 cp; u8 seen, needed, lower, upper;               // UTF-8
    int lead_byte; u32 lead_surrogate; bool pending_bmp;  // UTF-16
    u8 dlead, dflag;                                      // Big5 / EUC-JP: pending lead byte; EUC-JP: the lead is the 2nd byte of 8F xx
    // gb18030 (family 4 with four-byte tokens): dlead = "gb18030 first", gb2 / gb3 = second / third; rq = bytes the algorithm
    // "prepends to the stream" that an EARLIER call consumed: they are decoded again in front of the next call's input
    u8 gb2, gb3, rq[2], rq_n;
    // ISO-2022-JP: which set the bytes select (what an escape sequence changes), the set that was selected when an escape
    // sequence began, the byte in hand, "an escape sequence and no character since", and the `$` / `(` of a broken escape
    // sequence that the next call decodes in front of its input
    u8 iso_set, iso_out, iso_mid, iso_b1, iso_flag, iso_pend;
    const uint16_t* table;  // single byte: 128 entries (nullptr = x-user-defined); Big5 / EUC-JP: the blob
};

SXD void ddec_reset(DDecoder& d, int enc, const uint16_t* table) {
    d.enc = enc; d.cp = 0; d.seen = d.needed = 0; d.lower = 0x80; d.upper = 0xBF;
    d.lead_byte = -1; d.lead_surrogate = 0; d.pending_bmp = false; d.dlead = 0; d.dflag = 0; d.gb2 = d.gb3 = 0; d.rq[0] = d.rq[1] = 0; d.rq_n = 0; d.table = table;
    d.iso_set = d.iso_out = d.iso_mid = d.iso_b1 = d.iso_flag = d.iso_pend = 0;
}

SXD u32 dput_cp(u8* d, u32 c) {
    if (c < 0x80) { d[0] = (u8)c; return 1; }
    if (c < 0x800) { d[0] = (u8)(0xC0 | (c >> 6)); d[1] = (u8)(0x80 | (c & 0x3F)); return 2; }
    if (c < 0x10000) {
        d[0] = (u8)(0xE0 | (c >> 12)); d[1] = (u8)(0x80 | ((c >> 6) & 0x3F)); d[2] = (u8)(0x80 | (c & 0x3F));
        return 3;
    }
    d[0] = (u8)(0xF0 | (c >> 18)); d[1] = (u8)(0x80 | ((c >> 12) & 0x3F));
    d[2] = (u8)(0x80 | ((c >> 6) & 0x3F)); d[3] = (u8)(0x80 | (c & 0x3F));
    return 4;
}

SXD DStep ddec_utf8(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    u32 i = 0, w = 0;
    for (;;) {
        if (i >= n) {
            if (last && d.needed != 0) {
                d.cp = 0; d.needed = d.seen = 0; d.lower = 0x80; d.upper = 0xBF;
                return { RES_MALFORMED, i, w };
            }
            return { RES_INPUT_EMPTY, i, w };
        }
        if (cap - w < 4) return { RES_OUTPUT_FULL, i, w };
        const u8 b = src[i++];
        if (d.needed == 0) {
            if (b < 0x80) { dst[w++] = b; continue; }
            if (b >= 0xC2 && b <= 0xDF) { d.needed = 1; d.cp = b & 0x1F; continue; }
            if (b >= 0xE0 && b <= 0xEF) {
                if (b == 0xE0) d.lower = 0xA0;
                if (b == 0xED) d.upper = 0x9F;
                d.needed = 2; d.cp = b & 0x0F; continue;
            }
            if (b >= 0xF0 && b <= 0xF4) {
                if (b == 0xF0) d.lower = 0x90;
                if (b == 0xF4) d.upper = 0x8F;
                d.needed = 3; d.cp = b & 0x07; continue;
            }
            return { RES_MALFORMED, i, w };
        }
        if (b < d.lower || b > d.upper) {
            d.cp = 0; d.needed = d.seen = 0; d.lower = 0x80; d.upper = 0xBF;
            return { RES_MALFORMED, i - 1, w };  // un-read
        }
        d.lower = 0x80; d.upper = 0xBF;
        d.cp = (d.cp << 6) | (b & 0x3F);
        if (++d.seen != d.needed) continue;
        w += dput_cp(dst + w, d.cp);
        d.cp = 0; d.needed = d.seen = 0;
    }
}

template <bool BE>
SXD DStep ddec_utf16(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    constexpr bool be = BE;
    u32 i = 0, w = 0;
    if (d.pending_bmp) {
        if (cap - w < 3) return { RES_OUTPUT_FULL, 0, 0 };
        w += dput_cp(dst + w, d.lead_surrogate);
        d.pending_bmp = false; d.lead_surrogate = 0;
    }
    for (;;) {
        if (d.lead_byte < 0 && d.lead_surrogate == 0) {
            while (n - i >= 2 && cap - w >= 4) {
                const u32 u = be ? ((u32)src[i] << 8) | src[i + 1] : ((u32)src[i + 1] << 8) | src[i];
                if ((u & 0xF800) != 0xD800) { w += dput_cp(dst + w, u); i += 2; continue; }
                if ((u & 0xFC00) == 0xDC00) { i += 2; return { RES_MALFORMED, i, w }; }
                if (n - i < 4) break;
                const u32 v = be ? ((u32)src[i + 2] << 8) | src[i + 3] : ((u32)src[i + 3] << 8) | src[i + 2];
                if ((v & 0xFC00) != 0xDC00) { i += 2; return { RES_MALFORMED, i, w }; }
                w += dput_cp(dst + w, 0x10000u + ((u & 0x3FF) << 10) + (v & 0x3FF));
                i += 4;
            }
        }
        if (i >= n) {
            if (last && (d.lead_surrogate != 0 || d.lead_byte >= 0)) {
                d.lead_surrogate = 0; d.lead_byte = -1;
                return { RES_MALFORMED, i, w };
            }
            return { RES_INPUT_EMPTY, i, w };
        }
        if (cap - w < 4) return { RES_OUTPUT_FULL, i, w };
        const u8 b = src[i++];
        if (d.lead_byte < 0) { d.lead_byte = b; continue; }
        const u32 u = be ? ((u32)d.lead_byte << 8) | b : ((u32)b << 8) | (u32)d.lead_byte;
        d.lead_byte = -1;
        if ((u & 0xFC00) == 0xD800) {
            if (d.lead_surrogate != 0) { d.lead_surrogate = u; return { RES_MALFORMED, i, w }; }
            d.lead_surrogate = u;
            continue;
        }
        if ((u & 0xFC00) == 0xDC00) {
            if (d.lead_surrogate == 0) return { RES_MALFORMED, i, w };
            w += dput_cp(dst + w, 0x10000u + ((d.lead_surrogate & 0x3FF) << 10) + (u & 0x3FF));
            d.lead_surrogate = 0;
            continue;
        }
        if (d.lead_surrogate != 0) { d.lead_surrogate = u; d.pending_bmp = true; return { RES_MALFORMED, i, w }; }
        w += dput_cp(dst + w, u);
    }
}

constexpr int kEncReplacement = 70;   // == SX_ENC_REPLACEMENT
constexpr int kEncIso2022Jp = 71;     // == SX_ENC_ISO_2022_JP

// ISO-2022-JP (WHATWG "ISO-2022-JP decoder") as transitions over (selected set, where in a token we are):
//   iso_set: 0 ASCII, 1 JIS X 0201 Roman, 2 half-width katakana, 3 JIS X 0208 (two bytes per character);
//   iso_mid: 0 between tokens, 1 the first byte of a JIS X 0208 pair is in iso_b1, 2 ESC seen, 3 ESC and `$` / `(` (in iso_b1) seen.
// An escape sequence that does not complete is an error of the ESC alone: the byte that broke it is read again, and a `$` / `(`
// that was already consumed is decoded in front of the NEXT call's input (encoding_rs keeps it pending; its char then carries
// the position of that call).  Two escape sequences without a character between them: the second one is an error (iso_flag).
// The table is index jis0208 (the first kJisN words of the EUC-JP blob).  Never runs on the device: the set in force at a byte is
// not derivable from a bounded look-back, so such a Mission is ONE sequential pass on the host (sx_stage_b.cpp).
SXD DStep ddec_iso2022jp(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    u32 i = 0, w = 0;
    bool from_pend = d.iso_pend != 0;
    for (;;) {
        u8 b;
        if (from_pend) { b = d.iso_pend; d.iso_pend = 0; }
        else {
            if (i >= n) {
                if (!last || d.iso_mid == 0) return { RES_INPUT_EMPTY, i, w };
                // the input ends inside a token
                const u8 mid = d.iso_mid;
                d.iso_mid = 0;
                if (mid == 3) d.iso_pend = d.iso_b1;
                if (mid >= 2) { d.iso_flag = 0; d.iso_set = d.iso_out; }
                return { RES_MALFORMED, i, w };
            }
            if (cap - w < 3) return { RES_OUTPUT_FULL, i, w };
            b = src[i++];
        }
        const bool was_pend = from_pend;
        from_pend = false;
        if (was_pend && cap - w < 3) { d.iso_pend = b; return { RES_OUTPUT_FULL, i, w }; }
        switch (d.iso_mid) {
        case 2:   // after ESC
            if (b == 0x24 || b == 0x28) { d.iso_b1 = b; d.iso_mid = 3; continue; }
            i--;  // not an escape sequence: the byte is read again as what it is in the set that was in force
            d.iso_mid = 0; d.iso_flag = 0; d.iso_set = d.iso_out;
            return { RES_MALFORMED, i, w };
        case 3: { // after ESC `$` or ESC `(`
            int set = -1;
            if (d.iso_b1 == 0x28) set = b == 0x42 ? 0 : b == 0x4A ? 1 : b == 0x49 ? 2 : -1;
            else if (b == 0x40 || b == 0x42) set = 3;
            d.iso_mid = 0;
            if (set >= 0) {
                d.iso_set = d.iso_out = (u8)set;
                const bool twice = d.iso_flag != 0;
                d.iso_flag = 1;
                if (twice) return { RES_MALFORMED, i, w };
                continue;
            }
            i--;
            d.iso_pend = d.iso_b1; d.iso_flag = 0; d.iso_set = d.iso_out;
            return { RES_MALFORMED, i, w };
        }
        case 1: { // second byte of a JIS X 0208 pair
            d.iso_mid = 0;
            if (b == 0x1B) { d.iso_mid = 2; return { RES_MALFORMED, i, w }; }
            u32 cp = 0;
            if (b >= 0x21 && b <= 0x7E) cp = d.table[(u32)(d.iso_b1 - 0x21) * 94 + (b - 0x21)];
            if (!cp) return { RES_MALFORMED, i, w };
            w += dput_cp(dst + w, cp);
            continue;
        }
        default: break;
        }
        if (b == 0x1B) { d.iso_mid = 2; continue; }   // (iso_out already is the set in force)
        d.iso_flag = 0;
        u32 cp = 0;
        if (d.iso_set == 3) {
            if (b >= 0x21 && b <= 0x7E) { d.iso_b1 = b; d.iso_mid = 1; continue; }
        } else if (d.iso_set == 2) {
            if (b >= 0x21 && b <= 0x5F) cp = 0xFF61u - 0x21u + b;
        } else if (b < 0x80 && b != 0x0E && b != 0x0F) {
            cp = b;
            if (d.iso_set == 1) { if (b == 0x5C) cp = 0xA5; else if (b == 0x7E) cp = 0x203E; }
            if (b == 0) { dst[w++] = 0; continue; }
        }
        if (!cp) return { RES_MALFORMED, i, w };
        w += dput_cp(dst + w, cp);
    }
}

SXD DStep ddec_single(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last = false) {
    if (d.enc == kEncIso2022Jp) return ddec_iso2022jp(d, src, n, dst, cap, last);
    // WHATWG "replacement decoder": one error for the whole input and no character ever — nothing of it is observable
    // in the scan (no output, hence no finding, and no position anybody reads), so the input is just consumed
    if (d.enc == kEncReplacement) return { RES_INPUT_EMPTY, n, 0 };
    u32 i = 0, w = 0;
    for (;;) {
        if (i >= n) return { RES_INPUT_EMPTY, i, w };
        if (cap - w < 3) return { RES_OUTPUT_FULL, i, w };
        const u8 b = src[i++];
        if (b < 0x80) { dst[w++] = b; continue; }
        const u32 c = d.table ? d.table[b - 0x80] : 0xF780u + (b - 0x80u);
        if (c == 0) return { RES_MALFORMED, i, w };
        w += dput_cp(dst + w, c);
    }
}

// ---- double-byte encodings: a token is one byte, or a lead byte and the byte(s) after it ----
// Both decoders consume a token whole; only when the token is malformed AND its last byte is
// ASCII that byte stays unread (it starts the next call, where it is a character of its own).

// Big5 pointer of (lead, trail) -> up to two code points; 0 = no character
SXD u32 big5_lookup(const uint16_t* t, u32 lead, u32 trail, u32* second) {
    *second = 0;
    const bool trail_ok = (trail >= 0x40 && trail <= 0x7E) || (trail >= 0xA1 && trail <= 0xFE);
    if (!trail_ok) return 0;
    const u32 ptr = (lead - 0x81) * 157 + (trail - (trail < 0x7F ? 0x40u : 0x62u));
    if (ptr == 1133 || ptr == 1135) { *second = ptr == 1133 ? 0x0304u : 0x030Cu; return 0x00CA; }
    if (ptr == 1164 || ptr == 1166) { *second = ptr == 1164 ? 0x0304u : 0x030Cu; return 0x00EA; }
    const u32 lo = t[ptr];
    const u32 p2 = (t[kBig5N + (ptr >> 4)] >> (ptr & 15)) & 1u;
    return lo | (p2 << 17);
}

// Shift_JIS (WHATWG "Shift_JIS decoder"): pointer -> index jis0208, or the user-defined range 8836..10715 -> U+E000..
SXD u32 sjis_lookup(const uint16_t* t, u32 lead, u32 trail) {
    if (!((trail >= 0x40 && trail <= 0x7E) || (trail >= 0x80 && trail <= 0xFC))) return 0;
    const u32 ptr = (lead - (lead < 0xA0 ? 0x81u : 0xC1u)) * 188 + (trail - (trail < 0x7F ? 0x40u : 0x41u));
    if (ptr >= 8836 && ptr <= 10715) return 0xE000u - 8836u + ptr;
    return ptr < kSjisN ? t[ptr] : 0u;
}
// EUC-KR (WHATWG "EUC-KR decoder"): lead 81..FE, trail 41..FE
SXD u32 euckr_lookup(const uint16_t* t, u32 lead, u32 trail) {
    if (trail < 0x41 || trail > 0xFE) return 0;
    return t[(lead - 0x81) * 190 + (trail - 0x41)];
}
// gb18030 / GBK (WHATWG "gb18030 decoder"): two-byte cells, and the four-byte pointers through index gb18030 ranges
SXD u32 gb_lookup(const uint16_t* t, u32 lead, u32 trail) {
    if (trail < 0x40 || trail == 0x7F || trail > 0xFE) return 0;
    return t[(lead - 0x81) * 190 + (trail - (trail < 0x7F ? 0x40u : 0x41u))];
}
SXD u32 gb_ranges_cp(const uint16_t* t, u32 pointer) {   // 0 = null
    if ((pointer > 39419 && pointer < 189000) || pointer > 1237575) return 0;
    if (pointer == 7457) return 0xE7C7;
    if (pointer >= 189000) return 0x10000 + (pointer - 189000);
    const uint16_t* ptr = t + kGbN;
    const uint16_t* cp = ptr + kGbRanges;
    u32 lo = 0, hi = kGbRanges;   // the last breakpoint at or below pointer
    while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if (ptr[mid] <= pointer) lo = mid; else hi = mid; }
    return (u32)cp[lo] + (pointer - ptr[lo]);
}
SXD bool gb_digit(u8 b) { return b >= 0x30 && b <= 0x39; }

// The stream a call sees is [re-queued bytes..., src...] (position j < 0: in the queue).  "Prepend to the stream": what this call
// consumed is un-read (j goes back, possibly into the queue again); what an earlier call consumed is queued for the next call.
SXD DStep ddec_gb18030(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    const int q0 = d.rq_n;
    const u8 q[2] = { d.rq[0], d.rq[1] };
    int j = -q0;
    u32 w = 0;
    d.rq_n = 0;
    // leave with result r: positions below 0 that were not consumed stay queued
    auto leave = [&](int r) -> DStep {
        if (j < 0) { d.rq_n = (u8)(-j); for (int t = 0; t < -j; t++) d.rq[t] = q[q0 + j + t]; j = 0; }
        return { r, (u32)j, w };
    };
    // "restore the last k bytes to the stream" (the current byte and the pending ones before it; pre = the pending ones): the
    // current byte — always one of src — is read again by the next call; the pending ones stay consumed and are queued for it,
    // whichever call read them (encoding_rs keeps the digit / the third byte pending instead of un-reading them)
    auto prepend = [&](int k, u8 p1, u8 p2) -> bool {   // true: bytes were queued
        j -= 1;
        if (k < 2) return false;
        const u8 pre[2] = { p1, p2 };
        d.rq_n = (u8)(k - 1);
        for (int t = 0; t < k - 1; t++) d.rq[t] = pre[(k == 3 ? 0 : 1) + t];
        return true;
    };
    for (;;) {
        if (j >= (int)n) {
            if (last && (d.dlead || d.gb2 || d.gb3)) { d.dlead = d.gb2 = d.gb3 = 0; return leave(RES_MALFORMED); }
            return leave(RES_INPUT_EMPTY);
        }
        if (cap - w < 4) return leave(RES_OUTPUT_FULL);
        const u8 b = j < 0 ? q[q0 + j] : src[j];
        j++;
        if (d.gb3) {
            if (!gb_digit(b)) {
                const u8 b2 = d.gb2, b3 = d.gb3;
                d.dlead = d.gb2 = d.gb3 = 0;
                if (prepend(3, b2, b3)) return { RES_MALFORMED, (u32)j, w };
                return leave(RES_MALFORMED);
            }
            const u32 pointer = (u32)(d.dlead - 0x81) * 12600u + (u32)(d.gb2 - 0x30) * 1260u + (u32)(d.gb3 - 0x81) * 10u + (b - 0x30);
            d.dlead = d.gb2 = d.gb3 = 0;
            const u32 cp = gb_ranges_cp(d.table, pointer);
            if (!cp) return leave(RES_MALFORMED);
            w += dput_cp(dst + w, cp);
            continue;
        }
        if (d.gb2) {
            if (b >= 0x81 && b <= 0xFE) { d.gb3 = b; continue; }
            const u8 b2 = d.gb2;
            d.dlead = d.gb2 = 0;
            if (prepend(2, 0, b2)) return { RES_MALFORMED, (u32)j, w };
            return leave(RES_MALFORMED);
        }
        if (d.dlead) {
            if (gb_digit(b)) { d.gb2 = b; continue; }
            const u32 cp = gb_lookup(d.table, d.dlead, b);
            d.dlead = 0;
            if (cp) { w += dput_cp(dst + w, cp); continue; }
            if (b < 0x80) (void)prepend(1, 0, 0);
            return leave(RES_MALFORMED);
        }
        if (b < 0x80) { dst[w++] = b; continue; }
        if (b == 0x80) { w += dput_cp(dst + w, 0x20AC); continue; }
        if (b <= 0xFE) { d.dlead = b; continue; }
        return leave(RES_MALFORMED);   // 0xFF
    }
}

// the two-byte family: is b a lead byte / which character is (lead, trail) / which character is a single byte >= 0x80
SXD bool two_byte_lead(int enc, u8 b) {
    if (enc == kEncShiftJis) return (b >= 0x81 && b <= 0x9F) || (b >= 0xE0 && b <= 0xFC);
    return b >= 0x81 && b <= 0xFE;
}
SXD u32 two_byte_lookup(int enc, const uint16_t* t, u32 lead, u32 trail, u32* second) {
    *second = 0;
    if (enc == kEncShiftJis) return sjis_lookup(t, lead, trail);
    if (enc == kEncEucKr) return euckr_lookup(t, lead, trail);
    if (enc_is_gb(enc)) return gb_lookup(t, lead, trail);
    return big5_lookup(t, lead, trail, second);
}
SXD u32 two_byte_single(int enc, u8 b) {   // b >= 0x80 and not a lead byte: its character, 0 = malformed
    if (enc == kEncShiftJis) return b == 0x80 ? 0x80u : (b >= 0xA1 && b <= 0xDF) ? 0xFF61u - 0xA1u + b : 0u;
    if (enc_is_gb(enc)) return b == 0x80 ? 0x20ACu : 0u;
    return 0;
}

SXD DStep ddec_big5(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {   // Big5, Shift_JIS, EUC-KR (and gb18030: its own function)
    if (enc_is_gb(d.enc)) return ddec_gb18030(d, src, n, dst, cap, last);
    u32 i = 0, w = 0;
    for (;;) {
        if (i >= n) {
            if (last && d.dlead) { d.dlead = 0; return { RES_MALFORMED, i, w }; }
            return { RES_INPUT_EMPTY, i, w };
        }
        // room for one character before the next byte is read: Big5 may yield an astral one or two code points (2 + 2 bytes); Shift_JIS
        // and EUC-KR only yield BMP characters (encoding_rs: check_space_astral / check_space_bmp).  Observable through the
        // 8-byte probe of a slice start (finding_collection.rs:176-207) only.
        if (cap - w < (d.enc == kEncBig5 ? 4u : 3u)) return { RES_OUTPUT_FULL, i, w };
        const u8 b = src[i];
        if (d.dlead == 0) {
            i++;
            if (b < 0x80) { dst[w++] = b; continue; }
            if (two_byte_lead(d.enc, b)) { d.dlead = b; continue; }
            const u32 c1 = two_byte_single(d.enc, b);
            if (c1 == 0) return { RES_MALFORMED, i, w };
            w += dput_cp(dst + w, c1);
            continue;
        }
        u32 second;
        const u32 cp = two_byte_lookup(d.enc, d.table, d.dlead, b, &second);
        d.dlead = 0;
        if (cp) {
            i++;
            w += dput_cp(dst + w, cp);
            if (second) w += dput_cp(dst + w, second);
            continue;
        }
        if (b >= 0x80) i++;  // an ASCII second byte stays unread
        return { RES_MALFORMED, i, w };
    }
}

SXD DStep ddec_eucjp(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    u32 i = 0, w = 0;
    for (;;) {
        if (i >= n) {
            if (last && d.dlead) { d.dlead = 0; d.dflag = 0; return { RES_MALFORMED, i, w }; }
            return { RES_INPUT_EMPTY, i, w };
        }
        if (cap - w < 3) return { RES_OUTPUT_FULL, i, w };   // BMP characters only
        const u8 b = src[i];
        const bool b_hi = b >= 0xA1 && b <= 0xFE;
        if (d.dlead == 0) {
            i++;
            if (b < 0x80) { dst[w++] = b; continue; }
            if (b == 0x8E || b == 0x8F || b_hi) { d.dlead = b; continue; }
            return { RES_MALFORMED, i, w };
        }
        const u32 lead = d.dlead;
        if (lead == 0x8F && !d.dflag && b_hi) { d.dlead = b; d.dflag = 1; i++; continue; }  // three-byte form: 8F xx ..
        u32 cp = 0;
        if (lead == 0x8E && !d.dflag) { if (b >= 0xA1 && b <= 0xDF) cp = 0xFF61u - 0xA1u + b; }
        else if (lead >= 0xA1 && lead <= 0xFE && b_hi) cp = d.table[(d.dflag ? kJisN : 0u) + (lead - 0xA1) * 94 + (b - 0xA1)];
        d.dlead = 0; d.dflag = 0;
        if (cp) { i++; w += dput_cp(dst + w, cp); continue; }
        if (b >= 0x80) i++;
        return { RES_MALFORMED, i, w };
    }
}

template <int ENC>
SXD DStep ddecode(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    if (ENC == 1) return ddec_utf8(d, src, n, dst, cap, last);
    if (ENC == 2) return ddec_utf16<false>(d, src, n, dst, cap, last);
    if (ENC == 3) return ddec_utf16<true>(d, src, n, dst, cap, last);
    if (ENC == 4) return ddec_big5(d, src, n, dst, cap, last);
    if (ENC == 5) return ddec_eucjp(d, src, n, dst, cap, last);
    return ddec_single(d, src, n, dst, cap, last);
}
SXD DStep ddecode_any(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {
    switch (enc_family((u32)d.enc)) {
        case 1: return ddecode<1>(d, src, n, dst, cap, last);
        case 2: return ddecode<2>(d, src, n, dst, cap, last);
        case 3: return ddecode<3>(d, src, n, dst, cap, last);
        case 4: return ddecode<4>(d, src, n, dst, cap, last);
        case 5: return ddecode<5>(d, src, n, dst, cap, last);
        default: return ddecode<0>(d, src, n, dst, cap, last);
    }
}

// nothing pending: no partial UTF-8 sequence, no half UTF-16 unit, no surrogate waiting for its pair, no lead byte
template <int ENC>
SXD bool ddec_idle(const DDecoder& d) {
    if (ENC == 1) return d.needed == 0;
    if (ENC == 2 || ENC == 3) return d.lead_byte < 0 && d.lead_surrogate == 0 && !d.pending_bmp;
    if (ENC == 4 || ENC == 5) return d.dlead == 0 && d.gb2 == 0 && d.rq_n == 0;
    return true;
}
SXD bool ddec_idle_any(const DDecoder& d) {
    return d.needed == 0 && d.lead_byte < 0 && d.lead_surrogate == 0 && !d.pending_bmp && d.dlead == 0 && d.gb2 == 0 && d.rq_n == 0
           && d.iso_mid == 0 && d.iso_pend == 0;
}

// Token grammar of the double-byte encodings, used to find a character boundary without context:
// the byte after a byte outside the lead range always starts a token (WHATWG decoders: after a lead, any
// byte returns to neutral; a byte outside the lead range never becomes pending).
template <int ENC>
SXD bool dbcs_is_lead_range(u8 b, int enc) {
    if (ENC == 4) return two_byte_lead(enc, b);
    return b == 0x8E || b == 0x8F || (b >= 0xA1 && b <= 0xFE);
}
// length of the token that starts at s[0] when the decoder is neutral there (`avail` bytes are readable;
// a token cut short by the end of the input reports the full length it would have)
// ... with one exception, gb18030: an ASCII digit after a lead byte is the second byte of a four-byte token (or the lead is
// an error on its own and the digit is read again): the decoder is neutral after a byte that is neither in the lead range
// nor a digit.  "Could the decoder be in the middle of a token after b?"
template <int ENC>
SXD bool dbcs_may_be_pending_after(u8 b, int enc) {
    if (ENC == 4 && enc_is_gb(enc)) return two_byte_lead(enc, b) || gb_digit(b);
    return dbcs_is_lead_range<ENC>(b, enc);
}
template <int ENC>
SXD u32 dbcs_token_len(const u8* s, u64 avail, int enc) {
    if (!dbcs_is_lead_range<ENC>(s[0], enc)) return 1;
    if (ENC == 4 && enc_is_gb(enc) && avail >= 2 && gb_digit(s[1])) {
        // lead digit lead digit: four bytes (cut short by the end of the input: the length it would have); else the lead alone is
        // the error and the digit is read again
        if (avail >= 3 && !(s[2] >= 0x81 && s[2] <= 0xFE)) return 1;
        if (avail >= 4 && !gb_digit(s[3])) return 1;
        return 4;
    }
    if (ENC == 5 && s[0] == 0x8F && avail >= 2 && s[1] >= 0xA1 && s[1] <= 0xFE) return 3;
    return 2;
}

// How many of the next bytes finish the token that is pending in `d` (0: nothing pending, or the next byte
// will be given back because the token is malformed and that byte is ASCII).
template <int ENC>
SXD u32 dbcs_entry_skip(const DDecoder& d, const u8* s, u64 avail) {
    if (ENC == 4 && enc_is_gb(d.enc)) {
        if ((d.dlead == 0 && d.gb2 == 0 && d.rq_n == 0) || avail == 0) return 0;
        // ask the decoder: the shortest prefix of s after which it is neutral again — driven as the reference drives it (after
        // Malformed the decoder is called again on what was not read: an error of the pending token can queue bytes that are
        // decoded in front of s and take s[0] as their trail byte).  The result is a token boundary of the true grammar, the
        // first one the state on entry allows to name; the token grid of the buffer is counted from there.
        u8 sink[64];
        const u32 lim = avail < 8 ? (u32)avail : 8u;
        for (u32 m = 0; m <= lim; m++) {
            DDecoder c = d;
            u32 used = 0;
            for (int guard = 0; guard < 8; guard++) {
                const DStep r = ddec_gb18030(c, s + used, m - used, sink, sizeof sink, false);
                used += r.read;
                if (r.result != RES_MALFORMED) break;
            }
            if (used == m && c.dlead == 0 && c.gb2 == 0 && c.rq_n == 0) return m;
        }
        return lim;
    }
    if (d.dlead == 0 || avail == 0) return 0;
    const u8 b = s[0];
    if (ENC == 4) {
        u32 second;
        return (two_byte_lookup(d.enc, d.table, d.dlead, b, &second) || b >= 0x80) ? 1u : 0u;
    }
    if (b < 0x80) return 0;  // EUC-JP trails are >= 0xA1: the token is malformed, the ASCII byte is given back
    if (d.dlead == 0x8F && !d.dflag && b >= 0xA1 && b <= 0xFE) return (avail >= 2 && s[1] >= 0x80) ? 2u : 1u;
    return 1;
}

// ------------------------------------------------------------------------------------------
// Filter + SplitStr (reference src/helper.rs:206-433).  PM: anything with the fields
// af_lo, af_hi, ubf, grep_char, q, chars_min_nb, same_block (ReplayParams on the device,
// SplitParams below on the host).
// ------------------------------------------------------------------------------------------
struct SplitParams {
    u64 af_lo, af_hi, ubf;
    int32_t grep_char;
    u32 q, chars_min_nb, same_block;
};
template <class PM> SXD bool pass_af(const PM& p, u8 b) { b &= 127; return ((b < 64 ? p.af_lo >> b : p.af_hi >> (b - 64)) & 1) != 0; }
template <class PM> SXD bool pass_ubf(const PM& p, u8 b) { return ((p.ubf >> (b & 0x3F)) & 1) != 0; }
template <class PM> SXD bool pass_lead(const PM& p, u8 lead) { return (lead & 0x80) ? pass_ubf(p, lead) : pass_af(p, lead); }

struct DSplit {
    const u8 *inp_start, *inp_end, *p;
    bool last_cut, invalid_after;
};
struct DChunk { const u8* s; u32 len; bool completes, maybe_cut, again, min_ok, grep_ok; };

template <class PM>
SXD bool dsplit_next(const PM& m, DSplit& it, DChunk& out) {
    const bool grep_needed = m.grep_char >= 0;
    bool grep_ok = !grep_needed;
    const u8* ok_p = it.p;
    u32 ok_len = 0, ok_n = 0;
    u8 last_mb = 0;
    while (it.p < it.inp_end && ok_n < m.q) {  // exits 1 and 2, :237
        const u8 lead = *it.p;
        u32 cl = 1;
        if ((lead & 0x80) == 0) { if (!grep_ok && m.grep_char == (int)lead) grep_ok = true; }  // :252
        else if ((lead & 0xE0) == 0xC0) cl = 2;
        else if ((lead & 0xF0) == 0xE0) cl = 3;
        else if ((lead & 0xF8) == 0xF0) cl = 4;
        bool ok, advance = true;
        if (cl == 1) ok = pass_af(m, lead);  // :276
        else if (pass_ubf(m, lead)) {       // :279
            ok = !m.same_block || lead == last_mb || last_mb == 0;
            if (!ok) advance = false;  // the same char is scanned again as a string start, :289-291
            last_mb = lead;
        } else { ok = false; last_mb = 0; }
        if (ok) { ok_len += cl; ok_n++; it.p += cl; continue; }
        if (advance) it.p += cl;
        const bool exit3 = it.last_cut && ok_n > 0 && ok_p == it.inp_start;  // :315
        const bool exit4 = ok_n >= m.chars_min_nb && grep_ok;               // :317
        if (exit3 || exit4) break;
        ok_len = 0; ok_n = 0; ok_p = it.p; grep_ok = !grep_needed;  // :327-330
    }
    if (ok_len == 0) return false;  // :343
    const bool touches_left = ok_p == it.inp_start;
    const bool touches_right = ok_p + ok_len >= it.inp_end;
    const bool maybe_cut = ok_n >= m.q || (touches_right && !it.invalid_after);
    const bool completes = touches_left && it.last_cut;
    const bool again = !completes && touches_right && !it.invalid_after && (ok_n < m.q || !grep_ok);
    const bool min_rule = ok_n >= m.chars_min_nb;
    if (!completes && !again && (!grep_ok || !min_rule)) return false;  // :410-415
    if (ok_n >= m.q) it.inp_start = it.p;                               // :418-420
    it.last_cut = maybe_cut;                                            // :421
    out.s = ok_p; out.len = ok_len; out.completes = completes; out.maybe_cut = maybe_cut; out.again = again;
    out.min_ok = min_rule; out.grep_ok = grep_ok;
    return true;
}

}  // namespace sx
