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
constexpr int kEncBig5 = 64, kEncEucJp = 65, kEncShiftJis = 66, kEncEucKr = 67;  // == SX_ENC_* (include/stringsext_amd.h)
constexpr int enc_family(u32 encoding) {
    return encoding == 1 ? 1 : encoding == 2 ? 2 : encoding == 3 ? 3
         : (encoding == (u32)kEncBig5 || encoding == (u32)kEncShiftJis || encoding == (u32)kEncEucKr) ? 4 : encoding == (u32)kEncEucJp ? 5 : 0;
}
// Layout of the double-byte tables (csrc/gen_tables.py): one uint16_t blob per encoding.
//   Big5: [kBig5N low halves][kBig5P2Words plane-2 bitmap];  EUC-JP: [kJisN jis0208][kJisN jis0212];
//   Shift_JIS: [kSjisN: index jis0208 up to the IBM extension rows];  EUC-KR: [kEucKrN].
constexpr u32 kBig5N = 126 * 157, kBig5P2Words = (kBig5N + 15) / 16, kJisN = 94 * 94, kSjisN = 11280, kEucKrN = 126 * 190;

struct DDecoder {
    int enc;
    u32 cp; u8 seen, needed, lower, upper;               // UTF-8
    int lead_byte; u32 lead_surrogate; bool pending_bmp;  // UTF-16
    u8 dlead, dflag;                                      // Big5 / EUC-JP: pending lead byte; EUC-JP: the lead is the 2nd byte of 8F xx
    const uint16_t* table;  // single byte: 128 entries (nullptr = x-user-defined); Big5 / EUC-JP: the blob
};

SXD void ddec_reset(DDecoder& d, int enc, const uint16_t* table) {
    d.enc = enc; d.cp = 0; d.seen = d.needed = 0; d.lower = 0x80; d.upper = 0xBF;
    d.lead_byte = -1; d.lead_surrogate = 0; d.pending_bmp = false; d.dlead = 0; d.dflag = 0; d.table = table;
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
SXD DStep ddec_single(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap) {
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
// the two-byte family: is b a lead byte / which character is (lead, trail) / which character is a single byte >= 0x80
SXD bool two_byte_lead(int enc, u8 b) {
    if (enc == kEncShiftJis) return (b >= 0x81 && b <= 0x9F) || (b >= 0xE0 && b <= 0xFC);
    return b >= 0x81 && b <= 0xFE;
}
SXD u32 two_byte_lookup(int enc, const uint16_t* t, u32 lead, u32 trail, u32* second) {
    *second = 0;
    if (enc == kEncShiftJis) return sjis_lookup(t, lead, trail);
    if (enc == kEncEucKr) return euckr_lookup(t, lead, trail);
    return big5_lookup(t, lead, trail, second);
}
SXD u32 two_byte_single(int enc, u8 b) {   // b >= 0x80 and not a lead byte: its character, 0 = malformed
    if (enc == kEncShiftJis) return b == 0x80 ? 0x80u : (b >= 0xA1 && b <= 0xDF) ? 0xFF61u - 0xA1u + b : 0u;
    return 0;
}

SXD DStep ddec_big5(DDecoder& d, const u8* src, u32 n, u8* dst, u32 cap, bool last) {   // Big5, Shift_JIS, EUC-KR
    u32 i = 0, w = 0;
    for (;;) {
        if (i >= n) {
            if (last && d.dlead) { d.dlead = 0; return { RES_MALFORMED, i, w }; }
            return { RES_INPUT_EMPTY, i, w };
        }
        if (cap - w < 4) return { RES_OUTPUT_FULL, i, w };   // an astral character, or Big5's two code points (2 + 2 bytes): as the other two-byte decoders
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
        if (cap - w < 4) return { RES_OUTPUT_FULL, i, w };
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
    return ddec_single(d, src, n, dst, cap);
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
    if (ENC == 4 || ENC == 5) return d.dlead == 0;
    return true;
}
SXD bool ddec_idle_any(const DDecoder& d) {
    return d.needed == 0 && d.lead_byte < 0 && d.lead_surrogate == 0 && !d.pending_bmp && d.dlead == 0;
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
template <int ENC>
SXD u32 dbcs_token_len(const u8* s, u64 avail, int enc) {
    if (!dbcs_is_lead_range<ENC>(s[0], enc)) return 1;
    if (ENC == 5 && s[0] == 0x8F && avail >= 2 && s[1] >= 0xA1 && s[1] <= 0xFE) return 3;
    return 2;
}

// How many of the next bytes finish the token that is pending in `d` (0: nothing pending, or the next byte
// will be given back because the token is malformed and that byte is ASCII).
template <int ENC>
SXD u32 dbcs_entry_skip(const DDecoder& d, const u8* s, u64 avail) {
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
