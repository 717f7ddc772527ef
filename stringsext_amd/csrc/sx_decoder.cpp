// sx_decoder.cpp — the decoders the scan needs, with the call contract of encoding_rs
// 0.8.34 `Decoder::decode_to_str_without_replacement` (the crate is a Cargo dependency of
// the reference, Cargo.toml:19; call sites src/finding_collection.rs:138-143,180-194).
// Algorithms: WHATWG Encoding Standard "utf-8 decoder", "utf-16 decoder" (with the
// crate's streaming treatment of unpaired surrogates), "x-user-defined decoder",
// "single-byte decoder".  What matters to the caller is (result, read, written):
// a malformed sequence ends the call; `read` says where the next call starts.
#include <string.h>

#include "sx_host.hpp"

namespace sx {

#include "sx_tables.inc"

const uint16_t* single_byte_table(int enc) {
    if (enc >= SX_ENC_KOI8_R && enc < SX_ENC_KOI8_R + SX_N_SB_TABLES) return sx_sb_tables[enc - SX_ENC_KOI8_R];
    return nullptr;
}

const char* encoding_name(int enc) {
    switch (enc) {
    case SX_ENC_X_USER_DEFINED: return "x-user-defined";
    case SX_ENC_UTF8: return "UTF-8";
    case SX_ENC_UTF16LE: return "UTF-16LE";
    case SX_ENC_UTF16BE: return "UTF-16BE";
    default:
        if (enc >= SX_ENC_KOI8_R && enc < SX_ENC_KOI8_R + SX_N_SB_TABLES) return sx_sb_names[enc - SX_ENC_KOI8_R];
        return "?";
    }
}

void Decoder::reset(int encoding) {
    enc_ = encoding;
    cp_ = 0; seen_ = needed_ = 0; lower_ = 0x80; upper_ = 0xBF;
    lead_byte_ = -1; lead_surrogate_ = 0; pending_bmp_ = false;
    table_ = single_byte_table(encoding);
}

static inline size_t encode_utf8(uint8_t* d, uint32_t c) {
    if (c < 0x80) { d[0] = (uint8_t)c; return 1; }
    if (c < 0x800) { d[0] = (uint8_t)(0xC0 | (c >> 6)); d[1] = (uint8_t)(0x80 | (c & 0x3F)); return 2; }
    if (c < 0x10000) {
        d[0] = (uint8_t)(0xE0 | (c >> 12)); d[1] = (uint8_t)(0x80 | ((c >> 6) & 0x3F));
        d[2] = (uint8_t)(0x80 | (c & 0x3F));
        return 3;
    }
    d[0] = (uint8_t)(0xF0 | (c >> 18)); d[1] = (uint8_t)(0x80 | ((c >> 12) & 0x3F));
    d[2] = (uint8_t)(0x80 | ((c >> 6) & 0x3F)); d[3] = (uint8_t)(0x80 | (c & 0x3F));
    return 4;
}

DecodeStep Decoder::decode_to_str_without_replacement(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                                                      bool last) {
    switch (enc_) {
    case SX_ENC_UTF8: return utf8(src, n, dst, cap, last);
    case SX_ENC_UTF16LE:
    case SX_ENC_UTF16BE: return utf16(src, n, dst, cap, last);
    default: return single(src, n, dst, cap);
    }
}

// UTF-8.  A byte outside the expected continuation range ends the call as Malformed and
// is NOT consumed (the next call starts at it); a bad lead byte is consumed.
DecodeStep Decoder::utf8(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, bool last) {
    size_t i = 0, w = 0;
    auto clear = [this]() { cp_ = 0; needed_ = seen_ = 0; lower_ = 0x80; upper_ = 0xBF; };
    while (true) {
        if (needed_ == 0) {  // bulk ASCII
            while (i < n && src[i] < 0x80 && cap - w >= 4) dst[w++] = src[i++];
        }
        if (i >= n) {
            if (last && needed_ != 0) { clear(); return { DecoderResult::Malformed, i, w }; }
            return { DecoderResult::InputEmpty, i, w };
        }
        if (cap - w < 4) return { DecoderResult::OutputFull, i, w };
        const uint8_t b = src[i++];
        if (needed_ == 0) {
            if (b < 0x80) { dst[w++] = b; continue; }
            if (b >= 0xC2 && b <= 0xDF) { needed_ = 1; cp_ = b & 0x1F; continue; }
            if (b >= 0xE0 && b <= 0xEF) {
                if (b == 0xE0) lower_ = 0xA0;
                if (b == 0xED) upper_ = 0x9F;
                needed_ = 2; cp_ = b & 0x0F; continue;
            }
            if (b >= 0xF0 && b <= 0xF4) {
                if (b == 0xF0) lower_ = 0x90;
                if (b == 0xF4) upper_ = 0x8F;
                needed_ = 3; cp_ = b & 0x07; continue;
            }
            return { DecoderResult::Malformed, i, w };
        }
        if (b < lower_ || b > upper_) { clear(); return { DecoderResult::Malformed, i - 1, w }; }
        lower_ = 0x80; upper_ = 0xBF;
        cp_ = (cp_ << 6) | (b & 0x3F);
        if (++seen_ != needed_) continue;
        w += encode_utf8(dst + w, cp_);
        cp_ = 0; needed_ = seen_ = 0;
    }
}

// UTF-16.  Whole units are converted in bulk while nothing is pending; an unpaired
// surrogate met there ends the call right after that unit.  A high surrogate that is the
// last whole unit of the input becomes pending; if the following call then sees a BMP
// unit (or another high surrogate) the call ends Malformed with that unit consumed too —
// a BMP unit is remembered and written first thing by the next call.
DecodeStep Decoder::utf16(const uint8_t* src, size_t n, uint8_t* dst, size_t cap, bool last) {
    const bool be = enc_ == SX_ENC_UTF16BE;
    auto unit_at = [&](size_t k) -> uint16_t {
        return be ? (uint16_t)((src[k] << 8) | src[k + 1]) : (uint16_t)((src[k + 1] << 8) | src[k]);
    };
    size_t i = 0, w = 0;
    if (pending_bmp_) {
        if (cap - w < 3) return { DecoderResult::OutputFull, 0, 0 };
        w += encode_utf8(dst + w, lead_surrogate_);
        pending_bmp_ = false; lead_surrogate_ = 0;
    }
    while (true) {
        if (lead_byte_ < 0 && lead_surrogate_ == 0) {
            while (n - i >= 2 && cap - w >= 4) {
                const uint16_t u = unit_at(i);
                if ((u & 0xF800) != 0xD800) { w += encode_utf8(dst + w, u); i += 2; continue; }
                if ((u & 0xFC00) == 0xDC00) { i += 2; return { DecoderResult::Malformed, i, w }; }
                if (n - i < 4) break;  // high surrogate, last whole unit: goes pending below
                const uint16_t v = unit_at(i + 2);
                if ((v & 0xFC00) != 0xDC00) { i += 2; return { DecoderResult::Malformed, i, w }; }
                w += encode_utf8(dst + w, 0x10000u + (((uint32_t)u & 0x3FF) << 10) + (v & 0x3FF));
                i += 4;
            }
        }
        if (i >= n) {
            if (last && (lead_surrogate_ != 0 || lead_byte_ >= 0)) {
                lead_surrogate_ = 0; lead_byte_ = -1;
                return { DecoderResult::Malformed, i, w };
            }
            return { DecoderResult::InputEmpty, i, w };
        }
        if (cap - w < 4) return { DecoderResult::OutputFull, i, w };
        const uint8_t b = src[i++];
        if (lead_byte_ < 0) { lead_byte_ = b; continue; }
        const uint16_t u = be ? (uint16_t)((lead_byte_ << 8) | b) : (uint16_t)((b << 8) | lead_byte_);
        lead_byte_ = -1;
        if ((u & 0xFC00) == 0xD800) {
            if (lead_surrogate_ != 0) { lead_surrogate_ = u; return { DecoderResult::Malformed, i, w }; }
            lead_surrogate_ = u;
            continue;
        }
        if ((u & 0xFC00) == 0xDC00) {
            if (lead_surrogate_ == 0) return { DecoderResult::Malformed, i, w };
            w += encode_utf8(dst + w, 0x10000u + (((uint32_t)lead_surrogate_ & 0x3FF) << 10) + (u & 0x3FF));
            lead_surrogate_ = 0;
            continue;
        }
        if (lead_surrogate_ != 0) {
            lead_surrogate_ = u; pending_bmp_ = true;
            return { DecoderResult::Malformed, i, w };
        }
        w += encode_utf8(dst + w, u);
    }
}

// x-user-defined (0x80..0xFF -> U+F780..U+F7FF) and table-driven single-byte encodings.
DecodeStep Decoder::single(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    size_t i = 0, w = 0;
    while (true) {
        if (i >= n) return { DecoderResult::InputEmpty, i, w };
        if (cap - w < 3) return { DecoderResult::OutputFull, i, w };
        const uint8_t b = src[i++];
        if (b < 0x80) { dst[w++] = b; continue; }
        const uint32_t c = table_ ? table_[b - 0x80] : 0xF780u + (b - 0x80u);
        if (c == 0) return { DecoderResult::Malformed, i, w };
        w += encode_utf8(dst + w, c);
    }
}

}  // namespace sx
