// sx_decoder.cpp — the host-side `Decoder` and `SplitStr` classes of sx_host.hpp (the reference's
// operator interface: encoding_rs `Decoder` as used at src/finding_collection.rs:138-143,180-194,
// `SplitStr` src/helper.rs:58-433) as thin wrappers around the product's ONE implementation,
// sx_codec_core.hpp — the same source the device replay kernels are compiled from.
#include <string.h>

#include "sx_host.hpp"

namespace sx {

#include "sx_tables.inc"

static_assert(SX_BIG5_N == kBig5N && SX_BIG5_P2_WORDS == kBig5P2Words && SX_JIS_N == kJisN, "table layout");
static_assert(SX_ENC_BIG5 == kEncBig5 && SX_ENC_EUC_JP == kEncEucJp && SX_ENC_SHIFT_JIS == kEncShiftJis && SX_ENC_EUC_KR == kEncEucKr && SX_ENC_GB18030 == kEncGb18030 && SX_ENC_GBK == kEncGbk, "encoding ids");
static_assert(SX_ENC_REPLACEMENT == kEncReplacement && SX_ENC_ISO_2022_JP == kEncIso2022Jp, "encoding ids");
static_assert(SX_SJIS_N == kSjisN && SX_EUCKR_N == kEucKrN, "table layout");

const uint16_t* single_byte_table(int enc) {
    if (enc >= SX_ENC_KOI8_R && enc < SX_ENC_KOI8_R + SX_N_SB_TABLES) return sx_sb_tables[enc - SX_ENC_KOI8_R];
    return nullptr;
}

const uint16_t* decoder_table(int enc, size_t* n_words) {
    size_t n = 0;
    const uint16_t* t = nullptr;
    if (enc == SX_ENC_BIG5) { t = sx_big5; n = sizeof sx_big5 / sizeof sx_big5[0]; }
    else if (enc == SX_ENC_EUC_JP) { t = sx_eucjp; n = sizeof sx_eucjp / sizeof sx_eucjp[0]; }
    else if (enc == SX_ENC_SHIFT_JIS) { t = sx_sjis; n = sizeof sx_sjis / sizeof sx_sjis[0]; }
    else if (enc == SX_ENC_EUC_KR) { t = sx_euckr; n = sizeof sx_euckr / sizeof sx_euckr[0]; }
    else if (enc == SX_ENC_GB18030 || enc == SX_ENC_GBK) { t = sx_gb18030; n = sizeof sx_gb18030 / sizeof sx_gb18030[0]; }
    else if (enc == SX_ENC_ISO_2022_JP) { t = sx_eucjp; n = kJisN; }   // index jis0208: the first half of the EUC-JP blob
    else if ((t = single_byte_table(enc)) != nullptr) n = 128;
    if (n_words) *n_words = n;
    return t;
}

bool encoding_is_known(int enc) {
    return enc == SX_ENC_X_USER_DEFINED || enc == SX_ENC_UTF8 || enc == SX_ENC_UTF16LE || enc == SX_ENC_UTF16BE
           || enc == SX_ENC_REPLACEMENT || decoder_table(enc, nullptr) != nullptr;
}

const char* encoding_name(int enc) {
    switch (enc) {
    case SX_ENC_X_USER_DEFINED: return "x-user-defined";
    case SX_ENC_UTF8: return "UTF-8";
    case SX_ENC_UTF16LE: return "UTF-16LE";
    case SX_ENC_UTF16BE: return "UTF-16BE";
    case SX_ENC_BIG5: return "Big5";
    case SX_ENC_EUC_JP: return "EUC-JP";
    case SX_ENC_SHIFT_JIS: return "Shift_JIS";
    case SX_ENC_EUC_KR: return "EUC-KR";
    case SX_ENC_GB18030: return "gb18030";
    case SX_ENC_GBK: return "GBK";
    case SX_ENC_REPLACEMENT: return "replacement";
    case SX_ENC_ISO_2022_JP: return "ISO-2022-JP";
    default:
        if (enc >= SX_ENC_KOI8_R && enc < SX_ENC_KOI8_R + SX_N_SB_TABLES) return sx_sb_names[enc - SX_ENC_KOI8_R];
        return "?";
    }
}

void Decoder::reset(int encoding) { ddec_reset(d_, encoding, decoder_table(encoding, nullptr)); }

DecodeStep Decoder::decode_to_str_without_replacement(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                                                      bool last) {
    const DStep r = ddecode_any(d_, src, (u32)n, dst, (u32)(cap > 0xFFFFFFFFu ? 0xFFFFFFFFu : cap), last);
    return { r.result == RES_INPUT_EMPTY ? DecoderResult::InputEmpty
                                         : r.result == RES_MALFORMED ? DecoderResult::Malformed : DecoderResult::OutputFull,
             r.read, r.written };
}

bool Decoder::idle() const { return ddec_idle_any(d_); }

uint32_t Decoder::entry_skip(const uint8_t* s, uint64_t avail) const {
    switch (enc_family((u32)d_.enc)) {
    case 4: return dbcs_entry_skip<4>(d_, s, avail);
    case 5: return dbcs_entry_skip<5>(d_, s, avail);
    default: return 0;
    }
}

uint32_t Decoder::entry_skip_scan(const uint8_t* s, uint64_t avail) const {
    if (!enc_is_gb(d_.enc)) return entry_skip(s, avail);
    if (avail == 0) return 0;
    // the lead byte that is pending in the kernel's grammar: the third byte of a four-byte token, a lead waiting in the queue,
    // or a plain pending lead (lead + digit: the digit was a token of its own)
    uint8_t x = 0;
    if (d_.gb3) x = d_.gb3;
    else if (d_.rq_n) x = two_byte_lead(d_.enc, d_.rq[d_.rq_n - 1]) ? d_.rq[d_.rq_n - 1] : 0;
    else if (!d_.gb2) x = d_.dlead;
    if (!x) return 0;
    return (gb_lookup(d_.table, x, s[0]) || s[0] >= 0x80) ? 1u : 0u;
}

// SplitStr::next — src/helper.rs:206-433
bool SplitStr::next(SplitStrResult* out) {
    DChunk ch;
    if (!dsplit_next(pm_, it_, ch)) return false;
    out->s = ch.s; out->len = ch.len;
    out->s_completes_previous_s = ch.completes; out->s_is_maybe_cut = ch.maybe_cut;
    out->s_is_to_be_filtered_again = ch.again; out->s_satisfies_min_char_rule = ch.min_ok;
    out->s_satisfies_grep_char_rule = ch.grep_ok;
    return true;
}

}  // namespace sx
