// sx_mission.cpp — turns a Mission (src/mission.rs:382-421) into what the device needs:
// which classifier kernel fits its encoding + Utf8Filter, and that kernel's constants or
// lookup tables.  The filter acts on the UTF-8 lead byte of the decoded character
// (src/mission.rs:333-348, src/helper.rs:276-279), so for every encoding the accept set
// is first expressed in the encoding's own byte/unit terms.
#include <string.h>

#include "sx_host.hpp"
#include "sx_wave_core.hpp"

// SX_WAVE_SAME=0: Missions with -r as before round 5 (UTF-8 / UTF-16: the wave path for buffers with one kind of lead byte; single byte: not at all)
static bool wave_same_on() { const char* e = getenv("SX_WAVE_SAME"); return !(e && atoi(e) == 0); }
// (-g AND -r: the state has five bits for the lead code, sx_wave_core.hpp WvState)
static bool wave_same_fits(int grep_char, uint64_t ubf) { return grep_char < 0 || __builtin_popcountll(ubf & 0x001FFFFFFFFFFFFCull) <= 31; }

namespace sx {

const char* Mission::encoding_name() const { return sx::encoding_name(c.encoding); }

static uint8_t utf8_lead_of(uint32_t cp) {
    if (cp < 0x80) return (uint8_t)cp;
    if (cp < 0x800) return (uint8_t)(0xC0 | (cp >> 6));
    if (cp < 0x10000) return (uint8_t)(0xE0 | (cp >> 12));
    return (uint8_t)(0xF0 | (cp >> 18));
}

// [lo,hi] if the set bits of `bits` (positions first..last) are one contiguous run
static bool one_range(const bool* bits, int first, int last, int* lo, int* hi, bool* empty) {
    int l = -1, h = -1, n = 0;
    for (int i = first; i <= last; i++)
        if (bits[i]) { if (l < 0) l = i; h = i; n++; }
    *empty = n == 0;
    if (n == 0) return true;
    *lo = l; *hi = h;
    return n == h - l + 1;
}

int Mission::from_c(const sx_mission& in, bool force_generic, Mission* m, std::string* err) {
    m->c = in;
    m->filter.af_lo = in.af_lo; m->filter.af_hi = in.af_hi; m->filter.ubf = in.ubf;
    m->filter.grep_char = in.grep_char;
    m->q = in.output_line_char_nb_max;
    if (m->q < 6) { *err = "output_line_char_nb_max must be >= 6 (src/options.rs:33)"; return SX_E_INVALID; }
    if (in.grep_char > 127) { *err = "grep_char must be an ASCII code < 128 (src/mission.rs:547-555)"; return SX_E_INVALID; }
    m->window = 2 * m->q;
    m->long_run = (uint32_t)std::min<size_t>(in.chars_min_nb, m->q);
    if (m->long_run == 0) m->long_run = 1;
    const int enc = in.encoding;
    // -r (helper.rs:279-296) breaks a string between two multi-byte characters whose UTF-8 lead bytes pass ubf and differ.  If at most one
    // lead byte passes — `-u` with one bit; x-user-defined ("ascii"), whose characters >= 0x80 all begin with EF — it never does: such a
    // Mission is its own twin without -r, as far as the wave path is concerned (set below for the single-byte tables)
    uint32_t same_block = in.require_same_unicode_block ? 1u : 0u;
    if (same_block && __builtin_popcountll(in.ubf & 0x001FFFFFFFFFFFFCull) <= 1) same_block = 0;   // (leads C2..F4 <-> bits 2..52)
    if (!encoding_is_known(enc)) { *err = "unsupported encoding id " + std::to_string(enc); return SX_E_INVALID; }

    ScanParams& p = m->proto;
    memset(&p, 0, sizeof p);
    p.min_chars = m->long_run;
    const uint32_t min_bytes_per_char = m->is_utf16() ? 2 : 1;
    p.cand_bytes = std::min<uint32_t>(m->long_run * min_bytes_per_char, 14);
    p.big_endian = enc == SX_ENC_UTF16BE;
    p.a_lo = 1; p.a_hi = 0; p.u_lo = 0x81; p.u_hi = 0x80;  // empty ranges

    bool af[128], lead2[64] = { false };
    for (int b = 0; b < 128; b++) af[b] = m->filter.pass_af_filter((uint8_t)b);
    int alo = 0, ahi = 0;
    bool aempty = false;
    const bool af_is_range = one_range(af, 0, 127, &alo, &ahi, &aempty);
    if (!aempty) { p.a_lo = (uint32_t)alo; p.a_hi = (uint32_t)ahi; }
    // leads C2..DF <-> ubf bits 2..31
    for (int i = 2; i < 32; i++) lead2[i] = ((in.ubf >> i) & 1) != 0;
    int ulo = 0, uhi = 0;
    bool uempty = false;
    const bool ubf2_is_range = one_range(lead2, 2, 31, &ulo, &uhi, &uempty);
    const bool no_long_leads = ((in.ubf >> 32) & 0x1FFFFFull) == 0;  // E0..F4 <-> bits 32..52

    if (enc == SX_ENC_UTF8) {
        // the wave-cooperative stage B (sx_wave_dev.hip): kind of every byte + "a character with this lead byte passes the filter"
        // (-r: the kernels check per buffer that it cannot matter there — text without, or with one kind of, multi-byte characters)
        m->wave_ok = wv_mission_ok(in.grep_char, 0u, in.chars_min_nb, (uint32_t)m->q);
        m->wave_same = m->wave_ok && same_block != 0 && wave_same_on() && wave_same_fits(in.grep_char, in.ubf);   // (round 5: -r in the wave kernels, with or without -g)
        m->wave_lead_check = m->wave_ok && same_block != 0 && !m->wave_same;
        m->wave_family = 1;
        m->wave_lut.assign(256, 0);
        for (int b = 0; b < 256; b++) {
            uint8_t kind = WVU_BAD;
            if (b < 0x80) kind = WVU_ASCII; else if (b < 0xC0) kind = WVU_CONT; else if (b >= 0xC2 && b <= 0xDF) kind = WVU_LEAD2;
            else if (b >= 0xE0 && b <= 0xEF) kind = WVU_LEAD3; else if (b >= 0xF0 && b <= 0xF4) kind = WVU_LEAD4;
            const bool acc = kind == WVU_ASCII ? af[b] : (kind >= WVU_LEAD2 && m->filter.pass_ubf_filter((uint8_t)b));
            m->wave_lut[(size_t)b] = (uint8_t)(kind | (acc ? WVU_ACC : 0));
        }
        {   // the wave kernels' classes as ranges (sx_wave_core.hpp wv_classify16_utf8_swar): the accepted FIRST bytes — ASCII and lead bytes — at most six
            std::vector<std::pair<int, int>> ranges;
            for (int b = 0; b < 256; b++)
                if (m->wave_lut[(size_t)b] & WVU_ACC) {
                    if (!ranges.empty() && ranges.back().second == b - 1 && b != 0x80) ranges.back().second = b; else ranges.emplace_back(b, b);
                }
            if (ranges.size() <= 6) {
                WvSwar& R = m->wave_swar;
                R.cls = 1; R.n = (uint32_t)ranges.size();
                for (size_t k = 0; k < 6; k++) {
                    uint32_t lo = 1, hi = 0, high = 0;
                    if (k < ranges.size()) { lo = (uint32_t)ranges[k].first & 0x7F; hi = (uint32_t)ranges[k].second & 0x7F; high = ranges[k].first >= 0x80; }
                    R.c1[k] = (0x80u - lo) * 0x01010101u; R.c2[k] = (0x7Fu - hi) * 0x01010101u; R.hi[k] = high ? 0u : 0xFFFFFFFFu;
                }
            }
        }
        // three-byte leads E0..EF <-> ubf bits 32..47: the alias filters of mission.rs:167-218 (Cjk E4..E9, Kana E3, Hangul EB..ED, Asian E2..ED ...)
        // are one range of them; E0 (second byte A0..BF only) and the four-byte leads stay with the table kernel
        bool lead3[16];
        for (int i = 0; i < 16; i++) lead3[i] = ((in.ubf >> (32 + i)) & 1) != 0;
        int l3lo = 0, l3hi = 0;
        bool l3empty = false;
        const bool ubf3_is_range = one_range(lead3, 0, 15, &l3lo, &l3hi, &l3empty);
        const bool no_lead4 = ((in.ubf >> 48) & 0x1Full) == 0;   // F0..F4 <-> bits 48..52
        if (!force_generic && af_is_range && ubf2_is_range && no_long_leads) {
            m->kind = kClsUtf8Range2;
            if (!uempty) { p.u_lo = 0xC0u + (uint32_t)ulo; p.u_hi = 0xC0u + (uint32_t)uhi; }
        } else if (!force_generic && af_is_range && no_long_leads && [&] {   // two runs of two-byte leads: -u Latin = C2..C8 + CC..CD (mission.rs:63-67)
                       int runs = 0;
                       for (int i = 2; i < 32; i++) if (lead2[i] && !lead2[i - 1]) runs++;
                       return runs == 2;
                   }()) {
            m->kind = kClsUtf8Range2x2;   // (sx_classify_ranges.hpp)
            int i = 2;
            while (!lead2[i]) i++;
            p.u_lo = 0xC0u + (uint32_t)i; while (i < 32 && lead2[i]) i++; p.u_hi = 0xC0u + (uint32_t)i - 1;
            while (!lead2[i]) i++;
            p.l3_lo = 0xC0u + (uint32_t)i; while (i < 32 && lead2[i]) i++; p.l3_hi = 0xC0u + (uint32_t)i - 1;
        } else if (!force_generic && af_is_range && ubf2_is_range && ubf3_is_range && !l3empty && l3lo >= 1 && no_lead4) {
            m->kind = kClsUtf8Range3;   // (sx_classify_ranges.hpp)
            if (!uempty) { p.u_lo = 0xC0u + (uint32_t)ulo; p.u_hi = 0xC0u + (uint32_t)uhi; }
            p.l3_lo = 0xE0u + (uint32_t)l3lo; p.l3_hi = 0xE0u + (uint32_t)l3hi;
        } else {
            m->kind = kClsUtf8Lut;
            for (int b = 0; b < 256; b++) {
                uint8_t cls = 0;
                if (b < 0x80) cls = af[b] ? 0x38 : 0;
                else if (b < 0x90) cls = 1;
                else if (b < 0xA0) cls = 2;
                else if (b < 0xC0) cls = 4;
                else if (b >= 0xC2 && b <= 0xF4 && m->filter.pass_ubf_filter((uint8_t)b)) {
                    uint8_t allowed = 7, len1 = 1;
                    if (b >= 0xE0) { len1 = 2; if (b == 0xE0) allowed = 4; if (b == 0xED) allowed = 3; }
                    if (b >= 0xF0) { len1 = 3; allowed = b == 0xF0 ? 6 : (b == 0xF4 ? 1 : 7); }
                    cls = (uint8_t)((allowed << 3) | (len1 << 6));
                }
                p.lut[b] = cls;
            }
        }
    } else if (m->is_utf16()) {
        {   // the wave-cooperative stage B (sx_wave_core.hpp wv_utf16_unit): per high byte — the low byte's quadrants that pass, hb == 0, high / low surrogate —, per low byte of U+0000..U+00FF
            // (windows of >= 10 bytes: the slice-start probe, finding_collection.rs:176-207, lets a fresh decoder run over the slice until 8 bytes are
            // written — five units at most; in a shorter window the real decoder has already met the window's last unit, which it treats differently)
            m->wave_ok = wv_mission_ok(in.grep_char, 0u, in.chars_min_nb, (uint32_t)m->q) && m->window % 2 == 0 && m->window >= 10;
            m->wave_same = m->wave_ok && same_block != 0 && wave_same_on() && wave_same_fits(in.grep_char, in.ubf);
            m->wave_lead_check = m->wave_ok && same_block != 0 && !m->wave_same;   // (SX_WAVE_SAME=0: per buffer, as for UTF-8)
            m->wave_family = 2;
            m->wave_lut.assign(512, 0);
            for (int lb = 0; lb < 256; lb++) m->wave_lut[256 + (size_t)lb] = m->filter.pass_lead(utf8_lead_of((uint32_t)lb)) ? 1 : 0;
            for (int hb = 0; hb < 256; hb++) {
                uint8_t v = 0;
                if (hb == 0) v = WVW_ZERO;
                else if (hb >= 0xD8 && hb <= 0xDB) {
                    v = WVW_HIGH;
                    for (int qd = 0; qd < 4; qd++) {   // (the lead byte of an astral character's UTF-8 form depends on the high surrogate only)
                        const uint32_t u = (uint32_t)(hb << 8) | (uint32_t)(qd << 6);
                        if (m->filter.pass_ubf_filter(utf8_lead_of(0x10000u + ((u & 0x3FF) << 10)))) v |= (uint8_t)(1 << qd);
                    }
                } else if (hb >= 0xDC && hb <= 0xDF) v = WVW_LOW;
                else
                    for (int qd = 0; qd < 4; qd++)
                        if (m->filter.pass_ubf_filter(utf8_lead_of((uint32_t)(hb << 8) | (uint32_t)(qd << 6)))) v |= (uint8_t)(1 << qd);
                m->wave_lut[(size_t)hb] = v;
            }
        }
        const bool no_bmp3 = ((in.ubf >> 32) & 0xFFFFull) == 0;   // U+0800..U+FFFF <-> bits 32..47
        const bool no_astral = ((in.ubf >> 48) & 0x1Full) == 0;   // bits 48..52
        // the accepted units as ranges (no surrogate inside): what af and the lead bytes of ubf say unit by unit — C2..DF <-> U+0080..U+07FF,
        // E0..EF <-> U+0800..U+FFFF; with an astral plane accepted a surrogate pair can be a character, which the range kernels do not
        // know.  Up to two ranges below U+8000, one that straddles it, one above (sx_classify_ranges.hpp Utf16RangesT)
        std::vector<std::pair<uint32_t, uint32_t>> uranges;
        if (!force_generic)
            for (uint32_t u = 0; u < 0x10000u && uranges.size() <= 4; u++) {
                const bool acc = u < 0x80 ? m->filter.pass_lead((uint8_t)u)
                                          : (u < 0xD800u || u > 0xDFFFu) && m->filter.pass_ubf_filter(utf8_lead_of(u));
                if (!acc) continue;
                if (!uranges.empty() && uranges.back().second + 1 == u) uranges.back().second = u;
                else uranges.emplace_back(u, u);
            }
        uint32_t n_lo = 0, n_st = 0, n_hi = 0;
        for (const auto& r : uranges) { if (r.second < 0x8000u) n_lo++; else if (r.first >= 0x8000u) n_hi++; else n_st++; }
        // an astral plane passes: the high surrogates of such planes (the pair's UTF-8 lead byte F0..F4 follows from the high surrogate alone) must be one range
        uint32_t hs_lo = 0, hs_hi = 0, hs_runs = 0;
        for (uint32_t u = 0xD800u; u <= 0xDBFFu && !no_astral; u++) {
            if (!m->filter.pass_ubf_filter(utf8_lead_of(0x10000u + ((u & 0x3FFu) << 10)))) continue;
            if (hs_runs && hs_hi + 1 == u) hs_hi = u; else { hs_runs++; hs_lo = hs_hi = u; }
        }
        const bool uranges_fit = !force_generic && (no_astral || hs_runs == 1) && (!uranges.empty() || hs_runs == 1) && n_st <= 1 && n_hi <= 1 &&
                                 (n_lo <= 2 || (n_lo == 3 && n_st + n_hi == 0 && no_astral));   // (a third range below U+8000 takes the high surrogates' slot)
        if (!force_generic && af_is_range && ubf2_is_range && no_bmp3 && no_astral) {
            m->kind = kClsUtf16Range;
            if (!uempty) { p.u_lo = (uint32_t)ulo << 6; p.u_hi = ((uint32_t)uhi << 6) | 0x3F; }
            else { p.u_lo = 1; p.u_hi = 0; }
        } else if (uranges_fit) {
            m->kind = kClsUtf16Ranges;   // (sx_classify_ranges.hpp)
            p.n_ranges = n_lo | (n_st << 4) | (n_hi << 8) | (hs_runs ? 1u << 12 : 0u);
            for (int k = 0; k < 6; k++) { p.rng_c1[k] = 0u; p.rng_c2[k] = 0x7FFFu * 0x00010001u; p.rng_hi[k] = 0u; }   // (empty slots)
            uint32_t il = 0, ih = 3;
            for (const auto& r : uranges) {
                const uint32_t slot = r.second < 0x8000u ? (il < 2 ? il++ : (il++, 5u)) : r.first >= 0x8000u ? ih++ : 2u;
                p.rng_c1[slot] = (0x8000u - (r.first & 0x7FFFu)) * 0x00010001u;
                p.rng_c2[slot] = (0x8000u + (r.second & 0x7FFFu)) * 0x00010001u;
            }
            if (hs_runs) { p.rng_c1[5] = (0x8000u - (hs_lo & 0x7FFFu)) * 0x00010001u; p.rng_c2[5] = (0x8000u + (hs_hi & 0x7FFFu)) * 0x00010001u; }
        } else {
            m->kind = kClsUtf16Lut;
            uint8_t* H = p.lut;
            uint8_t* L = p.lut + 256;
            for (int lb = 0; lb < 256; lb++) L[lb] = m->filter.pass_lead(utf8_lead_of((uint32_t)lb)) ? 1 : 0;
            for (int hb = 0; hb < 256; hb++) {
                uint8_t v = 0;
                if (hb == 0) v = 0x10;
                else if (hb >= 0xD8 && hb <= 0xDB) {
                    v = 0x20;
                    for (int q = 0; q < 4; q++) {
                        const uint32_t u = (uint32_t)(hb << 8) | (uint32_t)(q << 6);
                        const uint32_t cp = 0x10000u + ((u & 0x3FF) << 10);  // lead depends on the high surrogate only
                        if (m->filter.pass_ubf_filter(utf8_lead_of(cp))) v |= (uint8_t)(1 << q);
                    }
                } else if (hb >= 0xDC && hb <= 0xDF) v = 0x40;
                else {
                    for (int q = 0; q < 4; q++) {
                        const uint32_t u = (uint32_t)(hb << 8) | (uint32_t)(q << 6);
                        if (m->filter.pass_ubf_filter(utf8_lead_of(u))) v |= (uint8_t)(1 << q);
                    }
                }
                H[hb] = v;
            }
        }
    } else if (enc == SX_ENC_REPLACEMENT || enc == SX_ENC_ISO_2022_JP) {
        // (ISO-2022-JP never reaches a scan kernel: Mission::host_sequential)
        // the replacement decoder emits nothing but one error: no byte is ever part of a character
        m->kind = kClsSingleByteRange;
        p.a_lo = 1; p.a_hi = 0; p.high_all = 0;
    } else if (m->is_dbcs()) {
        // Token classifier: accepted ASCII as a range (or a 256-entry LUT), and per byte pair a 2-bit code
        // (sx_device.hpp ScanParams::pair_lut) from the decoder's own lookup + the filter on the UTF-8 lead byte.
        const bool two_byte = enc_family((uint32_t)enc) == 4;
        m->kind = two_byte ? kClsBig5 : kClsEucJp;
        for (int b = 0; b < 256; b++) p.lut[b] = 0;
        for (int b = 0; b < 128; b++) p.lut[b] = af[b] ? 0x80 : 0;
        // lead byte ranges (two-byte family) and the single bytes >= 0x80 that are characters (Shift_JIS)
        const uint32_t lr[2][2] = { { enc == SX_ENC_SHIFT_JIS ? 0x81u : 0x81u, enc == SX_ENC_SHIFT_JIS ? 0x9Fu : 0xFEu },
                                    { enc == SX_ENC_SHIFT_JIS ? 0xE0u : 1u, enc == SX_ENC_SHIFT_JIS ? 0xFCu : 0u } };
        for (int r = 0; r < 2; r++) {
            const bool empty = lr[r][0] > lr[r][1];
            p.lr_c1[r] = (0x80u - (empty ? 1u : (lr[r][0] & 0x7F))) * 0x01010101u;
            p.lr_c2[r] = (0x7Fu - (empty ? 0u : (lr[r][1] & 0x7F))) * 0x01010101u;
        }
        p.high1 = 0;
        if (two_byte)
            for (int b = 0x80; b < 256; b++) {
                if (two_byte_lead(enc, (uint8_t)b)) continue;
                const uint32_t cp = two_byte_single(enc, (uint8_t)b);
                if (cp) { p.high1 = 1; if (m->filter.pass_lead(utf8_lead_of(cp))) p.lut[b] = 0x80; }
            }
        if (two_byte && !enc_is_gb(enc)) {
            // the wave-cooperative stage B (sx_wave_dev.hip): a class per byte on its own, 4 bits per byte pair.  Big5's four pointers that
            // yield two code points (U+00CA / U+00EA + U+0304 / U+030C) stand as ONE rejected char there: only for Missions that
            // reject both (UTF-8 lead bytes C3 and CC)
            m->wave_ok = wv_mission_ok(in.grep_char, same_block, in.chars_min_nb, (uint32_t)m->q)
                         && (enc != SX_ENC_BIG5 || (!m->filter.pass_ubf_filter(0xC3) && !m->filter.pass_ubf_filter(0xCC)));
            m->wave_family = 4;
            m->wave_lut.assign(256, 0);
            const uint16_t* t4 = decoder_table(enc, nullptr);
            for (int b = 0; b < 256; b++) {
                uint8_t c = 0;
                if (b < 0x80) c = (uint8_t)(WVC_VALID | (af[b] ? WVC_ACC : 0));
                else if (two_byte_lead(enc, (uint8_t)b)) c = WVC_LEAD;
                else if (const uint32_t cp = two_byte_single(enc, (uint8_t)b))
                    c = (uint8_t)(WVC_VALID | (m->filter.pass_lead(utf8_lead_of(cp)) ? WVC_ACC : 0) | (cp >= 0x800 ? WVC_O3 : WVC_O2));
                m->wave_lut[(size_t)b] = c;
            }
            m->wave_pairs.assign(8192, 0u);
            for (uint32_t b0 = 0x81; b0 <= 0xFE; b0++) {
                if (!two_byte_lead(enc, (uint8_t)b0)) continue;
                for (uint32_t b1 = 0; b1 < 256; b1++) {
                    uint32_t second = 0;
                    const uint32_t cp = two_byte_lookup(enc, t4, b0, b1, &second);
                    if (!cp) continue;
                    const uint32_t len = second ? 3u : cp < 0x800 ? 0u : cp < 0x10000 ? 1u : 2u;
                    const uint32_t code = 1u | ((!second && m->filter.pass_lead(utf8_lead_of(cp))) ? 2u : 0u) | (len << 2);
                    const uint32_t idx = b0 | (b1 << 8);
                    m->wave_pairs[idx >> 3] |= code << ((idx & 7u) * 4);
                }
            }
        }
        if (two_byte && !enc_is_gb(enc) && m->wave_ok && !p.high1) {
            // the same classes without the byte table (sx_wave_core.hpp wv_classify16_dbcs_swar): bytes below 0x80 are characters on their
            // own, the accepted ones at most six ranges; every accepted pair's UTF-8 form has the same length (Cjk, Asian, Kana, Hangul:
            // three bytes) -> 2 bits per pair do, and O2 / O3 / O4 follow from the masks of the character ends
            std::vector<std::pair<int, int>> ranges;
            for (int b = 0; b < 128; b++)
                if (af[b]) { if (!ranges.empty() && ranges.back().second == b - 1) ranges.back().second = b; else ranges.emplace_back(b, b); }
            uint32_t len_seen = 0;
            bool one_len = true;
            m->wave_pairs2.assign(4096, 0u);
            for (uint32_t idx = 0; idx < 65536; idx++) {
                const uint32_t code = (m->wave_pairs[idx >> 3] >> ((idx & 7u) * 4)) & 15u;
                if (!(code & 1u)) continue;
                const bool acc = (code & 2u) != 0 && (code >> 2) != 3u;
                if (acc) { const uint32_t len = 2u + (code >> 2); if (!len_seen) len_seen = len; else if (len_seen != len) one_len = false; }
                m->wave_pairs2[idx >> 4] |= (1u | (acc ? 2u : 0u)) << ((idx & 15u) * 2);
            }
            if (one_len && ranges.size() <= 6) {
                WvSwar& R = m->wave_swar;
                R.cls = 1; R.n = (uint32_t)ranges.size(); R.hi_len = len_seen ? len_seen : 3u;
                for (size_t k = 0; k < 6; k++) {
                    uint32_t lo = 1, hi = 0;
                    if (k < ranges.size()) { lo = (uint32_t)ranges[k].first; hi = (uint32_t)ranges[k].second; }
                    R.c1[k] = (0x80u - lo) * 0x01010101u; R.c2[k] = (0x7Fu - hi) * 0x01010101u; R.hi[k] = 0xFFFFFFFFu;
                }
                for (int r = 0; r < 2; r++) { R.lr_c1[r] = p.lr_c1[r]; R.lr_c2[r] = p.lr_c2[r]; }
            } else m->wave_pairs2.clear();
        }
        if (!two_byte) {
            // EUC-JP on the wave path (sx_wave_core.hpp wv_classify16_eucjp_swar): Missions whose accepted multi-byte characters all have
            // UTF-8 forms of one length (Asian, Cjk, Kana: three bytes; half-width katakana, 8E xx, are U+FF61.. = three bytes too) and whose
            // accepted ASCII bytes are at most six ranges.  2 bits per cell of index jis0208 and of index jis0212 (bit 0 mapped, bit 1 accepted).
            const uint16_t* tj = decoder_table(enc, nullptr);
            std::vector<std::pair<int, int>> ranges;
            for (int b = 0; b < 128; b++)
                if (af[b]) { if (!ranges.empty() && ranges.back().second == b - 1) ranges.back().second = b; else ranges.emplace_back(b, b); }
            uint32_t len_seen = 0;
            bool one_len = true;
            const bool kana = m->filter.pass_lead(utf8_lead_of(0xFF61u));
            if (kana) len_seen = 3;
            m->wave_pairs2.assign(4096, 0u);
            for (uint32_t cell = 0; cell < 2 * kJisN; cell++) {
                const uint32_t cp = tj[cell];
                if (!cp) continue;
                const bool acc = m->filter.pass_lead(utf8_lead_of(cp));
                if (acc) { const uint32_t len = cp < 0x800 ? 2u : 3u; if (!len_seen) len_seen = len; else if (len_seen != len) one_len = false; }
                m->wave_pairs2[cell >> 4] |= (1u | (acc ? 2u : 0u)) << ((cell & 15u) * 2);
            }
            m->wave_family = 5;
            m->wave_lut.assign(256, 0);
            for (int b = 0; b < 256; b++)
                m->wave_lut[(size_t)b] = b < 0x80 ? (uint8_t)(WVC_VALID | (af[b] ? WVC_ACC : 0)) : ((b >= 0xA1 && b <= 0xFE) || b == 0x8E || b == 0x8F) ? (uint8_t)WVC_LEAD : (uint8_t)0;
            m->wave_ok = wv_mission_ok(in.grep_char, same_block, in.chars_min_nb, (uint32_t)m->q) && one_len && ranges.size() <= 6;
            if (m->wave_ok) {
                WvSwar& R = m->wave_swar;
                R.cls = 1; R.n = (uint32_t)ranges.size(); R.hi_len = len_seen ? len_seen : 3u; R.kana = kana ? 1u : 0u;
                for (size_t k = 0; k < 6; k++) {
                    uint32_t lo = 1, hi = 0;
                    if (k < ranges.size()) { lo = (uint32_t)ranges[k].first; hi = (uint32_t)ranges[k].second; }
                    R.c1[k] = (0x80u - lo) * 0x01010101u; R.c2[k] = (0x7Fu - hi) * 0x01010101u; R.hi[k] = 0xFFFFFFFFu;
                }
            } else m->wave_pairs2.clear();
        }
        p.gb4 = (enc_is_gb(enc) && m->c.ubf != 0) ? 1u : 0u;   // some character beyond ASCII is accepted: four-byte tokens may be
        p.af_is_range = (!force_generic && af_is_range && !p.high1) ? 1u : 0u;
        const uint16_t* t = decoder_table(enc, nullptr);
        const size_t n_tables = two_byte ? 1 : 2;
        m->pair_lut.assign(n_tables * 4096, 0u);
        auto put = [&](size_t table, uint32_t b0, uint32_t b1, uint32_t code) {
            const uint32_t idx = b0 | (b1 << 8);
            m->pair_lut[table * 4096 + (idx >> 4)] |= code << ((idx & 15u) * 2);
        };
        for (uint32_t b0 = 0x81; b0 <= 0xFE; b0++)
            for (uint32_t b1 = 0; b1 < 256; b1++) {
                if (two_byte) {
                    if (!two_byte_lead(enc, (uint8_t)b0)) continue;
                    uint32_t second = 0;
                    const uint32_t cp = two_byte_lookup(enc, t, b0, b1, &second);
                    if (!cp) continue;
                    // two characters (Big5 pointers 1133..1166): a break between them cannot be expressed per byte, so the
                    // pair counts as accepted when either passes — a superset of the true runs (stage B is exact)
                    const bool ok = m->filter.pass_lead(utf8_lead_of(cp)) || (second && m->filter.pass_lead(utf8_lead_of(second)));
                    put(0, b0, b1, ok ? (second ? 2u : 3u) : 1u);
                } else {
                    if (b1 < 0xA1 || b1 > 0xFE) continue;
                    if (b0 == 0x8E) { if (b1 <= 0xDF) put(0, b0, b1, m->filter.pass_lead(utf8_lead_of(0xFF61u - 0xA1u + b1)) ? 3u : 1u); continue; }
                    if (b0 < 0xA1) continue;
                    for (size_t tb = 0; tb < 2; tb++) {
                        const uint32_t cp = t[tb * kJisN + (b0 - 0xA1) * 94 + (b1 - 0xA1)];
                        if (cp) put(tb, b0, b1, m->filter.pass_lead(utf8_lead_of(cp)) ? 3u : 1u);
                    }
                }
            }
    } else {  // x-user-defined and single-byte tables
        bool acc[256];
        const uint16_t* tab = single_byte_table(enc);
        // the wave-cooperative stage B (sx_wave_dev.hip) reads a class per byte: valid / accepted / bytes of its UTF-8 form
        if (same_block) {   // -r: the lead bytes of the characters this table can yield and the filter lets pass
            uint64_t leads = 0;
            for (int b = 0x80; b < 256; b++) {
                const uint32_t cp = tab ? tab[b - 0x80] : 0xF780u + (uint32_t)(b - 0x80);
                const uint8_t lead = utf8_lead_of(cp);
                if (cp >= 0x80 && m->filter.pass_ubf_filter(lead)) leads |= 1ull << (lead & 0x3F);
            }
            if (__builtin_popcountll(leads) <= 1) same_block = 0;
        }
        m->wave_same = same_block != 0 && wave_same_on() && wave_same_fits(in.grep_char, in.ubf) && wv_mission_ok(in.grep_char, 0u, in.chars_min_nb, (uint32_t)m->q);
        m->wave_ok = wv_mission_ok(in.grep_char, m->wave_same ? 0u : same_block, in.chars_min_nb, (uint32_t)m->q);
        m->wave_lut.assign(256, 0);
        for (int b = 0; b < 256; b++) {
            uint32_t cp = (uint32_t)b;
            if (b < 0x80) acc[b] = af[b];
            else {
                cp = tab ? tab[b - 0x80] : 0xF780u + (uint32_t)(b - 0x80);
                acc[b] = cp != 0 && m->filter.pass_lead(utf8_lead_of(cp));
            }
            if (b < 0x80 || cp != 0)
                m->wave_lut[(size_t)b] = (uint8_t)(WVC_VALID | (acc[b] ? WVC_ACC : 0) | (cp >= 0x800 ? WVC_O3 : cp >= 0x80 ? WVC_O2 : 0));
        }
        int hl = 0, hh = 0;
        bool hempty = false;
        one_range(acc, 128, 255, &hl, &hh, &hempty);
        const bool high_all = !hempty && hl == 128 && hh == 255 && [&] { for (int b = 128; b < 256; b++) if (!acc[b]) return false; return true; }();
        // maximal ranges of accepted bytes, cut at 0x80
        std::vector<std::pair<int, int>> ranges;
        for (int b = 0; b < 256; b++)
            if (acc[b]) {
                if (!ranges.empty() && ranges.back().second == b - 1 && b != 0x80) ranges.back().second = b;
                else ranges.emplace_back(b, b);
            }
        // the wave kernels' classes as ranges (sx_wave_core.hpp wv_classify16_single_swar): every byte a character, the accepted ones at
        // most six ranges, the accepted ones >= 0x80 with UTF-8 forms of one length
        {
            bool all_valid = true;
            int hi_len = 0;
            bool one_len = true;
            for (int b = 0; b < 256; b++) {
                const uint8_t c = m->wave_lut[(size_t)b];
                all_valid = all_valid && (c & WVC_VALID);
                if (b >= 0x80 && (c & WVC_ACC)) {
                    const int len = (c & WVC_O3) ? 3 : (c & WVC_O2) ? 2 : 1;
                    if (hi_len == 0) hi_len = len; else if (hi_len != len) one_len = false;
                }
            }
            if (all_valid && one_len && hi_len != 1 && ranges.size() <= 6) {
                WvSwar& R = m->wave_swar;
                R.cls = 1; R.n = (uint32_t)ranges.size(); R.hi_len = (uint32_t)hi_len;
                for (size_t k = 0; k < 6; k++) {
                    uint32_t lo = 1, hi = 0, high = 0;   // empty
                    if (k < ranges.size()) { lo = (uint32_t)ranges[k].first & 0x7F; hi = (uint32_t)ranges[k].second & 0x7F; high = ranges[k].first >= 0x80; }
                    R.c1[k] = (0x80u - lo) * 0x01010101u; R.c2[k] = (0x7Fu - hi) * 0x01010101u; R.hi[k] = high ? 0u : 0xFFFFFFFFu;
                }
            }
        }
        if (!force_generic && af_is_range && (hempty || high_all)) {
            m->kind = kClsSingleByteRange;
            p.high_all = high_all ? 1 : 0;
        } else if (!force_generic && ranges.size() <= 6) {
            m->kind = kClsSingleByteRanges;
            p.n_ranges = (uint32_t)ranges.size();
            for (size_t k = 0; k < 6; k++) {
                uint32_t lo = 1, hi = 0, high = 0;   // empty
                if (k < ranges.size()) { lo = (uint32_t)ranges[k].first & 0x7F; hi = (uint32_t)ranges[k].second & 0x7F; high = ranges[k].first >= 0x80; }
                p.rng_c1[k] = (0x80u - lo) * 0x01010101u;
                p.rng_c2[k] = (0x7Fu - hi) * 0x01010101u;
                p.rng_hi[k] = high ? 0u : 0xFFFFFFFFu;
            }
        } else {
            m->kind = kClsSingleByteLut;
            for (int b = 0; b < 256; b++) p.lut[b] = acc[b] ? 0x80 : 0;
        }
    }
    return SX_OK;
}

}  // namespace sx
