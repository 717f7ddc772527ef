// sx_front.cpp — the Mission front end: from the reference's command-line option *strings* to
// sx_mission[], with the reference's rules, defaults, quirks and error texts.
//
// Follows (re-stated, not copied):
//   src/options.rs:12-33        defaults (ENCODING_DEFAULT "UTF-8", CHARS_MIN_DEFAULT 4, line length 64, minimum 6)
//   src/mission.rs:32-53        default filters for "ascii" and for every other encoding
//   src/mission.rs:72-161       UBF_* constants;  :225-253 AF_* constants
//   src/mission.rs:167-218      unicode-block-filter aliases, :255-274 ascii-filter aliases (order matters)
//   src/mission.rs:448-462      parse_integer!   (empty -> None, "0x.." hex, else decimal, typed overflow)
//   src/mission.rs:474-504      parse_filter_parameter!  ("0x.." hex first, empty -> None, else the FIRST alias the
//                               trimmed text is a prefix of — "All" therefore means "All-Asian")
//   src/mission.rs:514-703      Missions::new;  :713-749 parse_enc_opt (split_terminator(','), <= 5 items)
// Encoding::for_label comes from encoding_rs (not vendored): the WHATWG label table, restated here for the
// encodings this library decodes; a valid label of any other encoding is reported as unsupported.
#include <ctype.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/stringsext_amd.h"

namespace {

typedef unsigned __int128 u128;

constexpr uint64_t UBF_ALL = 0xffffffffffffffffull, UBF_NONE = 0, UBF_INVALID = 0xffe0000000000003ull;
constexpr uint64_t UBF_LATIN = 0x1fc, UBF_ACCENTS = 0x3000, UBF_GREEK = 0xC000, UBF_CYRILLIC = 0x1f0000, UBF_ARMENIAN = 0x200000,
                   UBF_HEBREW = 0xc00000, UBF_ARABIC = 0x2f000000, UBF_SYRIAC = 0x10000000, UBF_AFRICAN = 0xffe00000,
                   UBF_COMMON = 0xfffffffc, UBF_KANA = 0x800000000ull, UBF_CJK = 0x3f000000000ull, UBF_HANGUL = 0x380000000000ull,
                   UBF_ASIAN = 0x3ffc00000000ull, UBF_PUA = 0x10400000000000ull, UBF_UNCOMMON = 0xf000000000000ull;
const u128 AF_ALL = ((u128)0xffffffffffffffffull << 64) | 0xfffffffffffffffeull;
const u128 AF_CTRL = ((u128)0x8000000000000000ull << 64) | 0x00000000ffffffffull;
const u128 AF_WHITESPACE = ((u128)0 << 64) | 0x0000000100001e00ull;

struct Alias64 { const char* name; uint64_t v; };
struct Alias128 { const char* name; u128 v; };
// names are 12 bytes, space padded: the prefix test below runs over the padding too
const Alias64 kUbfAliases[] = {
    { "African     ", UBF_AFRICAN }, { "All-Asian   ", UBF_ALL & ~UBF_INVALID & ~UBF_ASIAN }, { "All         ", UBF_ALL & ~UBF_INVALID },
    { "Arabic      ", UBF_ARABIC | UBF_SYRIAC }, { "Armenian    ", UBF_ARMENIAN }, { "Asian       ", UBF_ASIAN },
    { "Cjk         ", UBF_CJK }, { "Common      ", UBF_COMMON }, { "Cyrillic    ", UBF_CYRILLIC },
    { "Default     ", UBF_ALL & ~UBF_INVALID }, { "Greek       ", UBF_GREEK }, { "Hangul      ", UBF_HANGUL },
    { "Hebrew      ", UBF_HEBREW }, { "Kana        ", UBF_KANA }, { "Latin       ", UBF_LATIN | UBF_ACCENTS },
    { "None        ", ~UBF_ALL }, { "Private     ", UBF_PUA }, { "Uncommon    ", UBF_UNCOMMON | UBF_PUA },
};
std::vector<Alias128> af_aliases() {
    return { { "All         ", AF_ALL }, { "All-Ctrl    ", AF_ALL & ~AF_CTRL }, { "All-Ctrl+Wsp", (AF_ALL & ~AF_CTRL) | AF_WHITESPACE },
             { "Default     ", AF_ALL & ~AF_CTRL }, { "None        ", 0 }, { "Wsp         ", AF_WHITESPACE } };
}

struct Err {
    std::string msg;
    bool set = false;
    bool fail(const std::string& m) { if (!set) { msg = m; set = true; } return false; }
};

std::string trim(const std::string& s) {  // str::trim: Unicode White_Space; the options are ASCII in practice
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) a++;
    while (b > a && isspace((unsigned char)s[b - 1])) b--;
    return s.substr(a, b - a);
}

// {integer}::from_str_radix: an optional '+', then at least one digit of the radix, overflow is an error
bool parse_radix(const std::string& t, int radix, u128 max, u128* out) {
    size_t i = 0;
    if (i < t.size() && t[i] == '+') i++;
    if (i >= t.size()) return false;
    u128 v = 0;
    for (; i < t.size(); i++) {
        int d;
        const char c = t[i];
        if (c >= '0' && c <= '9') d = c - '0';
        else if (c >= 'a' && c <= 'z') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'Z') d = c - 'A' + 10;
        else return false;
        if (d >= radix) return false;
        if (v > (max - (u128)d) / (u128)radix) return false;
        v = v * (u128)radix + (u128)d;
    }
    *out = v;
    return true;
}

struct OptStr { bool some = false; std::string s; };
OptStr some(const char* p) { OptStr o; if (p) { o.some = true; o.s = p; } return o; }

// parse_integer!  -> (ok, has, value)
bool parse_integer(const OptStr& s, u128 max, bool* has, u128* v, Err* e) {
    *has = false;
    if (!s.some || s.s.empty()) return true;
    const std::string t = trim(s.s);
    if (t.size() >= 2 && t.compare(0, 2, "0x") == 0) {
        if (!parse_radix(t.substr(2), 16, max, v)) return e->fail("failed to parse hexadecimal number: `" + s.s + "`");
    } else if (!parse_radix(t, 10, max, v)) return e->fail("failed to parse number: " + s.s);
    *has = true;
    return true;
}

template <class A>
bool parse_filter(const OptStr& s, u128 max, const A* list, size_t n, bool* has, u128* v, Err* e) {
    *has = false;
    if (!s.some) return true;
    const std::string t = trim(s.s);
    if (t.size() >= 2 && t.compare(0, 2, "0x") == 0) {
        if (!parse_radix(t.substr(2), 16, max, v)) return e->fail("failed to parse hexadecimal number: `" + s.s + "`");
        *has = true;
        return true;
    }
    if (s.s.empty()) return true;
    for (size_t i = 0; i < n; i++)
        if (t.size() <= 12 && memcmp(t.data(), list[i].name, t.size()) == 0) { *v = (u128)list[i].v; *has = true; return true; }
    return e->fail("filter name `" + t + "` is not valid, try `--list-encodings`");
}

// str::split_terminator(','): a trailing empty piece is dropped; "" yields nothing
std::vector<std::string> split_terminator(const std::string& s) {
    std::vector<std::string> v;
    size_t a = 0;
    for (;;) {
        const size_t c = s.find(',', a);
        if (c == std::string::npos) { v.push_back(s.substr(a)); break; }
        v.push_back(s.substr(a, c - a));
        a = c + 1;
    }
    if (!v.empty() && v.back().empty()) v.pop_back();
    return v;
}

struct EncOpt {
    bool has_name = false; std::string name;
    bool has_min = false, has_af = false, has_ubf = false, has_grep = false;
    u128 min = 0, af = 0, ubf = 0, grep = 0;
};

bool parse_enc_opt(const std::string& enc_opt, EncOpt* o, Err* e) {
    const std::vector<std::string> it = split_terminator(enc_opt);
    auto item = [&](size_t i) { OptStr s; if (i < it.size()) { s.some = true; s.s = it[i]; } return s; };
    if (!it.empty() && !it[0].empty()) { o->has_name = true; o->name = trim(it[0]); }
    const auto af = af_aliases();
    if (!parse_integer(item(1), 0xff, &o->has_min, &o->min, e)) return false;
    if (!parse_filter(item(2), ~(u128)0, af.data(), af.size(), &o->has_af, &o->af, e)) return false;
    if (!parse_filter(item(3), (u128)0xffffffffffffffffull, kUbfAliases, sizeof kUbfAliases / sizeof kUbfAliases[0], &o->has_ubf, &o->ubf, e))
        return false;
    if (!parse_integer(item(4), 0xff, &o->has_grep, &o->grep, e)) return false;
    if (it.size() > 5) return e->fail("Too many items in `" + enc_opt + "`.");
    return true;
}

// Encoding::for_label (WHATWG Encoding Standard, "names and labels"): ASCII case-insensitive, surrounding
// ASCII whitespace ignored.  >= 0: SX_ENC_*;  -1: not a label;  -2: a label of an encoding not built in.
struct Label { const char* label; int enc; };
const Label kLabels[] = {
    { "unicode-1-1-utf-8", SX_ENC_UTF8 }, { "unicode11utf8", SX_ENC_UTF8 }, { "unicode20utf8", SX_ENC_UTF8 }, { "utf-8", SX_ENC_UTF8 },
    { "utf8", SX_ENC_UTF8 }, { "x-unicode20utf8", SX_ENC_UTF8 },
    { "unicodefffe", SX_ENC_UTF16BE }, { "utf-16be", SX_ENC_UTF16BE },
    { "csunicode", SX_ENC_UTF16LE }, { "iso-10646-ucs-2", SX_ENC_UTF16LE }, { "ucs-2", SX_ENC_UTF16LE }, { "unicode", SX_ENC_UTF16LE },
    { "unicodefeff", SX_ENC_UTF16LE }, { "utf-16", SX_ENC_UTF16LE }, { "utf-16le", SX_ENC_UTF16LE },
    { "x-user-defined", SX_ENC_X_USER_DEFINED },
    { "cskoi8r", SX_ENC_KOI8_R }, { "koi", SX_ENC_KOI8_R }, { "koi8", SX_ENC_KOI8_R }, { "koi8-r", SX_ENC_KOI8_R }, { "koi8_r", SX_ENC_KOI8_R },
    { "866", SX_ENC_IBM866 }, { "cp866", SX_ENC_IBM866 }, { "csibm866", SX_ENC_IBM866 }, { "ibm866", SX_ENC_IBM866 },
    { "csisolatin2", SX_ENC_ISO_8859_2 }, { "iso-8859-2", SX_ENC_ISO_8859_2 }, { "iso-ir-101", SX_ENC_ISO_8859_2 },
    { "iso8859-2", SX_ENC_ISO_8859_2 }, { "iso88592", SX_ENC_ISO_8859_2 }, { "iso_8859-2", SX_ENC_ISO_8859_2 },
    { "iso_8859-2:1987", SX_ENC_ISO_8859_2 }, { "l2", SX_ENC_ISO_8859_2 }, { "latin2", SX_ENC_ISO_8859_2 },
    { "csisolatincyrillic", SX_ENC_ISO_8859_5 }, { "cyrillic", SX_ENC_ISO_8859_5 }, { "iso-8859-5", SX_ENC_ISO_8859_5 },
    { "iso-ir-144", SX_ENC_ISO_8859_5 }, { "iso8859-5", SX_ENC_ISO_8859_5 }, { "iso88595", SX_ENC_ISO_8859_5 },
    { "iso_8859-5", SX_ENC_ISO_8859_5 }, { "iso_8859-5:1988", SX_ENC_ISO_8859_5 },
    { "csisolatin9", SX_ENC_ISO_8859_15 }, { "iso-8859-15", SX_ENC_ISO_8859_15 }, { "iso8859-15", SX_ENC_ISO_8859_15 },
    { "iso885915", SX_ENC_ISO_8859_15 }, { "iso_8859-15", SX_ENC_ISO_8859_15 }, { "l9", SX_ENC_ISO_8859_15 },
    { "cp1251", SX_ENC_WINDOWS_1251 }, { "windows-1251", SX_ENC_WINDOWS_1251 }, { "x-cp1251", SX_ENC_WINDOWS_1251 },
    { "ansi_x3.4-1968", SX_ENC_WINDOWS_1252 }, { "ascii", SX_ENC_WINDOWS_1252 }, { "cp1252", SX_ENC_WINDOWS_1252 },
    { "cp819", SX_ENC_WINDOWS_1252 }, { "csisolatin1", SX_ENC_WINDOWS_1252 }, { "ibm819", SX_ENC_WINDOWS_1252 },
    { "iso-8859-1", SX_ENC_WINDOWS_1252 }, { "iso-ir-100", SX_ENC_WINDOWS_1252 }, { "iso8859-1", SX_ENC_WINDOWS_1252 },
    { "iso88591", SX_ENC_WINDOWS_1252 }, { "iso_8859-1", SX_ENC_WINDOWS_1252 }, { "iso_8859-1:1987", SX_ENC_WINDOWS_1252 },
    { "l1", SX_ENC_WINDOWS_1252 }, { "latin1", SX_ENC_WINDOWS_1252 }, { "us-ascii", SX_ENC_WINDOWS_1252 },
    { "windows-1252", SX_ENC_WINDOWS_1252 }, { "x-cp1252", SX_ENC_WINDOWS_1252 },
    { "csisolatin3", SX_ENC_ISO_8859_3 }, { "iso-8859-3", SX_ENC_ISO_8859_3 }, { "iso-ir-109", SX_ENC_ISO_8859_3 },
    { "iso8859-3", SX_ENC_ISO_8859_3 }, { "iso88593", SX_ENC_ISO_8859_3 }, { "iso_8859-3", SX_ENC_ISO_8859_3 },
    { "iso_8859-3:1988", SX_ENC_ISO_8859_3 }, { "l3", SX_ENC_ISO_8859_3 }, { "latin3", SX_ENC_ISO_8859_3 },
    { "csisolatin4", SX_ENC_ISO_8859_4 }, { "iso-8859-4", SX_ENC_ISO_8859_4 }, { "iso-ir-110", SX_ENC_ISO_8859_4 },
    { "iso8859-4", SX_ENC_ISO_8859_4 }, { "iso88594", SX_ENC_ISO_8859_4 }, { "iso_8859-4", SX_ENC_ISO_8859_4 },
    { "iso_8859-4:1988", SX_ENC_ISO_8859_4 }, { "l4", SX_ENC_ISO_8859_4 }, { "latin4", SX_ENC_ISO_8859_4 },
    { "arabic", SX_ENC_ISO_8859_6 }, { "asmo-708", SX_ENC_ISO_8859_6 }, { "csiso88596e", SX_ENC_ISO_8859_6 },
    { "csiso88596i", SX_ENC_ISO_8859_6 }, { "csisolatinarabic", SX_ENC_ISO_8859_6 }, { "ecma-114", SX_ENC_ISO_8859_6 },
    { "iso-8859-6", SX_ENC_ISO_8859_6 }, { "iso-8859-6-e", SX_ENC_ISO_8859_6 }, { "iso-8859-6-i", SX_ENC_ISO_8859_6 },
    { "iso-ir-127", SX_ENC_ISO_8859_6 }, { "iso8859-6", SX_ENC_ISO_8859_6 }, { "iso88596", SX_ENC_ISO_8859_6 },
    { "iso_8859-6", SX_ENC_ISO_8859_6 }, { "iso_8859-6:1987", SX_ENC_ISO_8859_6 },
    { "csisolatingreek", SX_ENC_ISO_8859_7 }, { "ecma-118", SX_ENC_ISO_8859_7 }, { "elot_928", SX_ENC_ISO_8859_7 },
    { "greek", SX_ENC_ISO_8859_7 }, { "greek8", SX_ENC_ISO_8859_7 }, { "iso-8859-7", SX_ENC_ISO_8859_7 },
    { "iso-ir-126", SX_ENC_ISO_8859_7 }, { "iso8859-7", SX_ENC_ISO_8859_7 }, { "iso88597", SX_ENC_ISO_8859_7 },
    { "iso_8859-7", SX_ENC_ISO_8859_7 }, { "iso_8859-7:1987", SX_ENC_ISO_8859_7 }, { "sun_eu_greek", SX_ENC_ISO_8859_7 },
    { "csiso88598e", SX_ENC_ISO_8859_8 }, { "csisolatinhebrew", SX_ENC_ISO_8859_8 }, { "hebrew", SX_ENC_ISO_8859_8 },
    { "iso-8859-8", SX_ENC_ISO_8859_8 }, { "iso-8859-8-e", SX_ENC_ISO_8859_8 }, { "iso-ir-138", SX_ENC_ISO_8859_8 },
    { "iso8859-8", SX_ENC_ISO_8859_8 }, { "iso88598", SX_ENC_ISO_8859_8 }, { "iso_8859-8", SX_ENC_ISO_8859_8 },
    { "iso_8859-8:1988", SX_ENC_ISO_8859_8 }, { "visual", SX_ENC_ISO_8859_8 },
    { "csiso88598i", SX_ENC_ISO_8859_8_I }, { "iso-8859-8-i", SX_ENC_ISO_8859_8_I }, { "logical", SX_ENC_ISO_8859_8_I },
    { "csisolatin6", SX_ENC_ISO_8859_10 }, { "iso-8859-10", SX_ENC_ISO_8859_10 }, { "iso-ir-157", SX_ENC_ISO_8859_10 },
    { "iso8859-10", SX_ENC_ISO_8859_10 }, { "iso885910", SX_ENC_ISO_8859_10 }, { "l6", SX_ENC_ISO_8859_10 },
    { "latin6", SX_ENC_ISO_8859_10 },
    { "iso-8859-13", SX_ENC_ISO_8859_13 }, { "iso8859-13", SX_ENC_ISO_8859_13 }, { "iso885913", SX_ENC_ISO_8859_13 },
    { "iso-8859-14", SX_ENC_ISO_8859_14 }, { "iso8859-14", SX_ENC_ISO_8859_14 }, { "iso885914", SX_ENC_ISO_8859_14 },
    { "iso-8859-16", SX_ENC_ISO_8859_16 },
    { "koi8-ru", SX_ENC_KOI8_U }, { "koi8-u", SX_ENC_KOI8_U },
    { "csmacintosh", SX_ENC_MACINTOSH }, { "mac", SX_ENC_MACINTOSH }, { "macintosh", SX_ENC_MACINTOSH },
    { "x-mac-roman", SX_ENC_MACINTOSH },
    { "dos-874", SX_ENC_WINDOWS_874 }, { "iso-8859-11", SX_ENC_WINDOWS_874 }, { "iso8859-11", SX_ENC_WINDOWS_874 },
    { "iso885911", SX_ENC_WINDOWS_874 }, { "tis-620", SX_ENC_WINDOWS_874 }, { "windows-874", SX_ENC_WINDOWS_874 },
    { "cp1250", SX_ENC_WINDOWS_1250 }, { "windows-1250", SX_ENC_WINDOWS_1250 }, { "x-cp1250", SX_ENC_WINDOWS_1250 },
    { "cp1253", SX_ENC_WINDOWS_1253 }, { "windows-1253", SX_ENC_WINDOWS_1253 }, { "x-cp1253", SX_ENC_WINDOWS_1253 },
    { "cp1254", SX_ENC_WINDOWS_1254 }, { "csisolatin5", SX_ENC_WINDOWS_1254 }, { "iso-8859-9", SX_ENC_WINDOWS_1254 },
    { "iso-ir-148", SX_ENC_WINDOWS_1254 }, { "iso8859-9", SX_ENC_WINDOWS_1254 }, { "iso88599", SX_ENC_WINDOWS_1254 },
    { "iso_8859-9", SX_ENC_WINDOWS_1254 }, { "iso_8859-9:1989", SX_ENC_WINDOWS_1254 }, { "l5", SX_ENC_WINDOWS_1254 },
    { "latin5", SX_ENC_WINDOWS_1254 }, { "windows-1254", SX_ENC_WINDOWS_1254 }, { "x-cp1254", SX_ENC_WINDOWS_1254 },
    { "cp1255", SX_ENC_WINDOWS_1255 }, { "windows-1255", SX_ENC_WINDOWS_1255 }, { "x-cp1255", SX_ENC_WINDOWS_1255 },
    { "cp1256", SX_ENC_WINDOWS_1256 }, { "windows-1256", SX_ENC_WINDOWS_1256 }, { "x-cp1256", SX_ENC_WINDOWS_1256 },
    { "cp1257", SX_ENC_WINDOWS_1257 }, { "windows-1257", SX_ENC_WINDOWS_1257 }, { "x-cp1257", SX_ENC_WINDOWS_1257 },
    { "cp1258", SX_ENC_WINDOWS_1258 }, { "windows-1258", SX_ENC_WINDOWS_1258 }, { "x-cp1258", SX_ENC_WINDOWS_1258 },
    { "x-mac-cyrillic", SX_ENC_X_MAC_CYRILLIC }, { "x-mac-ukrainian", SX_ENC_X_MAC_CYRILLIC },
    { "big5", SX_ENC_BIG5 }, { "big5-hkscs", SX_ENC_BIG5 }, { "cn-big5", SX_ENC_BIG5 }, { "csbig5", SX_ENC_BIG5 },
    { "x-x-big5", SX_ENC_BIG5 },
    { "cseucpkdfmtjapanese", SX_ENC_EUC_JP }, { "euc-jp", SX_ENC_EUC_JP }, { "x-euc-jp", SX_ENC_EUC_JP },
    { "csshiftjis", SX_ENC_SHIFT_JIS }, { "ms932", SX_ENC_SHIFT_JIS }, { "ms_kanji", SX_ENC_SHIFT_JIS }, { "shift-jis", SX_ENC_SHIFT_JIS },
    { "shift_jis", SX_ENC_SHIFT_JIS }, { "sjis", SX_ENC_SHIFT_JIS }, { "windows-31j", SX_ENC_SHIFT_JIS }, { "x-sjis", SX_ENC_SHIFT_JIS },
    { "cseuckr", SX_ENC_EUC_KR }, { "csksc56011987", SX_ENC_EUC_KR }, { "euc-kr", SX_ENC_EUC_KR }, { "iso-ir-149", SX_ENC_EUC_KR },
    { "korean", SX_ENC_EUC_KR }, { "ks_c_5601-1987", SX_ENC_EUC_KR }, { "ks_c_5601-1989", SX_ENC_EUC_KR }, { "ksc5601", SX_ENC_EUC_KR },
    { "ksc_5601", SX_ENC_EUC_KR }, { "windows-949", SX_ENC_EUC_KR },
    { "chinese", SX_ENC_GBK }, { "csgb2312", SX_ENC_GBK }, { "csiso58gb231280", SX_ENC_GBK }, { "gb2312", SX_ENC_GBK }, { "gb_2312", SX_ENC_GBK },
    { "gb_2312-80", SX_ENC_GBK }, { "gbk", SX_ENC_GBK }, { "iso-ir-58", SX_ENC_GBK }, { "x-gbk", SX_ENC_GBK }, { "gb18030", SX_ENC_GB18030 },
    { "csiso2022kr", SX_ENC_REPLACEMENT }, { "hz-gb-2312", SX_ENC_REPLACEMENT }, { "iso-2022-cn", SX_ENC_REPLACEMENT },
    { "iso-2022-cn-ext", SX_ENC_REPLACEMENT }, { "iso-2022-kr", SX_ENC_REPLACEMENT }, { "replacement", SX_ENC_REPLACEMENT },
    { "csiso2022jp", SX_ENC_ISO_2022_JP }, { "iso-2022-jp", SX_ENC_ISO_2022_JP },
};
int for_label(const std::string& raw) {
    size_t a = 0, b = raw.size();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\f' || c == '\r'; };
    while (a < b && ws(raw[a])) a++;
    while (b > a && ws(raw[b - 1])) b--;
    std::string l = raw.substr(a, b - a);
    for (char& c : l) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    for (const Label& k : kLabels) if (l == k.label) return k.enc;
    return -1;   // (every encoding of help.rs:54-96 is built in: -2, "known to the reference, not built in", no longer occurs)
}

void put_err(const Err& e, char* err, size_t cap) {
    if (err && cap) { snprintf(err, cap, "%s", e.msg.c_str()); }
}

}  // namespace

extern "C" {

const char* sx_encoding_name(uint32_t encoding) {  // Encoding::name()
    switch (encoding) {
        case SX_ENC_X_USER_DEFINED: return "x-user-defined";
        case SX_ENC_UTF8: return "UTF-8";
        case SX_ENC_UTF16LE: return "UTF-16LE";
        case SX_ENC_UTF16BE: return "UTF-16BE";
        case SX_ENC_KOI8_R: return "KOI8-R";
        case SX_ENC_IBM866: return "IBM866";
        case SX_ENC_ISO_8859_2: return "ISO-8859-2";
        case SX_ENC_ISO_8859_5: return "ISO-8859-5";
        case SX_ENC_ISO_8859_15: return "ISO-8859-15";
        case SX_ENC_WINDOWS_1251: return "windows-1251";
        case SX_ENC_WINDOWS_1252: return "windows-1252";
        case SX_ENC_ISO_8859_3: return "ISO-8859-3";
        case SX_ENC_ISO_8859_4: return "ISO-8859-4";
        case SX_ENC_ISO_8859_6: return "ISO-8859-6";
        case SX_ENC_ISO_8859_7: return "ISO-8859-7";
        case SX_ENC_ISO_8859_8: return "ISO-8859-8";
        case SX_ENC_ISO_8859_8_I: return "ISO-8859-8-I";
        case SX_ENC_ISO_8859_10: return "ISO-8859-10";
        case SX_ENC_ISO_8859_13: return "ISO-8859-13";
        case SX_ENC_ISO_8859_14: return "ISO-8859-14";
        case SX_ENC_ISO_8859_16: return "ISO-8859-16";
        case SX_ENC_KOI8_U: return "KOI8-U";
        case SX_ENC_MACINTOSH: return "macintosh";
        case SX_ENC_WINDOWS_874: return "windows-874";
        case SX_ENC_WINDOWS_1250: return "windows-1250";
        case SX_ENC_WINDOWS_1253: return "windows-1253";
        case SX_ENC_WINDOWS_1254: return "windows-1254";
        case SX_ENC_WINDOWS_1255: return "windows-1255";
        case SX_ENC_WINDOWS_1256: return "windows-1256";
        case SX_ENC_WINDOWS_1257: return "windows-1257";
        case SX_ENC_WINDOWS_1258: return "windows-1258";
        case SX_ENC_BIG5: return "Big5";
        case SX_ENC_EUC_JP: return "EUC-JP";
        case SX_ENC_SHIFT_JIS: return "Shift_JIS";
        case SX_ENC_EUC_KR: return "EUC-KR";
        case SX_ENC_GB18030: return "gb18030";
        case SX_ENC_GBK: return "GBK";
        case SX_ENC_REPLACEMENT: return "replacement";
        case SX_ENC_ISO_2022_JP: return "ISO-2022-JP";
        case SX_ENC_X_MAC_CYRILLIC: return "x-mac-cyrillic";
        default: return nullptr;
    }
}

int sx_encoding_for_label(const char* label) { return label ? for_label(label) : -1; }

int sx_parse_enc_opt(const char* enc_opt, sx_enc_opt* out, char* err, size_t err_cap) {
    if (!enc_opt || !out) return SX_E_INVALID;
    EncOpt o;
    Err e;
    if (!parse_enc_opt(enc_opt, &o, &e)) { put_err(e, err, err_cap); return SX_E_INVALID; }
    memset(out, 0, sizeof *out);
    out->has_name = o.has_name;
    snprintf(out->name, sizeof out->name, "%s", o.name.c_str());
    out->has_chars_min = o.has_min; out->chars_min = (uint8_t)o.min;
    out->has_af = o.has_af; out->af_lo = (uint64_t)o.af; out->af_hi = (uint64_t)(o.af >> 64);
    out->has_ubf = o.has_ubf; out->ubf = (uint64_t)o.ubf;
    out->has_grep_char = o.has_grep; out->grep_char = (uint8_t)o.grep;
    return SX_OK;
}

int sx_missions_from_flags(const sx_cli_flags* f, sx_mission* out, int cap, int* n_out, char* err, size_t err_cap) {
    if (!f || !out || !n_out || cap <= 0) return SX_E_INVALID;
    Err e;
    auto bail = [&](int rc) { put_err(e, err, err_cap); return rc; };
    bool has_off, has_min, has_af, has_ubf, has_grep, has_q;
    u128 off = 0, min = 0, af = 0, ubf = 0, grep = 0, q = 0;
    const auto afl = af_aliases();
    if (!parse_integer(some(f->counter_offset), (u128)0xffffffffffffffffull, &has_off, &off, &e)) return bail(SX_E_INVALID);
    if (!parse_integer(some(f->chars_min), 0xff, &has_min, &min, &e)) return bail(SX_E_INVALID);
    if (!parse_filter(some(f->ascii_filter), ~(u128)0, afl.data(), afl.size(), &has_af, &af, &e)) return bail(SX_E_INVALID);
    if (!parse_filter(some(f->unicode_block_filter), (u128)0xffffffffffffffffull, kUbfAliases,
                      sizeof kUbfAliases / sizeof kUbfAliases[0], &has_ubf, &ubf, &e)) return bail(SX_E_INVALID);
    if (!parse_integer(some(f->grep_char), 0xff, &has_grep, &grep, &e)) return bail(SX_E_INVALID);
    if (has_grep && grep > 127) {
        e.fail("you can only `--grep-char` for ASCII codes < 128, you tried: `" + std::to_string((unsigned)grep) + "`.");
        return bail(SX_E_INVALID);
    }
    if (!parse_integer(some(f->output_line_len), (u128)0xffffffffffffffffull, &has_q, &q, &e)) return bail(SX_E_INVALID);
    if (has_q && q < 6) {
        e.fail("minimum for `--output-line-len` is `6`, you tried: `" + std::to_string((unsigned long long)q) + "`.");
        return bail(SX_E_INVALID);
    }
    std::vector<std::string> encs;
    for (int i = 0; i < f->n_encodings; i++) encs.push_back(f->encodings && f->encodings[i] ? f->encodings[i] : "");
    if (encs.empty()) encs.push_back("UTF-8");
    if ((int)encs.size() > cap) { e.fail("more encodings than room for missions"); return bail(SX_E_INVALID); }
    for (size_t id = 0; id < encs.size(); id++) {
        EncOpt o;
        if (!parse_enc_opt(encs[id], &o, &e)) return bail(SX_E_INVALID);
        const std::string scanner = std::string("Scanner ") + (char)(id + 97) + ": ";
        std::string name = o.has_name ? o.name : "UTF-8";
        const bool is_ascii = name == "ascii";
        sx_mission m;
        memset(&m, 0, sizeof m);
        m.mission_id = (uint8_t)id;
        m.counter_offset = has_off ? (uint64_t)off : 0;
        m.chars_min_nb = (uint8_t)(o.has_min ? o.min : has_min ? min : 4);
        m.require_same_unicode_block = f->same_unicode_block ? 1 : 0;
        m.output_line_char_nb_max = (uint32_t)(has_q ? q : 64);
        const u128 af_default = AF_ALL & ~AF_CTRL;                       // both default filters
        const u128 a = o.has_af ? o.af : has_af ? af : af_default;
        m.af_lo = (uint64_t)a; m.af_hi = (uint64_t)(a >> 64);
        m.ubf = (uint64_t)(o.has_ubf ? o.ubf : has_ubf ? ubf : (u128)(is_ascii ? UBF_NONE : UBF_COMMON));
        const bool g_has = o.has_grep || has_grep;
        const u128 g = o.has_grep ? o.grep : grep;
        if (g_has && g > 127) {
            e.fail(scanner + "you can only grep for ASCII codes < 128, you tried: `" + std::to_string((unsigned)g) + "`.");
            return bail(SX_E_INVALID);
        }
        m.grep_char = g_has ? (int16_t)g : (int16_t)-1;
        if (is_ascii) { m.print_encoding_as_ascii = 1; name = "x-user-defined"; }
        const int enc = for_label(name);
        if (enc == -1) {
            e.fail(scanner + "invalid input encoding name `" + name + "`, try flag `--list-encodings`.");
            return bail(SX_E_INVALID);
        }
        m.encoding = (uint8_t)enc;
        out[id] = m;
    }
    *n_out = (int)encs.size();
    return SX_OK;
}

}  // extern "C"
