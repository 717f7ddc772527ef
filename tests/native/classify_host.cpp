// Test-only harness: the range classifiers of stage A (stringsext_amd/csrc/sx_classify_ranges.hpp) compiled as host code and
// driven the way scan_kernel drives them — 16 bytes per lane, the dword behind them as look-ahead, `avail` at the end of the
// input, the spill bits of a lane ORed onto the first bytes of the next.  The three gfx950 builtins the header uses are
// restated below bit for bit (v_alignbyte_b32, v_perm_b32 for selectors 0..7, v_dot4_u32_u8); the helpers are those of
// sx_kernels.hip.  Output: per input byte "belongs to an accepted valid character" and "starts one", which
// tests/test_classify_ranges.py compares with a byte-by-byte statement of the decoders' rules.
#include <stdint.h>
#include <string.h>

#include "../../stringsext_amd/csrc/sx_device.hpp"

namespace sx {
typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
struct u32x4 { u32 x, y, z, w; };
#define SX_DEV inline
constexpr u32 kM = 0x80808080u;

static inline u32 emu_alignbyte(u32 hi, u32 lo, u32 n) { return (u32)(((((u64)hi) << 32) | lo) >> (8 * (n & 3))); }
static inline u32 emu_perm(u32 a, u32 b, u32 sel) {
    const u64 src = (((u64)a) << 32) | b;
    u32 r = 0;
    for (int i = 0; i < 4; i++) { const u32 s = (sel >> (8 * i)) & 0xFF; r |= (u32)((src >> (8 * (s & 7))) & 0xFF) << (8 * i); }
    return r;
}
static inline u32 emu_udot4(u32 a, u32 b, u32 c, bool) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return c;
}
#define __builtin_amdgcn_alignbyte emu_alignbyte
#define __builtin_amdgcn_perm emu_perm
#define __builtin_amdgcn_udot4 emu_udot4

SX_DEV u32 rep4(u32 b) { return b * 0x01010101u; }
SX_DEV u32 fill_ff(u32 v, int nb) { return nb >= 4 ? v : (nb <= 0 ? 0xFFFFFFFFu : (v | (0xFFFFFFFFu << (8 * nb)))); }
SX_DEV u32 low_mask(u32 n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }
SX_DEV u32 movemask16(u32 f0, u32 f1, u32 f2, u32 f3) {
    u32 lo = emu_udot4(f0, 0x08040201u, 0u, false);
    lo = emu_udot4(f1, 0x80402010u, lo, false);
    u32 hi = emu_udot4(f2, 0x08040201u, 0u, false);
    hi = emu_udot4(f3, 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);
}
#include "../../stringsext_amd/csrc/sx_classify_ranges.hpp"

static u32 rd32(const u8* d, u64 len, u64 off) {   // a buffer load: bytes past the end read as 0
    u32 v = 0;
    for (int i = 0; i < 4; i++) if (off + i < len) v |= (u32)d[off + i] << (8 * i);
    return v;
}

template <class CLS>
static void run(const ScanParams& p, const u8* d, u64 len, int always_near_end, u8* good, u8* start) {
    CLS cls;
    cls.init(p, nullptr);
    u32 spill = 0;
    for (u64 base = 0; base < len; base += 16) {
        const u32x4 x{ rd32(d, len, base), rd32(d, len, base + 4), rd32(d, len, base + 8), rd32(d, len, base + 12) };
        const u32 nx = rd32(d, len, base + 16);
        const u32 avail = len - base > 32 ? 32u : (u32)(len - base);
        const bool ne = always_near_end || base + 32 > len;   // (scan_kernel: tiles whose bytes + 16 lie inside the chunk are "safe")
        const u32 g = cls.template classify<false>(x, nx, avail, ne);
        const u32 s = cls.template classify<true>(x, nx, avail, ne);
        const u32 gf = (g & 0xFFFFu) | spill;
        spill = g >> 16;
        for (u32 j = 0; j < 16 && base + j < len; j++) { good[base + j] = (gf >> j) & 1; start[base + j] = (s >> j) & 1; }
    }
}
}  // namespace sx

using namespace sx;

extern "C" {
// UTF-8: af = [a_lo, a_hi], 2-byte leads [u_lo, u_hi] (empty: u_lo > u_hi), 3-byte leads [l3_lo, l3_hi]; has2 / ed pick the instantiation as launch_scan does
int sxh_classify_utf8_range3(uint32_t a_lo, uint32_t a_hi, uint32_t u_lo, uint32_t u_hi, uint32_t l3_lo, uint32_t l3_hi, int has2, int ed,
                             const uint8_t* d, uint64_t len, int always_near_end, uint8_t* good, uint8_t* start) {
    ScanParams p;
    memset(&p, 0, sizeof p);
    p.a_lo = a_lo; p.a_hi = a_hi; p.u_lo = u_lo; p.u_hi = u_hi; p.l3_lo = l3_lo; p.l3_hi = l3_hi;
    switch (has2 * 3 + ed) {
    case 0: run<Utf8Range3T<false, 0>>(p, d, len, always_near_end, good, start); break;
    case 1: run<Utf8Range3T<false, 1>>(p, d, len, always_near_end, good, start); break;
    case 2: run<Utf8Range3T<false, 2>>(p, d, len, always_near_end, good, start); break;
    case 3: run<Utf8Range3T<true, 0>>(p, d, len, always_near_end, good, start); break;
    case 4: run<Utf8Range3T<true, 1>>(p, d, len, always_near_end, good, start); break;
    case 5: run<Utf8Range3T<true, 2>>(p, d, len, always_near_end, good, start); break;
    default: return -1;
    }
    return 0;
}
// UTF-16: n unit ranges [lo[k], hi[k]] (ascending, no surrogate inside): at most two below U+8000, one across it, one above; general = 1: the
// instantiation with every slot (the unused ones empty) instead of the one launch_scan picks; [hs_lo, hs_hi]: the high surrogates of the
// astral planes that pass (0, 0: none)
int sxh_classify_utf16_ranges(const uint32_t* lo, const uint32_t* hi, int n, uint32_t hs_lo, uint32_t hs_hi, int general, int be, int odd,
                              const uint8_t* d, uint64_t len, int always_near_end, uint8_t* good, uint8_t* start) {
    ScanParams p;
    memset(&p, 0, sizeof p);
    p.big_endian = (uint32_t)be; p.parity = (uint32_t)odd;
    for (int k = 0; k < 6; k++) { p.rng_c1[k] = 0u; p.rng_c2[k] = 0x7FFFu * 0x00010001u; }
    uint32_t il = 0, ih = 3, nl = 0, ns = 0, nh = 0;
    for (int k = 0; k < n; k++) {   // (as sx_mission.cpp fills the slots)
        uint32_t slot;
        if (hi[k] < 0x8000u) { slot = il++; nl++; } else if (lo[k] >= 0x8000u) { slot = ih++; nh++; } else { slot = 2; ns++; }
        if (nl > 2 || ns > 1 || nh > 1) return -2;
        p.rng_c1[slot] = (0x8000u - (lo[k] & 0x7FFFu)) * 0x00010001u;
        p.rng_c2[slot] = (0x8000u + (hi[k] & 0x7FFFu)) * 0x00010001u;
    }
    p.n_ranges = nl | (ns << 4) | (nh << 8) | (hs_hi ? 1u << 12 : 0u);
    if (hs_hi) { p.rng_c1[5] = (0x8000u - (hs_lo & 0x7FFFu)) * 0x00010001u; p.rng_c2[5] = (0x8000u + (hs_hi & 0x7FFFu)) * 0x00010001u; }
    if (hs_hi || general) {   // surrogate pairs: the one instantiation with every slot
#define SXH_AST(B, O) if (be == B && odd == O) { run<Utf16RangesT<B, O, 2, 1, 1, 1>>(p, d, len, always_near_end, good, start); return 0; }
        SXH_AST(0, 0) SXH_AST(0, 1) SXH_AST(1, 0) SXH_AST(1, 1)
#undef SXH_AST
    }
    nl = nl <= 1 && (ns | nh) ? 1u : 2u;   // (launch_scan)
#define SXH_BO(NL, NS, NH, B, O) if (be == B && odd == O) { run<Utf16RangesT<B, O, NL, NS, NH>>(p, d, len, always_near_end, good, start); return 0; }
#define SXH_CASE(NL, NS, NH) if (nl == NL && ns == NS && nh == NH) { SXH_BO(NL, NS, NH, 0, 0) SXH_BO(NL, NS, NH, 0, 1) SXH_BO(NL, NS, NH, 1, 0) SXH_BO(NL, NS, NH, 1, 1) }
    SXH_CASE(1, 1, 0) SXH_CASE(1, 0, 1) SXH_CASE(1, 1, 1) SXH_CASE(2, 0, 0) SXH_CASE(2, 1, 0) SXH_CASE(2, 0, 1) SXH_CASE(2, 1, 1)
#undef SXH_CASE
#undef SXH_BO
    return -1;
}

// The product's own choice: kind and parameters as sx_scan_classifier hands them out (out20), dispatched as launch_scan does (sx_kernels.hip).
// Returns -3 for a kind that is not one of this header's classifiers.
int sxh_classify_product(int kind, const uint32_t* p20, int be, int odd, const uint8_t* d, uint64_t len, int always_near_end, uint8_t* good, uint8_t* start) {
    ScanParams p;
    memset(&p, 0, sizeof p);
    p.a_lo = p20[0]; p.a_hi = p20[1]; p.u_lo = p20[2]; p.u_hi = p20[3]; p.l3_lo = p20[4]; p.l3_hi = p20[5]; p.n_ranges = p20[6];
    for (int k = 0; k < 6; k++) { p.rng_c1[k] = p20[7 + k]; p.rng_c2[k] = p20[13 + k]; }
    p.big_endian = (uint32_t)be; p.parity = (uint32_t)odd;
    if (kind == (int)kClsUtf8Range3) {
        const int ed = p.l3_hi < 0xEDu || p.l3_lo > 0xEDu ? 0 : p.l3_hi == 0xEDu ? 1 : 2;
        switch ((p.u_lo <= p.u_hi ? 3 : 0) + ed) {
        case 0: run<Utf8Range3T<false, 0>>(p, d, len, always_near_end, good, start); break;
        case 1: run<Utf8Range3T<false, 1>>(p, d, len, always_near_end, good, start); break;
        case 2: run<Utf8Range3T<false, 2>>(p, d, len, always_near_end, good, start); break;
        case 3: run<Utf8Range3T<true, 0>>(p, d, len, always_near_end, good, start); break;
        case 4: run<Utf8Range3T<true, 1>>(p, d, len, always_near_end, good, start); break;
        default: run<Utf8Range3T<true, 2>>(p, d, len, always_near_end, good, start); break;
        }
        return 0;
    }
    if (kind == (int)kClsUtf8Range2x2) { run<Utf8Range2x2>(p, d, len, always_near_end, good, start); return 0; }
    if (kind != (int)kClsUtf16Ranges) return -3;
    const uint32_t ns = (p.n_ranges >> 4) & 1u, nh = (p.n_ranges >> 8) & 1u, nl = (p.n_ranges & 15u) == 3u ? 3u : (p.n_ranges & 15u) <= 1u && (ns | nh) ? 1u : 2u;
    if (p.n_ranges >> 12) {
#define SXH_AST(B, O) if (be == B && odd == O) { run<Utf16RangesT<B, O, 2, 1, 1, 1>>(p, d, len, always_near_end, good, start); return 0; }
        SXH_AST(0, 0) SXH_AST(0, 1) SXH_AST(1, 0) SXH_AST(1, 1)
#undef SXH_AST
    }
#define SXH_BO(NL, NS, NH, B, O) if (be == B && odd == O) { run<Utf16RangesT<B, O, NL, NS, NH>>(p, d, len, always_near_end, good, start); return 0; }
#define SXH_CASE(NL, NS, NH) if (nl == NL && ns == NS && nh == NH) { SXH_BO(NL, NS, NH, 0, 0) SXH_BO(NL, NS, NH, 0, 1) SXH_BO(NL, NS, NH, 1, 0) SXH_BO(NL, NS, NH, 1, 1) }
    SXH_CASE(3, 0, 0) SXH_CASE(1, 1, 0) SXH_CASE(1, 0, 1) SXH_CASE(1, 1, 1) SXH_CASE(2, 0, 0) SXH_CASE(2, 1, 0) SXH_CASE(2, 0, 1) SXH_CASE(2, 1, 1)
#undef SXH_CASE
#undef SXH_BO
    return -1;
}
}
