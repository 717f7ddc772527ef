// sx_scan_core.hpp — device code the stage-A kernels share (sx_kernels.hip: one launch per Mission; sx_fused.hip: one launch that reads
// the buffer once for several Missions): cross-lane helpers, the classifiers, the tile-to-tile carry, the record emitter and the two
// slow paths that resolve stretches across lanes.  Included by .hip translation units only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include <type_traits>

#include "sx_device.hpp"

namespace sx {

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

#define SX_DEV __device__ __forceinline__

constexpr u32 kM = 0x80808080u;  // byte flag position

// ------------------------------------------------------------------------------------------
// cross-lane helpers (wave64)
// ------------------------------------------------------------------------------------------
SX_DEV u32 lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// lane i <- lane i+1 ; lane 63 <- edge
SX_DEV u32 from_next(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xF, 0xF, false); }
// lane i <- lane i-1 ; lane 0 <- edge
SX_DEV u32 from_prev(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xF, 0xF, false); }
SX_DEV u32 bcast(u32 v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
SX_DEV u32 shfl(u32 v, u32 src_lane) { return __builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v); }
SX_DEV u32 uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }

// byte flags (bit 7 of each byte) of four dwords -> 16-bit mask, bit j = byte j
SX_DEV u32 movemask16(u32 f0, u32 f1, u32 f2, u32 f3) {
    u32 lo = __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, 0x80402010u, lo, false);
    u32 hi = __builtin_amdgcn_udot4(f2, 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(f3, 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);  // each flag byte is 0x80: sums are 128 * mask
}
SX_DEV u32 movemask4(u32 f) { return __builtin_amdgcn_udot4(f, 0x08040201u, 0u, false) >> 7; }
// unit flags (bits 15 and 31 of four dwords, 0x80 in bytes 1 and 3) -> 16-bit byte mask; W0 / W1: the weights of the dword's two
// units in the first / second dword of a pair (Utf16RangeT)
template <u32 W0, u32 W1>
SX_DEV u32 movemask_units(u32 f0, u32 f1, u32 f2, u32 f3) {
    u32 lo = __builtin_amdgcn_udot4(f0, W0, 0u, false);
    lo = __builtin_amdgcn_udot4(f1, W1, lo, false);
    u32 hi = __builtin_amdgcn_udot4(f2, W0, 0u, false);
    hi = __builtin_amdgcn_udot4(f3, W1, hi, false);
    return (lo >> 7) | (hi << 1);
}
// two 16-bit additions in one instruction (v_pk_add_u16)
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
SX_DEV u32 pk_add16(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b)); }

SX_DEV u32 rep4(u32 b) { return b * 0x01010101u; }
// continuation bytes (10xxxxxx) of a dword as byte flags: v & ~(v << 1) & 0x80808080 — two operations (left to itself the compiler
// shifts ~v and spends three)
SX_DEV u32 cont_flags(u32 v) { return __builtin_amdgcn_bitop3_b32(v, v << 1, kM, 0x20); }

// ------------------------------------------------------------------------------------------
// Classifiers.  Input: the lane's 16 bytes (x), the dword that follows them (nx) and
// `avail` = how many bytes from the lane's first byte on are inside the chunk (only looked
// at when `near_end`, which is wave-uniform).  classify<false> returns g: bits 0..15
// "byte j belongs to an accepted valid char", bits 16.. = bits that spill onto the first
// bytes of the next lane.  classify<true> returns the start mask s: bit j = "byte j is the
// first byte of such a char" (only the slow path asks for it).
// A character counts only if ALL of its bytes are inside the chunk.
// ------------------------------------------------------------------------------------------
SX_DEV u32 fill_ff(u32 v, int nb) {  // keep the low nb bytes, set the others to 0xFF
    return nb >= 4 ? v : (nb <= 0 ? 0xFFFFFFFFu : (v | (0xFFFFFFFFu << (8 * nb))));
}
SX_DEV u32 low_mask(u32 n) { return n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u); }

// --- single byte, accept set = [a_lo,a_hi] below 0x80, all-or-none above -----------------
struct SingleByteRange {
    u32 c1, c2, high;
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
        c1 = rep4(0x80u - p.a_lo);
        c2 = rep4(0x7Fu - p.a_hi);
        high = p.high_all ? kM : 0u;
    }
    SX_DEV u32 flags(u32 x) const {
        u32 t = x & 0x7F7F7F7Fu;
        return ((((t + c1) & ~(t + c2)) & ~x) | (x & high)) & kM;
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32, u32 avail, bool near_end) const {
        u32 g = movemask16(flags(x.x), flags(x.y), flags(x.z), flags(x.w));
        if (near_end) g &= low_mask(avail);
        return g;  // one byte per char: starts == good bytes
    }
};

// --- single byte, accept set = up to K byte ranges (each on one side of 0x80) ---------------
// What a filter made of a few Unicode blocks turns into in a legacy code page (KOI8-R + Cyrillic: 20..7E, A3,
// B3, C0..FF): three SWAR operations per range and dword instead of four LDS reads per dword.
template <int K>
struct SingleByteRanges {
    u32 c1[K], c2[K], hi[K];
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
#pragma unroll
        for (int k = 0; k < K; k++) { c1[k] = p.rng_c1[k]; c2[k] = p.rng_c2[k]; hi[k] = p.rng_hi[k]; }
    }
    SX_DEV u32 flags(u32 x) const {
        const u32 t = x & 0x7F7F7F7Fu;
        u32 f = 0;
#pragma unroll
        for (int k = 0; k < K; k++) f |= (t + c1[k]) & ~(t + c2[k]) & (x ^ hi[k]);   // hi = 0: bytes >= 0x80; ~0: bytes < 0x80
        return f & kM;
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32, u32 avail, bool near_end) const {
        u32 g = movemask16(flags(x.x), flags(x.y), flags(x.z), flags(x.w));
        if (near_end) g &= low_mask(avail);
        return g;
    }
};

// --- single byte, 256-entry accept LUT in LDS (entries 0x80 / 0) ---------------------------
struct SingleByteLut {
    const uint8_t* lut;
    SX_DEV void init(const ScanParams&, const uint8_t* lds) { lut = lds; }
    SX_DEV u32 look4(u32 x) const {
        u32 a = lut[x & 0xFF], b = lut[(x >> 8) & 0xFF], c = lut[(x >> 16) & 0xFF], d = lut[x >> 24];
        return a | (b << 8) | (c << 16) | (d << 24);
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32, u32 avail, bool near_end) const {
        u32 g = movemask16(look4(x.x), look4(x.y), look4(x.z), look4(x.w));
        if (near_end) g &= low_mask(avail);
        return g;
    }
};

// --- UTF-8, af = one range, ubf = one range of 2-byte leads (C2..DF), nothing longer -----
// A byte is good iff it is an accepted ASCII byte, or an accepted lead followed by a
// continuation byte, or the continuation byte of such a pair.  No table, no LDS.
struct Utf8Range2 {
    u32 a1, a2, l1, l2;
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
        a1 = rep4(0x80u - p.a_lo);
        a2 = rep4(0x7Fu - p.a_hi);
        l1 = rep4(0x80u - (p.u_lo & 0x7F));  // leads are >= 0x80: compare the low 7 bits
        l2 = rep4(0x7Fu - (p.u_hi & 0x7F));
    }
    // Round 3: l[] keeps the garbage of its two additions below bit 7 — it only ever meets c[], which is clean (one AND less per dword);
    // LA: the continuation flags behind my 16 bytes are the NEXT lane's c[0], fetched by DPP (c_edge = the flags of the dword behind
    // lane 63's bytes, wave-uniform) instead of the next lane's raw dword classified a second time (3 VALU less per tile); not for tiles
    // near the end of the input (`avail`), nor where all lanes hold the same bytes (starts_before).
    static constexpr bool kLa = true;
    template <bool WANT_S>
    SX_DEV u32 classify_la(u32x4 x, u32 c_edge) const { return classify_impl<WANT_S, true>(x, c_edge, 32u, false); }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const { return classify_impl<WANT_S, false>(x, nx, avail, near_end); }
    template <bool WANT_S, bool LA>
    SX_DEV u32 classify_impl(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 xs[5] = { x.x, x.y, x.z, x.w, nx };
        if (near_end) {
#pragma unroll
            for (int k = 0; k < 5; k++) xs[k] = fill_ff(xs[k], (int)avail - 4 * k);
        }
        u32 a[4], l[4], c[5];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 v = xs[k], t = v & 0x7F7F7F7Fu;
            a[k] = ((t + a1) & ~(t + a2)) & ~v & kM;
            l[k] = ((t + l1) & ~(t + l2)) & v;
            c[k] = cont_flags(v);
        }
        c[4] = LA ? from_next(c[0], nx) : cont_flags(xs[4]);
        u32 p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) p[k] = l[k] & __builtin_amdgcn_alignbyte(c[k + 1], c[k], 1);
        if (WANT_S) return movemask16(a[0] | p[0], a[1] | p[1], a[2] | p[2], a[3] | p[3]);
        u32 g0 = a[0] | p[0] | (p[0] << 8);
        u32 g1 = a[1] | p[1] | __builtin_amdgcn_alignbyte(p[1], p[0], 3);
        u32 g2 = a[2] | p[2] | __builtin_amdgcn_alignbyte(p[2], p[1], 3);
        u32 g3 = a[3] | p[3] | __builtin_amdgcn_alignbyte(p[3], p[2], 3);
        return movemask16(g0, g1, g2, g3) | ((p[3] >> 31) << 16);
    }
    // Round 6, the fused kernel's fast path (sx_fused.hip): the good mask of a tile that lies fully inside the chunk, 13 operations per
    // dword instead of 14 and none for the look-ahead dword.  The accepted-lead test and "the byte behind it is a continuation byte"
    // meet in one v_bitop3 on the raw bytes moved up by one (nv & ~(nv << 1): 10xxxxxx at bit 7) instead of continuation flags of their
    // own per dword; the byte flags stay dirty below bit 7 until one AND in front of the v_dot4s.
    SX_DEV u32 classify_g(u32x4 x, u32 nx) const {
        const u32 xs[5] = { x.x, x.y, x.z, x.w, nx };
        u32 a[4], p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 v = xs[k], t = v & 0x7F7F7F7Fu;
            a[k] = (t + a1) & ~(t + a2) & ~v;
            const u32 nv = __builtin_amdgcn_alignbyte(xs[k + 1], v, 1);
            p[k] = ((t + l1) & ~(t + l2) & v) & nv & ~(nv << 1);
        }
        const u32 g0 = (a[0] | p[0] | (p[0] << 8)) & kM;
        const u32 g1 = (a[1] | p[1] | __builtin_amdgcn_alignbyte(p[1], p[0], 3)) & kM;
        const u32 g2 = (a[2] | p[2] | __builtin_amdgcn_alignbyte(p[2], p[1], 3)) & kM;
        const u32 g3 = (a[3] | p[3] | __builtin_amdgcn_alignbyte(p[3], p[2], 3)) & kM;
        u32 lo = __builtin_amdgcn_udot4(g0, 0x08040201u, 0u, false);
        lo = __builtin_amdgcn_udot4(g1, 0x80402010u, lo, false);
        u32 hi = __builtin_amdgcn_udot4(g2, 0x08040201u, 0u, false);
        hi = __builtin_amdgcn_udot4(g3, 0x80402010u, hi, false);
        const u32 m = (hi << 1) | (lo >> 7);
        return (__builtin_amdgcn_ubfe(p[3], 31u, 1u) << 16) | m;   // (v_lshl_or_b32 twice)
    }
    // ... and the start mask of such a tile from its FINAL good mask gf: a character is one accepted ASCII byte or an accepted lead with its
    // continuation byte, so the good bytes that are continuation bytes are exactly the good bytes that start nothing (15 operations
    // instead of a second classification)
    static constexpr bool kStartsFromGood = true;
    SX_DEV u32 starts_from_good(u32x4 x, u32 gf) const {
        return gf & ~movemask16(cont_flags(x.x), cont_flags(x.y), cont_flags(x.z), cont_flags(x.w)) & 0xFFFFu;
    }
};

template <class T, class = void> struct has_la : std::false_type {};
template <class T> struct has_la<T, std::void_t<decltype(T::kLa)>> : std::true_type {};
template <class C> SX_DEV std::enable_if_t<has_la<C>::value, u32> classify_la_of(const C& c, u32x4 x, u32 c_edge) { return c.template classify_la<false>(x, c_edge); }
template <class C> SX_DEV std::enable_if_t<!has_la<C>::value, u32> classify_la_of(const C&, u32x4, u32) { return 0u; }

// --- UTF-8, any af/ubf: class LUT (LDS, 256 B: one dword per bank -> conflict free) -------
// class byte: bits 0-2 continuation class one-hot (80-8F, 90-9F, A0-BF);
//             bits 3-5 (accepted starts only) which continuation classes may follow;
//             bits 6-7 (accepted starts only) length - 1.   Everything else is 0.
struct Utf8Lut {
    const uint8_t* lut;
    SX_DEV void init(const ScanParams&, const uint8_t* lds) { lut = lds; }
    SX_DEV u32 look4(u32 x) const {
        u32 a = lut[x & 0xFF], b = lut[(x >> 8) & 0xFF], c = lut[(x >> 16) & 0xFF], d = lut[x >> 24];
        return a | (b << 8) | (c << 16) | (d << 24);
    }
    static SX_DEV u32 nz(u32 v) { return (v + 0x7F7F7F7Fu) & kM; }  // bytes <= 0x3F
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 xs[5] = { x.x, x.y, x.z, x.w, nx };
        if (near_end) {
#pragma unroll
            for (int k = 0; k < 5; k++) xs[k] = fill_ff(xs[k], (int)avail - 4 * k);
        }
        u32 c[5];
#pragma unroll
        for (int k = 0; k < 5; k++) c[k] = look4(xs[k]);
        u32 v[4], len2[4], len3[4], len4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 ck = c[k];
            u32 c1 = __builtin_amdgcn_alignbyte(c[k + 1], ck, 1);
            u32 c2 = __builtin_amdgcn_alignbyte(c[k + 1], ck, 2);
            u32 c3 = __builtin_amdgcn_alignbyte(c[k + 1], ck, 3);
            u32 st = nz(ck & 0x38383838u);
            u32 k1 = nz((ck >> 3) & c1 & 0x07070707u);
            u32 k2 = nz(c2 & 0x07070707u);
            u32 k3 = nz(c3 & 0x07070707u);
            u32 lb0 = (ck << 1) & kM, lb1 = ck & kM;  // bits of (length - 1)
            u32 need1 = lb0 | lb1, need2 = lb1, need3 = lb0 & lb1;
            u32 ok = st & (k1 | ~need1) & (k2 | ~need2) & (k3 | ~need3);
            v[k] = ok; len2[k] = ok & need1; len3[k] = ok & need2; len4[k] = ok & need3;
        }
        u32 A = movemask16(v[0], v[1], v[2], v[3]);
        if (WANT_S) return A;
        u32 A2 = movemask16(len2[0], len2[1], len2[2], len2[3]);
        u32 A3 = movemask16(len3[0], len3[1], len3[2], len3[3]);
        u32 A4 = movemask16(len4[0], len4[1], len4[2], len4[3]);
        return A | (A2 << 1) | (A3 << 2) | (A4 << 3);  // up to bit 18
    }
};

// --- UTF-16, af = one range, accepted non-ASCII units = one range below U+0800 -----------
// Units sit at stream parity; every surrogate and everything outside the two ranges is a
// break.  Both bytes of an accepted unit are good; the unit's first byte is the start.
// BE_T / ODD_T: byte order and unit parity as compile-time constants (-1: read from the parameters) — the
// kernels are bound by VALU issue, and a run-time byte order costs a v_perm + v_cndmask per dword.
template <int BE_T, int ODD_T>
struct Utf16RangeT {
    u32 a1, a2, u1, u2, odd_rt, be_rt;
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
        a1 = (0x8000u - p.a_lo) * 0x00010001u;
        a2 = (0x7FFFu - p.a_hi) * 0x00010001u;
        u1 = (0x8000u - p.u_lo) * 0x00010001u;
        u2 = (0x7FFFu - p.u_hi) * 0x00010001u;
        odd_rt = p.parity & 1;
        be_rt = p.big_endian;
    }
    SX_DEV u32 odd() const { return ODD_T < 0 ? odd_rt : (u32)ODD_T; }
    SX_DEV bool be() const { return BE_T < 0 ? be_rt != 0 : BE_T != 0; }
    // Round 5: the range compares as PACKED 16-bit additions (v_pk_add_u16: no carry from unit to unit, so the unit's bit 15 need not
    // be cleared first — a unit >= 0x8000 wraps, and is masked out by ~v anyway), and the byte masks straight from the UNIT flags:
    // the v_dot4 weights 3 / 12 / 48 / 192 set both bytes' bits of a good unit (1 / 4 / 16 / 64: its first byte's, the start mask)
    // — rounds 1-4 spread the flags onto both bytes first (a shift and an OR per dword).  62 -> 50 vector instructions per tile.
    SX_DEV u32 unit_flags(u32 v) const {  // two units per dword -> flags at bits 15 and 31
        if (be()) v = __builtin_amdgcn_perm(0u, v, 0x02030001u);
        u32 ra = pk_add16(v, a1) & ~pk_add16(v, a2);
        u32 ru = pk_add16(v, u1) & ~pk_add16(v, u2);
        return (ra | ru) & ~v & 0x80008000u;
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 d0 = x.x, d1 = x.y, d2 = x.z, d3 = x.w;
        if (odd()) {  // unit k of this lane = bytes 2k+1, 2k+2
            d0 = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
            d1 = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
            d2 = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
            d3 = __builtin_amdgcn_alignbyte(nx, x.w, 1);
        }
        u32 f0 = unit_flags(d0), f1 = unit_flags(d1), f2 = unit_flags(d2), f3 = unit_flags(d3);
        u32 m = WANT_S ? movemask_units<0x04000100u, 0x40001000u>(f0, f1, f2, f3)     // the unit's first byte
                       : movemask_units<0x0C000300u, 0xC0003000u>(f0, f1, f2, f3);    // both bytes of the unit
        if (near_end) {  // whole units only
            u32 nu = avail > odd() ? (avail - odd()) >> 1 : 0u;
            m &= low_mask(2 * nu);
        }
        return m << odd();  // odd parity: everything sits one byte later (bit 16 spills)
    }
};
using Utf16Range = Utf16RangeT<-1, -1>;

// --- UTF-16, any af/ubf incl. astral: two 256-entry LUTs in LDS ---------------------------
// lutH[hi byte]: bits 0-3 accept per (lo byte >> 6) quadrant; bit 4 "hi byte is 0: use
// lutL[lo byte] bit 0"; bit 5 high surrogate (bits 0-3 then say whether the PAIR is
// accepted); bit 6 low surrogate.
struct Utf16Lut {
    const uint8_t *lutH, *lutL;
    u32 odd, be;
    SX_DEV void init(const ScanParams& p, const uint8_t* lds) {
        lutH = lds; lutL = lds + 256; odd = p.parity & 1; be = p.big_endian;
    }
    // one unit -> bit0 accepted BMP char, bit1 high surrogate of an accepted plane, bit2 low surrogate
    SX_DEV u32 unit_class(u32 u) const {
        u32 hb = u >> 8, lb = u & 0xFF;
        u32 h = lutH[hb];
        u32 q = (h >> (lb >> 6)) & 1;
        u32 l = lutL[lb] & 1;
        u32 acc = (h & 0x10) ? l : q;
        u32 hs = (h >> 5) & 1, ls = (h >> 6) & 1;
        return (hs | ls) ? ((hs & q) << 1) | (ls << 2) : acc;
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 d[5] = { x.x, x.y, x.z, x.w, nx };
        if (odd) {
            d[0] = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
            d[1] = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
            d[2] = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
            d[3] = __builtin_amdgcn_alignbyte(nx, x.w, 1);
            d[4] = nx >> 8;
        }
        u32 nu = 9;  // whole units available (8 own + 1 look-ahead)
        if (near_end) { nu = avail > odd ? (avail - odd) >> 1 : 0u; if (nu > 9) nu = 9; }
        u32 bmp = 0, hs = 0, ls = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            u32 v = d[k >> 1];
            if (be) v = __builtin_amdgcn_perm(0u, v, 0x02030001u);
            u32 u = (k & 1) ? (v >> 16) : (v & 0xFFFF);
            u32 cl = unit_class(u);
            bmp |= (cl & 1) << k; hs |= ((cl >> 1) & 1) << k; ls |= ((cl >> 2) & 1) << k;
        }
        u32 um = low_mask(nu);
        bmp &= um; hs &= um; ls &= um;
        u32 pair = hs & (ls >> 1) & 0xFF;   // accepted pair whose high surrogate is own unit k
        u32 start_u = (bmp & 0xFF) | pair;   // char starts, unit granularity
        u32 m;
        if (WANT_S) {
            m = start_u;  // bit k -> bit 2k
            m = (m | (m << 4)) & 0x0F0Fu; m = (m | (m << 2)) & 0x3333u; m = (m | (m << 1)) & 0x5555u;
        } else {
            m = start_u | (pair << 1);  // bit 8 = first unit of the next lane
            m = (m | (m << 8)) & 0x00FF00FFu; m = (m | (m << 4)) & 0x0F0F0F0Fu;
            m = (m | (m << 2)) & 0x33333333u; m = (m | (m << 1)) & 0x55555555u;
            m |= m << 1;
        }
        return m << odd;
    }
};

// --- UTF-8 with three-byte leads / UTF-16 with up to four unit ranges: the alias filters (Cjk, Asian, Kana, Hangul ...) as SWAR ranges
#include "sx_classify_ranges.hpp"

// ------------------------------------------------------------------------------------------
// Tile-to-tile state of one wavefront.  Everything here is wave-uniform (SGPRs).
// ------------------------------------------------------------------------------------------
struct Carry {
    u32 g63;      // previous tile, lane 63: final 16-bit good mask | own spill bits << 16
    u32 tracked;  // 1: a stretch is open at the tile start and described below
    u32 t_chars, t_flags;
    u64 t_start;
};

// Record output.  One global atomic word sustains only ~90 returning atomics per
// microsecond on this chip (MI355X_MICROARCH.md, row "dequeue"), which is less than the
// record rate of a text-rich input.  So a wavefront reserves record slots in blocks: one
// atomic per kRecBlock records.  Slots of a block that stay unused are marked invalid
// (kRecInvalid) so that the host can skip them.  All fields are wave-uniform (SGPRs).
constexpr u32 kRecBlock = 16;
// Region mode (region_cap > 0): no shared pool and no atomics — the records of sub-chunk w go
// to slots [w*region_cap, (w+1)*region_cap) in the order they are found, and their number to
// region_counts[w]; a compaction pass (sx_sort.hip) then yields the records sorted by position
// without sorting.  Records beyond region_cap are only counted (counters[0]): the host then
// repeats the launch with the shared pool.
struct Emitter {
    DevRun* recs;
    u32* counters;
    u32 capacity;
    u32 base, left;  // my current block: slots [base, base+left) are still free
    u32 region_cap, rcount;
    u32* region_counts;
    u32 heavy_n = 0;   // tiles of this sub-chunk that took the general path (a statistic: one atomic per sub-chunk, not per tile)

    SX_DEV void begin_region(u64 wave) {
        if (region_cap) { base = (u32)wave * region_cap; rcount = 0; }
    }
    // (records beyond the region's room are only counted; one atomic per sub-chunk — one per append made the scan of a
    // string-dense buffer 30x slower: a single word takes ~90 atomics per microsecond)
    // Round 5: the two statistics (tiles on the general path, records of the launch) are added up in kStatShards (16) words 128 bytes apart
    // (sx_device.hpp), picked by the sub-chunk's number, and summed by the host.  One word takes ~90 atomics per microsecond: with a
    // record in nearly every sub-chunk (the headline's UTF-8 Mission: 262 144 sub-chunks per 64 GiB, 1.4 atomics each) that is 4 ms
    // of one L2 channel's time inside a 12 ms launch, and 64 KiB sub-chunks made the launch 20.8 ms long (r05 probe).
    SX_DEV void end_region(u64 wave) {
        u32* const shard = counters + kStatBase + ((u32)wave & (kStatShards - 1u)) * kStatStride;
        if (heavy_n && lane_id() == 0) atomicAdd(shard, heavy_n);
        heavy_n = 0;
        if (region_cap && lane_id() == 0) {
            region_counts[wave] = rcount < region_cap ? rcount : region_cap;
            if (rcount) atomicAdd(shard + 1, rcount);                  // all records of the launch (stage A's host side: how dense is the input?)
            if (rcount > region_cap) {
                atomicAdd(counters, rcount - region_cap);              // records that found no room
                atomicMax(counters + 3, rcount);                       // how much room the fullest sub-chunk needs
            }
        }
    }
    SX_DEV void invalidate_rest() {
        if (region_cap) return;
        u32 lane = lane_id();
        if (lane < left && base + lane < capacity) {
            DevRun r; r.start = 0; r.len = kRecInvalidLen; r.chars_flags = kRecInvalidFlags;
            recs[base + lane] = r;
        }
        left = 0;
    }
    // all lanes call (convergent); `want` lanes append one record each
    SX_DEV void append(bool want, u64 start, u64 end, u32 chars, u32 flags) {
        u64 m = __ballot(want);
        if (m == 0) return;
        u32 lane = lane_id();
        u32 n = (u32)__popcll(m);
        u32 idx;
        bool room;
        if (region_cap) {
            const u32 k = rcount + (u32)__popcll(m & ((1ull << lane) - 1ull));
            idx = base + k;
            room = k < region_cap;
            rcount += n;
        } else {
            if (n > left) {
                invalidate_rest();
                u32 grab = n > kRecBlock ? n : kRecBlock, b = 0;
                if (lane == 0) b = atomicAdd(counters, grab);
                base = __builtin_amdgcn_readfirstlane(b);
                left = grab;
            }
            idx = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
            room = idx < capacity;
            base += n; left -= n;
        }
        if (want && room) {
            DevRun r;
            r.start = start;
            r.len = (u32)(end - start);
            r.chars_flags = (chars > kRecCharsMask ? kRecCharsMask : chars) | flags;
            recs[idx] = r;
        }
    }
};

SX_DEV u32 wave_inclusive_scan(u32 v) {
    u32 lane = lane_id();
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        u32 o = shfl(v, lane >= d ? lane - d : lane);
        if (lane >= d) v += o;
    }
    return v;
}

SX_DEV u32 trailing_ones16(u32 g16) {  // ones from bit 15 downwards
    u32 inv = (~g16) & 0xFFFFu;
    return inv ? (u32)__clz((int)inv) - 16u : 16u;
}

// Light path: every stretch of >= cand_bytes bytes that ENDS in this tile lies inside the
// 32-bit window (previous lane | own lane) of the lane it ends in, and nothing long is
// open at either tile edge.  Each such lane resolves its stretch with a few bit operations.
// Returns false (nothing emitted) if the tile needs the general path instead.
//   w  = own good mask << 16 | previous lane's good mask;  sw = the same for start masks
//   r  = bit p set iff bits p-cand_bytes+1..p of w are all set
template <class EM>
SX_DEV bool light_path(u32 w, u32 sw, u32 r, u64 lane_base, EM& em, u32 min_chars) {
    const u32 lane = lane_id();
    const u32 gf = w >> 16;
    const u32 n0 = from_next(gf & 1u, 1u);  // lane 63: the next tile is unknown -> "goes on"
    const u32 ends = gf & ~((gf >> 1) | (n0 << 15));
    u32 cand = ends & (r >> 16);
    const bool open_long = lane == 63 && (r >> 31);  // >= cand_bytes already and still open at the tile end
    const u32 e0 = 16u + (cand ? (u32)__builtin_ctz(cand) : 0u);
    const bool unresolved = cand && ((~w) & ((1u << e0) - 1u)) == 0u;  // reaches beyond the window
    if (__ballot(open_long || unresolved)) return false;
    while (__ballot(cand != 0)) {
        const bool has = cand != 0;
        const u32 e = 16u + (has ? (u32)__builtin_ctz(cand) : 0u);
        const u32 below = (~w) & ((1u << e) - 1u);            // has a set bit whenever `has`
        const u32 st = below ? 32u - (u32)__clz((int)below) : 0u;
        const u32 field = (e >= 31u ? 0xFFFFFFFFu : ((1u << (e + 1u)) - 1u)) & ~((1u << st) - 1u);
        const u32 ch = (u32)__popc(sw & field);
        em.append(has && ch >= min_chars, lane_base - 16 + st, lane_base - 16 + e + 1, ch, 0u);
        cand &= cand - 1u;
    }
    return true;
}

// General path: exact resolution of all stretches that END inside this tile (the one still
// open at the tile end goes into the carry).  g: final 16-bit good mask; s: start mask;
// g_raw: the classifier's output (for its spill bits); s63: start mask of the previous
// tile's lane 63 (only read if an untracked stretch is open on entry).
// `first_tile`: the tile starts a sub-chunk: a stretch that is open on entry is clipped to
// the sub-chunk start and flagged kRecStartOpen (the previous wave reports the part before).
template <class EM>
SX_DEV void heavy_path(u32 g, u32 s, u32 g_raw, u32 g63_in, u32 s63, u32 r16, u64 tile_base, u64 tile_end, Carry& c,
                       EM& em, u32 min_chars, u32 cand_bytes, bool first_tile) {
    const u32 lane = lane_id();
    g &= 0xFFFFu;
    s &= 0xFFFFu;

    // -- the stretch that is open when the tile begins
    bool open = false;
    u64 ostart = 0;
    u32 ochars = 0, oflags = 0;
    if (c.tracked) {
        open = true; ostart = c.t_start; ochars = c.t_chars; oflags = c.t_flags;
    } else if (g63_in & 0x8000u) {
        open = true;
        if (first_tile) { ostart = tile_base; ochars = 0; oflags = kRecStartOpen; }
        else {  // shorter than cand_bytes <= 14 bytes: it lies inside lane 63 of the previous tile
            u32 suf = trailing_ones16(g63_in & 0xFFFFu);
            ostart = tile_base - suf;
            ochars = (u32)__popc((s63 & 0xFFFFu) >> (16u - suf));
        }
    }
    u32 g0 = bcast(g, 0) & 1u;
    if (open && !g0) {  // it ended exactly at the tile boundary
        em.append(lane == 0 && (ochars >= min_chars || oflags), ostart, tile_base, ochars, oflags);
        open = false;
    }

    // -- per-lane summaries
    bool all = g == 0xFFFFu;
    u32 cnt = (u32)__popc(s);
    u32 trail1 = trailing_ones16(g);
    u32 trail_chars = (u32)__popc(s >> (16u - trail1));
    u64 zmask = __ballot(!all);
    u32 P = wave_inclusive_scan(cnt);
    u32 Pex = P - cnt;

    // -- what lies to the left of my byte 0 (used only if my bit 0 is set)
    u64 below = zmask & ((1ull << lane) - 1ull);
    int j = below ? 63 - __clzll((long long)below) : -1;
    u32 jj = j < 0 ? 0u : (u32)j;
    u32 tj = shfl(trail1, jj), tcj = shfl(trail_chars, jj), Pj = shfl(P, jj);
    u64 left_start;
    u32 left_chars, left_flags = 0;
    if (j >= 0) {
        left_start = tile_base + 16ull * jj + (16u - tj);
        left_chars = tcj + (Pex - Pj);
    } else {
        left_start = open ? ostart : tile_base;
        left_chars = (open ? ochars : 0u) + Pex;
        left_flags = open ? oflags : 0u;
    }

    // -- my stretches that can matter (round 3; before, every stretch of the lane went through the loop: up to eight rounds of
    //    ballot + append per tile where one or two matter).  r16: bit e set iff the cand_bytes bytes up to my byte e are all good —
    //    a record needs min_chars characters, i.e. at least cand_bytes bytes.  So: the closed stretches whose end bit is in r16, the
    //    closed stretch at my byte 0 that carries flags from the left (the part of a stretch cut at the sub-chunk start is reported
    //    whatever its length), and — without a loop — the stretch still open at the tile end.
    const u32 nb0 = from_next(g & 1u, 2u);  // lane 63: the next tile is not classified yet -> "goes on"
    const u64 lane_base = tile_base + 16ull * lane;
    u32 ends = g & ~(g >> 1);
    if (nb0) ends &= 0x7FFFu;
    u32 cand = ends & r16;
    if ((g & 1u) && left_flags) cand |= ends & (1u << ((u32)__builtin_ctz(~g) - 1u));   // (~g has a bit above 15 at the latest)
    while (__ballot(cand != 0)) {
        const bool has = cand != 0;
        const u32 e = has ? (u32)__builtin_ctz(cand) : 0u;
        const u32 zb = ~g & ((1u << e) - 1u);
        const u32 st = zb ? 32u - (u32)__clz((int)zb) : 0u;
        const u32 field = ((2u << e) - 1u) & ~((1u << st) - 1u);
        u32 ch = (u32)__popc(s & field);
        u64 start = lane_base + st;
        u32 flags = 0;
        if (st == 0) { ch += left_chars; start = left_start; flags = left_flags; }
        em.append(has && (ch >= min_chars || flags), start, lane_base + e + 1u, ch, flags);
        cand &= cand - 1u;
    }
    u32 my_open = 0, my_och = 0, my_ofl = 0;
    u64 my_ostart = 0;
    if (lane == 63 && (g & 0x8000u)) {   // open at the tile end
        my_open = 1;
        if (all) { my_ostart = left_start; my_och = left_chars + cnt; my_ofl = left_flags; }
        else { my_ostart = lane_base + (16u - trail1); my_och = trail_chars; }
    }

    // -- state for the next tile
    c.g63 = bcast(g | (g_raw & 0xFFFF0000u), 63);
    c.tracked = 0;
    if (bcast(my_open, 63)) {
        u64 os = ((u64)bcast((u32)(my_ostart >> 32), 63) << 32) | bcast((u32)my_ostart, 63);
        u32 ofl = bcast(my_ofl, 63);
        if (tile_end - os >= cand_bytes || ofl) {  // short and plain: re-derived from g63 when needed
            c.tracked = 1; c.t_start = os; c.t_chars = bcast(my_och, 63); c.t_flags = ofl;
        }
    }
}

}  // namespace sx
