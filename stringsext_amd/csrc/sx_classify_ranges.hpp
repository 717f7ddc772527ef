// sx_classify_ranges.hpp — range classifiers of stage A for the alias filters of reference src/mission.rs:167-218
// (-u Cjk, Kana, Hangul, Asian, Misc, and their unions with Common / Latin ...): every one of them is a few contiguous
// ranges of UTF-8 lead bytes, i.e. a few ranges of UTF-16 units, so the byte classes are SWAR compares on registers —
// no table in LDS, no ds_read per byte (the LUT classifiers of sx_kernels.hip spend 213 vector instructions per KiB tile).
//
// Included by sx_kernels.hip (device) after its helpers (u32, u32x4, SX_DEV, kM, rep4, fill_ff, low_mask, movemask16)
// and, with those helpers and the three builtins restated for the host, by tests/native/classify_host.cpp, which
// compares every classifier lane by lane with the byte-by-byte statement of the decoders' rules.
//
// The contract of classify<WANT_S>() is the one of sx_kernels.hip: classify<false> returns g, bits 0..15 "byte j of the
// lane belongs to an accepted valid character that STARTS in this lane", bits 16.. the bytes of such characters that
// lie in the next lane; classify<true> returns the start mask.  A character counts only if all of its bytes are inside
// the chunk (`avail` bytes from the lane's first byte on, looked at when `near_end`).
#pragma once

// --- UTF-8: af = one range, 2-byte leads = one range of C2..DF (HAS2), 3-byte leads = one range of E1..EF, no 4-byte leads ---------
// What the decoder checks beyond "lead + the right number of continuation bytes" (encoding_rs utf_8.rs, restated in
// sx_codec_core.hpp ddec_utf8 and oracle/sxo.c): E0 wants A0..BF as its second byte, ED wants 80..9F.  E0 is not in the range by
// construction (a filter with E0 takes the LUT kernel); ED: 0 = not in the range; 1 = the range's last lead; 2 = inside the range.
template <bool HAS2, int ED>
struct Utf8Range3T {
    u32 a1, a2, l1, l2, m1, m2;
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
        a1 = rep4(0x80u - p.a_lo);
        a2 = rep4(0x7Fu - p.a_hi);
        l1 = rep4(0x80u - (p.u_lo & 0x7F));   // leads are >= 0x80: compare the low 7 bits
        l2 = rep4(0x7Fu - (p.u_hi & 0x7F));
        m1 = rep4(0x80u - (p.l3_lo & 0x7F));
        m2 = rep4(0x7Fu - (p.l3_hi & 0x7F));
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 xs[5] = { x.x, x.y, x.z, x.w, nx };
        if (near_end) {
#pragma unroll
            for (int k = 0; k < 5; k++) xs[k] = fill_ff(xs[k], (int)avail - 4 * k);
        }
        u32 c[5];
#pragma unroll
        for (int k = 0; k < 5; k++) c[k] = xs[k] & ~(xs[k] << 1);   // 10xxxxxx: bit 7 of every byte says so (the other bits are garbage until the last
                                                                 // AND with kM: every step below is bitwise or moves whole bytes)
        u32 a[4], q[4], p3[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 v = xs[k], t = v & 0x7F7F7F7Fu;
            a[k] = ((t + a1) & ~(t + a2)) & ~v;
            u32 l3 = ((t + m1) & ~(t + m2)) & v;
            const u32 c1 = __builtin_amdgcn_alignbyte(c[k + 1], c[k], 1);
            const u32 c2 = __builtin_amdgcn_alignbyte(c[k + 1], c[k], 2);
            if (ED) {   // ED A0..BF would be a surrogate: the second byte's bit 5, brought to the lead's flag bit
                const u32 b5 = __builtin_amdgcn_alignbyte(xs[k + 1], v, 1) << 2;
                u32 is_ed = t + 0x13131313u;                                  // bit 7: low 7 bits >= 0x6D (inside l3: the lead is >= ED)
                if (ED == 2) is_ed &= ~(t + 0x12121212u);                     // ... and not >= 0x6E
                l3 &= ~(is_ed & b5);
            }
            p3[k] = l3 & c1 & c2;
            q[k] = p3[k];
            if (HAS2) q[k] |= ((t + l1) & ~(t + l2)) & v & c1;
        }
        if (WANT_S) return movemask16((a[0] | q[0]) & kM, (a[1] | q[1]) & kM, (a[2] | q[2]) & kM, (a[3] | q[3]) & kM);
        // a lead spreads onto its one or two continuation bytes
        u32 g0 = a[0] | q[0] | (q[0] << 8) | (p3[0] << 16);
        u32 g1 = a[1] | q[1] | __builtin_amdgcn_alignbyte(q[1], q[0], 3) | __builtin_amdgcn_alignbyte(p3[1], p3[0], 2);
        u32 g2 = a[2] | q[2] | __builtin_amdgcn_alignbyte(q[2], q[1], 3) | __builtin_amdgcn_alignbyte(p3[2], p3[1], 2);
        u32 g3 = a[3] | q[3] | __builtin_amdgcn_alignbyte(q[3], q[2], 3) | __builtin_amdgcn_alignbyte(p3[3], p3[2], 2);
        // next lane: its byte 0 after a lead in my byte 15 or a 3-byte lead in my byte 14, its byte 1 after a 3-byte lead in my byte 15
        const u32 sp0 = ((q[3] >> 31) | (p3[3] >> 23)) & 1u, sp1 = p3[3] >> 31;
        return movemask16(g0 & kM, g1 & kM, g2 & kM, g3 & kM) | (sp0 << 16) | (sp1 << 17);
    }
};

// --- UTF-8: af = one range, TWO ranges of 2-byte leads (C2..DF), nothing longer: `-u Latin` (mission.rs:63-67: C2..C8 and CC..CD) ---------
// Utf8Range2 of sx_kernels.hip with one more compare for the lead byte (the second range travels in ScanParams::l3_lo / l3_hi).
struct Utf8Range2x2 {
    u32 a1, a2, l1, l2, m1, m2;
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
        a1 = rep4(0x80u - p.a_lo);
        a2 = rep4(0x7Fu - p.a_hi);
        l1 = rep4(0x80u - (p.u_lo & 0x7F));
        l2 = rep4(0x7Fu - (p.u_hi & 0x7F));
        m1 = rep4(0x80u - (p.l3_lo & 0x7F));
        m2 = rep4(0x7Fu - (p.l3_hi & 0x7F));
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 xs[5] = { x.x, x.y, x.z, x.w, nx };
        if (near_end) {
#pragma unroll
            for (int k = 0; k < 5; k++) xs[k] = fill_ff(xs[k], (int)avail - 4 * k);
        }
        u32 c[5];
#pragma unroll
        for (int k = 0; k < 5; k++) c[k] = xs[k] & ~(xs[k] << 1);   // (bit 7 of every byte; cleaned in front of the v_dot4s)
        u32 a[4], p[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 v = xs[k], t = v & 0x7F7F7F7Fu;
            a[k] = ((t + a1) & ~(t + a2)) & ~v;
            const u32 l = (((t + l1) & ~(t + l2)) | ((t + m1) & ~(t + m2))) & v;
            p[k] = l & __builtin_amdgcn_alignbyte(c[k + 1], c[k], 1);
        }
        if (WANT_S) return movemask16((a[0] | p[0]) & kM, (a[1] | p[1]) & kM, (a[2] | p[2]) & kM, (a[3] | p[3]) & kM);
        const u32 g0 = (a[0] | p[0] | (p[0] << 8)) & kM;
        const u32 g1 = (a[1] | p[1] | __builtin_amdgcn_alignbyte(p[1], p[0], 3)) & kM;
        const u32 g2 = (a[2] | p[2] | __builtin_amdgcn_alignbyte(p[2], p[1], 3)) & kM;
        const u32 g3 = (a[3] | p[3] | __builtin_amdgcn_alignbyte(p[3], p[2], 3)) & kM;
        return movemask16(g0, g1, g2, g3) | ((p[3] >> 31) << 16);
    }
};

// --- UTF-16LE/BE: the accepted units = NL ranges below U+8000, NS (0 / 1) ranges that straddle U+8000, NH ranges above it; no surrogate
// inside, no astral plane accepted (every surrogate is then a break, as in Utf16RangeT).  The ranges come from af (units below U+0080)
// and from ubf through the lead byte of the unit's UTF-8 form: C2..DF <-> U+0080..U+07FF in steps of 0x40, E0..EF <-> U+0800..U+FFFF
// in steps of 0x1000 (sx_mission.cpp builds them unit by unit from the Mission's own filter test): Cjk = 4E00-area leads E4..E9 =
// U+4000..U+9FFF, Asian = U+2000..U+D7FF, Hangul = U+B000..U+D7FF, Kana = U+3000..U+3FFF.
// Per range and pair of units, on the low 15 bits t of each unit: A = t + (0x8000 - lo15) has bit 15 set iff t >= lo15, B = (0x8000 +
// hi15) - t iff t <= hi15 (no carry or borrow leaves a 16-bit lane).  Below U+8000: A & B & ~unit; above: A & B & unit; a straddling
// range is "t >= lo15 where bit 15 is clear, t <= hi15 where it is set" = one v_bfi_b32 — three operations instead of the eight of two
// split ranges.  Slots in ScanParams::rng_c1 / rng_c2: 0, 1 the ranges below, 2 the straddling one, 3, 4 the ranges above; a third range below
// (`-u Latin`: af, U+0080..U+023F, U+0300..U+037F) takes slot 5, which is the high surrogates' otherwise (NL = 3 only without AST).
// AST: an astral plane passes the filter — a high surrogate whose plane does (one range of D800..DBFF, slot 5: the lead byte of the pair's
// UTF-8 form, F0..F4, follows from the high surrogate alone) followed by a low surrogate is a character of four bytes; every other
// surrogate stays a break (utf_16.rs: a lone or reversed surrogate is an error, the unit behind a lone high surrogate is read again).
template <int BE_T, int ODD_T, int NL, int NS, int NH, int AST = 0>
struct Utf16RangesT {
    u32 c1[6], c2[6];
    SX_DEV void init(const ScanParams& p, const uint8_t*) {
#pragma unroll
        for (int k = 0; k < 6; k++) { c1[k] = p.rng_c1[k]; c2[k] = p.rng_c2[k]; }
    }
    static SX_DEV u32 units(u32 v) { return BE_T ? __builtin_amdgcn_perm(0u, v, 0x02030001u) : v; }
    SX_DEV u32 bmp_flags(u32 v, u32 t) const {  // two units per dword -> flags at bits 15 and 31 (and garbage below them)
        u32 lo = 0, hi = 0, f = 0;
#pragma unroll
        for (int k = 0; k < NL; k++) lo |= (t + c1[k < 2 ? k : 5]) & (c2[k < 2 ? k : 5] - t);
#pragma unroll
        for (int k = 0; k < NH; k++) hi |= (t + c1[3 + k]) & (c2[3 + k] - t);
        if (NS) f = (v & (c2[2] - t)) | (~v & (t + c1[2]));   // v_bfi_b32
        if (NL) f |= lo & ~v;
        if (NH) f |= hi & v;
        return f;
    }
    template <bool WANT_S>
    SX_DEV u32 classify(u32x4 x, u32 nx, u32 avail, bool near_end) const {
        u32 d[5] = { x.x, x.y, x.z, x.w, nx };
        if (ODD_T) {  // unit k of this lane = bytes 2k+1, 2k+2
            d[0] = __builtin_amdgcn_alignbyte(x.y, x.x, 1);
            d[1] = __builtin_amdgcn_alignbyte(x.z, x.y, 1);
            d[2] = __builtin_amdgcn_alignbyte(x.w, x.z, 1);
            d[3] = __builtin_amdgcn_alignbyte(nx, x.w, 1);
            d[4] = nx >> 8;
        }
        const u32 nu = near_end ? (avail > (u32)ODD_T ? (avail - (u32)ODD_T) >> 1 : 0u) : 16u;   // whole units inside the chunk
        u32 f[4], pair[4] = { 0, 0, 0, 0 };
        if (AST) {
            constexpr u32 kLs1 = (0x8000u - 0x5C00u) * 0x00010001u, kLs2 = (0x8000u + 0x5FFFu) * 0x00010001u;   // DC00..DFFF
            u32 ls[5], hs[4];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const u32 v = units(d[k]), t = v & 0x7FFF7FFFu;
                ls[k] = (t + kLs1) & (kLs2 - t) & v;
                if (near_end)   // a unit that is not whole inside the chunk is no low surrogate (UTF-16BE: DC + the zero behind the end)
                    ls[k] &= ((u32)(2 * k) < nu ? 0x8000u : 0u) | ((u32)(2 * k + 1) < nu ? 0x80000000u : 0u);
                if (k < 4) { f[k] = bmp_flags(v, t); hs[k] = (t + c1[5]) & (c2[5] - t) & v; }
            }
#pragma unroll
            for (int k = 0; k < 4; k++) pair[k] = hs[k] & __builtin_amdgcn_alignbyte(ls[k + 1], ls[k], 2) & 0x80008000u;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { const u32 v = units(d[k]); f[k] = bmp_flags(v, v & 0x7FFF7FFFu); }
        }
        u32 b[4];   // byte flags: both bytes of a good unit (WANT_S: the first byte of a character's first unit)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32 u = (f[k] & 0x80008000u) | pair[k];
            if (AST && !WANT_S) u |= k ? __builtin_amdgcn_alignbyte(pair[k], pair[k - 1], 2) : pair[0] << 16;   // the pair's second unit
            b[k] = WANT_S ? u >> 8 : u | (u >> 8);
        }
        u32 m = movemask16(b[0], b[1], b[2], b[3]);
        if (AST && !WANT_S) m |= (pair[3] >> 31) * 0x30000u;   // a pair that begins in my last unit: the next lane's first unit
        if (near_end) m &= low_mask(2 * nu);   // whole units only
        return m << ODD_T;  // odd parity: everything sits one byte later (bit 16 spills)
    }
};
