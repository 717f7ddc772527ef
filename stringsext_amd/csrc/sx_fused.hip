// sx_fused.hip — stage A for several Missions in ONE pass over the buffer (round 6).
//
// The reference reads a slice once and hands the same bytes to every Mission (src/main.rs:153-168; src/input.rs:121-123).  Rounds 1-5
// launched one scan kernel per Mission, each fetching the whole buffer: three reads of 64 GiB for BASELINE's headline (`-e utf-8
// -e utf-16le -e utf-16be`).  Here one wavefront still streams one private sub-chunk in 1 KiB tiles through three rotating register
// sets (sx_kernels.hip), but every tile it has fetched is classified for up to kFusedMax Missions back to back: a classifier, a
// tile-to-tile carry and a record emitter per Mission ("slot"), each writing that Mission's own record regions, counters and
// statistics — what stage A's host side (sx_stage_a.cpp) reads per Mission is exactly what the per-Mission launches leave.
//
// With one read instead of three the kernel is bound by VALU issue, so the slots must be cheap.  The UTF-16 range classifiers get a
// PREFILTER that settles almost every tile of binary data in four vector instructions:
//
//   every accepted unit of Utf16RangeT lies below U+0800, so its high byte has no bit outside M (the smallest 2^k - 1 >= the top
//   unit's high byte; -u African: 7).  A stretch that can yield a record holds >= min_chars >= 7 units (cand_bytes == 14: the
//   prefilter is only switched on then).  The high bytes of 7 consecutive units are 7 consecutive even (or odd) byte positions, and
//   among any 7 of those four share one aligned 8-byte group.  So: a tile in which no aligned 8-byte group has all four high-byte
//   positions inside M — (d0 | d1) & ZZ != 0 for every pair of dwords — holds no aligned group of ANY stretch of >= 7 units.
//
//   Such a tile is skipped: no classification, nothing carried ("context unknown").  A stretch of >= 7 units has an aligned group in
//   at least one tile; that tile is classified in full (its entry context — lane 63 of the tile before — recomputed from memory if that
//   tile was skipped), sees the stretch end or leaves it open in its carry (g63 bit 15 / tracked), and an open carry forces the
//   next tile to be classified whatever its prefilter says.  By induction every tile from the one with the group to the one in which
//   the stretch ENDS is classified, and the tile of the end emits the record exactly as the per-Mission kernel does.  The first tile
//   of a sub-chunk is always looked at with a recomputed context (a stretch that crosses the sub-chunk start is reported in two
//   flagged parts, whatever its length), and so is the sub-chunk's end (the kRecEndOpen part).  On random bytes (8 / 256)^4 per
//   group = 1.2e-4 of the tiles pass: the two UTF-16 Missions of the headline cost 8 vector instructions per tile instead of 120.
//
// Records, flags and statistics are those of scan_kernel (same light / heavy paths, sx_scan_core.hpp); stage B never sees a difference.
#include "sx_scan_core.hpp"

namespace sx {

struct NoCls {};   // an empty slot

template <class CLS>
struct Prefilter {
    static constexpr bool kHas = false;
    u32 on = 0;
    SX_DEV void init(const ScanParams&) {}
    SX_DEV bool hit(u32x4) const { return true; }
};
template <int BE_T, int ODD_T>
struct Prefilter<Utf16RangeT<BE_T, ODD_T>> {
    static_assert(BE_T >= 0 && ODD_T >= 0, "the fused kernel knows byte order and parity at compile time");
    static constexpr bool kHas = true;
    u32 zz, on;
    SX_DEV void init(const ScanParams& p) {
        u32 top = 0;   // the highest accepted unit
        if (p.a_lo <= p.a_hi) top = p.a_hi;
        if (p.u_lo <= p.u_hi && p.u_hi > top) top = p.u_hi;
        u32 mh = top >> 8;
        mh |= mh >> 1; mh |= mh >> 2; mh |= mh >> 4;
        const u32 z = 0xFFu & ~mh;
        // where the units' high bytes sit in a raw dword: LE at even parity and BE at odd parity in bytes 1 and 3, else in bytes 0 and 2
        zz = (BE_T ^ ODD_T) ? z * 0x00010001u : z * 0x01000100u;
        on = (p.cand_bytes >= 14u && z != 0u) ? 1u : 0u;
    }
    SX_DEV bool hit(u32x4 x) const {
        const u32 q0 = (x.x | x.y) & zz, q1 = (x.z | x.w) & zz;
        return (q0 < q1 ? q0 : q1) == 0u;
    }
};

template <class CLS>
struct Slot {
    static constexpr bool kUsed = !std::is_same<CLS, NoCls>::value;
    using PF = Prefilter<CLS>;
    CLS cls;
    PF pf;
    Carry c;
    Emitter em;
    u32 known;   // c.g63 describes the tile before the current one (prefilter: not after a skipped tile)
};

template <class C0, class C1, class C2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6))) void scan_kernel_fused(const FusedParams fp) {
    const ScanParams& p = fp.m[0];   // data, len, subchunk: the same for every slot
    const u32 lane = lane_id();
    Slot<C0> s0; Slot<C1> s1; Slot<C2> s2;
    auto setup = [&](auto& S, const ScanParams& mp) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            S.cls.init(mp, nullptr);
            S.pf.init(mp);
            S.em = Emitter{ mp.recs, mp.counters, mp.capacity, 0u, 0u, mp.region_cap, 0u, mp.region_counts };
        }
    };
    setup(s0, fp.m[0]); setup(s1, fp.m[1]); setup(s2, fp.m[2]);

    const u64 wave = (u64)blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    const u64 sub_start = wave * (u64)p.subchunk;
    if (sub_start >= p.len) return;
    const u64 sub_end = (sub_start + p.subchunk < p.len) ? sub_start + p.subchunk : p.len;

    // the window [win_lo, win_hi) and its buffer descriptor: as in scan_kernel
    const bool has_pre = sub_start >= kTileBytes;
    const u64 win_lo = has_pre ? sub_start - kTileBytes : 0;
    u64 win_hi = sub_end + 2 * kTileBytes;
    if (win_hi > p.len) win_hi = p.len;
    const uint8_t* base_ptr = p.data + win_lo;
    const u32 base_lo = uniform((u32)(uintptr_t)base_ptr), base_hi = uniform((u32)((uintptr_t)base_ptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((u64)base_hi << 32) | base_lo), 0, (int)uniform(((u32)(win_hi - win_lo) + 15u) & ~15u), 0x00020000);
    auto load = [&](u32 off) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0); };
    struct TileRegs { u32x4 d; u32 e; };
    auto fetch = [&](u32 off) -> TileRegs {
        TileRegs r;
        const u32 a = off + lane * 16u;
        r.d = load(a);
        r.e = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(a + 16u), 0, 0);
        return r;
    };

    const int n_tiles = (int)((sub_end - sub_start + kTileBytes - 1) / kTileBytes);
    int n_safe = n_tiles;
    while (n_safe > 0 && sub_start + (u64)n_safe * kTileBytes + 16 > p.len) n_safe--;

    int t = has_pre ? -1 : 0;
    u32 toff = 0;
    TileRegs R0 = fetch(0u), R1 = fetch(kTileBytes), R2;

    auto begin = [&](auto& S) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            S.em.begin_region(wave);
            S.c.g63 = 0; S.c.tracked = 0; S.c.t_chars = 0; S.c.t_flags = 0; S.c.t_start = 0;
            S.known = 1;   // (nothing lies before window offset 0)
        }
    };
    begin(s0); begin(s1); begin(s2);

    // bytes of the chunk from the lane's first byte on, for a tile that begins at chunk offset `tile_base` (<= 32: all a classifier looks at)
    auto avail_at = [&](u64 tile_base) -> u32 {
        const u64 b = tile_base + 16ull * lane;
        return b >= p.len ? 0u : (p.len - b > 32 ? 32u : (u32)(p.len - b));
    };
    // start mask of the 16 bytes right before the tile at window offset `off` (scan_kernel's starts_before)
    auto starts_before = [&](auto& S, u32 off, u64 tile_base) -> u32 {
        if (off < 16u) return 0u;
        asm volatile("" : "+s"(off));
        const u32x4 x = load(off - 16u);
        const u32 nx = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
        const u64 b = tile_base - 16;
        const u32 avail = b >= p.len ? 0u : (p.len - b > 32 ? 32u : (u32)(p.len - b));
        return uniform(S.cls.template classify<true>(x, nx, avail, true));
    };
    // the carry word of the tile before the one at window offset `off` (g63: lane 63's final good mask | its spill bits << 16),
    // recomputed from memory: that tile was skipped by the prefilter.  0 if the window holds no such tile (chunk start).
    auto context_before = [&](auto& S, u32 off, u64 tile_base) -> u32 {
        if (off < kTileBytes) return 0u;
        u32 o = off - kTileBytes;
        asm volatile("" : "+s"(o));
        const TileRegs P = fetch(o);
        const u32 g = S.cls.template classify<false>(P.d, P.e, avail_at(tile_base - kTileBytes), true);
        const u32 pg = from_prev(g, 0u);
        const u32 gf = (g & 0xFFFFu) | (pg >> 16);
        return bcast(gf | (g & 0xFFFF0000u), 63);
    };

    // one slot's share of tile t (scan_kernel's body)
    auto step = [&](auto& S, const ScanParams& mp, auto near_tag, const TileRegs& X, u64 tile_base, u32 avail) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            constexpr bool NE = decltype(near_tag)::value;
            if constexpr (ST::PF::kHas) {
                if (S.pf.on) {
                    if (t < 0) { S.known = 0; return; }   // (the look-back tile only yields a context: recomputed when somebody wants it)
                    const bool any = __ballot(S.pf.hit(X.d)) != 0;
                    const bool must = any || S.c.tracked || (S.known ? (S.c.g63 & 0x8000u) != 0u : t == 0);
                    if (!must) { S.known = 0; return; }
                    if (!S.known) S.c.g63 = context_before(S, toff, tile_base);   // (tracked implies known)
                }
            }
            const u32x4 cur = X.d;
            const u64 lane_base = tile_base + 16ull * lane;
            const u32 g63_in = S.c.g63;
            const bool tracked_in = S.c.tracked != 0;
            const u32 g = S.cls.template classify<false>(cur, X.e, avail, NE);
            const u32 pg = from_prev(g, g63_in);
            const u32 gf = (g & 0xFFFFu) | (pg >> 16);
            const u32 g63_out = bcast(gf | (g & 0xFFFF0000u), 63);
            u32 w = __builtin_amdgcn_perm(gf, pg, 0x05040100u);
            u32 r = w;
            r &= r << mp.cand_sh[0]; r &= r << mp.cand_sh[1]; r &= r << mp.cand_sh[2]; r &= r << mp.cand_sh[3];
            const bool any_cand = __ballot((r & 0xFFFF0000u) != 0) != 0;
            const bool first_tile = t == 0;
            const bool first_open = first_tile && (g63_in & 0x8000u);
            S.known = 1;
            if (t < 0 || (!any_cand && !tracked_in && !first_open)) {
                S.c.g63 = g63_out;
            } else {
                const u32 s = S.cls.template classify<true>(cur, X.e, avail, NE);
                w = (gf << 16) | (from_prev(gf, g63_in) & 0xFFFFu);
                const u32 s63 = (g63_in & 0x8000u) ? starts_before(S, toff, tile_base) : 0u;
                const u32 sw = (s << 16) | (from_prev(s, s63) & 0xFFFFu);
                bool done = false;
                if (!tracked_in && !first_open) done = light_path(w, sw, r, lane_base, S.em, mp.min_chars);
                if (done) S.c.g63 = g63_out;
                else {
                    S.em.heavy_n++;
                    const u64 tile_end = tile_base + kTileBytes < sub_end ? tile_base + kTileBytes : sub_end;
                    heavy_path(gf, s, g, g63_in, s63, r >> 16, tile_base, tile_end, S.c, S.em, mp.min_chars, mp.cand_bytes, first_tile);
                }
            }
        }
    };

    auto body = [&](auto near_tag, const TileRegs& X, TileRegs& Z) {
        constexpr bool NE = decltype(near_tag)::value;
        Z = fetch(toff + 2 * kTileBytes);
        const u64 tile_base = sub_start + (u64)((long long)t * (long long)kTileBytes);
        u32 avail = 32;
        if (NE) avail = avail_at(tile_base);
        step(s0, fp.m[0], near_tag, X, tile_base, avail);
        step(s1, fp.m[1], near_tag, X, tile_base, avail);
        step(s2, fp.m[2], near_tag, X, tile_base, avail);
        toff += kTileBytes; t++;
    };

    while (t + 3 <= n_safe) {
        body(std::false_type{}, R0, R2);
        body(std::false_type{}, R1, R0);
        body(std::false_type{}, R2, R1);
    }
    while (t < n_tiles) { body(std::true_type{}, R0, R2); R0 = R1; R1 = R2; }

    // the stretch that is still open where the sub-chunk ends
    auto finish = [&](auto& S) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            const u64 after = sub_start + (u64)n_tiles * kTileBytes;
            if (!S.known) S.c.g63 = context_before(S, toff, after);
            if (S.c.tracked || (S.c.g63 & 0x8000u)) {
                u64 os; u32 och, ofl;
                if (S.c.tracked) { os = S.c.t_start; och = S.c.t_chars; ofl = S.c.t_flags; }
                else {
                    const u32 suf = trailing_ones16(S.c.g63 & 0xFFFFu);
                    const u32 s63 = starts_before(S, toff, after);
                    os = after - suf;
                    och = (u32)__popc((s63 & 0xFFFFu) >> (16u - suf));
                    ofl = 0;
                }
                S.em.append(lane == 0, os, sub_end, och, ofl | kRecEndOpen);
            }
            S.em.end_region(wave);
            S.em.invalidate_rest();
        }
    };
    finish(s0); finish(s1); finish(s2);
}

template <class C0, class C1, class C2>
static hipError_t launch_f(const FusedParams& fp, hipStream_t stream) {
    const ScanParams& p = fp.m[0];
    const u64 waves = (p.len + p.subchunk - 1) / p.subchunk;
    const u64 blocks = (waves + 3) / 4;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL((scan_kernel_fused<C0, C1, C2>), dim3((unsigned)blocks), dim3(256), 0, stream, fp);
    return hipGetLastError();
}

// Which slot a Mission's classifier can take in the fused kernel: 0 UTF-8 (Utf8Range2), 1 UTF-16LE, 2 UTF-16BE (Utf16RangeT); -1: none
// (the Mission keeps its own launch).
int fused_slot_of(ClassifierKind kind, const ScanParams& p) {
    if (kind == kClsUtf8Range2) return 0;
    if (kind == kClsUtf16Range) return p.big_endian ? 2 : 1;
    return -1;
}

// used: bit s set = slot s holds a Mission (fp.m[s]); the others are ignored.  All used slots share data, len and subchunk.
hipError_t launch_scan_fused(const FusedParams& fp, uint32_t used, hipStream_t stream) {
    u32 parity = 0;
    if (used & 2u) parity = fp.m[1].parity & 1u; else if (used & 4u) parity = fp.m[2].parity & 1u;
    FusedParams q = fp;
    {   // slot 0's geometry is what the kernel reads
        const int s = (used & 1u) ? 0 : ((used & 2u) ? 1 : 2);
        q.m[0].data = fp.m[s].data; q.m[0].len = fp.m[s].len; q.m[0].subchunk = fp.m[s].subchunk;
    }
#define SX_ARG(...) __VA_ARGS__
#define SX_F(U, P, A, B, C) if (used == U && parity == P) return launch_f<A, B, C>(q, stream);
    SX_F(7u, 0u, Utf8Range2, SX_ARG(Utf16RangeT<0, 0>), SX_ARG(Utf16RangeT<1, 0>))
    SX_F(7u, 1u, Utf8Range2, SX_ARG(Utf16RangeT<0, 1>), SX_ARG(Utf16RangeT<1, 1>))
    SX_F(3u, 0u, Utf8Range2, SX_ARG(Utf16RangeT<0, 0>), NoCls)
    SX_F(3u, 1u, Utf8Range2, SX_ARG(Utf16RangeT<0, 1>), NoCls)
    SX_F(5u, 0u, Utf8Range2, NoCls, SX_ARG(Utf16RangeT<1, 0>))
    SX_F(5u, 1u, Utf8Range2, NoCls, SX_ARG(Utf16RangeT<1, 1>))
    SX_F(6u, 0u, NoCls, SX_ARG(Utf16RangeT<0, 0>), SX_ARG(Utf16RangeT<1, 0>))
    SX_F(6u, 1u, NoCls, SX_ARG(Utf16RangeT<0, 1>), SX_ARG(Utf16RangeT<1, 1>))
#undef SX_F
#undef SX_ARG
    return hipErrorInvalidValue;
}

}  // namespace sx
