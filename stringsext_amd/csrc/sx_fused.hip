// sx_fused.hip — stage A for several Missions in ONE pass over the buffer (round 6).
//
// The reference reads a slice once and hands the same bytes to every Mission (src/main.rs:153-168; src/input.rs:121-123).  Rounds 1-5
// launched one scan kernel per Mission, each fetching the whole buffer: three reads of 64 GiB for BASELINE's headline (`-e utf-8
// -e utf-16le -e utf-16be`).  Here one wavefront still streams one private sub-chunk in 1 KiB tiles through three rotating register
// sets (sx_kernels.hip), but every tile it has fetched is classified for up to kFusedMax Missions back to back: a classifier, a
// tile-to-tile carry and a record emitter per Mission ("slot"), each writing that Mission's own record regions, counters and
// statistics — what stage A's host side (sx_stage_a.cpp) reads per Mission is exactly what the per-Mission launches leave.
//
// With one read instead of three the kernel is bound by VALU issue (a vector instruction takes a SIMD four cycles: 1024 SIMDs at
// ~2.2 GHz issue 560 G of them per second, the 67 M tiles of 64 GiB want to be through in 11 ms = 92 per tile), so:
//
// (1) A FAST LOOP that only knows the fast path.  Three tiles per trip (the three register sets rotate without moves); per slot a
//     handful of instructions decide "nothing in this tile can yield a record"; a tile in which some slot is not sure goes through
//     generic_tile for those slots — everything else: exact windows, start masks, light / heavy path —, out of whichever register set
//     holds it (no move touches a set with a load in flight); the look-back tile, tile 0 of the sub-chunk and the tiles near the end
//     of the input go through generic_tile one at a time.  What is live across the fast loop is small: per slot the carry, two counters and — instead of the parameters of the slow
//     paths (record pointers, thresholds, the UTF-16 classifiers' constants) — nothing: those are read from the kernel-argument
//     segment where they are used (late_params).
// (2) FULL slots (UTF-8): classify, one DPP shift whose lane 0 takes the previous tile's lane 63 out of a register that the tile
//     before prepared with a wave rotate (no v_readlane / v_mov per tile), a 7-operation candidate test (two 2-input steps and one
//     3-input step of the shift-AND chain: reach 12 bytes).
// (3) PREFILTER slots (the UTF-16 range classifiers with min_chars >= 7) settle almost every tile of binary data in four vector
//     instructions:
//
//   every accepted unit of Utf16RangeT lies below U+0800, so its high byte has no bit outside M (the smallest 2^k - 1 >= the top
//   unit's high byte; -u African: 7).  A stretch that can yield a record holds >= min_chars >= 7 units.  The high bytes of 7
//   consecutive units are 7 consecutive even (or odd) byte positions, and among any 7 of those four share one aligned 8-byte group.
//   So: a tile in which no aligned 8-byte group has all four high-byte positions inside M — (d0 | d1) & ZZ != 0 for every pair of
//   dwords — holds no aligned group of ANY stretch of >= 7 units.
//
//   Such a tile is skipped: no classification, nothing carried ("context unknown").  A stretch of >= 7 units has an aligned group in
//   at least one tile; that tile is classified in full (its entry context — lane 63 of the tile before — recomputed from memory if that
//   tile was skipped), sees the stretch end or leaves it open in its carry (g63 bit 15 / tracked), and an open carry forces the
//   next tile to be classified whatever its prefilter says.  By induction every tile from the one with the group to the one in which
//   the stretch ENDS is classified, and the tile of the end emits the record exactly as the per-Mission kernel does.  The first tile
//   of a sub-chunk is always classified (a stretch that crosses the sub-chunk start is reported in two flagged parts, whatever its
//   length), and the sub-chunk's end recomputes the context it needs for the kRecEndOpen part.  On random bytes (8 / 256)^4 per
//   group = 1.2e-4 of the tiles pass.
//
//   For 3 <= min_chars <= 6 (the reference's default -n 4) the same argument holds with PAIRS: the high bytes of 3 consecutive units
//   cover one aligned dword's two high-byte positions; 22 % of the tiles of random bytes pass.  There the skipped tile's carry word
//   usually needs no recomputing: if the last byte of that tile cannot be good — the high byte of the unit it belongs to is outside M,
//   read from the register set that still holds that tile — the word is 0.  (Kernel alone at -n 7, 64 GiB: no prefilter 25.0 ms, pairs
//   21.2, groups of four 17.7.)
//
// Records, flags and statistics are those of scan_kernel (same light / heavy paths, sx_scan_core.hpp); stage B never sees a difference.
#include "sx_scan_core.hpp"

namespace sx {

struct NoCls {};   // an empty slot

// lane i <- lane i-1 of v; lane 0 keeps old's lane 0
SX_DEV u32 shr1_keep(u32 v, u32 old) { return __builtin_amdgcn_update_dpp(old, v, 0x138, 0xF, 0xF, false); }
// lane i <- lane i-1; lane 0 <- lane 63 (wave_ror:1)
SX_DEV u32 ror1(u32 v) { return __builtin_amdgcn_mov_dpp(v, 0x13C, 0xF, 0xF, false); }   // (every lane has a source: nothing of the target is kept)

// Parameters of slot `slot`, read from the kernel-argument segment at the point of use (FusedParams is the kernel's only argument and
// m[] its first member, so m[slot] sits at slot * sizeof(ScanParams)); the empty asm keeps the compiler from loading them once at the
// kernel's top and carrying — i.e. spilling — them through the fast loop.
static_assert(offsetof(FusedParams, m) == 0, "late_params addresses m[slot] from the start of the kernel-argument segment");
typedef const __attribute__((address_space(4))) ScanParams* KArgMission;
SX_DEV KArgMission late_params(u32 slot) {
    asm volatile("" : "+s"(slot));
    return (KArgMission)__builtin_amdgcn_kernarg_segment_ptr() + slot;
}

// Region-mode record emitter whose pointers are late parameters (the fused kernel is only launched for Missions in region mode).
struct LateEmitter {
    u32 slot, wave;
    u32 rcount, heavy_n;
    SX_DEV void append(bool want, u64 start, u64 end, u32 chars, u32 flags) {
        const u64 m = __ballot(want);
        if (m == 0) return;
        const KArgMission mp = late_params(slot);
        const u32 cap = mp->region_cap;
        DevRun* const recs = mp->recs;
        const u32 k = rcount + (u32)__popcll(m & ((1ull << lane_id()) - 1ull));
        rcount += (u32)__popcll(m);
        if (want && k < cap) {
            DevRun r;
            r.start = start;
            r.len = (u32)(end - start);
            r.chars_flags = (chars > kRecCharsMask ? kRecCharsMask : chars) | flags;
            recs[wave * cap + k] = r;
        }
    }
    // (Emitter::end_region, region mode)
    SX_DEV void end_region() {
        const KArgMission mp = late_params(slot);
        const u32 cap = mp->region_cap;
        u32* const counters = mp->counters;
        u32* const shard = counters + kStatBase + (wave & (kStatShards - 1u)) * kStatStride;
        if (lane_id() == 0) {
            if (heavy_n) atomicAdd(shard, heavy_n);
            mp->region_counts[wave] = rcount < cap ? rcount : cap;
            if (rcount) atomicAdd(shard + 1, rcount);
            if (rcount > cap) {
                atomicAdd(counters, rcount - cap);
                atomicMax(counters + 3, rcount);
            }
        }
    }
};

template <class CLS>
struct Prefilter {
    static constexpr bool kHas = false;
    SX_DEV void init(u32) {}
    template <int MODE> SX_DEV bool hit(u32x4) const { return true; }
    SX_DEV bool last_unit_may_pass(u32, u32) const { return true; }
    static int mode_for(const ScanParams&) { return 0; }
    static u32 zero_bits(const ScanParams&) { return 0; }
};
template <int BE_T, int ODD_T>
struct Prefilter<Utf16RangeT<BE_T, ODD_T>> {
    static_assert(BE_T >= 0 && ODD_T >= 0, "the fused kernel knows byte order and parity at compile time");
    static constexpr bool kHas = true;
    u32 zz;
    static u32 zero_bits(const ScanParams& p) {   // (host) bits that are clear in the high byte of every accepted unit
        u32 top = 0;   // the highest accepted unit
        if (p.a_lo <= p.a_hi) top = p.a_hi;
        if (p.u_lo <= p.u_hi && p.u_hi > top) top = p.u_hi;
        u32 mh = top >> 8;
        mh |= mh >> 1; mh |= mh >> 2; mh |= mh >> 4;
        return 0xFFu & ~mh;
    }
    // where the units' high bytes sit in a raw dword: LE at even parity and BE at odd parity in bytes 1 and 3, else in bytes 0 and 2
    SX_DEV void init(u32 z) { zz = (BE_T ^ ODD_T) ? z * 0x00010001u : z * 0x01000100u; }
    // (host) which prefilter this Mission allows: 1 = a stretch that can yield a record holds >= 7 units (aligned groups of FOUR high
    // bytes), 2 = >= 3 units (aligned PAIRS: the high bytes of 3 consecutive units are 3 consecutive even — or odd — positions, two
    // of which share an aligned dword; the reference's default -n 4), 0 = none (fewer units, or the mask says nothing)
    static int mode_for(const ScanParams& p) { return zero_bits(p) == 0u ? 0 : p.min_chars >= 7u ? 1 : p.min_chars >= 3u ? 2 : 0; }
    template <int MODE> SX_DEV bool hit(u32x4 x) const {
        if (MODE == 1) {
            const u32 q0 = __builtin_amdgcn_bitop3_b32(x.x, zz, x.y, 0xC8), q1 = __builtin_amdgcn_bitop3_b32(x.z, zz, x.w, 0xC8);   // (a | c) & b
            return (q0 < q1 ? q0 : q1) == 0u;
        }
        const u32 a = x.x & zz, b = x.y & zz, c = x.z & zz, d = x.w & zz;
        const u32 ab = a < b ? a : b, cd = c < d ? c : d;
        return (ab < cd ? ab : cd) == 0u;
    }
    // Can the LAST byte of the tile in front of a tile be good at all?  It belongs to the unit whose high byte is the tile's byte 1023
    // (LE at even parity, BE at odd), its byte 1022 (BE, even) or the next tile's byte 0 (LE, odd).  prev_w: the last dword of that tile
    // (lane 63's x.w), cur_x: the first dword of the tile behind it (lane 0's x.x).  False: nothing is open at the tile's end and
    // nothing of it spills over — the carry word is 0, no classification of that tile is needed to know it.
    SX_DEV bool last_unit_may_pass(u32 prev_w, u32 cur_x) const {
        const u32 z8 = (zz | (zz >> 8)) & 0xFFu;
        const u32 hi = BE_T == ODD_T ? prev_w >> 24 : (BE_T ? prev_w >> 16 : cur_x);
        return (hi & z8) == 0u;
    }
};

template <class T, class = void> struct has_classify_g : std::false_type {};
template <class T> struct has_classify_g<T, std::void_t<decltype(std::declval<const T&>().classify_g(u32x4{}, 0u))>> : std::true_type {};
template <class CLS>
SX_DEV u32 classify_fast(const CLS& c, u32x4 x, u32 nx) {
    if constexpr (has_classify_g<CLS>::value) return c.classify_g(x, nx);
    else return c.template classify<false>(x, nx, 32u, false);
}

template <class T, class = void> struct has_starts_from_good : std::false_type {};
template <class T> struct has_starts_from_good<T, std::void_t<decltype(T::kStartsFromGood)>> : std::true_type {};

constexpr u32 kUnknown = 0x80000000u;   // c.g63 of a prefilter slot: the tile in front was skipped (bit 15 — "open" — is clear, like every bit a test looks at)

template <class CLS, int PFM, int SLOT>
struct Slot {
    static constexpr bool kUsed = !std::is_same<CLS, NoCls>::value;
    static constexpr u32 kSlot = SLOT;
    using PF = Prefilter<CLS>;
    static constexpr bool kPf = PFM != 0 && PF::kHas;
    static constexpr int kPfMode = PFM;
    CLS cls;         // FULL slots: the classifier's constants stay in SGPRs; prefilter slots build theirs where a tile is classified
    PF pf;
    Carry c;         // (FULL slots: c.g63 is only valid inside generic_tile; between tiles lane 0 of E holds it)
    LateEmitter em;
    // (prefilter slots: after a skipped tile the carry word is not known — c.g63 == kUnknown, a value no carry word takes: one state
    //  word instead of two keeps it in an SGPR through the fast loop)
    u32 E;           // FULL slots: lane 0 = the previous tile's lane 63 (final good mask | its spill bits << 16)
    u32 gtmp, pgtmp, rtmp; // FULL slots: what the fast loop knows of the tile it hands to generic_tile: the classification, the same one lane up (lane 0: the carry), the candidate test
    u32 sh0, sh1, sh2, sh3;   // the candidate test's shifts
};

// PFM: the prefilter of the slots that have one (Prefilter::mode_for; the host picks the weakest any such Mission allows: 0 none, 1 groups
// of four, 2 pairs)
template <class C0, class C1, class C2, int PFM>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PFM ? 6 : 5))) void scan_kernel_fused(const FusedParams fp) {
    const u32 lane = lane_id();
    Slot<C0, PFM, 0> s0; Slot<C1, PFM, 1> s1; Slot<C2, PFM, 2> s2;

    const u32 wave = blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    const u64 len = fp.len;
    const u64 sub_start = (u64)wave * (u64)fp.subchunk;
    if (sub_start >= len) return;
    const u64 sub_end = (sub_start + fp.subchunk < len) ? sub_start + fp.subchunk : len;

    auto make_cls = [&](auto& cls, u32 slot) {   // a classifier from the late parameters
        using CLS = std::decay_t<decltype(cls)>;
        if constexpr (!std::is_same<CLS, NoCls>::value) {
            const KArgMission mp = late_params(slot);
            ScanParams q;
            q.a_lo = mp->a_lo; q.a_hi = mp->a_hi; q.u_lo = mp->u_lo; q.u_hi = mp->u_hi; q.l3_lo = mp->l3_lo; q.l3_hi = mp->l3_hi;
            q.parity = mp->parity; q.big_endian = mp->big_endian; q.high_all = mp->high_all;
            cls.init(q, nullptr);
        }
    };
    auto setup = [&](auto& S) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            constexpr u32 slot = ST::kSlot;
            if constexpr (ST::kPf) S.pf.init(fp.pf_zero[slot]);
            else make_cls(S.cls, slot);
            S.sh0 = fp.fsh[slot][0]; S.sh1 = fp.fsh[slot][1]; S.sh2 = fp.fsh[slot][2]; S.sh3 = fp.fsh[slot][3];
            S.em.slot = slot; S.em.wave = wave; S.em.rcount = 0; S.em.heavy_n = 0;
            S.c.g63 = 0; S.c.tracked = 0; S.c.t_chars = 0; S.c.t_flags = 0; S.c.t_start = 0;
            S.E = 0; S.gtmp = 0; S.pgtmp = 0; S.rtmp = 0;
        }
    };
    setup(s0); setup(s1); setup(s2);

    // the window [win_lo, win_hi) and its buffer descriptor: as in scan_kernel
    const bool has_pre = sub_start >= kTileBytes;
    const u64 win_lo = has_pre ? sub_start - kTileBytes : 0;
    u64 win_hi = sub_end + 2 * kTileBytes;
    if (win_hi > len) win_hi = len;
    const uint8_t* base_ptr = fp.data + win_lo;
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
    while (n_safe > 0 && sub_start + (u64)n_safe * kTileBytes + 16 > len) n_safe--;

    int t = has_pre ? -1 : 0;
    u32 toff = 0;
    TileRegs R0 = fetch(0u), R1 = fetch(kTileBytes), R2;

    // bytes of the chunk from the lane's first byte on, for a tile that begins at chunk offset `tile_base` (<= 32: all a classifier looks at)
    auto avail_at = [&](u64 tile_base) -> u32 {
        const u64 b = tile_base + 16ull * lane;
        return b >= len ? 0u : (len - b > 32 ? 32u : (u32)(len - b));
    };
    // start mask of the 16 bytes right before the tile at window offset `off` (scan_kernel's starts_before)
    auto starts_before = [&](const auto& cls, u32 off, u64 tile_base) -> u32 {
        if (off < 16u) return 0u;
        asm volatile("" : "+s"(off));
        const u32x4 x = load(off - 16u);
        const u32 nx = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
        const u64 b = tile_base - 16;
        const u32 avail = b >= len ? 0u : (len - b > 32 ? 32u : (u32)(len - b));
        return uniform(cls.template classify<true>(x, nx, avail, true));
    };
    // the carry word of the tile before the one at window offset `off` (lane 63's final good mask | its spill bits << 16),
    // recomputed from memory: that tile was skipped by the prefilter.  0 if the window holds no such tile (chunk start).
    auto context_before = [&](const auto& cls, u32 off, u64 tile_base) -> u32 {
        if (off < kTileBytes) return 0u;
        u32 o = off - kTileBytes;
        asm volatile("" : "+s"(o));
        const TileRegs P = fetch(o);
        const u32 g = cls.template classify<false>(P.d, P.e, avail_at(tile_base - kTileBytes), true);
        const u32 pg = from_prev(g, 0u);
        const u32 gf = (g & 0xFFFFu) | (pg >> 16);
        return bcast(gf | (g & 0xFFFF0000u), 63);
    };

    // ---- everything but the fast path: one slot's share of tile t, classified here or (have_g) by the fast loop ----
    auto generic_slot = [&](auto& S, const TileRegs& X, u64 tile_base, u32 avail, bool ne, bool have_g) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            constexpr u32 slot = ST::kSlot;
            using CLS = std::decay_t<decltype(S.cls)>;
            CLS local;
            const CLS* cls = &S.cls;
            if constexpr (ST::kPf) {
                // (unknown context: c.g63 is kept at 0 and nothing is tracked)
                const u32 open = S.c.tracked | (S.c.g63 & 0x8000u) | ((S.c.g63 == kUnknown && t == 0) ? 1u : 0u);
                if (t < 0 || (__ballot(S.pf.template hit<ST::kPfMode>(X.d)) == 0 && open == 0u)) { S.c.g63 = kUnknown; return; }
                make_cls(local, slot);
                cls = &local;
                if (S.c.g63 == kUnknown) S.c.g63 = context_before(local, toff, tile_base);   // (tracked implies known)
            } else {
                S.c.g63 = bcast(have_g ? S.pgtmp : S.E, 0);   // (the fast loop has replaced E already)
            }
            const KArgMission mp = late_params(slot);
            const u32x4 cur = X.d;
            const u32 g63_in = S.c.g63;
            const bool tracked_in = S.c.tracked != 0;
            const bool fast_g = have_g && !ST::kPf;   // (the fast loop classified this tile: it is not near the end of the input)
            const u32 g = fast_g ? S.gtmp : cls->template classify<false>(cur, X.e, avail, ne);
            const u32 pg = fast_g ? S.pgtmp : from_prev(g, g63_in);
            const u32 gf = (g & 0xFFFFu) | (pg >> 16);
            const u32 g63_out = bcast(gf | (g & 0xFFFF0000u), 63);
            u32 w = (g << 16) | pg;
            u32 r = w;
            if (fast_g) r = S.rtmp;
            else { r &= r << S.sh0; r &= r << S.sh1; r = r & (r << S.sh2) & (r << S.sh3); }
            const bool any_cand = __ballot((r & 0xFFFF0000u) != 0) != 0;
            const bool first_tile = t == 0;
            const bool first_open = first_tile && (g63_in & 0x8000u);
            if (t < 0 || (!any_cand && !tracked_in && !first_open)) {
                S.c.g63 = g63_out;
            } else {
                const u32 min_chars = mp->min_chars;
                const u64 lane_base = tile_base + 16ull * lane;
                u32 s;
                if constexpr (has_starts_from_good<CLS>::value) s = ne ? cls->template classify<true>(cur, X.e, avail, true) : cls->starts_from_good(cur, gf);
                else s = cls->template classify<true>(cur, X.e, avail, ne);
                w = (gf << 16) | (from_prev(gf, g63_in) & 0xFFFFu);
                const u32 s63 = (g63_in & 0x8000u) ? starts_before(*cls, toff, tile_base) : 0u;
                const u32 sw = (s << 16) | (from_prev(s, s63) & 0xFFFFu);
                bool done = false;
                if (!tracked_in && !first_open) done = light_path(w, sw, r, lane_base, S.em, min_chars);
                if (done) S.c.g63 = g63_out;
                else {
                    S.em.heavy_n++;
                    const u64 tile_end = tile_base + kTileBytes < sub_end ? tile_base + kTileBytes : sub_end;
                    heavy_path(gf, s, g, g63_in, s63, r >> 16, tile_base, tile_end, S.c, S.em, min_chars, fp.cand_f[slot], first_tile);
                }
            }
            if constexpr (!ST::kPf) S.E = S.c.g63;
        }
    };
    // tile t out of X for the slots of `which`
    auto generic_tile = [&](const TileRegs& X, u32 which, bool have_g) {
        const u64 tile_base = sub_start + (u64)((long long)t * (long long)kTileBytes);
        const bool ne = t >= n_safe;
        const u32 avail = ne ? avail_at(tile_base) : 32u;
        if (which & 1u) generic_slot(s0, X, tile_base, avail, ne, have_g);
        if (which & 2u) generic_slot(s1, X, tile_base, avail, ne, have_g);
        if (which & 4u) generic_slot(s2, X, tile_base, avail, ne, have_g);
        toff += kTileBytes; t++;
    };

    // ---- the fast path of one slot: true = the tile is settled for this slot ----
    // (Zp: the register set that still holds the tile in front of X — zprev —, or anything)
    auto fast_slot = [&](auto& S, const TileRegs& X, const TileRegs& Zp, bool zprev) -> bool {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (!ST::kUsed) return true;
        else if constexpr (ST::kPf) {
            const u32 open = S.c.tracked | (S.c.g63 & 0x8000u);   // (unknown context: both are 0)
            if ((__ballot(S.pf.template hit<ST::kPfMode>(X.d)) | (u64)open) != 0) {
                // The tile will be classified.  If the one in front was skipped, its carry word is unknown: 0 for certain if its last
                // byte cannot be good (no classification, no load), else generic_tile recomputes it from memory.
                if (S.c.g63 == kUnknown && zprev && !S.pf.last_unit_may_pass(bcast(Zp.d.w, 63), bcast(X.d.x, 0))) S.c.g63 = 0;
                return false;
            }
            S.c.g63 = kUnknown;
            return true;
        } else {
            const u32 g = classify_fast(S.cls, X.d, X.e);
            const u32 pg = shr1_keep(g, S.E);
            // the window: my 16 bits above the previous lane's 16 — and what the previous lane's last character spills onto my first
            // bytes sits in pg's bits 16.. already, which ARE the window's bits 16..: one v_lshl_or_b32 (my own spill bits shift out)
            const u32 w = (g << 16) | pg;
            u32 r = w;
            r &= r << S.sh0; r &= r << S.sh1; r = r & (r << S.sh2) & (r << S.sh3);
            S.gtmp = g; S.pgtmp = pg; S.rtmp = r;   // (names, not moves: generic_tile reads them if the tile goes there)
            // Lane 0 of E: this tile's lane 63 as the next tile's lane 0 wants it — the RAW mask (its bits 0..2 lack what lane 62's last
            // character spills onto them), as every other lane gets its neighbour's.  The fast test never looks below bit 3 of the
            // window's lower half; generic_tile could only care if the stretch that is open at the tile start reached down to those
            // bits, i.e. held >= 13 bytes of the tile before: >= cand_f (<= 12, launch_scan_fused), so that tile left it TRACKED and
            // the heavy path takes its start from the carry, not from this mask.
            S.E = ror1(g);
            return (__ballot((r & 0xFFFF0000u) != 0) | (u64)S.c.tracked) == 0;
        }
    };
    // tile t out of X, Z <- tile t + 2.  0: settled; else the slots that want generic_tile (the others are done with the tile)
    auto fast_tile = [&](const TileRegs& X, TileRegs& Z, bool zprev) -> u32 {
        u32 zero = 0, one = 1, two = 2;
        asm volatile("" : "+s"(zero), "+s"(one), "+s"(two));   // (constants the compiler cannot see through: `slow` stays a scalar select, not a v_cndmask + v_readfirstlane of a bool)
        // the prefilter slots first: Z still holds the tile in front of X
        u32 slow = fast_slot(s1, X, Z, zprev) ? zero : two;
        if (!fast_slot(s2, X, Z, zprev)) slow |= 4u;
        Z = fetch(toff + 2 * kTileBytes);
        slow |= fast_slot(s0, X, Z, false) ? zero : one;
        if (slow == 0) { toff += kTileBytes; t++; }
        return slow;
    };
    constexpr u32 kAll = (Slot<C0, PFM, 0>::kUsed ? 1u : 0u) | (Slot<C1, PFM, 1>::kUsed ? 2u : 0u) | (Slot<C2, PFM, 2>::kUsed ? 4u : 0u);

    for (;;) {
        // one tile at a time, the register sets rotating by moves: the look-back tile (classification state only), tile 0 (a stretch
        // that is open on entry is flagged) and the tiles near the end of the input
        while (t < n_tiles && (t <= 0 || t + 3 > n_safe)) {
            R2 = fetch(toff + 2 * kTileBytes);
            generic_tile(R0, kAll, false);
            R0 = R1; R1 = R2;
        }
        if (t >= n_tiles) break;
        // the fast loop: three tiles per trip, no register moves; a tile that some slot is not sure of goes through generic_tile
        // for those slots, out of the register set it is in
        bool zprev = false;   // (at the loop's entry R2 is a copy of R1, not the tile in front of R0)
        do {
            u32 slow;
            if ((slow = fast_tile(R0, R2, zprev)) != 0) generic_tile(R0, slow, true);
            if ((slow = fast_tile(R1, R0, true)) != 0) generic_tile(R1, slow, true);
            if ((slow = fast_tile(R2, R1, true)) != 0) generic_tile(R2, slow, true);
            zprev = true;
        } while (t + 3 <= n_safe);
    }

    // the stretch that is still open where the sub-chunk ends
    auto finish = [&](auto& S) {
        using ST = std::decay_t<decltype(S)>;
        if constexpr (ST::kUsed) {
            constexpr u32 slot = ST::kSlot;
            using CLS = std::decay_t<decltype(S.cls)>;
            const u64 after = sub_start + (u64)n_tiles * kTileBytes;
            CLS local;
            const CLS* cls = &S.cls;
            if constexpr (ST::kPf) {
                make_cls(local, slot);
                cls = &local;
                if (S.c.g63 == kUnknown) S.c.g63 = context_before(local, toff, after);
            } else S.c.g63 = bcast(S.E, 0);
            if (S.c.tracked || (S.c.g63 & 0x8000u)) {
                u64 os; u32 och, ofl;
                if (S.c.tracked) { os = S.c.t_start; och = S.c.t_chars; ofl = S.c.t_flags; }
                else {
                    const u32 suf = trailing_ones16(S.c.g63 & 0xFFFFu);
                    const u32 s63 = starts_before(*cls, toff, after);
                    os = after - suf;
                    och = (u32)__popc((s63 & 0xFFFFu) >> (16u - suf));
                    ofl = 0;
                }
                S.em.append(lane == 0, os, sub_end, och, ofl | kRecEndOpen);
            }
            S.em.end_region();
        }
    };
    finish(s0); finish(s1); finish(s2);
}

template <class C0, class C1, class C2>
static hipError_t launch_f(const FusedParams& fp, uint32_t used, hipStream_t stream) {
    const u64 waves = (fp.len + fp.subchunk - 1) / fp.subchunk;
    const u64 blocks = (waves + 3) / 4;
    if (blocks == 0) return hipSuccess;
    FusedParams q = fp;
    // the prefilter: the weakest one any slot that has one allows (pairs are right wherever groups of four are); SX_FUSED_PREFILTER=0 / 2: none / pairs
    int mode = 1;
    if ((used & 2u) && Prefilter<C1>::mode_for(fp.m[1]) != 1) mode = Prefilter<C1>::mode_for(fp.m[1]);
    if ((used & 4u) && mode && Prefilter<C2>::mode_for(fp.m[2]) != 1) mode = Prefilter<C2>::mode_for(fp.m[2]) == 0 ? 0 : 2;
    if (!(used & 6u)) mode = 0;
    if (const char* e = getenv("SX_FUSED_PREFILTER")) { const int v = atoi(e); if (v == 0) mode = 0; else if (v == 2 && mode == 1) mode = 2; }
    q.pf_zero[0] = 0; q.pf_zero[1] = Prefilter<C1>::zero_bits(fp.m[1]); q.pf_zero[2] = Prefilter<C2>::zero_bits(fp.m[2]);
    if (mode == 1) hipLaunchKernelGGL((scan_kernel_fused<C0, C1, C2, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else if (mode == 2) hipLaunchKernelGGL((scan_kernel_fused<C0, C1, C2, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
    else hipLaunchKernelGGL((scan_kernel_fused<C0, C1, C2, 0>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
    return hipGetLastError();
}

// Which slot a Mission's classifier can take in the fused kernel: 0 UTF-8 (Utf8Range2), 1 UTF-16LE, 2 UTF-16BE (Utf16RangeT); -1: none
// (the Mission keeps its own launch).  Region mode only (LateEmitter).
int fused_slot_of(ClassifierKind kind, const ScanParams& p) {
    if (p.region_cap == 0 || p.persistent) return -1;
    if (kind == kClsUtf8Range2) return 0;
    if (kind == kClsUtf16Range) return p.big_endian ? 2 : 1;
    return -1;
}

// used: bit s set = slot s holds a Mission (fp.m[s]); the others are ignored.  All used slots share data, len and subchunk.
hipError_t launch_scan_fused(const FusedParams& fp, uint32_t used, hipStream_t stream) {
    u32 parity = 0;
    if (used & 2u) parity = fp.m[1].parity & 1u; else if (used & 4u) parity = fp.m[2].parity & 1u;
    FusedParams q = fp;
    for (int s = 0; s < kFusedMax; s++) {
        if (!((used >> s) & 1u)) continue;
        q.data = fp.m[s].data; q.len = fp.m[s].len; q.subchunk = fp.m[s].subchunk;
        // the candidate test of the fast loop: b = w & (w << sh0); b &= b << sh1; b = b & (b << sh2) & (b << sh3) — reach 12 bytes
        const u32 c = std::min<u32>(fp.m[s].cand_bytes, 12u);
        q.cand_f[s] = c;
        u32 have = 1;
        const u32 sh0 = have < c ? 1u : 0u; have += sh0;
        const u32 sh1 = have < c ? std::min(have, c - have) : 0u; have += sh1;
        const u32 rem = c - have;
        const u32 sh2 = std::min(have, rem), sh3 = rem;
        q.fsh[s][0] = sh0; q.fsh[s][1] = sh1; q.fsh[s][2] = sh2; q.fsh[s][3] = sh3;
    }
#define SX_ARG(...) __VA_ARGS__
#define SX_F(U, P, A, B, C) if (used == U && parity == P) return launch_f<A, B, C>(q, used, stream);
    SX_F(1u, 0u, Utf8Range2, NoCls, NoCls)   // (one Mission: the fast loop alone is worth it — 82 instead of 91 vector instructions per tile)
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
