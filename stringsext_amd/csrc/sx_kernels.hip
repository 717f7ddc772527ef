// sx_kernels.hip — hand-written HIP for gfx950 (MI355X / CDNA4).  Stage A of the scan.
//
// What the reference does per Mission and per byte (src/finding_collection.rs:134-143
// decoder call, src/helper.rs:237-332 filter loop, src/mission.rs:333-348 bit tests) is
// split here into a data-parallel part that runs on the device and an exact sequential
// replay that only runs where a Finding can actually arise (host, sx_replay.cpp):
//
//   device: per byte, G = "belongs to a valid character of the encoding whose UTF-8 lead
//           byte passes af/ubf" and S = "is the first byte of such a character";
//           report every maximal stretch of G with >= min_chars set S bits.
//
// Memory-bound integer work (no MFMA): one wavefront streams a contiguous sub-chunk in
// 1 KiB tiles, 16 bytes per lane (`buffer_load_dwordx4`, fully coalesced, hardware
// bounds check at the chunk end).  Classification is SWAR on 32-bit registers; the four
// byte-flag dwords become a 16-bit lane mask with `v_dot4_u32_u8`; neighbouring lanes
// exchange their edge bits with DPP wave shifts; tile-to-tile state lives in SGPRs
// (readlane).  A wave-uniform ballot test ("is there any stretch of >= cand_bytes
// bytes, or an open long stretch carried in?") keeps >99 % of tiles of binary data on
// a short fast path; the slow path resolves stretches across lanes with a ballot-guided
// look-back and appends records with one atomic per wave.
#include "sx_scan_core.hpp"

namespace sx {

// ------------------------------------------------------------------------------------------
// The scan kernel: one wavefront per sub-chunk.
// ------------------------------------------------------------------------------------------
template <class CLS, bool NEEDS_LUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NEEDS_LUT ? 6 : 7))) void scan_kernel(const ScanParams p) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_lut[512];
    if (NEEDS_LUT) {
        lds_lut[threadIdx.x] = p.lut[threadIdx.x];
        lds_lut[threadIdx.x + 256] = p.lut[threadIdx.x + 256];
        __syncthreads();
    }
    const u32 lane = lane_id();
    if (p.wave_prio) __builtin_amdgcn_s_setprio(3);   // scan waves issue before another stream's waves on the same SIMD (stage B)
    CLS cls;
    cls.init(p, lds_lut);
    Emitter em{ p.recs, p.counters, p.capacity, 0u, 0u, p.region_cap, 0u, p.region_counts };

    // Classic grid: wavefront w owns sub-chunk w.  Persistent grid (p.persistent): fewer
    // wavefronts than sub-chunks, each takes the next free sub-chunk from a device counter
    // until none is left — the grid then leaves wave slots (and all LDS) to other streams.
    for (bool first_round = true;; first_round = false) {
    u64 wave;
    if (p.persistent) {
        u32 w = 0;
        if (lane == 0) w = atomicAdd(p.counters + 3, 1u);
        wave = uniform(w);
    } else {
        if (!first_round) break;
        wave = (u64)blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    }
    const u64 sub_start = wave * (u64)p.subchunk;
    if (sub_start >= p.len) break;
    const u64 sub_end = (sub_start + p.subchunk < p.len) ? sub_start + p.subchunk : p.len;
    em.begin_region(wave);

    // Buffer descriptor over [win_lo, win_hi): one tile of look-back (classification state
    // at the sub-chunk start) and two of look-ahead; reads beyond it return 0.
    const bool has_pre = sub_start >= kTileBytes;
    const u64 win_lo = has_pre ? sub_start - kTileBytes : 0;
    u64 win_hi = sub_end + 2 * kTileBytes;
    if (win_hi > p.len) win_hi = p.len;
    const uint8_t* base_ptr = p.data + win_lo;
    const u32 base_lo = uniform((u32)(uintptr_t)base_ptr), base_hi = uniform((u32)((uintptr_t)base_ptr >> 32));
    // The range check is per dword: a dword that straddles the end would read as 0, so the
    // size is rounded up to the 16-byte block (same page: `data` is 16-byte aligned); bytes
    // past the chunk end are masked by the classifiers (`avail`), never trusted.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((u64)base_hi << 32) | base_lo), 0, (int)uniform(((u32)(win_hi - win_lo) + 15u) & ~15u), 0x00020000);
    auto load = [&](u32 off) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0); };
    // Round 5: a tile in flight is its 1 KiB (16 bytes per lane, `d`) plus the dword behind every lane's 16 bytes (`e`: the
    // look-ahead of the classifiers — the next lane's first dword, for lane 63 the next TILE's), fetched by a second load from the
    // same address register + 16 (the same eight cache lines).  Rounds 1-4 took the look-ahead from the next lane by DPP and, for
    // lane 63, out of the next tile's registers (v_readlane) — which made every tile wait for the tile behind it: one tile in
    // flight per wavefront whatever the code said (the compiler's s_waitcnt vmcnt(0) at the loop head), ≈ 7 MB in flight on the
    // whole chip where 8 TB/s x 1.5 us want 12.  Now three register sets rotate through a loop unrolled three times (no register
    // moves: rounds 1-4 spent four v_mov_b64 per tile on the rotation), the set being classified depends on nothing younger, and two
    // tiles are in flight behind it.
    struct TileRegs { u32x4 d; u32 e; };
    auto fetch = [&](u32 off) -> TileRegs {
        TileRegs r;
        const u32 a = off + lane * 16u;
        r.d = load(a);
        r.e = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(a + 16u), 0, 0);
        return r;
    };

    const int n_tiles = (int)((sub_end - sub_start + kTileBytes - 1) / kTileBytes);
    // tiles whose 1 KiB + look-ahead lie fully inside the chunk need no end-of-input care
    int n_safe = n_tiles;
    while (n_safe > 0 && sub_start + (u64)n_safe * kTileBytes + 16 > p.len) n_safe--;

    int t = has_pre ? -1 : 0;  // -1 = the look-back tile
    u32 toff = 0;              // byte offset of tile t inside the window
    TileRegs R0 = fetch(0u), R1 = fetch(kTileBytes), R2;
    Carry c;
    c.g63 = 0; c.tracked = 0; c.t_chars = 0; c.t_flags = 0; c.t_start = 0;

    // start mask of the 16 bytes right before the tile at window offset `off` (= lane 63 of
    // the previous tile), recomputed on demand: every lane reads the same 20 bytes
    auto starts_before = [&](u32 off, u64 tile_base, auto near_tag) -> u32 {
        constexpr bool NE = decltype(near_tag)::value;
        if (off < 16u) return 0u;
        asm volatile("" : "+s"(off));  // do not speculate these loads into the per-tile fast path
        const u32x4 x = load(off - 16u);
        const u32 nx = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, 0, 0);
        u32 avail = 32;
        if (NE) { const u64 b = tile_base - 16; avail = b >= p.len ? 0u : (p.len - b > 32 ? 32u : (u32)(p.len - b)); }
        return uniform(cls.template classify<true>(x, nx, avail, NE));
    };

    // X: the tile to classify (arrived or about to); Z: the free set, which takes the tile two behind X
    auto body = [&](auto near_tag, const TileRegs& X, TileRegs& Z) {
        constexpr bool NE = decltype(near_tag)::value;
        Z = fetch(toff + 2 * kTileBytes);  // in flight while this tile and the next are classified
        const u32x4 cur = X.d;
        const u64 tile_base = sub_start + (u64)((long long)t * (long long)kTileBytes);
        const u64 lane_base = tile_base + 16ull * lane;
        u32 avail = 32;
        if (NE) avail = lane_base >= p.len ? 0u : (p.len - lane_base > 32 ? 32u : (u32)(p.len - lane_base));

        const u32 g63_in = c.g63;  // previous tile's lane 63 (by value: c is rewritten below)
        const bool tracked_in = c.tracked != 0;
        const u32 g = cls.template classify<false>(cur, X.e, avail, NE);
        const u32 pg = from_prev(g, g63_in);
        const u32 gf = (g & 0xFFFFu) | (pg >> 16);       // final good mask of my 16 bytes
        const u32 g63_out = bcast(gf | (g & 0xFFFF0000u), 63);

        // r: bit q set iff bits q-cand_bytes+1 .. q of the window (my 16 bytes above the 16 before them) are all set.  The window's
        // lowest bits are the bits of the previous lane that ITS predecessor's spill can set (one bit for the range classifiers, up to
        // three for the LUT ones); cand_bytes is <= 14 (sx_stage_a.cpp), so no bit of r above 15 looks below bit 3, and the candidate
        // test takes the previous lane's mask as the first DPP delivered it (the slow paths get the exact window); four doubling
        // steps reach 14.
        u32 w = __builtin_amdgcn_perm(gf, pg, 0x05040100u);   // gf << 16 | pg & 0xFFFF
        u32 r = w;
        r &= r << p.cand_sh[0]; r &= r << p.cand_sh[1]; r &= r << p.cand_sh[2]; r &= r << p.cand_sh[3];   // (1 + 1 + 2 + 4 + 6 = 14 >= cand_bytes)
        const bool any_cand = __ballot((r & 0xFFFF0000u) != 0) != 0;
        const bool first_tile = t == 0;
        const bool first_open = first_tile && (g63_in & 0x8000u);

        if (t < 0 || (!any_cand && !tracked_in && !first_open)) {
            // fast path (and the look-back tile, of which only the classification state matters)
            c.g63 = g63_out;
        } else {
            const u32 s = cls.template classify<true>(cur, X.e, avail, NE);
            w = (gf << 16) | (from_prev(gf, g63_in) & 0xFFFFu);   // the exact window
            const u32 s63 = (g63_in & 0x8000u) ? starts_before(toff, tile_base, near_tag) : 0u;
            const u32 sw = (s << 16) | (from_prev(s, s63) & 0xFFFFu);
            bool done = false;
            if (!tracked_in && !first_open) done = light_path(w, sw, r, lane_base, em, p.min_chars);
            if (done) c.g63 = g63_out;
            else {
                em.heavy_n++;
                const u64 tile_end = tile_base + kTileBytes < sub_end ? tile_base + kTileBytes : sub_end;
                heavy_path(gf, s, g, g63_in, s63, r >> 16, tile_base, tile_end, c, em, p.min_chars, p.cand_bytes, first_tile);
            }
        }
        toff += kTileBytes; t++;
    };

    while (t + 3 <= n_safe) {
        body(std::false_type{}, R0, R2);
        body(std::false_type{}, R1, R0);
        body(std::false_type{}, R2, R1);
    }
    // (what is left — at most two tiles — and the tiles near the end of the input: one copy of the code, the sets rotate by moves)
    while (t < n_tiles) { body(std::true_type{}, R0, R2); R0 = R1; R1 = R2; }

    // the stretch that is still open where the sub-chunk ends
    if (c.tracked || (c.g63 & 0x8000u)) {
        u64 os; u32 och, ofl;
        if (c.tracked) { os = c.t_start; och = c.t_chars; ofl = c.t_flags; }
        else {
            const u32 suf = trailing_ones16(c.g63 & 0xFFFFu);
            const u64 after = sub_start + (u64)n_tiles * kTileBytes;
            const u32 s63 = starts_before(toff, after, std::true_type{});
            os = after - suf;
            och = (u32)__popc((s63 & 0xFFFFu) >> (16u - suf));
            ofl = 0;
        }
        em.append(lane == 0, os, sub_end, och, ofl | kRecEndOpen);
    }
    em.end_region(wave);
    }  // next sub-chunk
    em.invalidate_rest();
}

// ------------------------------------------------------------------------------------------
// Double-byte legacy encodings (Big5, EUC-JP): table-driven token classifier with the index in LDS.
//
// WHATWG decoders of this family consume the input in TOKENS: one byte outside the lead range, or a
// lead byte with the byte after it (EUC-JP: 8F + A1..FE + one more byte).  After a lead, any byte
// returns the decoder to neutral, and a byte outside the lead range never becomes pending — so the
// byte after every such byte starts a token, and inside a stretch of lead-range bytes the token
// starts follow from the first one by jumping token lengths (2, or 3 after 8F + A1..FE).  Per lane:
//   1. SWAR byte classes -> 16-bit masks (lead range, ASCII accepted, 8F, A1..FE);
//   2. token starts = the orbit of the known starts under "jump by token length", computed for every
//      possible number of bytes (0..NS-1) that the previous lane's last token hangs over: NS orbits
//      advance together by bit operations until none changes (wave-uniform loop);
//   3. the lane's transfer function "hang-over in -> hang-over out" is constant when the lane holds a
//      byte outside the lead range (binary data: always); then the previous lane's value arrives with one
//      DPP move, else the functions are composed by a wave prefix scan (text without ASCII);
//   4. one LDS lookup per token: a 2-bit code per byte pair (unmapped / mapped / accepted / accepted,
//      two characters), index = the pair as a little-endian u16 — tokens never start at adjacent bytes,
//      so 8 (+1) lookups cover the 16 bytes;
//   5. good/start masks: all bytes of an accepted token; a malformed token's last byte, if ASCII, is
//      given back by the decoder and is a character of its own.
// The masks then take the same light/heavy paths as every other classifier.
// ------------------------------------------------------------------------------------------
template <int ENC>
struct DbcsTraits {
    static constexpr int NS = ENC == 4 ? 2 : 3;            // possible hang-overs: 0..NS-1 bytes
    static constexpr u32 kTableWords = ENC == 4 ? 4096u : 8192u;  // 2 bits x 65536 pairs (EUC-JP: jis0208+8E, then jis0212)
};

// byte flags (bit 7 of every byte)
SX_DEV u32 swar_range(u32 v7, u32 c1, u32 c2) { return (v7 + c1) & ~(v7 + c2); }  // low 7 bits in [lo,hi]: c1 = 0x80-lo, c2 = 0x7F-hi
SX_DEV u32 swar_eq(u32 v, u32 pat) {  // bytes equal to pat's bytes
    const u32 y = v ^ pat;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y);
}

// ---- gb18030 / GBK: four-byte tokens, exactly (round 4; rounds 2-3 marked `lead digit lead digit` good wherever it stood: a superset).
// The two-byte grammar reads L D L D as (L, D) malformed — D given back, a character of its own — twice: aligned with the true tokens,
// so its token starts are a superset of the true ones and only three things are missing (WHATWG "gb18030 decoder"; sx_codec_core.hpp
// ddec_gb18030): (1) the pattern is ONE token — a character if its pointer is in range, else an error of all four bytes; either way the
// digits inside it are no characters; (2) of two candidates two bytes apart the first one wins (L D L D L D:
// bytes 0..3 are a token, byte 4 starts the next) — along a row of candidates every other one, from the row's first; (3) the token's
// character passes the filter or not (index gb18030 ranges -> code point -> UTF-8 lead byte -> ubf).  (2) is solved per parity class of the position (candidates two apart have the same parity): 8 elements per lane and
// class, rows = runs of set bits, alternate elements from a run's first; what a lane hands on (is the class's last element a candidate,
// and was it taken) depends on what it was handed only if all 8 are candidates.  At the first tile of a sub-chunk nothing is known about
// a row that is already running: its members are marked as in rounds 2-3 (good if accepted, nothing cleared) — a superset there, never less.
SX_DEV u32 gb4_compress(u32 x) {   // the bits at even positions of 16 -> 8
    x &= 0x5555u; x = (x | (x >> 1)) & 0x3333u; x = (x | (x >> 2)) & 0x0F0Fu; x = (x | (x >> 4)) & 0x00FFu;
    return x;
}
SX_DEV u32 gb4_expand(u32 x) {     // 8 bits -> the even positions of 16
    x &= 0xFFu; x = (x | (x << 4)) & 0x0F0Fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u;
    return x;
}
// in-states: 0 the class's element in front is no candidate, 1 it is one and was taken, 2 it is one and was not, 3 not known
struct Gb4Class { u32 sel0, sel1, run0, out_tab; };
SX_DEV Gb4Class gb4_class(u32 v) {
    Gb4Class r;
    r.run0 = v & ~(v + 1u) & 0xFFu;                       // the run that begins at element 0
    const u32 pair = v & (v << 1);                         // element k and k - 1 are candidates
    auto alt = [&](u32 h) { u32 t = h; t |= (t << 2) & pair; t |= (t << 2) & pair; t |= (t << 2) & pair; return t & 0xFFu; };
    const u32 heads = v & ~(v << 1);
    r.sel0 = alt(heads);                                   // every run from its first element
    r.sel1 = alt((heads & ~1u) | ((v & 1u) ? (v & 2u) : 0u));   // ... the run at element 0 from its second (the element in front was taken)
    if (!(v & 0x80u)) r.out_tab = 0;
    else if (r.run0 == 0xFFu) r.out_tab = 2u | (1u << 2) | (2u << 4) | (3u << 6);
    else { const u32 o = (r.sel0 & 0x80u) ? 1u : 2u; r.out_tab = o * 0x55u; }
    return r;
}
SX_DEV u32 gb4_spread(u32 q) { return q | (q << 1) | (q << 2) | (q << 3); }
// the character of a four-byte token, 0 = the pointer is out of range (t: [208 breakpoint pointers][208 code points])
SX_DEV u32 gb4_code_point(const uint16_t* t, u32 pointer) {
    if ((pointer > 39419u && pointer < 189000u) || pointer > 1237575u) return 0;
    if (pointer == 7457u) return 0xE7C7u;
    if (pointer >= 189000u) return 0x10000u + (pointer - 189000u);
    u32 lo = 0, hi = 208;
    while (hi - lo > 1) { const u32 mid = (lo + hi) / 2; if (t[mid] <= pointer) lo = mid; else hi = mid; }
    return (u32)t[208 + lo] + (pointer - t[lo]);
}

// ENC 4: the two-byte encodings (Big5, Shift_JIS, EUC-KR: lead ranges and pair table are parameters), 5: EUC-JP.
// AF_RANGE: the accepted ASCII bytes are one range (else a 256-entry LUT in LDS).  HIGH1: bytes >= 0x80 outside the
// lead range can be characters too (Shift_JIS: 0x80, A1..DF) — they come from the LUT.
template <int ENC, bool AF_RANGE, bool HIGH1 = false>
__global__ __launch_bounds__(256) void scan_kernel_dbcs(const ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) u32 lds_tab[];
    using TR = DbcsTraits<ENC>;
    constexpr int NS = TR::NS;
    {
        const u32x4* src = (const u32x4*)p.pair_lut;
        u32x4* dst = (u32x4*)lds_tab;
        for (u32 i = threadIdx.x; i < TR::kTableWords / 4; i += 256) dst[i] = src[i];
        if (!AF_RANGE) ((u8*)(lds_tab + TR::kTableWords))[threadIdx.x] = p.lut[threadIdx.x];
        static_assert(!HIGH1 || !AF_RANGE, "single bytes >= 0x80 are looked up");
        __syncthreads();
    }
    const u8* aflut = (const u8*)(lds_tab + TR::kTableWords);
    const u32 lane = lane_id();
    const u32 a1 = rep4(0x80u - p.a_lo), a2 = rep4(0x7Fu - p.a_hi);
    Emitter em{ p.recs, p.counters, p.capacity, 0u, 0u, p.region_cap, 0u, p.region_counts };

    const u64 wave = (u64)blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    const u64 sub_start = wave * (u64)p.subchunk;
    if (sub_start >= p.len) return;
    const u64 sub_end = (sub_start + p.subchunk < p.len) ? sub_start + p.subchunk : p.len;
    em.begin_region(wave);

    // byte classes of one dword: flags at bit 7 of every byte
    const u32 lr1a = p.lr_c1[0], lr2a = p.lr_c2[0], lr1b = p.lr_c1[1], lr2b = p.lr_c2[1];
    auto cls_lr = [&](u32 v) -> u32 {
        const u32 t = v & 0x7F7F7F7Fu;
        if (ENC == 4) return (swar_range(t, lr1a, lr2a) | swar_range(t, lr1b, lr2b)) & v & kM;   // Big5 / EUC-KR 81..FE; Shift_JIS 81..9F, E0..FC
        return ((swar_range(t, rep4(0x80u - 0x21u), rep4(0x7Fu - 0x7Eu)) & v) | swar_eq(v & 0xFEFEFEFEu, 0x8E8E8E8Eu)) & kM;  // A1..FE, 8E, 8F
    };

    // Look-back: the classification state at the sub-chunk start.  One tile normally: it holds a byte outside the lead range, and the
    // byte behind such a byte starts a token whatever came before.  A tile of lead-range bytes only says nothing about the grid; the
    // walk then goes further back — through the sub-chunk in front at most (round 3; before, to the nearest such byte however far:
    // a format fill of 0xF6 or 0xE5 bytes, gigabytes in a disk image, cost every wavefront in it a walk to its beginning and the
    // classification of all of it).  In a stretch of lead-range bytes every token has two bytes (EUC-JP: unless an 8E / 8F occurs),
    // so the hang-over at a position follows from the distance to a known position by parity:
    //   * a byte outside the lead range found at s: tokens start at s + 1;
    //   * none in the whole sub-chunk in front: that sub-chunk's wavefront publishes the hang-over at ITS first byte
    //     (p.grid_flags, written as soon as it is known, i.e. before its main loop) and this one waits for it — wavefronts are
    //     dispatched in order, so the one in front is resident or done; the chain ends at a sub-chunk that holds such a byte or at
    //     the chunk start, where the carried decoder says how many bytes finish the pending token (p.parity).
    // Then one look-back tile is classified (for the stretch that may be open at the sub-chunk start), entered with that hang-over.
    // EUC-JP with an 8E / 8F among the walked bytes: token lengths differ, all walked tiles are classified as before.
    int pre = 0;
    u32 cov = 0;  // bytes at the start of the next tile that belong to a token begun before it
    {
        u64 lo = sub_start;
        const u64 stop = sub_start >= p.subchunk ? sub_start - p.subchunk : 0;
        bool found = false, special = false, gb_skipped = false;   // (gb_skipped: the look-back tile holds bytes outside the lead range after all)
        u64 known_at = 0;     // a position from which the hang-over is known_cov ...
        u32 known_cov = 0;
        while (lo > stop) {
            lo -= kTileBytes;
            pre++;
            const u32x4 x = *(const u32x4*)(p.data + lo + 16u * lane);
            const u32 f0 = cls_lr(x.x), f1 = cls_lr(x.y), f2 = cls_lr(x.z), f3 = cls_lr(x.w);
            if (ENC == 5) special = special || __ballot(((swar_eq(x.x & 0xFEFEFEFEu, 0x8E8E8E8Eu) | swar_eq(x.y & 0xFEFEFEFEu, 0x8E8E8E8Eu) |
                                                          swar_eq(x.z & 0xFEFEFEFEu, 0x8E8E8E8Eu) | swar_eq(x.w & 0xFEFEFEFEu, 0x8E8E8E8Eu)) & kM) != 0) != 0;
            const u32 out16 = movemask16(f0, f1, f2, f3) ^ 0xFFFFu;   // my bytes outside the lead range
            const u64 bal = __ballot(out16 != 0);
            if (bal) {
                if (HIGH1 && p.gb4 && pre == 1) {
                    // gb18030: the tile's FIRST byte outside the lead range is a digit — the lead byte in front of it may begin a four-byte
                    // token, if it begins a token at all: that depends on the grid in front of it, which the usual case below does not
                    // know.  Walk on as if this tile held no such byte: the hang-over at its start then comes out exactly (parity from a
                    // known position further back, or the wavefront in front)
                    const int lowl = __builtin_ctzll(bal);
                    const u32 m0 = bcast(out16, lowl);
                    const u8 fb = p.data[lo + 16ull * (u32)lowl + (u32)__builtin_ctz(m0)];
                    if (fb >= 0x30 && fb <= 0x39) { gb_skipped = true; continue; }
                }
                const int top = 63 - __clzll((long long)bal);
                const u32 m = bcast(out16, top);
                known_at = lo + 16ull * (u32)top + (32u - (u32)__clz((int)m));   // the byte behind the last such byte
                known_cov = 0;
                found = true;
                break;
            }
        }
        if (pre == 1 && found) cov = 0;                      // the usual case: the tile in front holds such a byte, its entry does not matter
        else if (pre >= 1) {
            if (!found) {
                if (lo == 0) { known_at = 0; known_cov = p.parity; }   // the chunk start: the token pending on entry takes this many bytes
                else {
                    const u32* flag = p.grid_flags + (wave - 1);
                    u32 v = 0;
                    if (lane == 0) { while (((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u) == 0) __builtin_amdgcn_s_sleep(2); }
                    v = uniform(v);
                    known_at = stop; known_cov = (v >> 1) & 3u;
                }
            }
            if (special) cov = found ? 0u : known_cov;       // (all walked tiles are classified; the first one's entry: nothing known / the known hang-over)
            else {
                // tokens start at known_at + known_cov and have two bytes each up to the look-back tile
                const u64 tile_lo = sub_start - kTileBytes, first = known_at + known_cov;
                cov = first >= tile_lo ? (u32)(first - tile_lo) : (u32)((tile_lo - first) & 1ull);
                pre = 1;
                // (the hang-over at my own first byte follows the same way: the wavefront behind me need not wait for my look-back tile)
                if (lane == 0 && first <= sub_start && !gb_skipped)
                    __hip_atomic_store(p.grid_flags + wave, 1u | ((u32)((sub_start - first) & 1ull) << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        else cov = p.parity;  // sub-chunk 0: the chunk start
    }
    const u64 win_lo = sub_start - (u64)pre * kTileBytes;
    u64 win_hi = sub_end + 2 * kTileBytes;
    if (win_hi > p.len) win_hi = p.len;
    const uint8_t* base_ptr = p.data + win_lo;
    const u32 base_lo = uniform((u32)(uintptr_t)base_ptr), base_hi = uniform((u32)((uintptr_t)base_ptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((u64)base_hi << 32) | base_lo), 0, (int)uniform(((u32)(win_hi - win_lo) + 15u) & ~15u), 0x00020000);
    auto load = [&](u32 off) -> u32x4 { return __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0); };

    const int n_tiles = (int)((sub_end - sub_start + kTileBytes - 1) / kTileBytes);
    int n_safe = n_tiles;
    while (n_safe > 0 && sub_start + (u64)n_safe * kTileBytes + 16 > p.len) n_safe--;

    int t = -pre;
    u32 toff = 0;
    u32x4 cur = load(lane * 16u);
    u32x4 nxt = load(lane * 16u + kTileBytes);
    Carry c;
    c.g63 = 0; c.tracked = 0; c.t_chars = 0; c.t_flags = 0; c.t_start = 0;
    u32 s63c = 0;  // lane 63 of the previous tile: final start mask | own spill bits << 16
    u32 gb_c63 = 0, gb_q63 = 0;   // gb18030: lane 63's four-byte tokens' bytes beyond its own (bits 16..18); what it hands on per parity class (2 x 2 bits)

    auto body = [&](auto near_tag) {
        constexpr bool NE = decltype(near_tag)::value;
        const u32x4 nn = load(toff + lane * 16u + 2 * kTileBytes);
        const u64 tile_base = sub_start + (u64)((long long)t * (long long)kTileBytes);
        const u64 lane_base = tile_base + 16ull * lane;
        u32 avail = 32;
        if (NE) avail = lane_base >= p.len ? 0u : (p.len - lane_base > 32 ? 32u : (u32)(p.len - lane_base));
        const u32 valid = NE ? low_mask(avail) : 0xFFFFFFFFu;
        const u32 nx = from_next(cur.x, bcast(nxt.x, 0));
        const u32 xs[5] = { cur.x, cur.y, cur.z, cur.w, nx };

        // ---- 1. byte classes (bits 0..15 own bytes, 16..19 the next lane's first four)
        u32 fl[5], fa[5], fh[5], f8[5], fx[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const u32 v = xs[k], t7 = v & 0x7F7F7F7Fu;
            fl[k] = cls_lr(v);
            if (AF_RANGE) fa[k] = swar_range(t7, a1, a2) & ~v & kM;
            else fa[k] = (aflut[v & 0xFF] | (aflut[(v >> 8) & 0xFF] << 8) | (aflut[(v >> 16) & 0xFF] << 16) | (aflut[v >> 24] << 24)) & kM;
            if (HIGH1) fx[k] = fa[k] & ~v;   // the ASCII ones among them
            if (ENC == 5) {
                fh[k] = swar_range(t7, rep4(0x80u - 0x21u), rep4(0x7Fu - 0x7Eu)) & v & kM;
                f8[k] = swar_eq(v, 0x8F8F8F8Fu) & kM;
            }
        }
        const u32 LR = (movemask16(fl[0], fl[1], fl[2], fl[3]) | (movemask4(fl[4]) << 16)) & valid;
        const u32 asc = (movemask16(fa[0], fa[1], fa[2], fa[3]) | (movemask4(fa[4]) << 16)) & valid;   // accepted one-byte characters
        const u32 asc7 = HIGH1 ? (movemask16(fx[0], fx[1], fx[2], fx[3]) | (movemask4(fx[4]) << 16)) & valid : asc;   // ... that are ASCII
        u32 L3 = 0;
        if (ENC == 5) {
            const u32 H = movemask16(fh[0], fh[1], fh[2], fh[3]) | (movemask4(fh[4]) << 16);
            const u32 F8 = movemask16(f8[0], f8[1], f8[2], f8[3]);
            L3 = F8 & (H >> 1) & 0xFFFFu;   // 8F followed by A1..FE: a three-byte token if it starts one
        }
        const u32 L2 = LR & ~L3 & 0xFFFFu;
        const u32 known = ((~LR) & 0xFFFFu) << 1;  // the byte after a byte outside the lead range starts a token

        // ---- 2. token starts for every possible hang-over
        u32 St[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) St[s] = known | (1u << s);
        for (;;) {
            bool changed = false;
#pragma unroll
            for (int s = 0; s < NS; s++) {
                const u32 nw = St[s] | ((St[s] & L2) << 2) | ((St[s] & L3) << 3);
                changed |= nw != St[s];
                St[s] = nw;
            }
            if (!__ballot(changed)) break;
        }
        // ---- 3. the hang-over this lane starts with
        u32 T = 0;  // T(s) at bits 2s..2s+1
#pragma unroll
        for (int s = 0; s < NS; s++) T |= (u32)__builtin_ctz(St[s] >> 16) << (2 * s);
        u32 cov_in;
        const bool has_reset = (LR & 0xFFFFu) != 0xFFFFu;
        if (!__ballot(!has_reset)) {
            cov_in = from_prev(T & 3u, cov);  // T is constant in every lane
            cov = bcast(T & 3u, 63);
        } else {
            // inclusive prefix composition F_i = T_i o ... o T_0 (Hillis-Steele over the wavefront)
            u32 F = T;
#pragma unroll
            for (u32 d = 1; d < 64; d <<= 1) {
                const u32 G = shfl(F, lane >= d ? lane - d : lane);
                u32 r = 0;
#pragma unroll
                for (int s = 0; s < NS; s++) r |= ((F >> (2 * ((G >> (2 * s)) & 3u))) & 3u) << (2 * s);
                if (lane >= d) F = r;
            }
            const u32 Fc = (F >> (2 * cov)) & 3u;  // hang-over after my lane, given the tile's
            cov_in = from_prev(Fc, cov);
            cov = bcast(Fc, 63);
        }
        u32 S0 = NS == 2 ? (cov_in ? St[1] : St[0]) : (cov_in == 0 ? St[0] : (cov_in == 1 ? St[1] : St[NS - 1]));
        S0 &= ~((1u << cov_in) - 1u);   // the hang-over bytes are not starts (a `known` bit may sit there)
        S0 &= 0xFFFFu;

        // ---- 4. one lookup per multi-byte token
        u32 M3 = S0 & L3, P0 = S0 & L2, P1 = M3 << 1;  // lookup positions: table 0 at the token start, table 1 one byte later
        if (NE) { P0 &= valid >> 1; M3 &= valid >> 2; P1 = M3 << 1; }
        const u32 P = P0 | P1;
        u32 A = 0, Mp = 0, Dbl = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const u32 w0 = xs[k >> 1], w1 = xs[(k >> 1) + 1];
            const u32 u_even = (k & 1) ? (w0 >> 16) : (w0 & 0xFFFFu);
            const u32 u_odd = (k & 1) ? (__builtin_amdgcn_alignbyte(w1, w0, 3) & 0xFFFFu) : ((w0 >> 8) & 0xFFFFu);
            const u32 odd = (P >> (2 * k + 1)) & 1u;
            u32 idx = odd ? u_odd : u_even;
            if (ENC == 5) idx |= ((P1 >> (2 * k)) & 3u) ? 0x10000u : 0u;
            const u32 word = lds_tab[idx >> 4];
            const u32 code = (word >> ((idx & 15u) << 1)) & 3u;
            const u32 sh = 2 * k + odd;
            A |= (code >> 1) << sh; Mp |= ((code | (code >> 1)) & 1u) << sh; Dbl |= ((code == 2u) ? 1u : 0u) << sh;
        }
        if (ENC == 5) {  // a three-byte token that starts at byte 15: its lookup position is 16
            const u32 idx = (nx & 0xFFFFu) | 0x10000u;
            const u32 word = lds_tab[idx >> 4];
            const u32 code = (word >> ((idx & 15u) << 1)) & 3u;
            A |= (code >> 1) << 16; Mp |= ((code | (code >> 1)) & 1u) << 16;
        }
        A &= P; Mp &= P; Dbl &= P0;
        const u32 A0 = A & P0, A1 = A & P1;
        const u32 bad_last = (((P0 & ~Mp) << 1) | ((P1 & ~Mp) << 1)) & asc7;  // malformed token, last byte ASCII: its own character
        const u32 S1 = S0 & ~LR & asc;                                         // one-byte tokens
        // ---- 5. good / start masks (bits 16.. spill onto the next lane's first bytes)
        u32 g = S1 | A0 | (A0 << 1) | (A1 >> 1) | A1 | (A1 << 1) | bad_last;
        u32 s = S1 | A0 | ((Dbl & A0) << 1) | (A1 >> 1) | bad_last;
        u32 gb_cover = 0;   // gb18030: bytes of this lane's four-byte tokens (bits 16..18: the next lane's first bytes)
        if (HIGH1 && p.gb4) {
            u32 fd[5];
#pragma unroll
            for (int k = 0; k < 5; k++) fd[k] = swar_range(xs[k] & 0x7F7F7F7Fu, rep4(0x80u - 0x30u), rep4(0x7Fu - 0x39u)) & ~xs[k] & kM;
            const u32 D = (movemask16(fd[0], fd[1], fd[2], fd[3]) | (movemask4(fd[4]) << 16)) & valid;
            const u32 Qc = LR & (D >> 1) & (LR >> 2) & (D >> 3) & S0;   // lead digit lead digit at a token start of the two-byte grammar
            const bool first = t == -pre;
            if (t == n_tiles - 1 && lane == 0)   // what I enter my last tile with: the wavefront behind me may need it (its look-back tile is this one)
                __hip_atomic_fetch_or(p.grid_flags + wave, 0x10u | (gb_q63 << 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(Qc != 0) || gb_q63 || first) {
                // (1), (3): the pointer, its character, the filter — a handful of tiles per megabyte of text, every third tile of random bytes
                const u32 V = Qc;   // (a pointer out of range is an error of all four bytes: a token all the same)
                u32 AC = 0;
                for (u32 rem = Qc; rem;) {
                    const u32 i = (u32)__builtin_ctz(rem);
                    rem &= rem - 1u;
                    const u8* b = p.data + lane_base + i;
                    const u32 pointer = ((((u32)b[0] - 0x81u) * 10u + ((u32)b[1] - 0x30u)) * 126u + ((u32)b[2] - 0x81u)) * 10u + ((u32)b[3] - 0x30u);
                    const u32 cp = gb4_code_point(p.gb_ranges, pointer);
                    if (!cp) continue;
                    const u32 lead = cp < 0x800u ? 0xC0u | (cp >> 6) : cp < 0x10000u ? 0xE0u | (cp >> 12) : 0xF0u | (cp >> 18);
                    if ((p.ubf >> (lead & 0x3Fu)) & 1ull) AC |= 1u << i;
                }
                // (2): every other candidate of a row, per parity class
                const Gb4Class ce = gb4_class(gb4_compress(V)), co = gb4_class(gb4_compress(V >> 1));
                // lane 0's in-state: what lane 63 of the tile before handed on.  A sub-chunk's first tile is the last tile of the sub-chunk in
                // front: that wavefront publishes what it entered it with (grid_flags bit 4, bits 5..8; wavefronts are dispatched in order), and
                // this one waits for it — only if a row is running there (a candidate at lane 0's first element).  The chunk's byte 0: nothing
                // lies in front — or a token is pending on entry, then a row that is running is marked as a superset
                u32 edge = gb_q63;
                if (first) {
                    edge = tile_base == 0 && p.parity == 0 ? 0u : 0xFu;
                    if (tile_base != 0 && wave > 0) {
                        edge = 0;
                        if (__ballot(lane == 0 && (ce.run0 | co.run0) != 0)) {
                            u32 v = 0;
                            // (bounded, ADVICE r4: the wavefront in front publishes on entering its LAST tile — a fill of `81 30 81 30 ...` would hand
                            // the grid through wavefront by wavefront, each waiting for a whole sub-chunk; after ~1 ms the row is marked as a
                            // superset instead, the state the chunk's byte 0 uses when a token is pending there: stage B decides exactly)
                            if (lane == 0) {
                                u32 spins = 0;
                                while (((v = __hip_atomic_load(p.grid_flags + (wave - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x10u) == 0 && ++spins < (1u << 14)) __builtin_amdgcn_s_sleep(2);
                                if ((v & 0x10u) == 0) v = 0xFu << 5;
                            }
                            edge = (uniform(v) >> 5) & 0xFu;
                        }
                    }
                }
                u32 in = from_prev(0u, edge);
                for (;;) {
                    const u32 out = ((ce.out_tab >> (2 * (in & 3u))) & 3u) | (((co.out_tab >> (2 * ((in >> 2) & 3u))) & 3u) << 2);
                    const u32 nin = from_prev(out, edge);
                    const bool ch = nin != in;
                    in = nin;
                    if (!__ballot(ch)) { gb_q63 = bcast(out, 63); break; }
                }
                const u32 ie = in & 3u, io = (in >> 2) & 3u;
                const u32 se = ie == 1u ? ce.sel1 : (ie == 3u ? ce.sel0 & ~ce.run0 : ce.sel0), so = io == 1u ? co.sel1 : (io == 3u ? co.sel0 & ~co.run0 : co.sel0);
                const u32 SEL = gb4_expand(se) | (gb4_expand(so) << 1);
                const u32 UNC = (ie == 3u ? gb4_expand(ce.run0) : 0u) | (io == 3u ? gb4_expand(co.run0) << 1 : 0u);
                gb_cover = gb4_spread(SEL);
                g = (g & ~gb_cover) | gb4_spread(SEL & AC) | gb4_spread(UNC & AC);
                s = (s & ~gb_cover) | (SEL & AC) | (UNC & AC);
            }
        }

        const u32 g63_in = c.g63;
        const bool tracked_in = c.tracked != 0;
        const u32 pg = from_prev(g, g63_in);
        u32 gb_kill = 0;   // gb18030: my first bytes that lie inside a four-byte token of the lane in front (its own marks for them follow in pg / ps)
        if (HIGH1 && p.gb4) { gb_kill = from_prev(gb_cover, gb_c63) >> 16; gb_c63 = bcast(gb_cover, 63); }
        const u32 gf = ((g & 0xFFFFu) & ~gb_kill) | (pg >> 16);
        const u32 pgf = from_prev(gf, g63_in) & 0xFFFFu;
        const u32 g63_out = bcast(gf | (g & 0xFFFF0000u), 63);
        const u32 ps = from_prev(s, s63c);
        const u32 sf = ((s & 0xFFFFu) & ~gb_kill) | (ps >> 16);
        const u32 s63_in = s63c & 0xFFFFu;
        s63c = bcast(sf | (s & 0xFFFF0000u), 63);

        const u32 w = (gf << 16) | pgf;
        u32 r = w;
        r &= r << p.cand_sh[0]; r &= r << p.cand_sh[1]; r &= r << p.cand_sh[2];
        r &= r << p.cand_sh[3]; r &= r << p.cand_sh[4];
        const bool any_cand = __ballot((r & 0xFFFF0000u) != 0) != 0;
        const bool first_tile = t == 0;
        const bool first_open = first_tile && (g63_in & 0x8000u);
        if (t < 0 || (!any_cand && !tracked_in && !first_open)) c.g63 = g63_out;
        else {
            const u32 sw = (sf << 16) | (from_prev(sf, s63_in) & 0xFFFFu);
            bool done = false;
            if (!tracked_in && !first_open) done = light_path(w, sw, r, lane_base, em, p.min_chars);
            if (done) c.g63 = g63_out;
            else {
                em.heavy_n++;
                const u64 tile_end = tile_base + kTileBytes < sub_end ? tile_base + kTileBytes : sub_end;
                heavy_path(gf, sf, g, g63_in, s63_in, r >> 16, tile_base, tile_end, c, em, p.min_chars, p.cand_bytes, first_tile);
            }
        }
        cur = nxt; nxt = nn; toff += kTileBytes; t++;
    };

    // the look-back tiles first; then the hang-over at my first byte is known: published for the wavefront behind me (its look-back may
    // end at my sub-chunk's start) — outside the tile loop, whose code a store with release semantics would slow down
    while (t < 0) { if (sub_start + 16 <= p.len) body(std::false_type{}); else body(std::true_type{}); }
    if (lane == 0) __hip_atomic_store(p.grid_flags + wave, 1u | (cov << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (relaxed: the word is all there is to see — a release would write the L2 back once per sub-chunk)
    while (t < n_safe) body(std::false_type{});
    while (t < n_tiles) body(std::true_type{});

    if (c.tracked || (c.g63 & 0x8000u)) {
        u64 os; u32 och, ofl;
        if (c.tracked) { os = c.t_start; och = c.t_chars; ofl = c.t_flags; }
        else {
            const u32 suf = trailing_ones16(c.g63 & 0xFFFFu);
            const u64 after = sub_start + (u64)n_tiles * kTileBytes;
            os = after - suf;
            och = (u32)__popc((s63c & 0xFFFFu) >> (16u - suf));
            ofl = 0;
        }
        em.append(lane == 0, os, sub_end, och, ofl | kRecEndOpen);
    }
    em.end_region(wave);
    em.invalidate_rest();
}

template <int ENC>
static hipError_t launch_dbcs(const ScanParams& p, hipStream_t stream) {
    u64 waves = (p.len + p.subchunk - 1) / p.subchunk;
    u64 blocks = (waves + 3) / 4;
    if (blocks == 0) return hipSuccess;
    ScanParams q = p;
    q.persistent = 0;
    const size_t lds = DbcsTraits<ENC>::kTableWords * 4 + 256;
    if (ENC == 4 && p.high1) hipLaunchKernelGGL((scan_kernel_dbcs<ENC == 4 ? 4 : ENC, false, ENC == 4>), dim3((unsigned)blocks), dim3(256), lds, stream, q);
    else if (p.af_is_range) hipLaunchKernelGGL((scan_kernel_dbcs<ENC, true>), dim3((unsigned)blocks), dim3(256), lds, stream, q);
    else hipLaunchKernelGGL((scan_kernel_dbcs<ENC, false>), dim3((unsigned)blocks), dim3(256), lds, stream, q);
    return hipGetLastError();
}

template <class CLS, bool LUT>
static hipError_t launch_t(const ScanParams& p, hipStream_t stream) {
    u64 waves = (p.len + p.subchunk - 1) / p.subchunk;
    u64 blocks = (waves + 3) / 4;
    if (blocks == 0) return hipSuccess;
    ScanParams q = p;
    if (q.persistent && blocks > q.persistent) blocks = q.persistent; else q.persistent = 0;
    hipLaunchKernelGGL((scan_kernel<CLS, LUT>), dim3((unsigned)blocks), dim3(256), 0, stream, q);
    return hipGetLastError();
}

hipError_t launch_scan(ClassifierKind kind, const ScanParams& p, hipStream_t stream) {
    if (kind == kClsBig5) return launch_dbcs<4>(p, stream);
    if (kind == kClsEucJp) return launch_dbcs<5>(p, stream);
    switch (kind) {
    case kClsSingleByteLut: return launch_t<SingleByteLut, true>(p, stream);
    case kClsUtf8Lut: return launch_t<Utf8Lut, true>(p, stream);
    case kClsUtf16Lut: return launch_t<Utf16Lut, true>(p, stream);
    case kClsUtf8Range2: return launch_t<Utf8Range2, false>(p, stream);
    case kClsUtf16Range:
        switch ((p.big_endian ? 2 : 0) | (p.parity & 1)) {
        case 0: return launch_t<Utf16RangeT<0, 0>, false>(p, stream);
        case 1: return launch_t<Utf16RangeT<0, 1>, false>(p, stream);
        case 2: return launch_t<Utf16RangeT<1, 0>, false>(p, stream);
        default: return launch_t<Utf16RangeT<1, 1>, false>(p, stream);
        }
    case kClsUtf8Range3: {
        // ED (second byte 80..9F only): 0 = not among the leads, 1 = the last of them, 2 = inside the range
        const int ed = p.l3_hi < 0xEDu || p.l3_lo > 0xEDu ? 0 : p.l3_hi == 0xEDu ? 1 : 2;
        const bool has2 = p.u_lo <= p.u_hi;
        switch ((has2 ? 3 : 0) + ed) {
        case 0: return launch_t<Utf8Range3T<false, 0>, false>(p, stream);
        case 1: return launch_t<Utf8Range3T<false, 1>, false>(p, stream);
        case 2: return launch_t<Utf8Range3T<false, 2>, false>(p, stream);
        case 3: return launch_t<Utf8Range3T<true, 0>, false>(p, stream);
        case 4: return launch_t<Utf8Range3T<true, 1>, false>(p, stream);
        default: return launch_t<Utf8Range3T<true, 2>, false>(p, stream);
        }
    }
    case kClsUtf8Range2x2: return launch_t<Utf8Range2x2, false>(p, stream);
    case kClsUtf16Ranges: {
        // ranges below U+8000 (1: af alone, 2: + a range of two-byte leads or of leads up to E7) / across it / above it
        const u32 ns = (p.n_ranges >> 4) & 1u, nh = (p.n_ranges >> 8) & 1u, nl = (p.n_ranges & 15u) == 3u ? 3u : (p.n_ranges & 15u) <= 1u && (ns | nh) ? 1u : 2u;   // (an unused slot is empty)
        const u32 bo = (p.big_endian ? 2u : 0u) | (p.parity & 1u);
        if (p.n_ranges >> 12) switch (bo) {   // an astral plane passes: surrogate pairs (one instantiation; unused slots are empty)
            case 0: return launch_t<Utf16RangesT<0, 0, 2, 1, 1, 1>, false>(p, stream);
            case 1: return launch_t<Utf16RangesT<0, 1, 2, 1, 1, 1>, false>(p, stream);
            case 2: return launch_t<Utf16RangesT<1, 0, 2, 1, 1, 1>, false>(p, stream);
            default: return launch_t<Utf16RangesT<1, 1, 2, 1, 1, 1>, false>(p, stream);
        }
#define SX_U16R(NL, NS, NH)                                                                                  \
        if (nl == NL && ns == NS && nh == NH) switch (bo) {                                                  \
            case 0: return launch_t<Utf16RangesT<0, 0, NL, NS, NH>, false>(p, stream);                       \
            case 1: return launch_t<Utf16RangesT<0, 1, NL, NS, NH>, false>(p, stream);                       \
            case 2: return launch_t<Utf16RangesT<1, 0, NL, NS, NH>, false>(p, stream);                       \
            default: return launch_t<Utf16RangesT<1, 1, NL, NS, NH>, false>(p, stream);                      \
        }
        SX_U16R(3, 0, 0) SX_U16R(1, 1, 0) SX_U16R(1, 0, 1) SX_U16R(1, 1, 1) SX_U16R(2, 0, 0) SX_U16R(2, 1, 0) SX_U16R(2, 0, 1) SX_U16R(2, 1, 1)
#undef SX_U16R
        break;
    }
    case kClsSingleByteRange: return launch_t<SingleByteRange, false>(p, stream);
    case kClsSingleByteRanges:
        if (p.n_ranges <= 2) return launch_t<SingleByteRanges<2>, false>(p, stream);
        if (p.n_ranges <= 4) return launch_t<SingleByteRanges<4>, false>(p, stream);
        return launch_t<SingleByteRanges<6>, false>(p, stream);
    default: break;
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------
// Synthetic background, read-bandwidth probe, sparse gather
// ------------------------------------------------------------------------------------------
SX_DEV u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void fill_kernel(uint8_t* dst, u64 first, u64 len, u64 seed) {
    // one 8-byte word of the stream per thread iteration; `first` and `dst` need not be aligned
    u64 w0 = first >> 3, w1 = (first + len + 7) >> 3;
    u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 w = w0 + (u64)blockIdx.x * blockDim.x + threadIdx.x; w < w1; w += stride) {
        u64 v = mix64(seed + (w + 1) * 0x9E3779B97F4A7C15ull);
        u64 i0 = w << 3;
        if (i0 >= first && i0 + 8 <= first + len && (((uintptr_t)(dst + (i0 - first))) & 7) == 0) {
            *(u64*)(dst + (i0 - first)) = v;
        } else {
            for (int k = 0; k < 8; k++) {
                u64 i = i0 + k;
                if (i >= first && i < first + len) dst[i - first] = (uint8_t)(v >> (8 * k));
            }
        }
    }
}

hipError_t launch_fill_background(uint8_t* dst, u64 first, u64 len, u64 seed, hipStream_t stream) {
    if (len == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, stream, dst, first, len, seed);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void read_sum_kernel(const u32x4* src, u64 n16, u64* out) {
    u64 stride = (u64)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        u32x4 v = __builtin_nontemporal_load(src + i);
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) atomicAdd((unsigned long long*)out, 1ull);  // keep the loads alive
}

// same traversal as scan_kernel (one wavefront streams a private sub-chunk in 1 KiB tiles, two
// tiles in flight) but no classification: the bandwidth this access pattern can reach at all
__global__ __launch_bounds__(256) void read_subchunk_kernel(const uint8_t* data, u64 len, u32 subchunk, u64* out, u32 rot) {
    const u32 lane = lane_id();
    const u64 wave = (u64)blockIdx.x * 4u + uniform(threadIdx.x >> 6);
    const u64 sub_start = wave * (u64)subchunk;
    if (sub_start >= len) return;
    const u64 sub_end = sub_start + subchunk < len ? sub_start + subchunk : len;
    const uint8_t* base_ptr = data + sub_start;
    const u32 base_lo = uniform((u32)(uintptr_t)base_ptr), base_hi = uniform((u32)((uintptr_t)base_ptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((u64)base_hi << 32) | base_lo), 0, (int)uniform((u32)(sub_end - sub_start)), 0x00020000);
    const int n_tiles = (int)((sub_end - sub_start) / kTileBytes);
    // rot (round 5, SX_PROBE_ROT): wavefront w begins at tile (w * rot) mod n_tiles of its sub-chunk and wraps around — the wavefronts
    // that are resident together then read at offsets spread over the whole sub-chunk instead of all near the same one
    const u32 span = (u32)n_tiles * kTileBytes;
    u32 o0 = rot && n_tiles ? (u32)((wave * rot) % (u64)n_tiles) * kTileBytes : 0u;
    auto wrap = [&](u32 o) { return o >= span ? o - span : o; };
    u32x4 cur = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(o0 + lane * 16u), 0, 0);
    u32x4 nxt = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(wrap(o0 + kTileBytes) + lane * 16u), 0, 0);
    u32 acc = 0, off = wrap(o0 + 2 * kTileBytes);
    for (int t = 0; t < n_tiles; t++) {
        u32x4 nn = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + lane * 16u), 0, 0);
        acc += cur.x ^ cur.y ^ cur.z ^ cur.w;
        cur = nxt; nxt = nn; off = wrap(off + kTileBytes);
    }
    if (acc == 0x12345678u) atomicAdd((unsigned long long*)out, 1ull);
}

// Access-pattern probe (tools/gpu_probe5.py): LOADS x 1 KiB per iteration and wavefront, two
// iterations in flight.  MODE 0: a private sub-chunk per wavefront (the scan kernels' traversal);
// MODE 1: the four wavefronts of a block share a sub-chunk of 4 x subchunk and take its tiles in turn.
// Measured on MI355X, 32 GiB: grid-stride 6.97 TB/s; sub-chunk 16 KiB 6.74, 64 KiB-1 MiB 6.1-6.3 TB/s
// whatever LOADS and MODE: the traversal itself caps a scan kernel at ~6.2 TB/s.
template <int LOADS, int MODE>
__global__ __launch_bounds__(256) void read_pattern_kernel(const uint8_t* data, u64 len, u32 subchunk, u64* out) {
    const u32 lane = lane_id();
    const u32 wib = uniform(threadIdx.x >> 6);
    u64 sub_start, span;
    u32 first, step;
    if (MODE == 0) { sub_start = ((u64)blockIdx.x * 4u + wib) * subchunk; span = subchunk; first = 0; step = LOADS * kTileBytes; }
    else { sub_start = (u64)blockIdx.x * 4u * subchunk; span = 4ull * subchunk; first = wib * LOADS * kTileBytes; step = 4u * LOADS * kTileBytes; }
    if (sub_start >= len) return;
    const u64 sub_end = sub_start + span < len ? sub_start + span : len;
    const uint8_t* base_ptr = data + sub_start;
    const u32 base_lo = uniform((u32)(uintptr_t)base_ptr), base_hi = uniform((u32)((uintptr_t)base_ptr >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((u64)base_hi << 32) | base_lo), 0, (int)uniform((u32)(sub_end - sub_start)), 0x00020000);
    u32x4 cur[LOADS], nxt[LOADS];
    u32 off = first + lane * 16u;
#pragma unroll
    for (int k = 0; k < LOADS; k++) {
        cur[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + k * kTileBytes), 0, 0);
        nxt[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + step + k * kTileBytes), 0, 0);
    }
    u32 acc = 0;
    for (u32 o = first; o < (u32)(sub_end - sub_start); o += step) {
        u32x4 nn[LOADS];
#pragma unroll
        for (int k = 0; k < LOADS; k++) nn[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(off + 2 * step + k * kTileBytes), 0, 0);
#pragma unroll
        for (int k = 0; k < LOADS; k++) { acc += cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w; cur[k] = nxt[k]; nxt[k] = nn[k]; }
        off += step;
    }
    if (acc == 0x12345678u) atomicAdd((unsigned long long*)out, 1ull);
}

hipError_t launch_read_sum(const uint8_t* src, u64 len, u64* out, hipStream_t stream, uint32_t subchunk) {
    if (subchunk) {
        const int loads = getenv("SX_PROBE_LOADS") ? atoi(getenv("SX_PROBE_LOADS")) : 0;
        const int mode = getenv("SX_PROBE_MODE") ? atoi(getenv("SX_PROBE_MODE")) : 0;
        u64 waves = (len + subchunk - 1) / subchunk;
        const dim3 grid((unsigned)((waves + 3) / 4));
        if (loads == 0 && mode == 0)
            hipLaunchKernelGGL(read_subchunk_kernel, grid, dim3(256), 0, stream, src, len, subchunk, out, (u32)(getenv("SX_PROBE_ROT") ? atoi(getenv("SX_PROBE_ROT")) : 0));
#define SX_RP(L, M) else if (loads == L && mode == M) hipLaunchKernelGGL((read_pattern_kernel<L, M>), grid, dim3(256), 0, stream, src, len, subchunk, out)
        SX_RP(1, 0); SX_RP(2, 0); SX_RP(4, 0); SX_RP(1, 1); SX_RP(2, 1); SX_RP(4, 1);
#undef SX_RP
        else return hipErrorInvalidValue;
    } else {
        hipLaunchKernelGGL(read_sum_kernel, dim3(256 * 8), dim3(256), 0, stream, (const u32x4*)src, len / 16, out);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gather_kernel(const uint8_t* src, uint8_t* dst, const u64* seg_src,
                                                     const u64* seg_dst, const u32* seg_len, u32 n) {
    // one wavefront per segment
    u32 wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    u32 nw = (gridDim.x * 256u) >> 6;
    for (u32 i = wave; i < n; i += nw) {
        const uint8_t* s = src + seg_src[i];
        uint8_t* d = dst + seg_dst[i];
        u32 l = seg_len[i];
        for (u32 k = lane; k < l; k += 64) d[k] = s[k];
    }
}

hipError_t launch_gather(const uint8_t* src, uint8_t* dst, const u64* seg_src, const u64* seg_dst,
                         const u32* seg_len, u32 n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    u32 blocks = (n + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(gather_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, seg_src, seg_dst, seg_len, n);
    return hipGetLastError();
}

}  // namespace sx
