// sx_wave_dev.hip — stage B for string-dense Missions, wave-cooperative (gfx950, wave64).
//
// The lane-per-region replay (sx_replay_dev.hip) decodes byte by byte under full divergence: 0.7 % of the HBM
// peak on the bytes it touches.  Here a wavefront takes 64 CONSECUTIVE windows (8 KiB at W = 128) together:
//   1. it streams their bytes as stage A does (16 bytes per lane, coalesced), classifies them, and leaves
//      the per-byte masks (valid / accepted / UTF-8 length) in LDS as bit arrays;
//   2. lane i pulls window i's 128 bits of every mask out of LDS and runs FindingCollection::from over them
//      as bit arithmetic (sx_wave_core.hpp: decoder calls from the "invalid" bits, SplitStr's stretches and
//      q-char cuts from find-first-set / popcount);
//   3. the state a window carries into the next one (leftover, cut flag) travels lane to lane with one DPP move;
//      the lanes iterate until no lane's entry state changes (two rounds on all but contrived input: only a
//      window's first stretch depends on what is carried in);
//   4. findings per lane -> wave prefix sum -> (pass 2) records and strings written in order.
// A wavefront owns `nwin` consecutive windows and walks them batch by batch (the last lane's state is the next
// batch's entry: an SGPR); it starts kWvWarm windows early with "nothing carried" — the state is a function of
// the three windows in front —, and wave_verify_kernel checks afterwards that every wavefront's assumed entry state is
// what its predecessor really left (if not — input built for it — the Mission falls back to the lane-per-region path).
// Two passes: count (per-wavefront totals -> exclusive scan), write.  No atomics; output order = position order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "sx_device.hpp"

namespace sx {
#define SXD __device__ __forceinline__
#define SXD_NOINLINE __device__ __attribute__((noinline))
}  // namespace sx
#include "sx_wave_core.hpp"

namespace sx {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

SXD u32 wv_lane() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
SXD u32 wv_from_prev(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xF, 0xF, false); }   // lane i <- lane i-1; lane 0 <- edge
SXD u32 wv_uniform(u32 v) { return __builtin_amdgcn_readfirstlane(v); }
SXD u32 wv_shfl(u32 v, u32 src) { return __builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v); }
SXD u32 wv_scan_incl(u32 v, u32 lane) {
#pragma unroll
    for (u32 d = 1; d < 64; d <<= 1) {
        const u32 o = wv_shfl(v, lane >= d ? lane - d : lane);
        if (lane >= d) v += o;
    }
    return v;
}

// the finding record and its string (single byte: transcoded byte by byte through the decoder table).  FAM: the family's code only —
// the two-byte family's probe brings 384 B of scratch with it.  f / a: where the record and the string go, a_off: the string's offset
// in the segment's arena, win_pos: buffer offset of the window the finding belongs to.
template <int FAM>
SXD void wv_write_finding(const WaveParams& P, u64 fi, u8* a, u64 a_off, u64 win_pos, u32 din, u32 prec, bool completes, i32 src_rel,
                          u32 src_len, u32 out_len) {
    const u64 soff = win_pos / kWvSlice * kWvSlice;
    sx_finding r;
    r.position = P.consumed0 + win_pos + din;
    r.str_off = (u32)(a_off + P.str_off_base);
    r.str_len = out_len;
    if ((prec & 0xFFu) == WV_PROBE) {   // sx_wave_core.hpp WV_PROBE: this call starts at the slice's byte 0
        const u32 lb = (prec >> 8) & 511u, lback = (prec >> 17) & 1023u, hb = wv_probe_hb(prec), pend = wv_probe_pend(prec);
        const u64 avail = P.len - win_pos;
        if (FAM >= 4)   // (a leftover at a second call at byte 0: the `pend` bytes in front of the slice were the pending token's, not the leftover's)
            prec = wv_resolve_probe_dbcs((int)P.encoding, P.table, P.data + win_pos, avail < 32 ? (u32)avail : 32u,
                                         P.data + (win_pos - lback), lb ? lback - pend : 0u, lb, hb);
        else if (FAM == 1) prec = wv_resolve_probe(P.data + win_pos, avail < 32 ? (u32)avail : 32u, P.data + (win_pos - lback), lb);
        else prec = WV_EXACT;   // (single-byte decoders never leave the probe open)
    }
    r.precision = (u8)prec;
    r.completes_previous = completes ? 1 : 0;
    r.mission_id = (u8)P.mission_id;
    r.reserved = 0;
    r.input_file_id = (int16_t)P.file_id;
    r.reserved2 = 0;
    r.slice_index = (u32)(soff / kWvSlice) + P.slice_base;
    if (P.packed) {   // the record as it crosses PCIe (include/stringsext_amd.h sx_finding16)
        sx_finding16 p;
        p.position = r.position; p.str_off = r.str_off; p.str_len = (uint16_t)out_len;
        p.flags = (u8)((prec & 3u) | (completes ? 4u : 0u)); p.mission_id = (u8)P.mission_id;
        ((sx_finding16*)P.findings)[fi] = p;
    } else P.findings[fi] = r;
    const u8* s = P.data + (u64)((long long)win_pos + src_rel);
    if (FAM >= 4) (void)wv_transcode_dbcs((int)P.encoding, P.table, s, src_len, a);
    else if (FAM == 2) (void)wv_transcode_utf16(P.encoding == (u32)kEncUtf16be, s, src_len, a);
    else if (out_len == src_len) {     // every char is one byte on both sides (ASCII; UTF-8 input): unaligned wide copies, 16 / 8 / 4 / 2 / 1 bytes
        u32 t = 0;
        for (; t + 16 <= src_len; t += 16) { u32x4 v; __builtin_memcpy(&v, s + t, 16); __builtin_memcpy(a + t, &v, 16); }
        if (t + 8 <= src_len) { u64 v; __builtin_memcpy(&v, s + t, 8); __builtin_memcpy(a + t, &v, 8); t += 8; }
        if (t + 4 <= src_len) { u32 v; __builtin_memcpy(&v, s + t, 4); __builtin_memcpy(a + t, &v, 4); t += 4; }
        if (t + 2 <= src_len) { uint16_t v; __builtin_memcpy(&v, s + t, 2); __builtin_memcpy(a + t, &v, 2); t += 2; }
        if (t < src_len) a[t] = s[t];
    } else {
        u32 w = 0;
        for (u32 t = 0; t < src_len; t += 4) {   // four source bytes per load
            u32 x4;
            const u32 k = src_len - t < 4 ? src_len - t : 4u;
            if (k == 4) __builtin_memcpy(&x4, s + t, 4);
            else { x4 = s[t]; if (k > 1) x4 |= (u32)s[t + 1] << 8; if (k > 2) x4 |= (u32)s[t + 2] << 16; }
            for (u32 j = 0; j < k; j++) {
                const u32 b = (x4 >> (8 * j)) & 0xFFu;
                if (b < 0x80) a[w++] = (u8)b;
                else w += dput_cp(a + w, P.table ? (u32)P.table[b - 0x80] : 0xF780u + (b - 0x80u));
            }
        }
    }
}

// the window-parallel writer's emitter (pass 2 = the count pass once more, writing)
template <int FAM> struct WriteEmit {
    const WaveParams* P;
    u64 f;                // next record of this lane (its index in the launch's segment)
    u8* a;                // next string byte of this lane
    u64 a_off;            // ... its offset in the segment's string arena
    u64 win_pos;          // buffer offset of the window
    SXD void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
        wv_write_finding<FAM>(*P, f, a, a_off, win_pos, din, prec, completes, src_rel, src_len, out_len);
        f++; a += out_len; a_off += out_len;
    }
};

// the lanes' LDS traffic of one wavefront in order (a block of several wavefronts shares only read-only tables: no block barrier)
template <int WPB> SXD void wave_lds_sync() {
    if (WPB == 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
}

// masks a wavefront keeps per batch (16 bits per lane and tile each).  CLS 1: the classes come from ranges (sx_device.hpp WvSwar): a
// single-byte Mission then stores accepted / >= 0x80 only, a two-byte one E, A, F, MA, MB (G and the lengths follow from them)
constexpr int wv_n_masks(int fam, int cls) { return fam == 5 ? 5 : fam == 4 ? (cls ? 5 : 9) : fam == 1 ? (cls ? 5 : 6) : fam == 2 ? 4 : cls ? 2 : 4; }
constexpr u32 kMaskWords = kWvMaxTiles * 32 + 8;
constexpr u32 wv_lds_words(int fam, int cls, int opt = 0) {   // ... and the descriptors staged in their place need kWvStage x 3 x 64 words (-r: kWvStageSame)
    return (u32)wv_n_masks(fam, cls) * kMaskWords > (opt >= 2 ? kWvStageSame : kWvStage) * 192u ? (u32)wv_n_masks(fam, cls) * kMaskWords : (opt >= 2 ? kWvStageSame : kWvStage) * 192u;
}

// MODE 0: count; 1: write.  FAM 0: single-byte decoders; 1: UTF-8; 4: the two-byte family (Big5, Shift_JIS, EUC-KR: 4 wavefronts
// per block share the 32 KB of pair codes in LDS).
// Round 6: NO kernel that holds the exchange loop may spill a vector register (tests/test_kernel_resources.py) — in round 5 the -r kernels,
// at 128 registers with 30 to 65 of them spilled, brought a value that lives across that loop back wrong in the lanes that had run it twice;
// the cause was never found (see profiles/r06_spill_note.md), so the loop only ever runs in kernels without spill code.  Single byte and
// UTF-8: three wavefronts per SIMD (168 registers; at four / 128 they spilled 2 - 28) — text, -e ascii -n 4 and Russian text run as fast
// (profiles/r06i_*: the A/B); UTF-16: two.  The one exception, named in the test: the two-byte families' COUNT kernels with SWAR classes
// (<0, 4, 4, 1, 0>, <0, 5, 4, 1, 0>: 65 / 63 spilled at four wavefronts, none at two) stay at four — at two BASELINE config 5 runs 387 -> 465 ms
// per step (r06i) —; 60 k GPU fuzz cases of round 5 and this round's ran clean on them, and their writers (<1, 4 / 5, ..>) do not spill.
#ifndef SX_WV_OCC0
#define SX_WV_OCC0 3   // wavefronts per SIMD the compiler is asked for, count pass: single-byte / UTF-8 / two-byte family
#define SX_WV_OCC1 3
#define SX_WV_OCC4 2
#endif
#ifndef SX_WV_OCC2
#define SX_WV_OCC2 2   // ... UTF-16
#endif
#ifndef SX_WV_OCCW0
#define SX_WV_OCCW0 SX_WV_OCC0   // ... write pass
#define SX_WV_OCCW1 SX_WV_OCC1
#define SX_WV_OCCW4 SX_WV_OCC4
#endif
#ifndef SX_WV_OCC4S
#define SX_WV_OCC4S 4   // ... the two-byte family with SWAR classes and 2-bit pair codes (40 KB of LDS per block of four wavefronts)
#endif
#ifndef SX_WV_OCCG
#define SX_WV_OCCG 2   // the -g kernels (OPT 1) likewise: no spilled register at 256 (17 to 81 at 128), and text with -g runs as fast (72 / 70 / 51 GiB/s)
#endif
#ifndef SX_WV_OCCS
#define SX_WV_OCCS 2   // the -r kernels (OPT 2): 256 registers each — at 128 they spill 30 to 65 of them, and came out wrong (see the writer's `ws2`)
#endif
constexpr int wv_occ(int mode, int fam, int cls, int opt = 0) {
    if (opt >= 2) return SX_WV_OCCS;
    if (opt == 1) return SX_WV_OCCG;
    if (mode == 0 && fam >= 4 && cls) return SX_WV_OCC4S;
    if (fam == 2) return SX_WV_OCC2;
    return mode == 0 ? (fam >= 4 ? SX_WV_OCC4 : fam == 1 ? SX_WV_OCC1 : SX_WV_OCC0) : (fam >= 4 ? SX_WV_OCCW4 : fam == 1 ? SX_WV_OCCW1 : SX_WV_OCCW0);
}
// GREP: the Mission has -g (round 5) — a compile-time constant: WvWin::GC and the grep rules cost registers a Mission without -g must not pay for
// (OPT 0: neither; 1: -g; 2: -r, families 0 - 2 — WvWin::MBA / D and wv_stretch_same, round 5; 3: both)
template <int MODE, int FAM, int WPB, int CLS, int OPT>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(wv_occ(MODE, FAM, CLS, OPT)))) void wave_replay_kernel(const WaveParams P) {
    // FAM 0: valid, accepted, O2, O3 (CLS 1: accepted, >= 0x80); FAM 1: E, A, F, MA, MB, G; FAM 4: E, A, F, MA, MB, G, O2, O3, O4 — 16 bits per lane and tile
    __shared__ u32 lds_all[WPB][wv_lds_words(FAM, CLS, OPT)];
    __shared__ u8 lds_lut[FAM == 2 ? 512 : CLS ? 4 : 256];
    __shared__ u32 lds_pairs[FAM == 5 ? kWvJisWords + 3 : FAM == 4 ? (CLS ? 4096 : 8192) : 1];
    const u32 lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
    if (!CLS && threadIdx.x < 64) ((u32*)lds_lut)[threadIdx.x] = ((const u32*)P.lut)[threadIdx.x];
    if (FAM == 2 && threadIdx.x < 64) ((u32*)lds_lut)[FAM == 2 ? 64 + threadIdx.x : 0] = ((const u32*)P.lut)[64 + threadIdx.x];   // (UTF-16: 512 bytes)
    if (FAM == 4) for (u32 i = threadIdx.x; i < (CLS ? 4096u : 8192u); i += 64 * WPB) lds_pairs[i] = CLS ? P.pairs2[i] : P.pairs[i];
    if (FAM == 5) for (u32 i = threadIdx.x; i < kWvJisWords; i += 64 * WPB) lds_pairs[i] = P.pairs2[i];
    __syncthreads();
    u32* const lds_base = lds_all[wib];
    auto lds_mask = [&](int k) -> u32* { return lds_base + (u32)k * kMaskWords; };
    const WvSwar& SW = P.swar;

    const u64 v = P.v0 + (u64)blockIdx.x * WPB + wib;
    const u64 own_start = P.g_lo + v * P.nwin;
    if (v >= P.v1 || own_start >= P.g_hi) return;
    const u64 own_end = own_start + P.nwin < P.g_hi ? own_start + P.nwin : P.g_hi;
    // (round 5) the entry state of the first own window is KNOWN — what the wavefront in front left: a repair launch of the count pass
    // (WaveParams::redo), the writer after repairs (use_entry) — and there are no warm-up windows
    bool known = false;
    u32 known_state = 0;
    if (v != 0 && ((MODE == 0 && P.redo) || (MODE == 1 && P.use_entry))) {
        known_state = __hip_atomic_load(P.wave_out + (v - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 0) {
            const u32 mine = P.wave_in[v];
            if (mine == known_state || mine == 0xFFFFFFFEu) return;   // its assumption was right (or it gave the buffer back: nothing to repair)
        }
        known = true;
    }
    const u64 gw = (v == 0 || known) ? own_start : own_start - kWvWarm;
    u32 carry = v == 0 ? P.inject : (known ? known_state : 0u);   // entry state of the batch's first window
    u32 assumed_in = carry;
    u32 tot_f = 0, tot_b = 0;
    u32 dbcs_cov = 0;   // FAM 4: bytes at the next tile's start that belong to a token begun before it (0 / 1)
    bool dbcs_valid = false;   // ... known for the next batch's first tile
    u64 ref_pos = ~0ull;       // two-byte family: a tile start of the batch before at which the hang-over (ref_cov) is known
    u32 ref_cov = 0;
    u64 fbase = 0, abase = 0;
    if (MODE == 1) { fbase = P.wave_fbase[v] - P.f_sub; abase = P.wave_abase[v] - P.a_sub; }   // relative to this launch's output segment
    constexpr bool GREP = OPT == 1 || OPT == 3, SAME = OPT == 2 || OPT == 3;
    const WvParams WP{ P.q, P.n_min, GREP ? 1u : 0u, SAME ? 1u : 0u };

    for (u64 g0 = gw; g0 < own_end; g0 += kWvBatch) {
        const u64 g = g0 + lane;
        const bool active = g < own_end, owned = active && g >= own_start;
        u64 ws = 0;
        u32 wn = 0;
        if (active) wv_window_at(g, P.W, P.wps, P.len, &ws, &wn);
        const u32 n_act = own_end - g0 < kWvBatch ? (u32)(own_end - g0) : kWvBatch;
        const u64 span_lo = ((u64)wv_uniform((u32)(ws >> 32)) << 32) | wv_uniform((u32)ws);
        const u64 we = ws + wn;
        const int last_lane = (int)wv_uniform(n_act - 1);
        // (the builtin returns int: without the casts a low half >= 2^31 sign-extends into the high one)
        const u64 span_hi = ((u64)(u32)__builtin_amdgcn_readlane((u32)(we >> 32), last_lane) << 32) | (u64)(u32)__builtin_amdgcn_readlane((u32)we, last_lane);
        const u64 tile0 = wv_tile0(span_lo);
        const u32 n_tiles = (u32)((span_hi - tile0 + kTileBytes - 1) / kTileBytes);

        // ---- 1. classify the batch's bytes; masks -> LDS
        wave_lds_sync<WPB>();   // (the previous batch's readers are done)
        // FAM 4: where tokens start at tile0 follows from the bytes in front of it: the wavefront's first batch walks back to a tile
        // that holds a byte outside the lead range (behind such a byte a token starts whatever came before); later batches go on
        // from the batch before.  At the buffer's byte 0 the token pending on entry ends after P.entry_skip bytes.
        int t_first = 0;
        const u64 next_t0 = wv_tile0(span_hi);   // the next batch's first tile (the batches' windows are contiguous)
        bool have_next = false;
        u32 cov_next = 0;
        if (FAM >= 4 && (g0 == gw || !dbcs_valid)) {
            dbcs_cov = 0;
            long long lo = (long long)tile0;
            // Two-byte family (round 4): inside a stretch of lead-range bytes every token has two bytes, so the hang-over at tile0 follows
            // from a position where it is KNOWN by the parity of the distance — the walk need not find the stretch's beginning (a format
            // fill of 0xF6 in a disk image is gigabytes of it; round 3: give up after 64 KiB, and the buffer went to the host replay at
            // 0.18 GiB/s).  Known: the last tile start of the batch before (later batches), or the first tile of the wavefront in front
            // (first batch): it publishes its hang-over there as soon as it has it (P.wave_grid; wavefronts are dispatched in order).
            long long stop = -1;
            bool special = false;   // EUC-JP: an 8E / 8F among the walked bytes — tokens of other lengths, the parity does not hold
            if ((FAM == 4 || FAM == 5) && P.wave_grid) {
                if (g0 != gw && ref_pos != ~0ull) stop = (long long)ref_pos;
                else if (g0 == gw && v > 0) {
                    const u64 gp = v - 1 == 0 ? P.g_lo : P.g_lo + (v - 1) * P.nwin - kWvWarm;
                    u64 wsp; u32 wnp;
                    wv_window_at(gp, P.W, P.wps, P.len, &wsp, &wnp);
                    stop = (long long)wv_tile0(wsp);
                }
            }
            while (lo > 0) {
                lo -= kTileBytes; t_first--;
                const long long o = lo + 16ll * lane;
                bool reset = o < 0;
                if (!reset) {
                    const u32x4 x = *(const u32x4*)(P.data + o);
                    const u32 xs[4] = { x.x, x.y, x.z, x.w };
                    if (FAM == 5) {
                        const u32 x5[5] = { xs[0], xs[1], xs[2], xs[3], 0u };
                        reset = (wv_eucjp_classes_swar<1>(SW, x5, 16u).lr & 0xFFFFu) != 0xFFFFu;
                        u32 sp8 = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) { const u32 y = (xs[k] & 0xFEFEFEFEu) ^ 0x8E8E8E8Eu; sp8 |= (y - 0x01010101u) & ~y & 0x80808080u; }   // a byte 8E or 8F
                        special = special || __ballot(sp8 != 0) != 0;
                    }
                    else if (CLS) reset = wv_dbcs_classes_swar<1>(SW, xs, 16u).lr != 0xFFFFu;
                    else for (int k = 0; k < 16; k++) reset = reset || !(lds_lut[(xs[k >> 2] >> (8 * (k & 3))) & 0xFFu] & WVC_LEAD);
                }
                if (__ballot(reset)) break;
                if (FAM == 5 && special) stop = -1;   // (as before round 5: walk on, give up after 64 KiB)
                if (stop >= 0 && lo <= stop) {   // nothing but lead-range bytes from the known position to tile0 (EUC-JP, round 5: and no 8E / 8F — two bytes per token)
                    u32 kc = ref_cov;
                    if (g0 == gw) {
                        u32 fv = 0;
                        if (lane == 0) { while (((fv = __hip_atomic_load(P.wave_grid + (v - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1u) == 0) __builtin_amdgcn_s_sleep(2); }
                        if (wv_uniform(fv) >> 31) {   // the wavefront in front gave the buffer back: so does this one
                            if (lane == 0) {
                                P.wave_in[v] = 0xFFFFFFFEu; P.wave_out[v] = 0xFFFFFFFDu; P.wave_nf[v] = 0; P.wave_nb[v] = 0;
                                if (!known) __hip_atomic_store(P.wave_grid + v, 0x80000001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                            return;
                        }
                        kc = (wv_uniform(fv) >> 1) & (FAM == 5 ? 3u : 1u);   // (EUC-JP: a token of three bytes can hang over by two)
                    }
                    const u64 first = (u64)stop + kc;                       // a token starts here, and every two bytes from here on
                    dbcs_cov = tile0 >= first ? (u32)((tile0 - first) & 1ull) : (u32)(first - tile0);
                    t_first = 0;                                           // (no look-back tile needs classifying)
                    break;
                }
                if (stop < 0 && t_first < -64) {   // 64 KiB of lead-range bytes and no end, and no known position to count from (EUC-JP: token lengths differ).  This wavefront gives up — an entry state no wavefront
                                       // ever leaves makes the verification fail, and the lane-per-region path takes the buffer
                    if (lane == 0) {
                        P.wave_in[v] = 0xFFFFFFFEu; P.wave_out[v] = 0xFFFFFFFDu; P.wave_nf[v] = 0; P.wave_nb[v] = 0;
                        // (the wavefront behind may be waiting for my hang-over: it gets "gave up" instead, and does the same)
                        if (P.wave_grid && g0 == gw && !known) __hip_atomic_store(P.wave_grid + v, 0x80000001u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
            }
        }
        // The batch's tiles through a buffer descriptor [tile0, the buffer's end rounded up to 16 — as the scan kernels read it): beyond it
        // the loads give zeros, and what lies between the buffer's end and the next multiple of 16 is masked by `avail`.  Four tiles are in
        // flight per wavefront (round 3: one, behind per-tile 64-bit bounds arithmetic and a byte-by-byte tail loop); bounds are 32-bit
        // offsets from tile0.
        const u64 rem0 = P.len - tile0;                                            // bytes from tile0 to the buffer's end (> 0: a window starts behind tile0)
        const u32 rem32 = rem0 > (1ull << 30) ? (1u << 30) : (u32)rem0;
        const u32 span_cap = kWvMaxTiles * kTileBytes + 16u;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(P.data + tile0), 0, (int)wv_uniform(rem32 > span_cap ? span_cap : ((rem32 + 15u) & ~15u)), 0x00020000);
        const bool full = rem32 >= n_tiles * kTileBytes + 4u;                      // no lane of this batch is near the buffer's end
        auto issue = [&](int t) -> u32x4 {
            u32x4 x = { 0, 0, 0, 0 };
            if (t < (int)n_tiles) x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((u32)t * kTileBytes + 16u * lane), 0, 0);
            return x;
        };
        u32x4 xa = issue(0), xb = issue(1), xc = issue(2), xd = issue(3);
        // the four bytes in front of tile0 and behind the batch's last tile (the same address in every lane)
        u32 edge_back = tile0 >= 4 ? *(const u32*)(P.data + (tile0 - 4)) : 0u;
        u32 lead_lo = 0, lead_hi = 0;   // -r: the lead bytes that pass ubf among this batch's bytes (wave-uniform)
        u32 u16_carry = 0;   // UTF-16: what lane 63 of the tile before hands on (bit 0 its last unit is a high surrogate, 1 whose character passes, 2 the unit behind it is read in slow mode)
        bool u16_exo = false;   // ... a case the masks cannot say: the wavefront gives the buffer back
        u32 euc_spill = 0;   // EUC-JP: marks of the tile's last tokens that lie on the next tile's first two bytes (five masks x 2 bits)
        const u32 edge_after = FAM == 0 ? 0u : (u32)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(n_tiles * kTileBytes), 0, 0);

        // one tile: t < 0 = the two-byte family's way back to a token boundary (nothing is stored for those)
        auto do_tile = [&](int t, u32x4 x, u32 back_edge, u32 ahead_edge) {
            const int rel = t * (int)kTileBytes + 16 * (int)lane;                  // the lane's first byte, from tile0
            const bool before = (long long)tile0 + rel < 0;                        // (in front of the buffer: a look-back tile that begins there)
            const int left = (int)rem32 - rel;
            const u32 avail = before ? 0u : (left >= 16 ? 16u : (left > 0 ? (u32)left : 0u));
            if (!full || t < 0) {   // near the buffer's ends: bytes that do not exist are zero
                const u32 xs[4] = { x.x, x.y, x.z, x.w };
                u32 ys[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { const int nb = (int)avail - 4 * k; ys[k] = nb >= 4 ? xs[k] : (nb <= 0 ? 0u : xs[k] & ((1u << (8 * nb)) - 1u)); }
                x.x = ys[0]; x.y = ys[1]; x.z = ys[2]; x.w = ys[3];
            }
            const u32 idx = (u32)(t < 0 ? 0 : t) * 64 + lane;
            if (FAM == 0 && CLS) {
                WvMasks16R m;
                if (SW.n <= 1) m = wv_classify16_single_swar<1>(SW, x.x, x.y, x.z, x.w, avail);
                else if (SW.n <= 3) m = wv_classify16_single_swar<3>(SW, x.x, x.y, x.z, x.w, avail);
                else if (SW.n <= 4) m = wv_classify16_single_swar<4>(SW, x.x, x.y, x.z, x.w, avail);   // (KOI8-R + Cyrillic: 20..7E, A3, B3, C0..FF)
                else m = wv_classify16_single_swar<6>(SW, x.x, x.y, x.z, x.w, avail);
                ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.a;
                ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.hi;
            } else if (FAM == 0) {
                const WvMasks16 m = wv_classify16_single(lds_lut, x.x, x.y, x.z, x.w, avail);
                ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.v;
                ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.a;
                ((uint16_t*)lds_mask(CLS ? 0 : 2))[idx] = (uint16_t)m.o2;
                ((uint16_t*)lds_mask(CLS ? 0 : 3))[idx] = (uint16_t)m.o3;
            } else {
                // the four bytes in front of the lane's 16 and the four behind them: the neighbours' registers (one DPP move each); lane 0
                // and lane 63 get the tile's edges
                const bool has_back = !before && (long long)tile0 + rel >= 4 && avail;
                u32 back = wv_from_prev(x.w, back_edge);
                if (!has_back) back = 0;
                const int left2 = left - 16;
                const u32 n_ahead = avail == 16 ? (left2 >= 4 ? 4u : (left2 > 0 ? (u32)left2 : 0u)) : 0u;
                u32 ahead = __builtin_amdgcn_update_dpp(ahead_edge, x.x, 0x130, 0xF, 0xF, false);   // lane i <- lane i + 1, lane 63 <- the edge
                if (n_ahead < 4) ahead = n_ahead ? ahead & ((1u << (8 * n_ahead)) - 1u) : 0u;
                const u32 ws6[6] = { back, x.x, x.y, x.z, x.w, ahead };
                const u32 have_lo = has_back ? 0u : 4u, have_hi = 4u + avail + n_ahead;
                if (FAM == 1 && P.lead_set) {
                    // -r (WaveParams::lead_set): bytes C0.. among mine?  Text: hardly ever, and then of one kind
                    const u32 hi2 = ((x.x & (x.x << 1)) | (x.y & (x.y << 1)) | (x.z & (x.z << 1)) | (x.w & (x.w << 1))) & 0x80808080u;
                    if (__ballot(hi2 != 0)) {
                        u32 slo = 0, shi = 0;
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const u32 b = (ws6[1 + (k >> 2)] >> (8 * (k & 3))) & 0xFFu;
                            if (b >= 0xC2u && b <= 0xF4u && ((P.ubf >> (b & 0x3Fu)) & 1ull)) { if ((b & 0x3Fu) < 32u) slo |= 1u << (b & 31u); else shi |= 1u << (b & 31u); }
                        }
#pragma unroll
                        for (u32 d = 1; d < 64; d <<= 1) { slo |= wv_shfl(slo, lane ^ d); shi |= wv_shfl(shi, lane ^ d); }
                        lead_lo |= wv_uniform(slo); lead_hi |= wv_uniform(shi);
                    }
                }
                if (FAM == 2) {
                    // UTF-16 (sx_wave_core.hpp wv_classify16_utf16): eight units per lane; which of them begin a window; from the lane in front:
                    // is its last unit a high surrogate, does the character that one begins pass, is my first unit read in slow mode
                    const bool be = P.encoding == (u32)kEncUtf16be;
                    const u32 n_units = avail >> 1;
                    const u32 xs4[4] = { x.x, x.y, x.z, x.w };
                    const WvU16Lane L = wv_utf16_lane_units(lds_lut, be, xs4, n_units);
                    if (P.lead_set && __ballot((L.len2 | L.hm) != 0)) {   // -r (WaveParams::lead_set): the UTF-8 lead bytes of the units beyond ASCII that pass ubf
                        u32 slo = 0, shi = 0;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            if ((u32)j >= n_units) continue;
                            const u32 raw = (xs4[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, u = be ? ((raw & 0xFFu) << 8) | (raw >> 8) : raw;
                            if (u < 0x80u || (u & 0xFC00u) == 0xDC00u) continue;   // (an astral character's lead byte follows from its high surrogate)
                            const u32 code = (u & 0xFC00u) == 0xD800u ? 0x30u | (((0x10000u + ((u & 0x3FFu) << 10)) >> 18) & 7u) : u < 0x800u ? u >> 6 : 0x20u | (u >> 12);
                            if ((P.ubf >> code) & 1ull) { if (code < 32u) slo |= 1u << code; else shi |= 1u << (code & 31u); }
                        }
#pragma unroll
                        for (u32 d = 1; d < 64; d <<= 1) { slo |= wv_shfl(slo, lane ^ d); shi |= wv_shfl(shi, lane ^ d); }
                        lead_lo |= wv_uniform(slo); lead_hi |= wv_uniform(shi);
                    }
                    const u32 r = (u32)((tile0 + (u64)rel) & (kWvSlice - 1));
                    const u32 rw = r % P.W;
                    u32 wbm = r == kWvSlice - 16 ? 0x100u : 0u;                        // (a slice begins behind the lane)
                    for (u32 d = rw ? P.W - rw : 0u; d <= 16u; d += P.W) wbm |= 1u << (d >> 1);
                    if (left >= 0 && left <= 16) wbm |= 1u << ((u32)left >> 1);        // the buffer's end ends a window too
                    if (t == 0) {   // the batch's first tile: from the dword in front of it
                        u16_carry = 0;
                        if (tile0 >= 4) {
                            const u32 raw = edge_back >> 16;
                            const WvU16Unit pu = wv_utf16_unit(lds_lut, be ? ((raw & 0xFFu) << 8) | (raw >> 8) : raw);
                            // (whether my first unit is read in slow mode is not known here; it cannot matter 16 bytes on — unless all of lane 0's units are high surrogates)
                            u16_carry = (pu.kind == 1 ? 5u : 0u) | (pu.acc << 1);
                        }
                    }
                    const bool transparent = L.hm == 0xFFu && (wbm & 0x1FEu) == 0;
                    if (transparent || (t == 0 && lane == 0 && (u16_carry & 1u) && L.hm == 0xFFu)) u16_exo = true;
                    const u32 own_info = ((L.hm >> 7) & 1u) | (((L.accm >> 7) & 1u) << 1) | ((wv_utf16_chain(L.hm, wbm, 0u) >> 8) << 2);
                    u32 in_info = wv_from_prev(own_info, u16_carry);   // (every lane takes part in the move: no lane-dependent condition around it)
                    if (!has_back) in_info = 0;
                    u16_carry = (u32)__builtin_amdgcn_readlane(own_info, 63);
                    u32 edge_l = 0;
                    { const u32 raw = ahead_edge & 0xFFFFu; edge_l = wv_utf16_unit(lds_lut, be ? ((raw & 0xFFu) << 8) | (raw >> 8) : raw).kind == 2 ? 1u : 0u; }
                    u32 next_l = __builtin_amdgcn_update_dpp(edge_l, L.lm & 1u, 0x130, 0xF, 0xF, false);   // lane i <- lane i + 1, lane 63 <- the edge
                    if (n_ahead < 2) next_l = 0;
                    const WvMasks16W m = wv_classify16_utf16(L, n_units, wbm, in_info & 1u, (in_info >> 1) & 1u, (in_info >> 2) & 1u, next_l);
                    if (m.exotic) u16_exo = true;
                    const WvU16Packed pk = wv_utf16_pack(m, L.hm);
                    ((uint16_t*)lds_mask(0))[idx] = (uint16_t)pk.m0;
                    ((uint16_t*)lds_mask(1))[idx] = (uint16_t)pk.m1;
                    ((uint16_t*)lds_mask(2))[idx] = (uint16_t)pk.m2;
                    ((uint16_t*)lds_mask(3))[idx] = (uint16_t)pk.m3;
                } else if (FAM == 5) {
                    // EUC-JP: token starts for a hang-over of 0 / 1 / 2 bytes (a wave-uniform loop: one round per token in a row of lead-range
                    // bytes), the hang-overs composed along the wavefront, a table cell per token, marks that lie beyond the lane handed on
                    WvEucPre pc;
                    if (SW.n <= 1) pc = wv_eucjp_classes_swar<1>(SW, &ws6[1], avail + n_ahead);
                    else if (SW.n <= 3) pc = wv_eucjp_classes_swar<3>(SW, &ws6[1], avail + n_ahead);
                    else pc = wv_eucjp_classes_swar<6>(SW, &ws6[1], avail + n_ahead);
                    WvEucOrbit ob = wv_eucjp_orbit_init(pc);
                    while (__ballot(wv_eucjp_orbit_step(ob))) {}
                    u32 T = wv_eucjp_over(ob, 0) | (wv_eucjp_over(ob, 1) << 2) | (wv_eucjp_over(ob, 2) << 4);   // hang-over out for hang-over in 0 / 1 / 2
                    const bool at_zero = (long long)tile0 + rel == 0;   // the buffer's byte 0: the token pending on entry decides
                    if (at_zero) { const u32 o = (T >> (2 * P.entry_skip)) & 3u; T = o | (o << 2) | (o << 4); }
                    u32 out_here;
                    if (!__ballot((pc.lr & 0xFFFFu) == 0xFFFFu && !at_zero)) out_here = T & 3u;   // every lane holds a byte outside the lead range
                    else {
                        u32 Fc = T;
#pragma unroll
                        for (u32 d = 1; d < 64; d <<= 1) {         // inclusive composition: T_i o ... o T_0
                            const u32 g = wv_shfl(Fc, lane >= d ? lane - d : lane);
                            u32 r = 0;
#pragma unroll
                            for (int q = 0; q < 3; q++) r |= ((Fc >> (2 * ((g >> (2 * q)) & 3u))) & 3u) << (2 * q);
                            if (lane >= d) Fc = r;
                        }
                        out_here = (Fc >> (2 * dbcs_cov)) & 3u;
                    }
                    u32 cov_in = wv_from_prev(out_here, dbcs_cov);
                    if (at_zero) cov_in = P.entry_skip;
                    dbcs_cov = (u32)__builtin_amdgcn_readlane(out_here, 63);
                    if ((long long)tile0 + (long long)(t + 1) * (long long)kTileBytes == (long long)next_t0) { cov_next = dbcs_cov; have_next = true; }
                    const WvMasks18 m = wv_classify16_eucjp_swar(lds_pairs, SW.kana, ws6, pc, ob, cov_in, avail + n_ahead);
                    const u32 sp = (m.e >> 16) | ((m.a >> 16) << 2) | ((m.f >> 16) << 4) | ((m.ma >> 16) << 6) | ((m.mb >> 16) << 8);
                    const u32 si = wv_from_prev(sp, euc_spill);     // what the lane in front marked on my first two bytes
                    euc_spill = (u32)__builtin_amdgcn_readlane(sp, 63);
                    if (t >= 0) {
                        ((uint16_t*)lds_mask(0))[idx] = (uint16_t)(m.e | (si & 3u));
                        ((uint16_t*)lds_mask(1))[idx] = (uint16_t)(m.a | ((si >> 2) & 3u));
                        ((uint16_t*)lds_mask(2))[idx] = (uint16_t)(m.f | ((si >> 4) & 3u));
                        ((uint16_t*)lds_mask(3))[idx] = (uint16_t)(m.ma | ((si >> 6) & 3u));
                        ((uint16_t*)lds_mask(FAM == 5 ? 4 : 0))[idx] = (uint16_t)(m.mb | ((si >> 8) & 3u));
                    }
                } else if (FAM == 1 && CLS) {
                    u32 wz[6] = { ws6[0], ws6[1], ws6[2], ws6[3], ws6[4], ws6[5] };
                    if (!has_back) wz[0] = 0;
                    WvMasks16V m;
                    if (SW.n <= 1) m = wv_classify16_utf8_swar<1>(SW, wz, avail);
                    else if (SW.n <= 3) m = wv_classify16_utf8_swar<3>(SW, wz, avail);
                    else m = wv_classify16_utf8_swar<6>(SW, wz, avail);
                    ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.e;
                    ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.a;
                    ((uint16_t*)lds_mask(2))[idx] = (uint16_t)m.f;
                    ((uint16_t*)lds_mask(3))[idx] = (uint16_t)m.ma;
                    ((uint16_t*)lds_mask(FAM == 1 ? 4 : 0))[idx] = (uint16_t)m.mb;
                } else if (FAM == 1) {
                    u8 b[24];
#pragma unroll
                    for (int k = 0; k < 24; k++) b[k] = (u8)(ws6[k >> 2] >> (8 * (k & 3)));
                    const WvMasks16U m = wv_classify16_utf8(lds_lut, b, have_lo, have_hi);
                    ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.e;
                    ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.a;
                    ((uint16_t*)lds_mask(2))[idx] = (uint16_t)m.f;
                    ((uint16_t*)lds_mask(3))[idx] = (uint16_t)m.ma;
                    ((uint16_t*)lds_mask(FAM == 1 ? 4 : 0))[idx] = (uint16_t)m.mb;
                    ((uint16_t*)lds_mask(FAM == 1 && !CLS ? 5 : 0))[idx] = (uint16_t)m.g;
                } else {
                    // token starts: the lane's trails for both cases (bit arithmetic, sx_wave_core.hpp wv_dbcs_trails), the cases
                    // composed along the wavefront
                    WvDbcsPre pc{};
                    WvDbcsPreS ps{};
                    if (CLS) {
                        if (SW.n <= 1) ps = wv_dbcs_classes_swar<1>(SW, &ws6[1], avail);
                        else if (SW.n <= 3) ps = wv_dbcs_classes_swar<3>(SW, &ws6[1], avail);
                        else ps = wv_dbcs_classes_swar<6>(SW, &ws6[1], avail);
                    } else pc = wv_dbcs_classes(lds_lut, &ws6[1], avail);
                    const u32 lr16 = CLS ? ps.lr : pc.lr;
                    const u32 tr0 = wv_dbcs_trails(lr16, 0u), tr1 = wv_dbcs_trails(lr16, 1u);
                    const u32 o0 = tr0 >> 16, o1 = tr1 >> 16;
                    u32 fn = o0 | (o1 << 1);                       // bit c: how far the lane's last token hangs over if its first byte is at c
                    const bool at_zero = (long long)tile0 + rel == 0;   // the buffer's byte 0: the token pending on entry decides
                    if (at_zero) { const u32 o = P.entry_skip ? o1 : o0; fn = o | (o << 1); }
                    u32 out_here;
                    if (!__ballot(o0 != o1 && !at_zero)) out_here = fn & 1u;   // every lane holds a byte outside the lead range (binary data: always): what it hands on does not depend on what it gets
                    else {
#pragma unroll
                        for (u32 d = 1; d < 64; d <<= 1) {         // inclusive composition: fn_i o ... o fn_0
                            const u32 g = wv_shfl(fn, lane >= d ? lane - d : lane);
                            const u32 r = ((fn >> (g & 1u)) & 1u) | (((fn >> ((g >> 1) & 1u)) & 1u) << 1);
                            if (lane >= d) fn = r;
                        }
                        out_here = (fn >> dbcs_cov) & 1u;
                    }
                    u32 cov_in = wv_from_prev(out_here, dbcs_cov);
                    if (at_zero) cov_in = P.entry_skip ? 1u : 0u;
                    dbcs_cov = (u32)__builtin_amdgcn_readlane(out_here, 63);
                    if ((long long)tile0 + (long long)(t + 1) * (long long)kTileBytes == (long long)next_t0) { cov_next = dbcs_cov; have_next = true; }
                    if (t >= 0 && CLS) {
                        const WvMasks16E m = wv_classify16_dbcs_swar(lds_pairs, ws6, ps, cov_in ? tr1 : tr0, cov_in, (long long)tile0 + rel > 0, avail + n_ahead);
                        ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.e;
                        ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.a;
                        ((uint16_t*)lds_mask(2))[idx] = (uint16_t)m.f;
                        ((uint16_t*)lds_mask(3))[idx] = (uint16_t)m.ma;
                        ((uint16_t*)lds_mask(FAM == 4 ? 4 : 0))[idx] = (uint16_t)m.mb;
                    } else if (t >= 0) {
                        const WvMasks16D m = wv_classify16_dbcs_bits(lds_pairs, ws6, pc, cov_in ? tr1 : tr0, cov_in, (long long)tile0 + rel > 0, avail + n_ahead);
                        ((uint16_t*)lds_mask(0))[idx] = (uint16_t)m.e;
                        ((uint16_t*)lds_mask(1))[idx] = (uint16_t)m.a;
                        ((uint16_t*)lds_mask(2))[idx] = (uint16_t)m.f;
                        ((uint16_t*)lds_mask(3))[idx] = (uint16_t)m.ma;
                        ((uint16_t*)lds_mask(FAM == 4 ? 4 : 0))[idx] = (uint16_t)m.mb;
                        ((uint16_t*)lds_mask(FAM == 4 && !CLS ? 5 : 0))[idx] = (uint16_t)m.g;
                        ((uint16_t*)lds_mask(FAM == 4 && !CLS ? 6 : 0))[idx] = (uint16_t)m.o2;
                        ((uint16_t*)lds_mask(FAM == 4 && !CLS ? 7 : 0))[idx] = (uint16_t)m.o3;
                        ((uint16_t*)lds_mask(FAM == 4 && !CLS ? 8 : 0))[idx] = (uint16_t)m.o4;
                    }
                }
            }
        };
        if (FAM >= 4 && t_first < 0) {   // the way back to a token boundary: plain loads, one tile after the other (a wavefront's first batch)
            u32 eb = 0;
            {
                const long long first = (long long)tile0 + (long long)t_first * (long long)kTileBytes;
                if (first >= 4) eb = *(const u32*)(P.data + (first - 4));
            }
            for (int t = t_first; t < 0; t++) {
                const long long soff = (long long)tile0 + (long long)t * (long long)kTileBytes + 16ll * lane;
                u32x4 x = { 0, 0, 0, 0 };
                if (soff >= 0) x = *(const u32x4*)(P.data + soff);
                // (the dword behind the tile's last lane: the next look-back tile's first, or tile 0's)
                const long long nxt = (long long)tile0 + (long long)(t + 1) * (long long)kTileBytes;
                const u32 ea = nxt >= 0 && (u64)nxt + 4 <= P.len ? *(const u32*)(P.data + nxt) : 0u;
                do_tile(t, x, eb, ea);
                eb = (u32)__builtin_amdgcn_readlane(x.w, 63);
            }
            edge_back = eb;
        }
        if ((FAM == 4 || FAM == 5) && P.wave_grid && g0 == gw && lane == 0 && !known)   // the hang-over at my first tile: the wavefront behind me may be waiting for it (a repair launch: published by the first one, at THAT launch's first tile)
            __hip_atomic_store(P.wave_grid + v, 1u | (dbcs_cov << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int t_ref = next_t0 >= tile0 ? (int)((next_t0 - tile0) / kTileBytes) : 0;   // the last tile that starts at or in front of the next batch's first
        for (int t = 0; t < (int)n_tiles; t++) {
            if ((FAM == 4 || FAM == 5) && t == t_ref) { ref_pos = tile0 + (u64)t * kTileBytes; ref_cov = dbcs_cov; }
            const u32x4 x = xa;
            xa = xb; xb = xc; xc = xd; xd = issue(t + 4);
            const u32 ea = t + 1 < (int)n_tiles ? (u32)__builtin_amdgcn_readlane(xa.x, 0) : edge_after;
            do_tile(t, x, edge_back, ea);
            edge_back = (u32)__builtin_amdgcn_readlane(x.w, 63);
        }
        if ((FAM == 1 || FAM == 2) && P.lead_set && (lead_lo | lead_hi)) {
            if (lane == 0) atomicOr((unsigned long long*)P.lead_set, ((unsigned long long)lead_hi << 32) | lead_lo);
            if (__popc(lead_lo) + __popc(lead_hi) > 1) {   // two kinds already: -r may matter, this buffer is not the wave path's
                if (lane == 0 && MODE == 0) { P.wave_in[v] = 0xFFFFFFFEu; P.wave_out[v] = 0xFFFFFFFDu; P.wave_nf[v] = 0; P.wave_nb[v] = 0; }
                return;
            }
        }
        if (FAM == 2 && __ballot(u16_exo)) {   // as EUC-JP's way out: an entry state no wavefront ever leaves makes the verification fail
            if (lane == 0 && MODE == 0) { P.wave_in[v] = 0xFFFFFFFEu; P.wave_out[v] = 0xFFFFFFFDu; P.wave_nf[v] = 0; P.wave_nb[v] = 0; }
            return;
        }
        wave_lds_sync<WPB>();
        if ((FAM == 4 || FAM == 5) && t_ref >= (int)n_tiles) { ref_pos = tile0 + (u64)n_tiles * kTileBytes; ref_cov = dbcs_cov; }
        if (FAM >= 4) { dbcs_valid = have_next; if (have_next) dbcs_cov = cov_next; }   // (else the next batch walks back again)

#if defined(SX_WV_EXP) && SX_WV_EXP == 1   // experiments (tools/build_variant.sh): what the classification alone costs
        { u32 acc = 0; for (int k = 0; k < wv_n_masks(FAM, CLS); k++) acc ^= lds_mask(k)[lane]; tot_f += acc & 1u; continue; }
#endif
        // ---- 2. lane = window
        WvWin w;
        {
            const u32 o = active ? (u32)(ws - tile0) : 0u;
            const u32 n = active ? wn : 0u;
            if (FAM == 0 && CLS) w = wv_win_single_swar(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), SW.hi_len, n, P.n_min);
            else if (FAM == 0)
                w = wv_win_single(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(CLS ? 0 : 2), o, n),
                                  wv_extract(lds_mask(CLS ? 0 : 3), o, n), n, P.n_min);
            else if (FAM == 2) {
                const u32 ob2 = o >= 2 ? o - 2 : 0u;
                const bool hb = active && ws >= 2 && o >= 2 && ((lds_mask(2)[ob2 >> 5] >> (ob2 & 31u)) & 1u);
                w = wv_win_utf16(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(2), o, n), wv_extract(lds_mask(3), o, n), hb,
                                 ws % kWvSlice == 0, n, P.n_min);
            } else if (FAM == 5) {
                // (the two bytes in front of the window: one bit each of three masks)
                auto bit = [&](int k, u32 at) -> u32 { return (lds_mask(k)[at >> 5] >> (at & 31u)) & 1u; };
                const u32 o1 = o >= 1 ? o - 1 : 0u, o2 = o >= 2 ? o - 2 : 0u;
                const bool has1 = ws >= 1 && o >= 1, has2 = ws >= 2 && o >= 2;
                const bool done1 = !has1 || (bit(0, o1) | bit(3, o1)) != 0, done2 = !has2 || (bit(0, o2) | bit(3, o2)) != 0;
                w = wv_win_eucjp_swar(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(2), o, n), wv_extract(lds_mask(3), o, n),
                                      wv_extract(lds_mask(FAM == 5 ? 4 : 0), o, n), SW.hi_len, done1, done2, has1 && bit(2, o1), has2 && bit(2, o2), has1, has2,
                                      ws % kWvSlice == 0, n, P.n_min);
            } else if (FAM == 4) {
                // (the byte in front of the window: one bit of three masks)
                const u32 ob = o >= 1 ? o - 1 : 0u;
                const u32 eb = o >= 1 ? (lds_mask(0)[ob >> 5] >> (ob & 31u)) & 1u : 1u, mab = o >= 1 ? (lds_mask(3)[ob >> 5] >> (ob & 31u)) & 1u : 0u;
                const u32 fb1 = o >= 1 ? (lds_mask(2)[ob >> 5] >> (ob & 31u)) & 1u : 0u;
                if (CLS) w = wv_win_dbcs_swar(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(2), o, n), wv_extract(lds_mask(3), o, n),
                                              wv_extract(lds_mask(FAM == 4 ? 4 : 0), o, n), SW.hi_len, (eb | mab) != 0, fb1 != 0, ws > 0, ws % kWvSlice == 0, n, P.n_min);
                else w = wv_win_dbcs(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(2), o, n),
                                wv_extract(lds_mask(FAM == 4 && !CLS ? 5 : 0), o, n), wv_extract(lds_mask(3), o, n), wv_extract(lds_mask(FAM == 4 ? 4 : 0), o, n),
                                wv_extract(lds_mask(FAM == 4 && !CLS ? 6 : 0), o, n), wv_extract(lds_mask(FAM == 4 && !CLS ? 7 : 0), o, n),
                                wv_extract(lds_mask(FAM == 4 && !CLS ? 8 : 0), o, n), (eb | mab) != 0, fb1 != 0, ws > 0, ws % kWvSlice == 0, n, P.n_min);
            } else {
                const u32 fb = o >= 3 ? (u32)wv_extract(lds_mask(2), o - 3, 3).lo : 0u;
                if (CLS) {
                    const WvMask A_ = wv_extract(lds_mask(1), o, n), F_ = wv_extract(lds_mask(2), o, n);
                    w = wv_win_utf8(wv_extract(lds_mask(0), o, n), A_, F_, wv_utf8_good_from(A_, F_), wv_extract(lds_mask(3), o, n),
                                    wv_extract(lds_mask(FAM == 1 ? 4 : 0), o, n), fb, ws % kWvSlice == 0, n, P.n_min);
                } else
                w = wv_win_utf8(wv_extract(lds_mask(0), o, n), wv_extract(lds_mask(1), o, n), wv_extract(lds_mask(2), o, n),
                                wv_extract(lds_mask(FAM == 1 && !CLS ? 5 : 0), o, n), wv_extract(lds_mask(3), o, n),
                                wv_extract(lds_mask(FAM == 1 ? 4 : 0), o, n), fb, ws % kWvSlice == 0, n, P.n_min);
            }
        }

        constexpr int KIND = FAM == 0 ? 0 : FAM == 1 ? 1 : FAM == 2 ? 3 : 2;
        // -g (round 5): which of the window's characters are the grep char — every lane from its own window's bytes (in the cache: the
        // batch's classification has just read them); a wave-uniform branch, nothing for Missions without -g
        if (GREP) { if (active && wn) wv_set_grep<KIND>(w, WP, P.data + ws, (u32)P.grep_char, P.encoding == (u32)kEncUtf16be); else w.GC = wm_zero(); }
        // -r (round 5): the accepted multi-byte characters and where their lead byte changes, the same way (a trip per such character)
        if (SAME) {
            if (active && wn) wv_set_same<KIND>(w, P.data + ws, P.ubf, P.encoding == (u32)kEncUtf16be, WvLeadOfTable{ P.table });
            else { w.MBA = wm_zero(); w.D = wm_zero(); w.mb0_e = 128; w.mb0_code = 0; w.mbl_code = 0; }
        }

        // ---- 3. entry states: iterate until they are consistent along the lanes
        // (the exchange starts from every window's guess of what it hands on — wv_exit_guess: exact unless the window's last stretch is
        // the text-start stretch of its call and something is carried into it —, not from "nothing carried": one round, not two or three)
        const WvTail tail = active ? wv_tail_g<KIND, GREP, SAME>(WP, w) : WvTail{ 128u, 0u };   // (the window's last stretch: looked at once, used by every replay of it)
        u32 out = tail.state;
#if defined(SX_WV_XCHG_BPERM)   // (tools/repro: the exchange through ds_bpermute instead of a DPP wave shift)
#define SX_XCHG(v, edge) (lane ? wv_shfl((v), lane - 1u) : (edge))
#else
#define SX_XCHG(v, edge) wv_from_prev((v), (edge))
#endif
        u32 in = SX_XCHG(out, carry);
        const bool injected = g == P.g_lo;   // the host's exact state
        if (injected) in = P.inject;
        u32 nf = 0, nb = 0;
        bool todo = true;
#if defined(SX_WV_EXP) && SX_WV_EXP == 2   // ... + the windows' masks out of LDS and the guess
        { tot_f += (out ^ in ^ tail.a ^ (u32)w.LS.lo ^ (u32)w.O3.hi ^ (u32)w.CS.hi) & 1u; continue; }
#endif
        // (MODE 0: the findings' descriptors are staged where the batch's masks lay — every lane holds its window in registers now; the
        // LDS traffic of one wavefront is in order)
        u32* const stage = lds_base;
        for (;;) {
            if (todo && active) {
                WvState st = wv_unpack(in);
#if defined(SX_WV_EXP) && SX_WV_EXP >= 3   // ... + the windows' state machine without descriptors
                WvCountEmit<0> ce;
#else
                WvStageEmit<u32*> ce;
                ce.cap = SAME ? kWvStageSame : kWvStage;
                ce.stage = stage; ce.lane = lane;
#endif
                ce.widx = (u32)(g - own_start);
                if (MODE == 0) { wv_window_g<KIND, GREP, SAME>(WP, w, st, ce, tail); nf = ce.nf; nb = ce.nb; }
                else { WvCountEmit<0> cc; wv_window_g<KIND, GREP, SAME>(WP, w, st, cc, tail); nf = cc.nf; nb = cc.nb; }
                out = wv_pack(st);
            } else if (!active) out = in;
            u32 pin = SX_XCHG(out, carry);
            if (injected) pin = P.inject;
            todo = active && pin != in;
            in = pin;
            if (!__ballot(todo)) break;
        }
        if (g0 == gw && v != 0 && !known) {   // the state this wavefront assumes for its first own window (lane kWvWarm of the first batch)
            assumed_in = (u32)__builtin_amdgcn_readlane(in, (int)kWvWarm);
        }
        carry = (u32)__builtin_amdgcn_readlane(out, 63);
        const u32 last_out = (u32)__builtin_amdgcn_readlane(out, last_lane);
        if (!owned) { nf = 0; nb = 0; }

        // ---- 4. counts -> offsets -> (pass 2) output
        const u32 packed = (nf << 18) | nb;   // per batch: nb <= 64 windows x 640 bytes < 2^18, nf <= 64 x 129 < 2^14
        const u32 incl = wv_scan_incl(packed, lane);
        const u32 bt = (u32)__builtin_amdgcn_readlane(incl, 63);
        if (MODE == 1 && (nf | nb)) {
            const u32 excl = incl - packed;
            const u64 fo = fbase + tot_f + (excl >> 18), ao = abase + tot_b + (excl & 0x3FFFFu);
            // (the window's offset once more, from its number: with the -r kernels — 30 to 45 spilled registers — `ws` came out of the loop
            // above as the lane's number less one in the lanes that had run it twice, on gfx950 with ROCm 7.2.0's compiler; nothing else
            // that lives across the loop is used here, and tools/gpu_fuzz.py compares every byte of the output)
            u64 ws2 = 0; u32 wn2 = 0;
            wv_window_at(g, P.W, P.wps, P.len, &ws2, &wn2);
#if defined(SX_WV_USE_WS)   // (tools/repro: the round-5 build — the offset as it comes out of the loop)
            ws2 = ws;
#endif
            WriteEmit<FAM> we_{ &P, fo, P.arena + ao, ao, ws2 };
            WvState st = wv_unpack(in);
            wv_window_g<KIND, GREP, SAME>(WP, w, st, we_, tail);
        }
#if defined(SX_WV_EXP) && SX_WV_EXP >= 3
        if (false) {
#else
        if (MODE == 0 && P.desc && nf) {   // the lane-per-finding writer's input (beyond desc_cap: counted only, the launch falls back)
#endif
            const u32 excl = incl - packed;
            const u32 at = tot_f + (excl >> 18), ab = tot_b + (excl & 0x3FFFFu);
            WvDesc* slot = (WvDesc*)P.desc + v * (u64)P.desc_cap + at;
            const u32 room = at < P.desc_cap ? P.desc_cap - at : 0u;
            if (nf <= (SAME ? kWvStageSame : kWvStage)) {   // the usual window: what the count staged, its string offsets moved to the wavefront's
                const u32 k = nf < room ? nf : room;
                for (u32 j = 0; j < k; j++) {
                    const u32* p = stage + j * 192u + lane;
                    slot[j] = WvDesc{ p[0] + ab, p[64], p[128] };
                }
            } else {
                WvDescEmit de{ slot, room, ab, (u32)(g - own_start) };
                WvState st = wv_unpack(in);
                wv_window_g<KIND, GREP, SAME>(WP, w, st, de, tail);
            }
        }
        tot_f += bt >> 18; tot_b += bt & 0x3FFFFu;
        if (g0 + kWvBatch >= own_end && MODE == 0) {
            // (the state after a buffer's last window also says whether it ends inside a token: the next buffer's decoder holds that byte)
            const u32 pend = FAM >= 4 ? (u32)__builtin_amdgcn_readlane(w.tail_pend, last_lane) : 0u;
            if (lane == 0) P.wave_out[v] = last_out | (own_end == P.g_hi ? pend << 27 : 0u);
        }
    }
    if (MODE == 0 && lane == 0) { P.wave_nf[v] = tot_f; P.wave_nb[v] = tot_b; P.wave_in[v] = assumed_in; }
}

// Every wavefront after the first assumed "nothing carried" kWvWarm windows in front of its own; is the state it
// reached for its first own window what its predecessor really left there?  totals: [0] findings, [1] string bytes,
// [2] wavefronts whose assumption was wrong (high half: wavefronts with more findings than descriptors), [3] the state after the last
// window (high half: wavefronts that gave the buffer back).
__global__ __launch_bounds__(256) void wave_verify_kernel(const u32* wave_in, const u32* wave_out, const u32* wave_nf, const u32* wave_nb,
                                                          const u64* fbase, const u64* abase, u64 v0, u64 v1, u64* totals, u32 desc_cap) {
    const u64 v = v0 + (u64)blockIdx.x * 256 + threadIdx.x;
    if (v >= v1) return;
    if (v > 0 && wave_in[v] != wave_out[v - 1]) atomicAdd((unsigned long long*)&totals[2], 1ull);
    if (desc_cap && wave_nf[v] > desc_cap) atomicAdd((unsigned long long*)&totals[2], 1ull << 32);   // more findings than descriptors
    if (wave_in[v] == 0xFFFFFFFEu) atomicAdd((unsigned long long*)&totals[3], 1ull << 32);         // it gave the buffer back (no repair launch can help)
    if (v + 1 == v1) { totals[0] = fbase[v] + wave_nf[v]; totals[1] = abase[v] + wave_nb[v]; atomicAdd((unsigned long long*)&totals[3], (unsigned long long)wave_out[v]); }
}

// The writer that works a lane per finding: wavefront v's descriptors -> records and strings.  No classification, no state machine, no
// LDS; every lane has work (the window-parallel writer: a lane per window, busy only where a window holds a finding).
template <int FAM>
__global__ __launch_bounds__(256) void wave_emit_kernel(const WaveParams P) {
    const u32 lane = threadIdx.x & 63u;
    const u64 v = P.v0 + (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (v >= P.v1) return;
    const u32 nf = P.wave_nf[v];
    const u64 own_start = P.g_lo + v * P.nwin;
    const u64 fbase = P.wave_fbase[v] - P.f_sub, abase = P.wave_abase[v] - P.a_sub;
    const WvDesc* d = (const WvDesc*)P.desc + v * (u64)P.desc_cap;
    for (u32 i = lane; i < nf; i += 64) {
        const WvDesc x = d[i];
        u64 ws; u32 wn;
        wv_window_at(own_start + wv_desc_widx(x), P.W, P.wps, P.len, &ws, &wn);
        const u64 ao = abase + wv_desc_a_local(x);
        wv_write_finding<FAM>(P, fbase + i, P.arena + ao, ao, ws, wv_desc_din(x), wv_desc_prec(x), wv_desc_completes(x),
                              wv_desc_src_rel(x), wv_desc_src_len(x), wv_desc_out_len(x));
    }
}

hipError_t launch_wave_emit(const WaveParams& P, uint64_t v0, uint64_t v1, hipStream_t stream) {
    if (v1 <= v0) return hipSuccess;
    WaveParams Q = P;
    Q.v0 = v0; Q.v1 = v1;
    const dim3 grid((unsigned)((v1 - v0 + 3) / 4));
    if (P.family == 5) hipLaunchKernelGGL((wave_emit_kernel<5>), grid, dim3(256), 0, stream, Q);
    else if (P.family == 4) hipLaunchKernelGGL((wave_emit_kernel<4>), grid, dim3(256), 0, stream, Q);
    else if (P.family == 2) hipLaunchKernelGGL((wave_emit_kernel<2>), grid, dim3(256), 0, stream, Q);
    else if (P.family == 1) hipLaunchKernelGGL((wave_emit_kernel<1>), grid, dim3(256), 0, stream, Q);
    else hipLaunchKernelGGL((wave_emit_kernel<0>), grid, dim3(256), 0, stream, Q);
    return hipGetLastError();
}

struct U32ToU64 {
    const u32* p;
    __device__ u64 operator()(u64 i) const { return (u64)p[i]; }
};

size_t wave_scratch_bytes(uint64_t n_waves) {
    size_t a = 0;
    auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), U32ToU64{ nullptr });
    (void)rocprim::exclusive_scan(nullptr, a, it, (u64*)nullptr, (u64)0, (size_t)n_waves, rocprim::plus<u64>(), (hipStream_t)0);
    return a + 512;
}

// (a launch of wave_replay_kernel<M, F, W, C, GREP>: with or without -g)
#define SX_WV_UNPACK(...) __VA_ARGS__
#define SX_WV_LAUNCH(targs, grid, block, dyn, stream, Q)                                                        \
    do {                                                                                                       \
        if ((Q).grep_char >= 0) hipLaunchKernelGGL((wave_replay_kernel<SX_WV_UNPACK targs, 1>), grid, block, dyn, stream, Q);   \
        else hipLaunchKernelGGL((wave_replay_kernel<SX_WV_UNPACK targs, 0>), grid, block, dyn, stream, Q);                      \
    } while (0)
// (families 0 - 2: with -r as well)
#define SX_WV_LAUNCH_S(targs, grid, block, dyn, stream, Q)                                                      \
    do {                                                                                                       \
        if ((Q).same && (Q).grep_char >= 0) hipLaunchKernelGGL((wave_replay_kernel<SX_WV_UNPACK targs, 3>), grid, block, dyn, stream, Q);   \
        else if ((Q).same) hipLaunchKernelGGL((wave_replay_kernel<SX_WV_UNPACK targs, 2>), grid, block, dyn, stream, Q);   \
        else SX_WV_LAUNCH(targs, grid, block, dyn, stream, Q);                                                  \
    } while (0)
// pass 1 of wavefronts [v0, v1): counts, their exclusive sums from v0 on (fbase[v], abase[v]), the verification against
// wavefront v0 - 1 (an earlier launch on the same stream) and among themselves
hipError_t launch_wave_count(const WaveParams& P, uint64_t v0, uint64_t v1, uint64_t* fbase, uint64_t* abase, uint64_t* totals,
                             void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (v1 <= v0) return hipSuccess;
    const uint64_t n = v1 - v0;
    WaveParams Q = P;
    Q.v0 = v0; Q.v1 = v1;
    const unsigned dyn = getenv("SX_WAVE_DYN_LDS") ? (unsigned)atoi(getenv("SX_WAVE_DYN_LDS")) : 0u;   // experiments: fewer wavefronts per CU
    if (P.wave_grid && !P.redo) { hipError_t ez = hipMemsetAsync(P.wave_grid + v0, 0, (size_t)n * 4, stream); if (ez != hipSuccess) return ez; }   // (a repair launch reads what the first one published)
    if (P.family == 5) SX_WV_LAUNCH((0, 5, 4, 1), dim3((unsigned)((n + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 4 && P.swar.cls) SX_WV_LAUNCH((0, 4, 4, 1), dim3((unsigned)((n + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 4) SX_WV_LAUNCH((0, 4, 4, 0), dim3((unsigned)((n + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 2) SX_WV_LAUNCH_S((0, 2, 1, 0), dim3((unsigned)n), dim3(64), dyn, stream, Q);
    else if (P.family == 1 && P.swar.cls) SX_WV_LAUNCH_S((0, 1, 1, 1), dim3((unsigned)n), dim3(64), dyn, stream, Q);
    else if (P.family == 1) SX_WV_LAUNCH_S((0, 1, 1, 0), dim3((unsigned)n), dim3(64), dyn, stream, Q);
    else if (P.swar.cls) SX_WV_LAUNCH_S((0, 0, 1, 1), dim3((unsigned)n), dim3(64), dyn, stream, Q);
    else SX_WV_LAUNCH_S((0, 0, 1, 0), dim3((unsigned)n), dim3(64), dyn, stream, Q);
    void* tmp = (void*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)((uint8_t*)tmp - (uint8_t*)scratch);
    auto itf = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), U32ToU64{ P.wave_nf + v0 });
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, itf, fbase + v0, (u64)0, (size_t)n, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    tmp_bytes = scratch_bytes - (size_t)((uint8_t*)tmp - (uint8_t*)scratch);
    auto itb = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), U32ToU64{ P.wave_nb + v0 });
    e = rocprim::exclusive_scan(tmp, tmp_bytes, itb, abase + v0, (u64)0, (size_t)n, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(totals, 0, 4 * sizeof(uint64_t), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wave_verify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P.wave_in, P.wave_out, P.wave_nf,
                       P.wave_nb, fbase, abase, v0, v1, totals, P.desc ? P.desc_cap : 0u);
    return hipGetLastError();
}

// wavefronts [v0, v1): their findings go to P.findings / P.arena at (their offset - P.f_sub / P.a_sub)
hipError_t launch_wave_write(const WaveParams& P, uint64_t v0, uint64_t v1, hipStream_t stream) {
    if (v1 <= v0) return hipSuccess;
    WaveParams Q = P;
    Q.v0 = v0; Q.v1 = v1;
    const unsigned dyn = getenv("SX_WAVE_DYN_LDS") ? (unsigned)atoi(getenv("SX_WAVE_DYN_LDS")) : 0u;
    if (P.wave_grid) { hipError_t ez = hipMemsetAsync(P.wave_grid + v0, 0, (size_t)(v1 - v0) * 4, stream); if (ez != hipSuccess) return ez; }
    if (P.family == 5) SX_WV_LAUNCH((1, 5, 4, 1), dim3((unsigned)(((v1 - v0) + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 4 && P.swar.cls) SX_WV_LAUNCH((1, 4, 4, 1), dim3((unsigned)(((v1 - v0) + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 4) SX_WV_LAUNCH((1, 4, 4, 0), dim3((unsigned)(((v1 - v0) + 3) / 4)), dim3(256), dyn, stream, Q);
    else if (P.family == 2) SX_WV_LAUNCH_S((1, 2, 1, 0), dim3((unsigned)(v1 - v0)), dim3(64), dyn, stream, Q);
    else if (P.family == 1 && P.swar.cls) SX_WV_LAUNCH_S((1, 1, 1, 1), dim3((unsigned)(v1 - v0)), dim3(64), dyn, stream, Q);
    else if (P.family == 1) SX_WV_LAUNCH_S((1, 1, 1, 0), dim3((unsigned)(v1 - v0)), dim3(64), dyn, stream, Q);
    else if (P.swar.cls) SX_WV_LAUNCH_S((1, 0, 1, 1), dim3((unsigned)(v1 - v0)), dim3(64), dyn, stream, Q);
    else SX_WV_LAUNCH_S((1, 0, 1, 0), dim3((unsigned)(v1 - v0)), dim3(64), dyn, stream, Q);
    return hipGetLastError();
}

}  // namespace sx
