// sx_replay_dev.hip — stage B on the device: the exact replay of FindingCollection::from
// (reference src/finding_collection.rs:84-342) around long runs, one lane per run.
//
// This is the device twin of sx_replay.cpp (RangeReplay): same rules, same order of operations
// (and the very same decoders and SplitStr, sx_codec_core.hpp), so that positions, precision marks,
// cuts and strings are identical to the host replay (which stays the fallback).
//   * lane i owns the region that begins in the window of run i's first byte, with the
//     carried state re-derived from the bytes before that window (decoder state; one
//     accepted char as leftover if a short run hangs over the edge);
//   * it follows runs i, i+1, .. until no cut string is pending and no long leftover is
//     carried (RangeReplay::scan_from's stop rule);
//   * pass 1 counts (findings, string bytes, end position); the host decides which regions
//     stand (a region is void if an earlier one ran over its start) and assigns output
//     offsets; pass 2 replays the standing regions again and writes findings + strings, in
//     order, at those offsets — no atomics, deterministic output order.
// Regions longer than kMaxWindows windows, or anything this path does not cover, are
// reported with a status and replayed by the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "sx_device.hpp"

namespace sx {

#define SXD __device__ __forceinline__
#define SXD_NOINLINE __device__ __attribute__((noinline))
}  // namespace sx
#include "sx_replay_core.hpp"
namespace sx {

#ifndef SX_REPLAY_WAVES
#define SX_REPLAY_WAVES 8   // the replay lives on occupancy: 64 VGPRs and a few spilled registers beat 128 VGPRs at 4 waves (measured)
#endif

// Pass 1: one lane per run.  A run that begins in the window where the run before it ends is
// certainly inside the region that holds that run (its window start lies inside or in front of that
// run): it is marked kRegionChained right away.  A run that begins in a later window starts a region
// of its own — the region before it stops at that window start if nothing is pending there
// (replay_region's region_over), else it runs on and the stitch voids this one.  On string-dense
// input (a run in nearly every window) this gives one short region per window instead of chains.
// (With -g the next window still belongs to the region: sx_replay_core.hpp regions_may_touch.)
SXD bool region_is_chained(const ReplayParams& P, u64 i, u64 want) { return run_is_chained(P, i, want); }
// which runs replay at all (for the cache slots): neither somebody else's nor chained
// (the runs that do not replay get their pass-1 record here: the count kernel only visits the others)
__global__ __launch_bounds__(256) void replay_heads_kernel(const ReplayParams P, u32* head, ReplayRegionOut* ro) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= P.n_runs) return;
    const u64 want = win_start(P.runs[i].start, P.W);
    const bool not_mine = want < P.lo || want >= P.hi;
    const bool h = !not_mine && !region_is_chained(P, i, want);
    head[i] = h ? 1u : 0u;
    if (!h) {
        ReplayRegionOut o;
        o.end = 0; o.n_find = 0; o.n_bytes = 0; o.status = not_mine ? kRegionNotMine : kRegionChained; o.pad = 0;
        ro[i] = o;
    }
}
__global__ __launch_bounds__(256) void replay_heads_list_kernel(const ReplayParams P, const u32* head, const u32* slot_of, u32* head_list) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < P.n_runs && head[i]) head_list[slot_of[i]] = (u32)i;
}
__global__ void replay_heads_total_kernel(const u32* head_last, const u32* slot_last, u32* n_heads) { *n_heads = *slot_last + *head_last; }

// (EQ: the encoding family, + 8 for 64 < q <= 255 — larger private buffers, sx_replay_core.hpp QBIG)
template <int EQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SX_REPLAY_WAVES))) void replay_count_kernel(
    const ReplayParams P, ReplayRegionOut* out) {
    constexpr int ENC = EQ & 7;
    constexpr bool QBIG = EQ >= 8;
    const u64 i = (u64)blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n_runs) return;
    ReplayRegionOut o;
    o.end = 0; o.n_find = 0; o.n_bytes = 0; o.status = kRegionOk; o.pad = 0;
    const u64 want = win_start(P.runs[i].start, P.W);
    if (want < P.lo || want >= P.hi) o.status = kRegionNotMine;
    else if (region_is_chained(P, i, want)) o.status = kRegionChained;
    else replay_region<0, ENC, false, QBIG>(P, i, o, nullptr, nullptr, 0);
    out[i] = o;
}
// Pass 1 with the output cache: one lane per replaying run (on dense input half of the runs are chained: no idle lanes).
// The window's staging copy lies in LDS, one row per lane (win_row_bytes: an odd number of dwords, the rows start in
// different banks) — as a private array it is scratch memory, and the scratch of all resident waves is far larger than L2.
__host__ __device__ inline u32 win_row_bytes(u32 W) { return (((W + kBackBytes + 3) / 4) | 1u) * 4; }
template <int EQ, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void replay_count_cached_kernel(
    const ReplayParams P, ReplayRegionOut* out) {
    constexpr int ENC = EQ & 7;
    constexpr bool QBIG = EQ >= 8;
    extern __shared__ __align__(16) u8 lds_win[];
    // (round 5) with the fast pre-pass in front only what it left is visited: the list's length is known on the device only, so the grid is
    // the whole head list's (a block that finds nothing to do returns at once) — bounded: the blocks stride through the list
    const u64 n_todo = P.hard_list ? (u64)*P.n_hard : (u64)*P.n_heads;
    for (u64 j = (u64)blockIdx.x * 64 + threadIdx.x; j < n_todo; j += (u64)gridDim.x * 64) {
        u64 i = P.hard_list ? P.hard_list[j] : j;
        i = P.head_list[i];
        ReplayRegionOut o;
        o.end = 0; o.n_find = 0; o.n_bytes = 0; o.status = kRegionOk; o.pad = 0;
        // slots have a minimum size: with more replaying regions than the arena has room for, the ones
        // behind its end go without (cap 0: nothing fits, o.pad stays 0, pass 2 replays them)
        const CacheGeom g = cache_geom(P.arena_bytes, *P.n_heads);
        const u64 off = (u64)P.slot_of[i] * g.slot_bytes;
        const bool room = off + g.slot_bytes <= P.arena_bytes;
        u8* slot = P.cache_arena + (room ? off : 0);
        replay_region<2, ENC, true, QBIG>(P, i, o, (sx_finding*)slot, slot + g.cap_f * sizeof(sx_finding), 0, room ? g.cap_f : 0u, room ? g.cap_b : 0u,
                                    lds_win + threadIdx.x * win_row_bytes(P.W));
        out[i] = o;
    }
}

// ---- Pass 1, the fast pre-pass (round 5) -----------------------------------------------------------------------------------------
// On sparse input — the headline: 2.8 M runs per 64 GiB of random bytes — nearly every region is ONE run of a dozen characters
// somewhere inside ONE window, and the general kernel above spends ~700 cycles per byte on it: it derives the state at the window
// start, stages the window, decodes the run's call byte by byte and walks SplitStr over it (2.1 ms of the whole chip per step, half
// of stage B's instructions, taken out of the scan launch that runs next to it).  For such a region everything follows from the
// run record and a short walk back from the run's first byte:
//   (1) the run [rs, re) is a plain run (no continuation piece), has fewer than q characters (one line, helper.rs:237), lies in the
//       window [want, wend) with re + 4 <= wend (whatever begins at re — a rejected character, a malformed sequence — is settled
//       inside the window: the run does not touch the end of its call's text with the call still open, helper.rs:353-355, 389-392),
//       and the next run of the list begins at or behind wend (nothing else in this window can yield: every stretch of >= n
//       characters is a run of the list);
//   (2) the call that holds rs starts at vs >= want + 4, behind a malformed sequence INSIDE the window (call_start_before; with
//       vs < want + 4 the walk may have met the tail of a character that began in front of the window, which is no error).  Then the
//       window's first call has taken whatever was carried in (leftover, cut flag) and — ending in an error, holding no run —
//       dropped it (helper.rs:315-330, 410-415); the calls in between yield nothing; this call's first and only yield is the run:
//       position = the call's start (finding_collection.rs:260), precision Exact (:146: nothing was prepended, no slice-start
//       probe: din > 0), completes = false, the string = the run's bytes (UTF-8 in, UTF-8 out);
//   (3) behind the run nothing is pending (the cut flag stays down: :353-355 needs q characters or an open text end), the tail of
//       the window holds no long stretch, and no run crosses wend: the region ends at wend (replay_region's shortcut (B)).
// Also the pieces of a run that was cut at a window start ((4), (5) in replay_fast_region: a run across one window start, 9 % of the headline's).
// A lane that finds (1)-(3) writes what replay_region<2> would have written — the region's record, the finding and the string in
// the cache slot — and is done; the others put their slot on a list (one atomic per wavefront) that the general kernel then visits
// instead of the whole head list.  tests/test_gpu_fast_replay.py runs every Mission shape with and without the pre-pass.
// ---------------------------------------------------------------------------------------------------------------------------------
// characters that BEGIN in [p, p + n): bytes that are no continuation bytes (valid UTF-8: a run's bytes)
SXD u32 fast_count_chars(const u8* p, u32 n) {
    u32 c = 0, t = 0;
    for (; t + 8 <= n; t += 8) {
        u64 v; __builtin_memcpy(&v, p + t, 8);
        const u64 cont = v & ~(v << 1) & 0x8080808080808080ull;
        c += 8u - (u32)__builtin_popcountll(cont);
    }
    for (; t < n; t++) c += (p[t] & 0xC0) != 0x80;
    return c;
}
template <int ENC>
SXD bool replay_fast_region(const ReplayParams& P, u64 i, ReplayRegionOut& o, sx_finding* fout, u8* aout, u32 cap_f, u32 cap_b) {
    static_assert(ENC == 1, "UTF-8: the string is the run's bytes");
    const sx_run r = P.runs[i];
    const u32 W = P.W;
    const u64 rs = r.start, re = r.end;
    const u64 want = win_start(rs, W), wend = next_win_start(rs, W);
    if (wend > P.len) return false;
    const u64 n_look = P.n_look ? P.n_look : P.n_runs;
    const bool has_next = i + 1 < n_look;
    u64 nx_start = ~0ull, nx_chars = 0;
    if (has_next) { nx_start = P.runs[i + 1].start; nx_chars = P.runs[i + 1].chars; }
    const bool goes_on = re == wend;                                   // the run crosses the window end: it was cut there, if its piece is next
    if (goes_on && !(has_next && nx_start == wend && (nx_chars & kPieceCont))) return false;
    if (!goes_on && (re + 4 > wend || (has_next && nx_start < wend))) return false;
    u64 vs = want, str0 = rs;
    u8 precision = SX_PRECISION_EXACT;
    if (r.chars & kPieceCont) {
        // (4) a continuation piece (sx_replay_core.hpp kPieceCont): the window starts inside the run, delta bytes behind its beginning.  Carried in:
        // the run's characters that complete in front of the window as the leftover — fewer than q: else the first line was cut and the
        // cut flag is pending, the general kernel's —, the bytes of a character across the window start in the decoder.  The window's
        // first call holds the rest of the run, which (leftover prepended) is its text-start stretch: position = the window start
        // (din = 0), precision Before (:214-221), the string = the run's bytes from its beginning.  A piece that goes on into the next
        // window only adds to the leftover (helper.rs:389-392): nothing is written.
        const u64 delta = r.chars & ~kPieceCont;
        if (delta == 0 || delta >= 4ull * P.q || delta > rs) return false;
        str0 = rs - delta;
        const u32 total = fast_count_chars(P.data + str0, (u32)(re - str0));     // characters of the run up to this piece's end
        if (total >= P.q) return false;
        if (!goes_on) {
            const u32 before = fast_count_chars(P.data + str0, (u32)delta) - ((P.data[rs] & 0xC0) == 0x80 ? 1u : 0u);   // ... that complete in front of the window
            if (before == 0) return false;                                      // (no leftover: the slice-start probe may speak, :176-207)
        }
        precision = SX_PRECISION_BEFORE;
    } else {
        if (r.chars >= P.q) return false;                                   // (saturating count: >= q means at least q)
        vs = call_start_before<ENC>(P, want, rs);
        if (vs < want + 4) return false;
    }
    if (goes_on) {   // (5) the run's first / a middle piece: it touches the end of its call's text with the call still open -> carried (helper.rs:389-392)
        o.end = wend; o.n_find = 0; o.n_bytes = 0; o.status = kRegionOk; o.pad = 1;
        return true;
    }
    const u32 nb = (u32)(re - str0);
    if (cap_f < 1 || nb > cap_b) return false;
    const u64 soff = want / kSliceLen * kSliceLen;
    sx_finding f;
    f.position = P.consumed0 + vs;
    f.str_off = 0;
    f.str_len = nb;
    f.precision = precision;
    f.completes_previous = 0;
    f.mission_id = (u8)P.mission_id;
    f.reserved = 0;
    f.input_file_id = (int16_t)P.file_id;
    f.reserved2 = 0;
    f.slice_index = (u32)(soff / kSliceLen) + P.slice_base;
    fout[0] = f;
    const u8* s = P.data + str0;
    u32 t = 0;
    for (; t + 16 <= nb; t += 16) { uint4 v; __builtin_memcpy(&v, s + t, 16); __builtin_memcpy(aout + t, &v, 16); }
    if (t + 8 <= nb) { u64 v; __builtin_memcpy(&v, s + t, 8); __builtin_memcpy(aout + t, &v, 8); t += 8; }
    if (t + 4 <= nb) { u32 v; __builtin_memcpy(&v, s + t, 4); __builtin_memcpy(aout + t, &v, 4); t += 4; }
    for (; t < nb; t++) aout[t] = s[t];
    o.end = wend; o.n_find = 1; o.n_bytes = nb; o.status = kRegionOk; o.pad = 1;
    return true;
}
template <int ENC>
__global__ __launch_bounds__(256) void replay_fast_kernel(const ReplayParams P, ReplayRegionOut* out) {
    const u64 s = (u64)blockIdx.x * 256 + threadIdx.x;   // slot of a replaying run
    const u32 nh = *P.n_heads;
    bool hard = false;
    if (s < nh) {
        const u64 i = P.head_list[s];
        const CacheGeom g = cache_geom(P.arena_bytes, nh);
        const u64 off = (u64)P.slot_of[i] * g.slot_bytes;
        const bool room = off + g.slot_bytes <= P.arena_bytes;
        u8* slot = P.cache_arena + (room ? off : 0);
        ReplayRegionOut o;
        if (room && replay_fast_region<ENC>(P, i, o, (sx_finding*)slot, slot + g.cap_f * sizeof(sx_finding), g.cap_f, g.cap_b)) out[i] = o;
        else hard = true;
    }
    const unsigned long long m = __ballot(hard);
    if (m) {
        const u32 lane = threadIdx.x & 63u;
        u32 base = 0;
        if (lane == (u32)__ffsll((long long)m) - 1u) base = atomicAdd(P.n_hard, (u32)__popcll(m));
        base = __shfl(base, (int)__ffsll((long long)m) - 1);
        if (hard) P.hard_list[base + (u32)__popcll(m & ((1ull << lane) - 1ull))] = (u32)s;
    }
}
bool replay_fast_covers(const ReplayParams& P) {
    return enc_family(P.encoding) == 1 && P.grep_char < 0 && !P.same_block && P.chars_min_nb >= 1 && P.chars_min_nb <= P.q && P.long_run >= 1
           && P.cache_arena && P.head_list && P.hard_list && P.n_hard;
}
hipError_t launch_replay_fast(const ReplayParams& P, ReplayRegionOut* out, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL((replay_fast_kernel<1>), dim3((unsigned)((P.n_runs + 255) / 256)), dim3(256), 0, stream, P, out);
    return hipGetLastError();
}

// Pass 2: the standing regions write their findings and strings at the offsets the host assigned.
template <int EQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SX_REPLAY_WAVES))) void replay_write_kernel(const ReplayParams P, const u64* region_index, const u64* fbase,
                                                          const u64* abase, u64 n_regions, sx_finding* findings, u8* arena) {
    const u64 k = (u64)blockIdx.x * 64 + threadIdx.x;
    if (k >= n_regions) return;
    ReplayRegionOut o;
    replay_region<1, EQ & 7, false, (EQ >= 8)>(P, region_index[k], o, findings + fbase[k], arena + abase[k], abase[k]);
}

// Pass 2, flagged form: one lane per run; the standing regions (stitch below) write at the
// offsets the device scans assigned.  A region whose output pass 1 kept in its cache slot is left to
// replay_copy_cached_kernel below; the others are replayed once more by their lane.
template <int EQ>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SX_REPLAY_WAVES))) void replay_write_flagged_kernel(
    const ReplayParams P, const ReplayRegionOut* ro, const u8* stands, const u64* fpos, const u64* apos, sx_finding* findings,
    u8* arena) {
    u64 i = (u64)blockIdx.x * 64 + threadIdx.x;
    if (P.head_list) {  // only a replaying run can stand
        if (i >= *P.n_heads) return;
        i = P.head_list[i];
    } else if (i >= P.n_runs) return;
    if (!stands[i]) return;
    if (P.cache_arena && ro[i].pad) return;
    const u64 fp = fpos[i], ap = apos[i];
    ReplayRegionOut o;
    replay_region<1, EQ & 7, false, (EQ >= 8)>(P, i, o, findings + fp, arena + ap, ap + P.str_off_base);
}

// The cached outputs go into place: G lanes per region (4, 16 or 64, by the average output size), 64 / G regions of a wave
// at a time — coalesced within a region (a lane of its own would copy byte by byte into 64 different lines per
// instruction), and no decoder in the kernel, so that many waves are resident.  A finding is two uint4 (str_off at offset 8
// of the first); the strings go as dwords aligned to the DESTINATION, each built from two source dwords, the up to three
// bytes on either side singly.
template <int G, int U>
__global__ __launch_bounds__(64) void replay_copy_cached_kernel(const ReplayParams P, const ReplayRegionOut* ro, const u8* stands,
                                                                const u64* fpos, const u64* apos, sx_finding* findings, u8* arena) {
    const u32 lane = threadIdx.x;
    u64 i = (u64)blockIdx.x * 64 + lane;
    bool mine;
    if (P.head_list) {
        mine = i < *P.n_heads;
        if (mine) i = P.head_list[i];
    } else mine = i < P.n_runs;
    u64 fp = 0, ap = 0;
    u32 nf = 0, nb = 0, slot_no = 0;
    if (mine && stands[i] && ro[i].pad) { fp = fpos[i]; ap = apos[i]; nf = ro[i].n_find; nb = ro[i].n_bytes; slot_no = P.slot_of[i]; }
    if (!__ballot(nf | nb)) return;
    const CacheGeom g = cache_geom(P.arena_bytes, *P.n_heads);
    const u32 bytes_at = g.cap_f * (u32)sizeof(sx_finding);
    const u32 sub = lane % G, group = lane / G;
#pragma unroll 1
    for (u32 it = 0; it < G; it += U) {
        // U regions per group at a time, all their loads first (the first G uint4 of findings and G dwords of strings
        // of each: what most regions have), then the stores; what is left of large regions follows in plain loops
        const u8* slot[U];
        u64 rfp[U], rap[U];
        u32 rnf[U], rnb[U], head[U], body[U], lo[U], hi[U];
        uint4 fv[U];
        u8 hb[U], tb[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int r = (int)((it + u) * (64 / G) + group);   // the region this group copies: lane r's
            rnf[u] = __shfl(nf, r); rnb[u] = __shfl(nb, r);
            rfp[u] = __shfl(fp, r); rap[u] = __shfl(ap, r);
            slot[u] = P.cache_arena + (u64)__shfl(slot_no, r) * g.slot_bytes;
            fv[u] = sub < 2 * rnf[u] ? ((const uint4*)slot[u])[sub] : uint4{ 0, 0, 0, 0 };
            const u8* src = slot[u] + bytes_at;   // 16-aligned
            head[u] = (u32)((4 - ((uintptr_t)(arena + rap[u]) & 3)) & 3);
            if (head[u] > rnb[u]) head[u] = rnb[u];
            body[u] = (rnb[u] - head[u]) >> 2;
            const u32* sw = (const u32*)src;   // the dword that holds src[head] (head < 4)
            lo[u] = sub < body[u] ? sw[sub] : 0;
            hi[u] = sub < body[u] && head[u] ? sw[sub + 1] : 0;   // (slots lie inside the arena: one dword on is readable)
            hb[u] = sub < head[u] ? src[sub] : (u8)0;
            const u32 tail = (rnb[u] - head[u]) & 3;
            tb[u] = sub < tail ? src[head[u] + body[u] * 4 + sub] : (u8)0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 add = (u32)rap[u] + P.str_off_base;
            uint4* of = (uint4*)(findings + rfp[u]);
            u8* dst = arena + rap[u];
            u32* d = (u32*)(dst + head[u]);
            if (sub < 2 * rnf[u]) {
                uint4 v = fv[u];
                if (!(sub & 1)) v.z += add;
                of[sub] = v;
            }
            if (sub < head[u]) dst[sub] = hb[u];
            if (sub < body[u]) d[sub] = head[u] ? __builtin_amdgcn_alignbyte(hi[u], lo[u], head[u]) : lo[u];
            const u32 tail = (rnb[u] - head[u]) & 3;
            if (sub < tail) dst[head[u] + body[u] * 4 + sub] = tb[u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u32 add = (u32)rap[u] + P.str_off_base;
            const uint4* cf = (const uint4*)slot[u];
            uint4* of = (uint4*)(findings + rfp[u]);
            for (u32 t = sub + G; t < 2 * rnf[u]; t += G) {
                uint4 v = cf[t];
                if (!(t & 1)) v.z += add;
                of[t] = v;
            }
            const u32* sw = (const u32*)(slot[u] + bytes_at);
            u32* d = (u32*)(arena + rap[u] + head[u]);
            for (u32 k = sub + G; k < body[u]; k += G) {
                const u32 l = sw[k];
                d[k] = head[u] ? __builtin_amdgcn_alignbyte(sw[k + 1], l, head[u]) : l;
            }
        }
    }
}

// ---- long runs cut into pieces at the window starts they cross (sx_replay_core.hpp kPieceCont) ----------
// P.runs = the joined runs; counts -> exclusive scan -> one thread per piece writes it.
template <int ENC>
__global__ __launch_bounds__(256) void split_count_kernel(const ReplayParams P, u64* counts) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < P.n_runs) counts[i] = split_count<ENC>(P, i);
}
__global__ void split_total_kernel(const u64* counts, const u64* offsets, u64 n, u64* total) { *total = offsets[n - 1] + counts[n - 1]; }
__global__ __launch_bounds__(256) void split_write_kernel(const ReplayParams P, const u64* offsets, u64 n_pieces, sx_run* out) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pieces) return;
    u64 lo = 0, hi = P.n_runs;   // the last run whose first piece is at or before j
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) / 2;
        if (offsets[mid] <= j) lo = mid; else hi = mid;
    }
    const u64 count = (lo + 1 < P.n_runs ? offsets[lo + 1] : n_pieces) - offsets[lo];
    out[j] = split_piece(P, lo, j - offsets[lo], count);
}
size_t split_scratch_bytes(uint64_t n_runs) {
    size_t a = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (const u64*)nullptr, (u64*)nullptr, (u64)0, (size_t)n_runs, rocprim::plus<u64>(), (hipStream_t)0);
    return a + 2 * n_runs * 8 + 1024;
}
// scratch: [counts n][offsets n][rocprim]; *d_total (device) = number of pieces
hipError_t launch_split_count(const ReplayParams& P, void* scratch, size_t scratch_bytes, uint64_t* d_total, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    u64* counts = (u64*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    u64* offsets = counts + P.n_runs;
    void* tmp = (void*)(((uintptr_t)(offsets + P.n_runs) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)((uint8_t*)tmp - (uint8_t*)scratch);
    const dim3 grid((unsigned)((P.n_runs + 255) / 256));
    switch (enc_family(P.encoding)) {
        case 1: hipLaunchKernelGGL(split_count_kernel<1>, grid, dim3(256), 0, stream, P, counts); break;
        case 2: hipLaunchKernelGGL(split_count_kernel<2>, grid, dim3(256), 0, stream, P, counts); break;
        case 3: hipLaunchKernelGGL(split_count_kernel<3>, grid, dim3(256), 0, stream, P, counts); break;
        case 4: hipLaunchKernelGGL(split_count_kernel<4>, grid, dim3(256), 0, stream, P, counts); break;
        case 5: hipLaunchKernelGGL(split_count_kernel<5>, grid, dim3(256), 0, stream, P, counts); break;
        default: hipLaunchKernelGGL(split_count_kernel<0>, grid, dim3(256), 0, stream, P, counts); break;
    }
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, (const u64*)counts, offsets, (u64)0, (size_t)P.n_runs, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(split_total_kernel, dim3(1), dim3(1), 0, stream, counts, offsets, P.n_runs, d_total);
    return hipGetLastError();
}
hipError_t launch_split_write(const ReplayParams& P, const void* scratch, uint64_t n_pieces, sx_run* out, hipStream_t stream) {
    if (n_pieces == 0) return hipSuccess;
    const u64* counts = (const u64*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    hipLaunchKernelGGL(split_write_kernel, dim3((unsigned)((n_pieces + 255) / 256)), dim3(256), 0, stream, P, counts + P.n_runs, n_pieces, out);
    return hipGetLastError();
}

// ---- which regions stand, on the device ---------------------------------------------------
// The rule is sequential (a region is void if an earlier standing one ran over its start:
// E = end of the last standing region; region i stands iff want_i >= E), but regions rarely
// reach their successor, so: every block of stitch_block_runs() runs is resolved on its own as if
// nothing reached into it (its first candidate stands); then one wavefront walks the block
// summaries in order and repairs the few blocks whose entry was overrun, following the true
// chain only until it meets the block's own chain again.
struct StitchBlock { u64 first_want, end; u64 last; };  // first candidate's window, E after the block, last standing run

SXD u64 wave_prefix_max_excl(u64 x, u32 lane) {  // max over lanes below `lane` (0 for lane 0)
    u64 v = x;
    for (int o = 1; o < 64; o <<= 1) {
        const u64 t = __shfl_up(v, o);
        if ((int)lane >= o && t > v) v = t;
    }
    const u64 up = __shfl_up(v, 1);
    return lane ? up : 0ull;
}

__global__ __launch_bounds__(64) void stitch_blocks_kernel(const ReplayParams P, const ReplayRegionOut* ro, u8* stands,
                                                           StitchBlock* blocks, u64 n_blocks, u32 per_block, u64* totals) {
    // Round 5: a WAVEFRONT per block, 64 consecutive runs at a time (rounds 2-4: a lane per block walked its 512 runs one after the
    // other, every load and store of the wavefront scattered over 64 cache lines — 87 wavefronts waiting on memory for 0.3 ms on an
    // idle chip and for 3.5 ms next to a scan kernel).  Inside a group the rule is settled the way stitch_chain_kernel settles
    // blocks: everything stands whose window start is not below the ends in front of it (a prefix maximum); the first run that is
    // overrun is void, its end leaves the maximum, and the lanes behind it are looked at again — one round per void run.
    const u64 b = blockIdx.x;
    if (b >= n_blocks) return;
    const u32 lane = threadIdx.x;
    const u64 i0 = b * per_block, i1 = i0 + per_block < P.n_runs ? i0 + per_block : P.n_runs;
    u64 first_want = ~0ull, E = 0, last = ~0ull;   // wave-uniform
    u32 too_long = 0;
    for (u64 g = i0; g < i1; g += 64) {
        const u64 i = g + lane;
        const bool in = i < i1;
        const u32 st = in ? ro[i].status : (u32)kRegionNotMine;
        const bool ok = in && st == kRegionOk;
        const u64 end = ok ? ro[i].end : 0ull;
        const u64 w = ok ? win_start(P.runs[i].start, P.W) : 0ull;
        too_long += (u32)__popcll(__ballot(in && st == kRegionTooLong));
        const unsigned long long okm = __ballot(ok);
        if (first_want == ~0ull && okm) first_want = __shfl(w, (int)__ffsll((long long)okm) - 1);
        bool f = false;
        u32 cur = 0;   // lanes below cur are settled
        while (cur < 64) {
            const bool live = ok && lane >= cur;
            u64 e_in = wave_prefix_max_excl(live ? end : 0ull, lane);
            if (e_in < E) e_in = E;
            const unsigned long long badmask = __ballot(live && e_in > w);
            const u32 v = badmask ? (u32)__ffsll((long long)badmask) - 1u : 64u;   // the first run that is overrun: void
            if (live && lane < v) f = true;
            const unsigned long long stood = __ballot(live && lane < v);
            if (stood) {
                const int top = 63 - __clzll((long long)stood);
                E = __shfl(end, top);
                last = g + (u64)top;
            }
            cur = v + 1;
        }
        if (in) stands[i] = f ? 1 : 0;
    }
    if (lane == 0) {
        StitchBlock sb; sb.first_want = first_want; sb.end = E; sb.last = last;
        blocks[b] = sb;
        if (too_long) atomicAdd((unsigned long long*)&totals[kTotTooLong], (unsigned long long)too_long);
    }
}

__global__ __launch_bounds__(64) void stitch_chain_kernel(const ReplayParams P, const ReplayRegionOut* ro, u8* stands,
                                                          const StitchBlock* blocks, u64 n_blocks, u32 per_block, u64 E0, u64* totals) {
    const u32 lane = threadIdx.x;
    u64 E = E0, last = ~0ull;  // wave-uniform
    StitchBlock next; next.first_want = ~0ull; next.end = 0; next.last = ~0ull;
    if (lane < n_blocks) next = blocks[lane];
    for (u64 base = 0; base < n_blocks; base += 64) {
        const StitchBlock mine = next;
        // the summaries of the next 64 blocks are on their way while these are chained (the loop is a
        // chain of dependent steps: without this every step waits for a DRAM round trip)
        next.first_want = ~0ull; next.end = 0; next.last = ~0ull;
        if (base + 64 + lane < n_blocks) next = blocks[base + 64 + lane];
        const bool has = mine.first_want != ~0ull;
        u32 cur = 0;  // lanes below cur are settled
        while (cur < 64) {
            // As long as nothing reaches into a block, the ends of the blocks' own chains grow
            // with the block index, so E at lane l = max(E, ends of the lanes in [cur, l)).
            const u64 contrib = (has && lane >= cur) ? mine.end : 0ull;
            u64 e_in = wave_prefix_max_excl(contrib, lane);
            if (e_in < E) e_in = E;
            const bool bad = has && lane >= cur && e_in > mine.first_want;
            const unsigned long long badmask = __ballot(bad);
            const u32 v = badmask ? (u32)__ffsll((long long)badmask) - 1u : 64u;  // first overrun block
            // settle lanes [cur, v): their own chains stand
            const unsigned long long okmask = __ballot(has && lane >= cur && lane < v);
            if (okmask) {
                const int top = 63 - __clzll((long long)okmask);
                E = __shfl(mine.end, top);
                last = __shfl(mine.last, top);
            }
            if (v >= 64) break;
            // repair block v: follow the true chain until it meets the block's own chain again
            // (every lane does the same work on the same data; lane 0 stores)
            const u64 be = __shfl(mine.end, (int)v), bl = __shfl(mine.last, (int)v);
            const u64 b = base + v, i0 = b * per_block, i1 = i0 + per_block < P.n_runs ? i0 + per_block : P.n_runs;
            for (u64 i = i0; i < i1; i++) {
                if (ro[i].status != kRegionOk) continue;
                const u64 w = win_start(P.runs[i].start, P.W);
                const u8 was = stands[i];
                if (w < E) { if (lane == 0 && was) stands[i] = 0; continue; }
                if (was) { E = be; last = bl; break; }
                if (lane == 0) stands[i] = 1;
                E = ro[i].end; last = i;
            }
            cur = v + 1;
        }
    }
    if (lane == 0) {
        totals[kTotEnd] = E; totals[kTotLast] = last;
        totals[kTotLastStart] = last == ~0ull ? 0ull : win_start(P.runs[last].start, P.W);   // (the host need not hold the run list for it)
    }
}

struct StandingFindings {
    const u8* stands; const ReplayRegionOut* ro;
    __device__ u64 operator()(u64 i) const { return stands[i] ? (u64)ro[i].n_find : 0ull; }
};
struct StandingBytes {
    const u8* stands; const ReplayRegionOut* ro;
    __device__ u64 operator()(u64 i) const { return stands[i] ? (u64)ro[i].n_bytes : 0ull; }
};
// (256-thread blocks: anything larger starves next to a scan kernel that keeps refilling the CUs)
__global__ __launch_bounds__(256) void stitch_totals_kernel(const ReplayParams P, const ReplayRegionOut* ro, const u8* stands,
                                                            const u64* fpos, const u64* apos, u64* totals) {
    __shared__ u64 s_cnt[4], s_rb[4];
    // four runs per thread: 1024 runs per block, one pair of atomics per block
    const u64 i0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 4;
    u64 cnt = 0, rb = 0;
    for (u32 j = 0; j < 4; j++) {
        const u64 i = i0 + j;
        if (i < P.n_runs && stands[i]) { cnt++; rb += ro[i].end - win_start(P.runs[i].start, P.W); }
        if (i + 1 == P.n_runs) {
            totals[kTotFindings] = fpos[i] + (stands[i] ? ro[i].n_find : 0);
            totals[kTotBytes] = apos[i] + (stands[i] ? ro[i].n_bytes : 0);
        }
    }
    for (int o = 32; o; o >>= 1) { cnt += __shfl_down(cnt, o); rb += __shfl_down(rb, o); }
    if ((threadIdx.x & 63) == 0) { s_cnt[threadIdx.x >> 6] = cnt; s_rb[threadIdx.x >> 6] = rb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const u64 c = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], r = s_rb[0] + s_rb[1] + s_rb[2] + s_rb[3];
        if (c) {
            atomicAdd((unsigned long long*)&totals[kTotStanding], (unsigned long long)c);
            atomicAdd((unsigned long long*)&totals[kTotReplayBytes], (unsigned long long)r);
        }
    }
}

size_t stitch_scratch_bytes(uint64_t n_runs) {
    size_t a = 0;
    auto it = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), StandingFindings{ nullptr, nullptr });
    (void)rocprim::exclusive_scan(nullptr, a, it, (u64*)nullptr, (u64)0, (size_t)n_runs, rocprim::plus<u64>(), (hipStream_t)0);
    return a + 512;
}
uint64_t stitch_block_count(uint64_t n_runs) { const uint64_t k = stitch_block_runs(n_runs); return (n_runs + k - 1) / k; }
size_t stitch_blocks_bytes(uint64_t n_runs) { return stitch_block_count(n_runs) * sizeof(StitchBlock) + 64; }

// stage 1 (no dependency on the entry region): the blocks' own chains.  totals must be zeroed.
hipError_t launch_stitch_blocks(const ReplayParams& P, const ReplayRegionOut* ro, uint8_t* stands, void* blocks,
                                uint64_t* totals, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    const u64 nb = stitch_block_count(P.n_runs);
    hipLaunchKernelGGL(stitch_blocks_kernel, dim3((unsigned)nb), dim3(64), 0, stream, P, ro, stands,
                       (StitchBlock*)blocks, nb, stitch_block_runs(P.n_runs), totals);
    return hipGetLastError();
}
// stage 2: chain the blocks from E0 (end of the host's entry region), assign output offsets, totals
hipError_t launch_stitch_finish(const ReplayParams& P, const ReplayRegionOut* ro, uint8_t* stands, const void* blocks,
                                uint64_t E0, uint64_t* fpos, uint64_t* apos, uint64_t* totals, void* scratch,
                                size_t scratch_bytes, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    const u64 nb = stitch_block_count(P.n_runs);
    hipLaunchKernelGGL(stitch_chain_kernel, dim3(1), dim3(64), 0, stream, P, ro, stands, (const StitchBlock*)blocks, nb, stitch_block_runs(P.n_runs), E0, totals);
    void* tmp = (void*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)((uint8_t*)tmp - (uint8_t*)scratch);
    auto itf = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), StandingFindings{ stands, ro });
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, itf, fpos, (u64)0, (size_t)P.n_runs, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    tmp_bytes = scratch_bytes - (size_t)((uint8_t*)tmp - (uint8_t*)scratch);
    auto itb = rocprim::make_transform_iterator(rocprim::counting_iterator<u64>(0), StandingBytes{ stands, ro });
    e = rocprim::exclusive_scan(tmp, tmp_bytes, itb, apos, (u64)0, (size_t)P.n_runs, rocprim::plus<u64>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stitch_totals_kernel, dim3((unsigned)((P.n_runs + 1023) / 1024)), dim3(256), 0, stream, P, ro, stands, fpos,
                       apos, totals);
    return hipGetLastError();
}
hipError_t launch_replay_write_flagged(const ReplayParams& P, const ReplayRegionOut* ro, const uint8_t* stands,
                                       const uint64_t* fpos, const uint64_t* apos, sx_finding* findings,
                                       uint8_t* arena, uint64_t avg_out_bytes, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    const dim3 grid((unsigned)((P.n_runs + 63) / 64));
    switch ((int)enc_family(P.encoding) + (P.q > 64 ? 8 : 0)) {
        case 1: hipLaunchKernelGGL(replay_write_flagged_kernel<1>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 2: hipLaunchKernelGGL(replay_write_flagged_kernel<2>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 3: hipLaunchKernelGGL(replay_write_flagged_kernel<3>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 4: hipLaunchKernelGGL(replay_write_flagged_kernel<4>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 5: hipLaunchKernelGGL(replay_write_flagged_kernel<5>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 9: hipLaunchKernelGGL(replay_write_flagged_kernel<9>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 10: hipLaunchKernelGGL(replay_write_flagged_kernel<10>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 11: hipLaunchKernelGGL(replay_write_flagged_kernel<11>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 12: hipLaunchKernelGGL(replay_write_flagged_kernel<12>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 13: hipLaunchKernelGGL(replay_write_flagged_kernel<13>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        case 8: hipLaunchKernelGGL(replay_write_flagged_kernel<8>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
        default: hipLaunchKernelGGL(replay_write_flagged_kernel<0>, grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena); break;
    }
    if (P.cache_arena) {
        if (avg_out_bytes <= 96) hipLaunchKernelGGL((replay_copy_cached_kernel<4, 2>), grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena);
        else if (avg_out_bytes <= 640) hipLaunchKernelGGL((replay_copy_cached_kernel<32, 4>), grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena);
        else hipLaunchKernelGGL((replay_copy_cached_kernel<64, 4>), grid, dim3(64), 0, stream, P, ro, stands, fpos, apos, findings, arena);
    }
    return hipGetLastError();
}

size_t replay_heads_scratch_bytes(uint64_t n_runs) {
    size_t a = 0;
    (void)rocprim::exclusive_scan(nullptr, a, (u32*)nullptr, (u32*)nullptr, 0u, (size_t)n_runs, rocprim::plus<u32>(), (hipStream_t)0);
    return n_runs * 4 + a + 1024;
}
hipError_t launch_replay_heads(const ReplayParams& P, uint32_t* slot_of, uint32_t* n_heads, uint32_t* head_list, ReplayRegionOut* ro,
                               void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (P.n_runs == 0) return hipMemsetAsync(n_heads, 0, 4, stream);
    if (scratch_bytes < replay_heads_scratch_bytes(P.n_runs)) return hipErrorInvalidValue;
    u32* head = (u32*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    void* tmp = (void*)(((uintptr_t)(head + P.n_runs) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)((u8*)tmp - (u8*)scratch);
    hipLaunchKernelGGL(replay_heads_kernel, dim3((unsigned)((P.n_runs + 255) / 256)), dim3(256), 0, stream, P, head, ro);
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, head, slot_of, 0u, (size_t)P.n_runs, rocprim::plus<u32>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(replay_heads_list_kernel, dim3((unsigned)((P.n_runs + 255) / 256)), dim3(256), 0, stream, P, head, slot_of, head_list);
    hipLaunchKernelGGL(replay_heads_total_kernel, dim3(1), dim3(1), 0, stream, head + (P.n_runs - 1), slot_of + (P.n_runs - 1), n_heads);
    return hipGetLastError();
}
hipError_t launch_replay_count(const ReplayParams& P, ReplayRegionOut* out, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    dim3 grid((unsigned)((P.n_runs + 63) / 64));
    const bool cache = P.cache_arena != nullptr;
    if (cache && P.hard_list && grid.x > 4096u) grid.x = 4096u;   // (behind the fast pre-pass: a short list, walked by a bounded grid)
    const unsigned lds = 64u * win_row_bytes(P.W);
    static const int waves = [] { const char* e = getenv("SX_COUNT_WAVES"); return e ? atoi(e) : 4; }();
#define SX_LAUNCH_COUNT(E)                                                                                              \
    do {                                                                                                                \
        if (!cache) hipLaunchKernelGGL((replay_count_kernel<E>), grid, dim3(64), 0, stream, P, out);                     \
        else if (waves >= 8) hipLaunchKernelGGL((replay_count_cached_kernel<E, 8>), grid, dim3(64), lds, stream, P, out); \
        else if (waves >= 6) hipLaunchKernelGGL((replay_count_cached_kernel<E, 6>), grid, dim3(64), lds, stream, P, out); \
        else hipLaunchKernelGGL((replay_count_cached_kernel<E, 4>), grid, dim3(64), lds, stream, P, out);                 \
    } while (0)
    switch ((int)enc_family(P.encoding) + (P.q > 64 ? 8 : 0)) {
        case 1: SX_LAUNCH_COUNT(1); break;
        case 2: SX_LAUNCH_COUNT(2); break;
        case 3: SX_LAUNCH_COUNT(3); break;
        case 4: SX_LAUNCH_COUNT(4); break;
        case 5: SX_LAUNCH_COUNT(5); break;
        case 8: SX_LAUNCH_COUNT(8); break;
        case 9: SX_LAUNCH_COUNT(9); break;
        case 10: SX_LAUNCH_COUNT(10); break;
        case 11: SX_LAUNCH_COUNT(11); break;
        case 12: SX_LAUNCH_COUNT(12); break;
        case 13: SX_LAUNCH_COUNT(13); break;
        default: SX_LAUNCH_COUNT(0); break;
    }
#undef SX_LAUNCH_COUNT
    return hipGetLastError();
}
// Where a run list may be cut into slabs that are replayed one after the other (sx_stage_b.cpp): at a run that starts a
// region of its own (not chained), the first one at or behind j/n_slabs of the list.  idx[j-1], hi[j-1] = that run and its
// window start, for j = 1 .. n_slabs-1.
__global__ void slab_cuts_kernel(const ReplayParams P, u32 n_slabs, u64* idx, u64* hi) {
    const u32 j = threadIdx.x + 1;
    if (j >= n_slabs) return;
    u64 i = P.n_runs / n_slabs * j;
    while (i < P.n_runs && run_is_chained(P, i, win_start(P.runs[i].start, P.W))) i++;
    idx[j - 1] = i;
    hi[j - 1] = i < P.n_runs ? win_start(P.runs[i].start, P.W) : ~0ull;
}
hipError_t launch_slab_cuts(const ReplayParams& P, uint32_t n_slabs, uint64_t* idx, uint64_t* hi, hipStream_t stream) {
    if (n_slabs < 2 || n_slabs > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(slab_cuts_kernel, dim3(1), dim3(64), 0, stream, P, n_slabs, idx, hi);
    return hipGetLastError();
}
hipError_t launch_replay_write(const ReplayParams& P, const u64* region_index, const u64* fbase, const u64* abase,
                               u64 n_regions, sx_finding* findings, u8* arena, hipStream_t stream) {
    if (n_regions == 0) return hipSuccess;
    const dim3 grid((unsigned)((n_regions + 63) / 64));
    switch ((int)enc_family(P.encoding) + (P.q > 64 ? 8 : 0)) {
        case 1: hipLaunchKernelGGL(replay_write_kernel<1>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 2: hipLaunchKernelGGL(replay_write_kernel<2>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 3: hipLaunchKernelGGL(replay_write_kernel<3>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 4: hipLaunchKernelGGL(replay_write_kernel<4>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 5: hipLaunchKernelGGL(replay_write_kernel<5>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 9: hipLaunchKernelGGL(replay_write_kernel<9>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 10: hipLaunchKernelGGL(replay_write_kernel<10>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 11: hipLaunchKernelGGL(replay_write_kernel<11>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 12: hipLaunchKernelGGL(replay_write_kernel<12>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 13: hipLaunchKernelGGL(replay_write_kernel<13>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        case 8: hipLaunchKernelGGL(replay_write_kernel<8>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
        default: hipLaunchKernelGGL(replay_write_kernel<0>, grid, dim3(64), 0, stream, P, region_index, fbase, abase, n_regions, findings, arena); break;
    }
    return hipGetLastError();
}

}  // namespace sx
