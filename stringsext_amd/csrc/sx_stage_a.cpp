// sx_stage_a.cpp — stage A on the host side: launch every mission's scan kernel, collect the run
// records (region mode or shared pool), order and join them into long runs on the device.
#include "sx_ctx.hpp"

using namespace sx;

namespace sx {

// Regions large enough for string-dense input cost memory: at most half the input's size (and 32-bit slot numbers).
static bool large_regions_fit(uint64_t len, uint64_t n_regions, uint32_t cap) {
    const uint64_t slots = n_regions * cap;
    return slots < (1ull << 31) && slots * sizeof(DevRun) <= std::max<uint64_t>(256ull << 20, len / 2);
}

ScanParams scan_params(const sx_ctx* ctx, int mission, const ScanSlot& s, const uint8_t* d_bytes, uint64_t len,
                       uint32_t parity, uint64_t min_chars) {
    const Mission& m = ctx->missions[(size_t)mission];
    uint32_t sub = ctx->opt.subchunk_bytes ? ctx->opt.subchunk_bytes : 256u * 1024u;
    sub = std::max<uint32_t>(kTileBytes, sub / kTileBytes * kTileBytes);
    ScanParams p = m.proto;
    p.data = d_bytes; p.len = len; p.subchunk = sub; p.parity = parity;
    p.pair_lut = ctx->dev[(size_t)mission].d_pair_lut;
    if (p.gb4) {   // (sx_codec_core.hpp: the decoder's blob is [kGbN two-byte cells][breakpoints][code points])
        p.gb_ranges = ctx->dev[(size_t)mission].d_table ? ctx->dev[(size_t)mission].d_table + kGbN : nullptr;
        p.ubf = m.c.ubf;
        if (!p.gb_ranges) p.gb4 = 0;
    }
    p.wave_prio = getenv("SX_SCAN_PRIO") ? (uint32_t)atoi(getenv("SX_SCAN_PRIO")) : 0u;
    p.min_chars = (uint32_t)std::min<uint64_t>(min_chars, kRecCharsMask);
    if (p.min_chars == 0) p.min_chars = 1;
    // (<= 14: the scan kernel's candidate test takes the previous lane's mask without what ITS predecessor spilled into it — at most
    // the window's bits 0..2, a three-byte continuation of the LUT classifiers —, and with 14 no bit of the test looks below bit 3)
    p.cand_bytes = std::min<uint32_t>(p.min_chars * (m.is_utf16() ? 2u : 1u), 14u);
    {   // r &= r << sh, doubling the proven run length until it reaches cand_bytes
        uint32_t have = 1;
        for (int i = 0; i < 5; i++) {
            const uint32_t sh = have < p.cand_bytes ? std::min(have, p.cand_bytes - have) : 0;
            p.cand_sh[i] = sh;
            have += sh;
        }
    }
    p.capacity = s.capacity; p.recs = s.d_recs; p.counters = s.d_counters;
    p.region_cap = s.region_cap; p.region_counts = s.d_cnt; p.grid_flags = s.d_grid;
    // Blocks (of 4 wavefronts) per CU the scan kernel occupies.  8 fills every wave slot; with
    // fewer the kernel runs as a persistent grid and leaves the rest to the second stream
    // (sort/join and stage B of a mission that is already scanned).
    unsigned occ = ctx->scan_blocks_per_cu;
    p.persistent = (occ >= 1 && occ < 8) ? occ * ctx->n_cus : 0u;
    return p;
}

// Stage A, first half: enqueue every mission's scan kernel over [d_bytes, d_bytes+len) on its
// scan stream, writing into record slot `si`.  Returns at once.
int stage_a_launch(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                   const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars, int si) {
    if (len == 0) return SX_OK;
    for (size_t k = 0; k < which.size(); k++) {
        MissionDev& d = ctx->dev[(size_t)which[k]];
        ScanSlot& s = d.slot[si];
        if (s.free_pending) { HIP_TRY(ctx, hipStreamWaitEvent(d.stream, s.ev_free, 0)); s.free_pending = false; }
        {   // region mode unless the mission's last buffer was too dense for it (or the options rule it out)
            uint32_t sub = ctx->opt.subchunk_bytes ? ctx->opt.subchunk_bytes : 256u * 1024u;
            sub = std::max<uint32_t>(kTileBytes, sub / kTileBytes * kTileBytes);
            const uint64_t n_regions = (len + sub - 1) / sub;
            if (ctx->dense.size() != ctx->missions.size()) ctx->dense.assign(ctx->missions.size(), 0);
            s.region_cap = 0; s.n_regions = n_regions;
            const uint32_t dn = ctx->dense[(size_t)which[k]];
            const uint32_t want_cap = dn > 1 ? dn : (dn == 0 ? ctx->region_cap : 0u);
            if (ctx->region_cap && want_cap &&
                (dn > 1 ? large_regions_fit(len, n_regions, want_cap) : n_regions * want_cap < (1ull << 28))) {
                s.region_cap = want_cap;
                int rc = ensure_capacity(ctx, s, (uint32_t)(n_regions * s.region_cap));
                if (rc != SX_OK && dn > 1) {  // no room for large regions in this buffer: the pool
                    (void)hipGetLastError();
                    ctx->dense[(size_t)which[k]] = 1;
                    s.region_cap = 0;
                    rc = SX_OK;
                }
                if (rc != SX_OK) return rc;
                if (s.region_cap == 0) goto launch;
                if (s.cnt_cap < n_regions) {
                    if (s.d_cnt) HIP_TRY(ctx, hipFree(s.d_cnt));
                    s.d_cnt = nullptr; s.cnt_cap = 0;
                    HIP_TRY(ctx, hipMalloc((void**)&s.d_cnt, (n_regions + n_regions / 4 + 64) * 4));
                    s.cnt_cap = n_regions + n_regions / 4 + 64;
                }
            }
        }
    launch:
        const bool dbcs = ctx->missions[(size_t)which[k]].is_dbcs();
        if (dbcs) {   // a flag word per sub-chunk: where its token grid stands, for the sub-chunks behind it (sx_kernels.hip scan_kernel_dbcs)
            uint32_t sub = ctx->opt.subchunk_bytes ? ctx->opt.subchunk_bytes : 256u * 1024u;
            sub = std::max<uint32_t>(kTileBytes, sub / kTileBytes * kTileBytes);
            const uint64_t n_sub = (len + sub - 1) / sub;
            if (s.grid_cap < n_sub) {
                if (s.d_grid) HIP_TRY(ctx, hipFree(s.d_grid));
                s.d_grid = nullptr; s.grid_cap = 0;
                HIP_TRY(ctx, hipMalloc((void**)&s.d_grid, (n_sub + n_sub / 4 + 64) * 4));
                s.grid_cap = n_sub + n_sub / 4 + 64;
            }
        }
    }
    // Round 6: the Missions whose classifiers have a slot in the fused kernel (sx_fused.hip) share ONE launch — the buffer is read once for
    // all of them, as the reference hands one slice to every Mission (src/main.rs:153-168).  Per Mission everything else is as before: its
    // own record slot, counters and events (ev0 / ev1 of every fused Mission bracket the same launch, whose time stats.fused_ms counts once).
    std::vector<char> fused(which.size(), 0);
    std::vector<ScanParams> params(which.size());
    for (size_t k = 0; k < which.size(); k++)
        params[k] = scan_params(ctx, which[k], ctx->dev[(size_t)which[k]].slot[si], d_bytes, len, parity[k], min_chars[k]);
    const bool may_fuse = !(ctx->opt.flags & (SX_OPT_NO_FUSED_SCAN | SX_OPT_MISSION_STREAMS)) && !getenv("SX_MISSION_STREAMS") &&
                          !(getenv("SX_FUSED") && !atoi(getenv("SX_FUSED"))) && !getenv("SX_SCAN_WARM");
    if (may_fuse) {
        FusedParams fp{};
        uint32_t used = 0;
        size_t member[kFusedMax] = { 0, 0, 0 };
        for (size_t k = 0; k < which.size(); k++) {
            if (params[k].persistent) continue;
            const int sl = fused_slot_of(ctx->missions[(size_t)which[k]].kind, params[k]);
            if (sl < 0 || (used >> sl) & 1u) continue;   // (a second Mission of the same slot keeps its own launch)
            // (the kernel knows ONE unit parity at compile time: the UTF-16 Missions of a stream share it — they have consumed the same
            //  bytes —; one that does not keeps its own launch)
            if (sl >= 1 && (used & 6u) && ((fp.m[(used & 2u) ? 1 : 2].parity ^ params[k].parity) & 1u)) continue;
            used |= 1u << sl; fp.m[sl] = params[k]; member[sl] = k;
        }
        // (two Missions or more — or the UTF-8 range Mission alone: the fused kernel's fast loop is the faster scan of it)
        if (__builtin_popcount(used) >= 2 || used == 1u) {
            hipStream_t st = nullptr;
            for (int sl = 0; sl < kFusedMax; sl++) if ((used >> sl) & 1u) {
                MissionDev& d = ctx->dev[(size_t)which[member[sl]]];
                if (!st) st = d.stream;
                HIP_TRY(ctx, hipMemsetAsync(d.slot[si].d_counters, 0, kCounterWords * sizeof(uint32_t), st));
            }
            for (int sl = 0; sl < kFusedMax; sl++) if ((used >> sl) & 1u) HIP_TRY(ctx, hipEventRecord(ctx->dev[(size_t)which[member[sl]]].slot[si].ev0, st));
            HIP_TRY(ctx, launch_scan_fused(fp, used, st));
            bool first = true;
            for (int sl = 0; sl < kFusedMax; sl++) if ((used >> sl) & 1u) {
                ScanSlot& s = ctx->dev[(size_t)which[member[sl]]].slot[si];
                HIP_TRY(ctx, hipEventRecord(s.ev1, st));
                s.fused = true; s.fused_first = first; first = false;
                fused[member[sl]] = 1;
            }
            ctx->stats.fused_launches++;
        }
    }
    for (size_t k = 0; k < which.size(); k++) {
        if (fused[k]) continue;
        MissionDev& d = ctx->dev[(size_t)which[k]];
        ScanSlot& s = d.slot[si];
        s.fused = false; s.fused_first = false;
        const bool dbcs = ctx->missions[(size_t)which[k]].is_dbcs();
        const ScanParams& p = params[k];
        const uint64_t n_sub = (len + p.subchunk - 1) / p.subchunk;
        // SX_SCAN_WARM=n (measurements): the launch is preceded by n identical ones, so that the timed one starts on a busy chip
        // (bench.py's "alone" launches start from an idle one and take ~1 ms longer: DESIGN §6)
        if (const char* e = getenv("SX_SCAN_WARM"))
            for (int w = atoi(e); w > 0; w--) {
                if (dbcs) HIP_TRY(ctx, hipMemsetAsync(s.d_grid, 0, n_sub * 4, d.stream));
                HIP_TRY(ctx, hipMemsetAsync(s.d_counters, 0, kCounterWords * sizeof(uint32_t), d.stream));
                HIP_TRY(ctx, launch_scan(ctx->missions[(size_t)which[k]].kind, p, d.stream));
            }
        if (dbcs) HIP_TRY(ctx, hipMemsetAsync(s.d_grid, 0, n_sub * 4, d.stream));
        HIP_TRY(ctx, hipMemsetAsync(s.d_counters, 0, kCounterWords * sizeof(uint32_t), d.stream));
        HIP_TRY(ctx, hipEventRecord(s.ev0, d.stream));
        HIP_TRY(ctx, launch_scan(ctx->missions[(size_t)which[k]].kind, p, d.stream));
        HIP_TRY(ctx, hipEventRecord(s.ev1, d.stream));
        if (dbcs) { d.grid_of_scan = s.d_grid; d.grid_sub = p.subchunk; d.grid_data = d_bytes; d.grid_len = len; }
    }
    return SX_OK;
}

// Stage A, second half: wait for slot `si`, re-run a mission whose record buffer overflowed,
// and turn the records into the mission's sorted long runs (joined on the device when there
// are many).  Only stream_b is used from here on: the scan streams may already hold the next piece.
int stage_a_finish(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                   const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars, int si,
                   std::vector<RunList>* out, bool cut_into_pieces, const ReplayJob* wave_job) {
    out->assign(which.size(), RunList{});
    if (len == 0) return SX_OK;
    const double t0 = now_ms();
    for (size_t k = 0; k < which.size(); k++) {
        MissionDev& d = ctx->dev[(size_t)which[k]];
        ScanSlot& s = d.slot[si];
        HIP_TRY(ctx, hipEventSynchronize(s.ev1));
        const double t_ev = now_ms();
        SX_TL("mission %d: scan kernel done", which[k]);
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, s.ev0, s.ev1));
        if (which[k] < 16) ctx->stats.kernel_ms[which[k]] += ms;
        if (s.fused && s.fused_first) ctx->stats.fused_ms += ms;
        if (which[k] < 64) { if (s.fused) ctx->stats.fused_mask |= 1ull << which[k]; else ctx->stats.fused_mask &= ~(1ull << which[k]); }
        uint32_t counters[4] = { 0, 0, 0, 0 };
        std::vector<uint32_t> hc(kCounterWords);
        bool skip_runs = false;
        for (int round = 0;; round++) {
            // (the two statistics arrive in shards — sx_device.hpp kStatBase —: one 2 KB copy, summed here)
            { const int rb = read_back_sync(ctx, d, d.stream_b, hc.data(), s.d_counters, kCounterWords * sizeof(uint32_t)); if (rb != SX_OK) return rb; }
            counters[0] = hc[0]; counters[3] = hc[3];
            {
                uint64_t heavy = 0, recs = 0;
                for (uint32_t sh = 0; sh < kStatShards; sh++) { heavy += hc[kStatBase + sh * kStatStride]; recs += hc[kStatBase + sh * kStatStride + 1]; }
                counters[1] = (uint32_t)std::min<uint64_t>(heavy, 0xFFFFFFFFull); counters[2] = (uint32_t)std::min<uint64_t>(recs, 0xFFFFFFFFull);
            }
            if (round == 0 && wave_job) {
                // String-dense input of a Mission whose stage B can replay every window (sx_wave.cpp): the records are only
                // counted — no second scan with larger regions, no sort, no join, no pieces.
                const uint64_t total = s.region_cap ? counters[2] : counters[0];
                if (wave_replay_wanted(ctx, *wave_job, (size_t)which[k], total, counters[1])) {
                    RunList& rl = (*out)[k];
                    // (n: what stage B sees as the buffer's density; a buffer of a few GIANT runs counts as dense too)
                    rl.n = wave_replay_wanted(ctx, *wave_job, (size_t)which[k], total) ? total : std::max<uint64_t>(total, len / 16);
                    rl.p = nullptr; rl.skipped = true; rl.complete = true;
                    skip_runs = true;
                    break;
                }
            }
            if (s.region_cap) {
                if (counters[0] == 0) break;  // every sub-chunk's records fit its region
                // Too dense for these regions.  If the fullest sub-chunk's records (+25 %) fit regions that
                // the memory allows, scan again with those — the kernel then appends without atomics, which
                // on string-dense input is 20x faster than the shared pool — else use the pool from now on.
                const uint32_t big = (counters[3] + counters[3] / 4 + 127) / 64 * 64;
                if (s.region_cap == ctx->region_cap && counters[3] && large_regions_fit(len, s.n_regions, big) && !getenv("SX_NO_LARGE_REGIONS")) {
                    ctx->dense[(size_t)which[k]] = big;
                    s.region_cap = big;
                    if (ensure_capacity(ctx, s, (uint32_t)(s.n_regions * big)) != SX_OK) {  // no room after all: the pool
                        (void)hipGetLastError();
                        ctx->dense[(size_t)which[k]] = 1;
                        s.region_cap = 0;
                    }
                } else {
                    ctx->dense[(size_t)which[k]] = 1;
                    s.region_cap = 0;
                }
            } else {
                if (counters[0] <= s.capacity) break;
                if (round >= 8) { ctx->err = "device run-record buffer kept overflowing"; return SX_E_NOMEM; }
                // overflow: grow the slot and scan this piece again for this mission
                int rc = ensure_capacity(ctx, s, counters[0] + counters[0] / 8 + 1024);
                if (rc != SX_OK) return rc;
            }
            const ScanParams p = scan_params(ctx, which[k], s, d_bytes, len, parity[k], min_chars[k]);
            const double tr0 = now_ms();
            if (p.grid_flags) HIP_TRY(ctx, hipMemsetAsync(p.grid_flags, 0, ((len + p.subchunk - 1) / p.subchunk) * 4, d.stream_b));
            HIP_TRY(ctx, hipMemsetAsync(s.d_counters, 0, kCounterWords * sizeof(uint32_t), d.stream_b));
            HIP_TRY(ctx, launch_scan(ctx->missions[(size_t)which[k]].kind, p, d.stream_b));
            if (p.grid_flags) { d.grid_of_scan = p.grid_flags; d.grid_sub = p.subchunk; d.grid_data = d_bytes; d.grid_len = len; }
            HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            ctx->stats.rescans++;
            ctx->stats.rescan_ms += now_ms() - tr0;
        }
        const double tc0 = now_ms();
        SX_TL("mission %d: counters read (%u records)", which[k], s.region_cap ? counters[2] : counters[0]);
        if (getenv("SX_TIMING2")) fprintf(stderr, "[sx]   mission %d: kernel done at +%.2f ms, counters at +%.2f ms\n", which[k], t_ev - t0, tc0 - t0);
        if (skip_runs) {
            HIP_TRY(ctx, hipEventRecord(s.ev_free, d.stream_b));
            s.free_pending = true;
            if (getenv("SX_TIMING")) fprintf(stderr, "[sx] mission %d: kernel %.2f ms, %llu records counted: string-dense, every window is replayed\n", which[k], ms, (unsigned long long)(*out)[k].n);
            ctx->stats.run_records += (*out)[k].n;
            ctx->stats.bytes_scanned += len;
            ctx->stats.heavy_tiles += counters[1];
            continue;
        }
        uint32_t nrec = counters[0];
        const DevRun* d_records = s.d_recs;   // sorted already in region mode
        const bool large_regions = s.region_cap > ctx->region_cap && ctx->region_cap;
        const bool regions = s.region_cap != 0 && !large_regions;
        if (large_regions) {
            // the slot array is treated like the pool: unused slots marked, everything sorted below
            HIP_TRY(ctx, invalidate_region_slack(s.d_recs, s.d_cnt, s.n_regions, s.region_cap, d.stream_b));
            nrec = (uint32_t)(s.n_regions * s.region_cap);
        }
        if (regions) {
            // pack the regions: the records come out ordered by position, no sort needed
            const uint64_t slots = s.n_regions * s.region_cap;
            if (s.packed_cap < slots) {
                if (s.d_packed) HIP_TRY(ctx, hipFree(s.d_packed));
                s.d_packed = nullptr; s.packed_cap = 0;
                HIP_TRY(ctx, hipMalloc((void**)&s.d_packed, (slots + slots / 8 + 64) * sizeof(DevRun)));
                s.packed_cap = slots + slots / 8 + 64;
            }
            int rc = ensure_scratch(ctx, compact_scratch_bytes(s.n_regions)); if (rc != SX_OK) return rc;
            HIP_TRY(ctx, compact_regions(s.d_recs, s.d_cnt, s.n_regions, s.region_cap, s.d_packed, s.d_counters + 1, ctx->d_scratch,
                                         ctx->d_scratch_cap, d.stream_b));
            { const int rb = read_back_sync(ctx, d, d.stream_b, &nrec, s.d_counters + 1, 4); if (rb != SX_OK) return rb; }
            d_records = s.d_packed;
            SX_TL("mission %d: regions packed (%u)", which[k], nrec);
        } else if (ctx->region_cap && !large_regions && nrec < s.n_regions * ctx->region_cap / 4)
            ctx->dense[(size_t)which[k]] = 0;  // sparse again: regions next time
        const uint32_t join_min = getenv("SX_DEVICE_JOIN_MIN") ? (uint32_t)atoi(getenv("SX_DEVICE_JOIN_MIN")) : 65536u;
        const bool dev_sorted = nrec >= join_min && nrec > 0;  // worth a handful of small kernels
        double tc1 = tc0;
        RunList& rl = (*out)[k];
        if (dev_sorted) {
            // sort the records and join them into runs on the device; only the runs travel
            const size_t sb = std::max(sort_scratch_bytes(nrec), merge_scratch_bytes(nrec));
            int rc = ensure_scratch(ctx, sb); if (rc != SX_OK) return rc;
            if (d.ev_runs) HIP_TRY(ctx, hipEventSynchronize(d.ev_runs));  // the previous list's copy (normally long done)
            rc = ensure_rp(ctx, d, 0, (uint64_t)nrec * sizeof(sx_run)); if (rc != SX_OK) return rc;
            if (!regions) HIP_TRY(ctx, sort_records(s.d_recs, nrec, len, ctx->d_scratch, ctx->d_scratch_cap, d.stream_b));
            if (getenv("SX_TIMING2")) { HIP_TRY(ctx, hipStreamSynchronize(d.stream_b)); fprintf(stderr, "[sx]   sort done +%.2f ms\n", now_ms() - tc0); }
            HIP_TRY(ctx, merge_sorted_records(d_records, nrec, min_chars[k], ctx->d_scratch, ctx->d_scratch_cap,
                                              (sx_run*)d.d_rp[0], s.d_counters + 2, d.stream_b));
            HIP_TRY(ctx, hipEventRecord(s.ev_free, d.stream_b));
            s.free_pending = true;
            if (getenv("SX_TIMING2")) { HIP_TRY(ctx, hipStreamSynchronize(d.stream_b)); fprintf(stderr, "[sx]   join done +%.2f ms\n", now_ms() - tc0); }
            uint32_t nruns32 = 0;
            { const int rb = read_back_sync(ctx, d, d.stream_b, &nruns32, s.d_counters + 2, 4); if (rb != SX_OK) return rb; }
            uint64_t nruns = nruns32;
            SX_TL("mission %d: joined (%u runs)", which[k], nruns32);
            const sx_run* d_list = (const sx_run*)d.d_rp[0];
            // Runs that cross window starts are cut into one piece per window where the state at those window
            // starts follows from the run alone (sx_replay_core.hpp kPieceCont): stage B then gets a region per
            // window instead of one serial replay per run (text: a run per line, most of them cross a window start).
            const Mission& mm = ctx->missions[(size_t)which[k]];
            if (cut_into_pieces && nruns && !getenv("SX_NO_PIECES") && mm.c.grep_char < 0 && !mm.c.require_same_unicode_block
                && mm.c.chars_min_nb >= 1 && mm.c.chars_min_nb <= mm.q && mm.q <= 255) {   // (q: what the replay kernels' buffers hold, sx_replay_core.hpp kObCapBig)
                ReplayParams SP{};
                SP.data = d_bytes; SP.len = len; SP.runs = d_list; SP.n_runs = nruns; SP.encoding = mm.c.encoding; SP.table = d.d_table;
                SP.chars_min_nb = mm.c.chars_min_nb; SP.same_block = 0; SP.q = (uint32_t)mm.q; SP.W = (uint32_t)mm.window; SP.grep_char = -1;
                rc = ensure_scratch(ctx, split_scratch_bytes(nruns) + 64); if (rc != SX_OK) return rc;
                uint64_t* d_total = (uint64_t*)ctx->d_scratch;   // the first 64 bytes; the rest is the split's scratch
                HIP_TRY(ctx, launch_split_count(SP, ctx->d_scratch + 64, ctx->d_scratch_cap - 64, d_total, d.stream_b));
                uint64_t total = 0;
                { const int rb = read_back_sync(ctx, d, d.stream_b, &total, d_total, 8); if (rb != SX_OK) return rb; }
                if (total > nruns && total < (1ull << 32) && ensure_rp(ctx, d, 9, total * sizeof(sx_run)) == SX_OK) {
                    HIP_TRY(ctx, launch_split_write(SP, ctx->d_scratch + 64, total, (sx_run*)d.d_rp[9], d.stream_b));
                    HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));   // the list's copy for the host runs on another stream: complete first
                    d_list = (const sx_run*)d.d_rp[9];
                    nruns = total;
                } else (void)hipGetLastError();
            }
            tc1 = now_ms();
            SX_TL("mission %d: cut into pieces (%llu)", which[k], (unsigned long long)nruns);
            if ((uint64_t)nruns * sizeof(sx_run) > d.h_runs_cap) {
                if (d.h_runs) HIP_TRY(ctx, hipHostFree(d.h_runs));
                d.h_runs = nullptr; d.h_runs_cap = 0;
                const uint64_t cap = (uint64_t)nruns * sizeof(sx_run) * 5 / 4 + 4096;
                HIP_TRY(ctx, hipHostMalloc((void**)&d.h_runs, cap, hipHostMallocNonCoherent));
                d.h_runs_cap = cap;
            }
            rl.p = d.h_runs; rl.n = nruns; rl.on_device = true; rl.dev_ptr = d_list;
            if (large_regions && nruns < s.n_regions * ctx->region_cap / 4) ctx->dense[(size_t)which[k]] = 0;  // sparse again
            if (nruns) {
                // The list is complete on the device (the count was just read).  Its copy for the host's
                // part of stage B runs on a stream of its own, started on demand: whoever reads rl.p[] calls
                // rl.wait() first; the device replay starts it once its first pass is under way.
                if (!ctx->d2h_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
                if (!d.ev_runs) HIP_TRY(ctx, hipEventCreateWithFlags(&d.ev_runs, hipEventDisableTiming));
                rl.dev_src = d_list; rl.copy_bytes = (size_t)nruns * sizeof(sx_run);
                rl.copy_stream = ctx->d2h_stream; rl.ready = d.ev_runs; rl.issued = false;
            }
        } else {
            int rc = ensure_pinned(ctx, (uint64_t)nrec * sizeof(DevRun) + 16);
            if (rc != SX_OK) return rc;
            DevRun* recs_p = (DevRun*)ctx->h_pin;
            if (nrec) {
                { const int rb = read_back_async(ctx, d.stream_b, recs_p, d_records, (size_t)nrec * sizeof(DevRun)); if (rb != SX_OK) return rb; }   // (pinned: sx_ctx::h_pin)
                HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            }
            tc1 = now_ms();
            if (getenv("SX_DEBUG_RECS")) {
                std::vector<DevRun> srt(recs_p, recs_p + nrec);
                std::sort(srt.begin(), srt.end(), [](const DevRun& a, const DevRun& b) { return a.start < b.start; });
                for (const DevRun& r : srt)
                    fprintf(stderr, "[sx] rec start=%llu len=%u chars=%u flags=%s%s\n", (unsigned long long)r.start, r.len,
                            r.chars_flags & kRecCharsMask, (r.chars_flags & kRecStartOpen) ? "S" : "-",
                            (r.chars_flags & kRecEndOpen) ? "E" : "-");
                fprintf(stderr, "[sx] slow tiles %u\n", counters[1]);
            }
            if (regions) merge_sorted_device_runs(recs_p, nrec, min_chars[k], &rl.own);
            else merge_device_runs(recs_p, nrec, min_chars[k], 64 * 1024, &rl.own);
            rl.use_own();
        }
        if (getenv("SX_TIMING"))
            fprintf(stderr, "[sx] mission %d: kernel %.2f ms, %u %s, %s %.2f ms, %s %.2f ms -> %zu runs\n", which[k], ms, nrec,
                    regions ? "records (regions)" : (large_regions ? "record slots (large regions)" : "record slots (pool)"), dev_sorted ? (regions ? "device pack+join" : "device sort+join") : "d2h",
                    tc1 - tc0, dev_sorted ? "d2h runs" : "host join", now_ms() - tc1, rl.size());
        ctx->stats.run_records += rl.size();
        ctx->stats.bytes_scanned += len;
        ctx->stats.heavy_tiles += counters[1];
    }
    ctx->stats.device_ms += now_ms() - t0;
    return SX_OK;
}

// Stage A over one buffer, start to end.
int device_runs(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars,
                std::vector<RunList>* out) {
    int rc = stage_a_launch(ctx, which, d_bytes, len, parity, min_chars, 0);
    if (rc != SX_OK) return rc;
    return stage_a_finish(ctx, which, d_bytes, len, parity, min_chars, 0, out);
}


}  // namespace sx
