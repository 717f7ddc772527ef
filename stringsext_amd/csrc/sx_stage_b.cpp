// sx_stage_b.cpp — stage B: the exact replay of FindingCollection::from around long runs, on the
// device (regions, which of them stand, output offsets, interleaving of missions) and on the host
// (entry/exit of every buffer, missions with few runs, regions the device gives back).
#include "sx_ctx.hpp"

using namespace sx;

namespace sx {

static inline uint64_t win_start_h(uint64_t p, size_t W) {
    const uint64_t s0 = p / kInputBufLen * kInputBufLen;
    return s0 + (p - s0) / W * W;
}

bool device_replay_wanted(const sx_ctx* ctx, const ReplayJob& job, size_t k, size_t n_runs) {
    if (!job.d_bytes || ctx->host_only || job.is_last || ctx->missions[k].host_sequential()) return false;
    if (ctx->opt.flags & SX_OPT_HOST_REPLAY) return false;
    if (ctx->missions[k].q > 255) return false;   // (64 < q <= 255: the replay kernels' QBIG instantiations, round 4)
    // -n 0: SplitStr's exit 4 (helper.rs:317) then fires on a rejected char with nothing collected, which ends the
    // iteration for the whole decoder call (helper.rs:343) — accepted chars behind it are never carried, so the
    // rule "the last accepted char in front of a window start is the leftover" (derive_at) does not hold.  Such a
    // mission is replayed by ONE exact sequential pass on the host: no derived states, no speculation.
    if (ctx->missions[k].c.chars_min_nb == 0) return false;
    if (getenv("SX_HOST_REPLAY")) return false;
    if (wave_replay_wanted(ctx, job, k, n_runs)) return true;
    return (ctx->opt.flags & SX_OPT_DEVICE_REPLAY) || getenv("SX_DEVICE_REPLAY") || n_runs >= 4096;
}

// Stage B of one mission on the device (sx_replay_dev.hip) + the little the host keeps:
// the chunk's strict entry region, regions the device gave back, the exact exit state.
// One slab = the runs [i0, i1) of the list (all of them unless the mission is replayed in slabs, see device_replay_mission):
// its regions begin in [job.lo, job.hi); what it writes goes to the host on the copy stream, behind `copied`.
struct Slab {
    size_t i0 = 0, i1 = 0;
    bool first = true, last = true;   // first: the host's entry part belongs to it; last: the exit state
    int out_slot = 5;                 // d_rp[] index of the output buffer (two in turn)
    bool async_copy = false;          // leave the copy to the host running (the caller waits for `copied`)
    hipEvent_t reuse_after = nullptr; // the output buffer is read by a copy until this event
    hipEvent_t copied = nullptr;
    // the mission's own start in this buffer (the job's lo / entry_exact are the slab's): the exit state is replayed from the
    // last region's start, which may lie in an earlier slab
    uint64_t lo0 = 0;
    bool entry_exact0 = false;
};
struct SlabCarry {                    // from slab to slab: where the replay stands
    uint64_t last_start = 0;
    bool last_is_entry = false, any = false;
};
constexpr int SX_RETRY_WHOLE = 1000;  // (internal) a slab met something only the whole-list path handles

static int device_replay_slab(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, const RunList& runs, const Slab& sl,
                              SlabCarry& carry, MissionFindings* out, uint64_t* end_pos, uint64_t defer_min_bytes) {
    const Mission& m = ctx->missions[k];
    MissionDev& d = ctx->dev[k];
    const size_t n_all = runs.size();
    const size_t n = sl.i1 - sl.i0;   // the runs this slab's kernels visit (lookups may go on to the list's end)
    const bool slabbed = !(sl.first && sl.last);
    const size_t W = m.window;
    const double t0 = now_ms();

    // ---- pass 1 on the device: every region's extent and output size; then (still on the
    // device) which regions stand and where each writes.  The host keeps its own version of
    // that step for buffers with regions the device gave back (kRegionTooLong).
    ReplayParams P{};
    bool fast_pass = false;
    bool dev_stitch = n > 0 && !getenv("SX_HOST_STITCH");
    uint64_t* h_tot = nullptr;
    ReplayRegionOut* ro = nullptr;
    {
        // (the regions' records only travel to the host when it decides what stands: 24 bytes per run, 6 GB for C5's floods)
        int rc = ensure_pinned2(ctx, (dev_stitch ? 0 : n * sizeof(ReplayRegionOut)) + 512);
        if (rc != SX_OK) return rc;
        h_tot = (uint64_t*)ctx->h_pin2;   // kTotCount totals; words 32, 33: the fast pre-pass' statistics
        ro = (ReplayRegionOut*)(ctx->h_pin2 + 384);
    }
    if (n) {
        int rc = ensure_rp(ctx, d, 0, n * sizeof(sx_run)); if (rc) return rc;
        rc = ensure_rp(ctx, d, 1, n * sizeof(ReplayRegionOut)); if (rc) return rc;
        rc = ensure_rp(ctx, d, 2, n * 8); if (rc) return rc;
        rc = ensure_rp(ctx, d, 3, (n + 1) * 8); if (rc) return rc;
        rc = ensure_rp(ctx, d, 4, (n + 1) * 8); if (rc) return rc;
        rc = ensure_rp(ctx, d, 6, stitch_blocks_bytes(n)); if (rc) return rc;
        rc = ensure_rp(ctx, d, 7, kTotCount * 8); if (rc) return rc;
        rc = ensure_scratch(ctx, stitch_scratch_bytes(n)); if (rc) return rc;
        if (!runs.on_device)   // (never slabbed)
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[0], runs.data(), n * sizeof(sx_run), hipMemcpyHostToDevice, d.stream_b));
        P.data = job.d_bytes; P.len = job.len; P.runs = (runs.on_device ? runs.dev_ptr : (const sx_run*)d.d_rp[0]) + sl.i0; P.n_runs = n;
        P.n_look = n_all - sl.i0;
        P.lo = job.lo[k]; P.hi = job.hi; P.consumed0 = job.consumed0[k]; P.stream0 = job.stream0[k];
        P.slice_base = job.slice_base; P.encoding = m.c.encoding; P.table = d.d_table;
        P.chars_min_nb = m.c.chars_min_nb; P.same_block = m.c.require_same_unicode_block; P.q = (uint32_t)m.q;
        P.W = (uint32_t)W; P.long_run = m.long_run; P.skip = getenv("SX_NO_REPLAY_SKIP") ? 0u : 1u; P.grep_char = m.c.grep_char; P.mission_id = m.c.mission_id;
        P.file_id = job.file_id; P.af_lo = m.c.af_lo; P.af_hi = m.c.af_hi; P.ubf = m.c.ubf;
        P.max_windows = kMaxRegionWindowsDefault;
        P.entry_skip = m.buf_entry_skip;
        P.grid_flags = nullptr; P.grid_sub = 0;
        // (not for gb18030 / GBK — ADVICE round 5: the scan kernel publishes the hang-over of the TWO-byte grammar; a four-byte token that
        //  straddles a sub-chunk start followed by lead / digit bytes only would start the walk two bytes off and never resynchronise)
        if (d.grid_of_scan && d.grid_data == job.d_bytes && d.grid_len == job.len && !getenv("SX_NO_GRID_BOUND") && !enc_is_gb(m.c.encoding)) { P.grid_flags = d.grid_of_scan; P.grid_sub = d.grid_sub; }
        if (const char* e = getenv("SX_MAX_REGION_WINDOWS")) P.max_windows = (uint32_t)std::max(1, atoi(e));
        // pass-1 output cache: one arena for all replaying regions (slot = arena / their number, on the device)
        if (dev_stitch && !getenv("SX_NO_REPLAY_CACHE")) {
            // the arena is at most 1 KiB per run; 8 GiB of the 288 GB, and for a flood of runs (a Mission with hundreds of
            // millions of them: slots of the minimum size, 160 bytes, for as many regions as possible — a region without slot
            // is replayed twice) up to a third of what is free, at most 40 GiB
            uint64_t budget = 8192ull << 20;
            if ((uint64_t)n * 160 > budget) {
                size_t fr = 0, tot = 0;
                if (hipMemGetInfo(&fr, &tot) == hipSuccess)
                    budget = std::max(budget, std::min<uint64_t>(40960ull << 20, ((uint64_t)fr + ctx->d_cache_cap) / 3));
            }
            if (const char* e = getenv("SX_REPLAY_CACHE_MIB")) budget = (uint64_t)atoll(e) << 20;
            uint64_t arena = std::max<uint64_t>(4096, std::min<uint64_t>(budget, (uint64_t)n * 1024));
            const uint64_t lists = (uint64_t)(3 * n + 8) * 4 + 512;   // slot_of, n_heads, head_list; (round 5) n_hard, hard_list
            rc = ensure_cache(ctx, arena + lists);
            while (rc != SX_OK && arena > (64ull << 20)) {   // the device is short of memory: a smaller arena (more regions are replayed twice)
                (void)hipGetLastError();
                arena /= 2;
                rc = ensure_cache(ctx, arena + lists);
            }
            if (rc) return rc;
            rc = ensure_scratch(ctx, std::max(stitch_scratch_bytes(n), replay_heads_scratch_bytes(n))); if (rc) return rc;
            uint8_t* base = ctx->d_cache;
            P.cache_arena = base; P.arena_bytes = arena & ~255ull;
            uint32_t* slot_of = (uint32_t*)(base + ((arena + 255) & ~255ull));
            P.slot_of = slot_of; P.n_heads = slot_of + n + 1; P.head_list = slot_of + n + 2;
            HIP_TRY(ctx, launch_replay_heads(P, slot_of, slot_of + n + 1, slot_of + n + 2, (ReplayRegionOut*)d.d_rp[1], ctx->d_scratch,
                                             ctx->d_scratch_cap, d.stream_b));
            // (round 5) the fast pre-pass settles the regions that are one run inside one window; the general kernel gets the rest
            P.n_hard = slot_of + 2 * n + 4; P.hard_list = slot_of + 2 * n + 5;
            if (replay_fast_covers(P) && !(getenv("SX_FAST_REPLAY") && !atoi(getenv("SX_FAST_REPLAY")))) {
                HIP_TRY(ctx, hipMemsetAsync(P.n_hard, 0, 4, d.stream_b));
                HIP_TRY(ctx, launch_replay_fast(P, (ReplayRegionOut*)d.d_rp[1], d.stream_b));
                fast_pass = true;
            } else { P.n_hard = nullptr; P.hard_list = nullptr; }
        }
        HIP_TRY(ctx, launch_replay_count(P, (ReplayRegionOut*)d.d_rp[1], d.stream_b));
        if (fast_pass) {   // (statistics: how many regions each pass took; read with the totals below)
            h_tot[32] = 0; h_tot[33] = 0;
            { int rb = read_back_async(ctx, d.stream_b, h_tot + 32, P.n_hard, 4); if (rb == SX_OK) rb = read_back_async(ctx, d.stream_b, h_tot + 33, P.n_heads, 4); if (rb != SX_OK) return rb; }
        }
        // The host's copy of a device-joined run list: its share of this replay (the buffer's entry region, the exit state)
        // reads only the list's two ends — a large list is not copied whole (68 MB for the headline's 2.8 M runs, 1.2 ms on
        // the critical path when nothing else runs); the rest follows on demand (regions the device gives back).
        if (runs.size() * sizeof(sx_run) <= (4u << 20)) HIP_TRY(ctx, runs.start_copy());
        if (dev_stitch) {
            HIP_TRY(ctx, hipMemsetAsync(d.d_rp[7], 0, kTotCount * 8, d.stream_b));
            HIP_TRY(ctx, launch_stitch_blocks(P, (const ReplayRegionOut*)d.d_rp[1], (uint8_t*)d.d_rp[2], d.d_rp[6],
                                              (uint64_t*)d.d_rp[7], d.stream_b));
        } else
            HIP_TRY(ctx, hipMemcpyAsync(ro, d.d_rp[1], n * sizeof(ReplayRegionOut), hipMemcpyDeviceToHost, d.stream_b));
    }

    // ---- meanwhile on the host: the strict entry region (exact carried state), if any
    constexpr size_t kEdgeRuns = 8192;
    if (dev_stitch) HIP_TRY(ctx, runs.fetch_edges(kEdgeRuns)); else HIP_TRY(ctx, runs.wait());   // (the host's stitch walks the whole list)
    // the part of the list the host may read without the whole copy: [0, kEdgeRuns) and [n - kEdgeRuns, n)
    const size_t head_n = runs.full() ? n_all : kEdgeRuns;
    std::deque<ReplayPart> host_parts;
    struct Seg { int host_part; size_t v0, v1; };  // host_part >= 0, or device regions [v0, v1) of `valid`
    std::vector<Seg> segs;
    uint64_t E = std::min(job.lo[k], job.hi);
    uint64_t last_start = carry.any ? carry.last_start : E;   // start of the last region of any kind (for the exit state)
    bool last_is_entry = carry.any ? carry.last_is_entry : false;
    if (job.entry_exact[k]) {
        // The chunk's first window belongs to the host: only it has the exact carried state
        // (leftover, cut flag, and the decoder's pending bytes, which cannot be re-derived here).
        host_parts.emplace_back();
        replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, runs.data(), head_n,
                    job.lo[k], job.lo[k] + 1, true, &host_parts.back());
        if (head_n < n_all && host_parts.back().end_pos >= runs[head_n - 1].start) {   // it ran into what was not copied: the whole list
            HIP_TRY(ctx, runs.wait());
            host_parts.pop_back();
            host_parts.emplace_back();
            replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, runs.data(), n_all,
                        job.lo[k], job.lo[k] + 1, true, &host_parts.back());
        }
        if (host_parts.back().regions.empty()) host_parts.pop_back();
        else { segs.push_back({ (int)host_parts.size() - 1, 0, 0 }); last_is_entry = true; E = std::max(E, host_parts.back().end_pos); }
        E = std::max(E, job.lo[k] + 1);
    }
    if (dev_stitch) {
        HIP_TRY(ctx, launch_stitch_finish(P, (const ReplayRegionOut*)d.d_rp[1], (uint8_t*)d.d_rp[2], d.d_rp[6], E,
                                          (uint64_t*)d.d_rp[3], (uint64_t*)d.d_rp[4], (uint64_t*)d.d_rp[7], ctx->d_scratch,
                                          ctx->d_scratch_cap, d.stream_b));
        { const int rb = read_back_async(ctx, d.stream_b, h_tot, d.d_rp[7], kTotCount * 8); if (rb != SX_OK) return rb; }
    }
    SX_TL("  replay: pass 1 + stitch queued, host entry part done");
    if (n) HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
    SX_TL("  replay: pass 1 + stitch done");
    if (dev_stitch && h_tot[kTotTooLong] && slabbed) return SX_RETRY_WHOLE;
    if (dev_stitch && h_tot[kTotTooLong]) {  // regions for the host: it also decides what stands
        dev_stitch = false;
        HIP_TRY(ctx, runs.wait());
        const uint64_t keep32 = h_tot[32], keep33 = h_tot[33];
        int rc = ensure_pinned2(ctx, n * sizeof(ReplayRegionOut) + 512);
        if (rc != SX_OK) return rc;
        h_tot = (uint64_t*)ctx->h_pin2;   // (its values are not read again)
        h_tot[32] = keep32; h_tot[33] = keep33;
        ro = (ReplayRegionOut*)(ctx->h_pin2 + 384);
        HIP_TRY(ctx, hipMemcpyAsync(ro, d.d_rp[1], n * sizeof(ReplayRegionOut), hipMemcpyDeviceToHost, d.stream_b));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
    }
    const double t1 = now_ms();
    if (fast_pass && n) {
        std::lock_guard<std::mutex> g(ctx->mu);
        ctx->stats.fast_regions += (uint32_t)h_tot[33] - (uint32_t)h_tot[32];
        ctx->stats.general_regions += (uint32_t)h_tot[32];
    }

    std::vector<uint64_t> valid, fbase, abase;
    uint64_t nf = 0, nb = 0, n_standing = 0;
    if (dev_stitch) {
        nf = h_tot[kTotFindings]; nb = h_tot[kTotBytes]; n_standing = h_tot[kTotStanding];
        out->replay_bytes += h_tot[kTotReplayBytes];
        if (h_tot[kTotLast] != ~0ull) {
            E = std::max(E, h_tot[kTotEnd]);
            last_start = h_tot[kTotLastStart];
            last_is_entry = false;
        }
    } else {
        // ---- which regions stand: a region is void if an earlier one ran over its start
        valid.reserve(n); fbase.reserve(n + 1); abase.reserve(n + 1);
        for (size_t i = 0; i < n; i++) {
            const uint32_t st = ro[i].status;
            if (st == kRegionChained || st == kRegionNotMine) continue;
            const uint64_t want = win_start_h(runs[i].start, W);
            if (want >= job.hi) break;
            if (want < E) continue;
            if (st == kRegionTooLong) {  // given back: the host replays it (and whatever it runs into)
                host_parts.emplace_back();
                replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, runs.data(),
                            n, want, want + 1, false, &host_parts.back());
                segs.push_back({ (int)host_parts.size() - 1, 0, 0 });
                E = std::max(E, host_parts.back().end_pos);
            } else {
                if (segs.empty() || segs.back().host_part >= 0) segs.push_back({ -1, valid.size(), valid.size() });
                valid.push_back(i); fbase.push_back(nf); abase.push_back(nb);
                segs.back().v1 = valid.size();
                nf += ro[i].n_find; nb += ro[i].n_bytes;
                E = std::max(E, ro[i].end);
            }
            last_start = want; last_is_entry = false;
        }
        fbase.push_back(nf); abase.push_back(nb);
        n_standing = valid.size();
        for (uint64_t v : valid) out->replay_bytes += ro[v].end - win_start_h(runs[v].start, W);
    }
    if (nb > 0xFFFFFFFFull) { ctx->err = "more than 4 GiB of strings in one chunk"; return SX_E_NOMEM; }
    const double t2 = now_ms();

    // ---- pass 2: the standing regions write findings and strings, in order; the D2H lands in a
    // pinned block that becomes the result's storage (no copy).  The host's entry part (the chunk's first
    // region, replayed from the exact carried state) goes in front of it in the same block: the device writes
    // its str_off already shifted by that part's strings.  Only regions the device gave back to the host
    // (somewhere in the middle) need the finding-by-finding splice.
    PinnedPool::Block blk{};
    bool deferred = false, keep_dev = false;
    std::vector<sx_finding> entry_f;
    const bool entry_only = dev_stitch && host_parts.size() == 1 && true;
    const MissionFindings* hf0 = entry_only ? &host_parts[0].findings : nullptr;
    const uint64_t nfh = hf0 ? hf0->v.size() : 0, nbh = hf0 ? hf0->arena.size() : 0;
    if (nb + nbh > 0xFFFFFFFFull) { ctx->err = "more than 4 GiB of strings in one chunk"; return SX_E_NOMEM; }
    if (n_standing) {
        // the device buffer has the block's layout: [host findings][device findings][host strings][device strings]
        const uint64_t out_bytes = (nfh + nf) * sizeof(sx_finding) + nbh + nb;
        if (sl.reuse_after) {   // a copy of an earlier slab may still read this buffer
            if (d.d_rp_cap[sl.out_slot] < out_bytes + 64) HIP_TRY(ctx, hipEventSynchronize(sl.reuse_after));
            else HIP_TRY(ctx, hipStreamWaitEvent(d.stream_b, sl.reuse_after, 0));
        }
        int rc = ensure_rp(ctx, d, sl.out_slot, out_bytes + 64); if (rc) return rc;
        uint8_t* d_all = (uint8_t*)d.d_rp[sl.out_slot];
        sx_finding* d_f = (sx_finding*)d_all + nfh;
        uint8_t* d_a = d_all + (nfh + nf) * sizeof(sx_finding) + nbh;
        if (dev_stitch) {
            P.str_off_base = (uint32_t)nbh;
            HIP_TRY(ctx, launch_replay_write_flagged(P, (const ReplayRegionOut*)d.d_rp[1], (const uint8_t*)d.d_rp[2],
                                                     (const uint64_t*)d.d_rp[3], (const uint64_t*)d.d_rp[4], d_f, d_a,
                                                     (nf * sizeof(sx_finding) + nb) / std::max<uint64_t>(1, n_standing), d.stream_b));
            if (nfh) {   // the host's entry part joins it there (a few findings): the missions can be interleaved on the device
                entry_f.assign(hf0->v.begin(), hf0->v.end());
                for (sx_finding& f : entry_f) f.slice_index += job.slice_base;
                HIP_TRY(ctx, hipMemcpyAsync(d_all, entry_f.data(), nfh * sizeof(sx_finding), hipMemcpyHostToDevice, d.stream_b));
                if (nbh) HIP_TRY(ctx, hipMemcpyAsync(d_all + (nfh + nf) * sizeof(sx_finding), hf0->arena.data(), nbh, hipMemcpyHostToDevice, d.stream_b));
            }
        } else {
            const size_t nv = valid.size();
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[2], valid.data(), nv * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[3], fbase.data(), (nv + 1) * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[4], abase.data(), (nv + 1) * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, launch_replay_write(P, (const uint64_t*)d.d_rp[2], (const uint64_t*)d.d_rp[3],
                                             (const uint64_t*)d.d_rp[4], nv, d_f, d_a, d.stream_b));
        }
        // a large output of one of several missions: the caller interleaves the missions on the device first, one copy instead of two
        deferred = defer_min_bytes && nf * sizeof(sx_finding) + nb >= defer_min_bytes && dev_stitch && (host_parts.empty() || entry_only);
        // SX_OPT_RESULT_ON_DEVICE (round 5): one Mission, the whole buffer in this one slab, nothing the host has to splice in — it stays here
        keep_dev = (ctx->opt.flags & SX_OPT_RESULT_ON_DEVICE) && ctx->missions.size() == 1 && defer_min_bytes == 0 && job.commit_state && !ctx->sharded_call &&
                   ctx->single_piece && sl.first && sl.last && !sl.async_copy && dev_stitch && (host_parts.empty() || entry_only);
        deferred = deferred || keep_dev;
        if (!deferred) {
            blk = ctx->pool->take(out_bytes + 64);
            if (!blk.p) { ctx->err = "hipHostMalloc failed"; return SX_E_NOMEM; }
            if (sl.async_copy) {   // on the copy stream: the next slab's kernels run meanwhile
                HIP_TRY(ctx, hipEventRecord(ctx->merge_ev[0], d.stream_b));
                HIP_TRY(ctx, hipStreamWaitEvent(ctx->merge_copy_stream, ctx->merge_ev[0], 0));
                HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_all, out_bytes, hipMemcpyDeviceToHost, ctx->merge_copy_stream));
                HIP_TRY(ctx, hipEventRecord(sl.copied, ctx->merge_copy_stream));
            } else {
                // (experiments, SX_REPLAY_COPY_WGS=n: the copy as a kernel of n workgroups instead of the runtime's blit, which slows every
                // kernel next to it — here: the last Mission's scan launch, DESIGN.md section 2 "Schedule")
                static const int wgs = [] { const char* e = getenv("SX_REPLAY_COPY_WGS"); return e ? atoi(e) : 0; }();
                if (wgs > 0 && out_bytes >= (1u << 20)) HIP_TRY(ctx, launch_copy_bytes(blk.p, d_all, out_bytes, (uint32_t)wgs, d.stream_b));
                else HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_all, out_bytes, hipMemcpyDeviceToHost, d.stream_b));
            }
        }
        SX_TL("  replay: pass 2 + copy queued (%llu bytes%s)", (unsigned long long)out_bytes, deferred ? ", stays on the device" : "");
        if (!sl.async_copy || nfh) HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));   // (nfh: the entry part's upload reads host vectors)
        SX_TL("  replay: pass 2 + copy done");
    }
    const double t3 = now_ms();

    // ---- splice (almost always: device findings only, or the entry part in front of them)
    if (host_parts.empty() || (entry_only && (blk.p || deferred))) {
        if (deferred) {
            out->dev_only = true; out->ext_nf = nfh + nf; out->ext_na = nbh + nb; out->dev_copy = d.d_rp[sl.out_slot];
            if (keep_dev) { out->keep_on_device = true; out->dev_epoch_ref = ctx->dev_epoch; out->dev_epoch = ctx->dev_epoch->load(); }
        }
        else if (blk.p) { out->ext = blk; out->ext_nf = nfh + nf; out->ext_na = nbh + nb; out->dev_copy = sl.async_copy ? nullptr : d.d_rp[sl.out_slot]; }
        if (hf0) out->replay_bytes += hf0->replay_bytes;
    } else {
        const sx_finding* dev_f = (const sx_finding*)blk.p;
        const char* dev_a = blk.p ? (const char*)blk.p + nf * sizeof(sx_finding) : nullptr;
        if (dev_stitch) {  // only the entry part can be here; everything the device wrote follows it
            fbase.assign({ 0, nf }); abase.assign({ 0, nb });
            if (n_standing) segs.push_back({ -1, 0, 1 });
        }
        for (const Seg& g : segs) {
            if (g.host_part >= 0) {
                const MissionFindings& hf = host_parts[(size_t)g.host_part].findings;
                const uint32_t base = (uint32_t)out->arena.size();
                out->arena += hf.arena;
                for (sx_finding f : hf.v) { f.str_off += base; f.slice_index += job.slice_base; out->v.push_back(f); }
                out->replay_bytes += hf.replay_bytes;
            } else if (g.v1 > g.v0) {
                const uint64_t f0 = fbase[g.v0], f1 = fbase[g.v1], a0 = abase[g.v0], a1 = abase[g.v1];
                const uint32_t base = (uint32_t)out->arena.size();
                out->arena.append(dev_a + a0, a1 - a0);
                for (uint64_t j = f0; j < f1; j++) { sx_finding f = dev_f[j]; f.str_off = f.str_off - (uint32_t)a0 + base; out->v.push_back(f); }
            }
        }
        ctx->pool->give(blk);
    }

    // ---- the state handed to the next chunk: replay the last region and the tail once more
    // on the host, only for its final state (RangeReplay's tail rule makes it exact)
    if (job.commit_state) {
        const uint64_t tail = job.len ? job.len - 1 : 0;
        uint64_t ts = win_start_h(tail, W);
        for (int t = 0; t < 3 && ts > 0; t++) ts = win_start_h(ts - 1, W);
        const uint64_t lo0 = slabbed ? sl.lo0 : job.lo[k];
        const bool exact0 = slabbed ? sl.entry_exact0 : job.entry_exact[k] != 0;
        uint64_t from = last_is_entry ? lo0 : (E > ts ? last_start : ts);
        if (from < lo0) from = lo0;
        ReplayPart fin;
        const sx_run* fr = runs.data();
        size_t fn = n_all;
        if (!runs.full()) {   // the list's tail is enough if it begins in front of `from`
            if (n_all > kEdgeRuns && runs[n_all - kEdgeRuns].start <= from) { fr = runs.data() + (n_all - kEdgeRuns); fn = kEdgeRuns; }
            else HIP_TRY(ctx, runs.wait());
        }
        replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, fr, fn,
                    from, job.len, exact0 && from == lo0, &fin);
        ctx->states[k] = fin.state;
        ctx->states[k].consumed_bytes = job.consumed0[k] + job.len;
        ctx->states[k].stream_bytes = job.stream0[k] + job.len;
        E = job.len;
    }
    if (end_pos) *end_pos = std::max(E, std::min(job.hi, job.len));
    carry.last_start = last_start; carry.last_is_entry = last_is_entry; carry.any = true;
    if (getenv("SX_TIMING"))
        fprintf(stderr, "[sx] device replay mission %zu: %zu runs, pass1+entry %.2f ms, validity %.2f ms (%zu standing, %zu host parts), "
                        "pass2+d2h %.2f ms (%llu findings), splice+state %.2f ms\n", k, n, t1 - t0, t2 - t1, (size_t)n_standing,
                host_parts.size(), t3 - t2, (unsigned long long)nf, now_ms() - t3);
    return SX_OK;
}

// The stream of the result copies (slabs, merge parts).  SX_COPY_PRIO=-1/0/1: its priority (experiments).
int ensure_copy_stream(sx_ctx* ctx) {
    if (ctx->merge_copy_stream) return SX_OK;
    int prio = 0;
    if (const char* e = getenv("SX_COPY_PRIO")) prio = atoi(e);
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->merge_copy_stream, hipStreamNonBlocking, prio));
    for (hipEvent_t& e : ctx->merge_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return SX_OK;
}

// Stage B of one mission on the device.  A single mission with a long run list is replayed in slabs (a quarter of the
// list each, cut where a region begins): a slab's findings travel to the host while the next slab is replayed, so that only
// the last slab's copy is not hidden (string-dense and text-like input: the output is as large as the input).  Slab j begins
// where slab j-1 stopped — its last region may run past the cut —, with the state derived there as at a shard start.
int device_replay_mission(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, const RunList& runs,
                          MissionFindings* out, uint64_t* end_pos, uint64_t defer_min_bytes) {
    const size_t n = runs.size();
    MissionDev& d = ctx->dev[k];
    if (wave_replay_wanted(ctx, job, k, n)) {   // string-dense: a lane per window instead of a lane per region (sx_wave.cpp)
        const int rc = wave_replay_mission(ctx, k, view, job, out, end_pos, defer_min_bytes);
        if (rc != SX_WAVE_FALLBACK) return rc;
        if (k < ctx->wave_off.size()) ctx->wave_off[k] = 1;
        if (k < ctx->wave_pred.size()) ctx->wave_pred[k] = 0;
    }
    if (runs.skipped) return SX_NEED_RUNS;
    size_t K = 1;
    const bool can = ctx->missions.size() == 1 && runs.on_device && !getenv("SX_HOST_STITCH") && defer_min_bytes == 0;
    const bool keep_dev_wanted = (ctx->opt.flags & SX_OPT_RESULT_ON_DEVICE) && ctx->missions.size() == 1 && defer_min_bytes == 0 && job.commit_state && !ctx->sharded_call && ctx->single_piece;   // (one slab then: one block of the context's)
    if (can && n >= (1u << 20) && !keep_dev_wanted) K = 3;   // (measured on string-dense and text-like input: 3 beats 2, 4 and 6)
    if (const char* e = getenv("SX_SLABS")) { K = (size_t)std::max(1, std::min(64, atoi(e))); if (!can || n < 8 * K) K = 1; }
    std::vector<uint64_t> idx{ 0 }, his{ 0 };
    if (K > 1) {   // where to cut
        int rc = ensure_copy_stream(ctx); if (rc != SX_OK) return rc;
        rc = ensure_scratch(ctx, 4096); if (rc != SX_OK) return rc;
        rc = ensure_pinned2(ctx, 4096); if (rc != SX_OK) return rc;
        ReplayParams P{};
        P.runs = runs.dev_ptr; P.n_runs = n; P.W = (uint32_t)ctx->missions[k].window; P.grep_char = ctx->missions[k].c.grep_char;
        uint64_t* d_cut = (uint64_t*)ctx->d_scratch;
        HIP_TRY(ctx, launch_slab_cuts(P, (uint32_t)K, d_cut, d_cut + 64, d.stream_b));
        { const int rb = read_back_async(ctx, d.stream_b, ctx->h_pin2, d_cut, 128 * 8); if (rb != SX_OK) return rb; }
        HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
        const uint64_t* h = (const uint64_t*)ctx->h_pin2;
        for (size_t j = 1; j < K; j++)
            if (h[j - 1] > idx.back() && h[j - 1] < n && h[64 + j - 1] < job.hi) { idx.push_back(h[j - 1]); his.push_back(h[64 + j - 1]); }
    }
    idx.push_back(n);
    const size_t ns = idx.size() - 1;
    if (ns > 1) {
        std::vector<MissionFindings> got;
        SlabCarry carry;
        hipEvent_t copied[2] = { ctx->merge_ev[1], ctx->merge_ev[2] };
        uint64_t e_prev = 0;
        int rc = SX_OK;
        for (size_t j = 0; j < ns && rc == SX_OK; j++) {
            ReplayJob sj = job;
            sj.hi = j + 1 < ns ? std::min(job.hi, his[j + 1]) : job.hi;
            if (j > 0) { sj.lo[k] = std::max(job.lo[k], e_prev); sj.entry_exact[k] = 0; }
            sj.commit_state = job.commit_state && j + 1 == ns;
            Slab sl;
            sl.i0 = idx[j]; sl.i1 = idx[j + 1]; sl.first = j == 0; sl.last = j + 1 == ns;
            sl.out_slot = (j & 1) ? 8 : 5; sl.async_copy = true;
            sl.reuse_after = j >= 2 ? copied[j & 1] : nullptr; sl.copied = copied[j & 1];
            sl.lo0 = job.lo[k]; sl.entry_exact0 = job.entry_exact[k] != 0;
            MissionFindings mf;
            uint64_t e_now = 0;
            rc = device_replay_slab(ctx, k, view, sj, runs, sl, carry, &mf, &e_now, 0);
            e_prev = std::max(e_prev, e_now);
            if (rc == SX_OK) {
                out->replay_bytes += mf.replay_bytes; mf.replay_bytes = 0;
                if (mf.count()) got.push_back(std::move(mf));
            } else if (mf.ext.p) got.push_back(std::move(mf));
        }
        (void)hipStreamSynchronize(ctx->merge_copy_stream);
        (void)hipStreamSynchronize(d.stream_b);
        if (rc == SX_OK) {
            const uint64_t rb = out->replay_bytes;
            if (!got.empty()) {
                *out = std::move(got[0]);
                for (size_t j = 1; j < got.size(); j++) out->more.push_back(std::move(got[j]));
            }
            out->replay_bytes = rb;
            if (end_pos) *end_pos = e_prev;
            return SX_OK;
        }
        for (auto& g : got) if (g.ext.p) ctx->pool->give(g.ext);
        out->replay_bytes = 0;
        if (rc != SX_RETRY_WHOLE) return rc;
    }
    Slab sl;
    sl.i1 = n;
    SlabCarry carry;
    return device_replay_slab(ctx, k, view, job, runs, sl, carry, out, end_pos, defer_min_bytes);
}

// A mission's findings that were left on the device (dev_only) come to the host after all.
static int fetch_deferred(sx_ctx* ctx, MissionFindings& mf) {
    if (!mf.dev_only) return SX_OK;
    const size_t bytes = mf.ext_nf * sizeof(sx_finding) + mf.ext_na;
    PinnedPool::Block blk = ctx->pool->take(bytes + 64);
    if (!blk.p) { ctx->err = "hipHostMalloc failed"; return SX_E_NOMEM; }
    HIP_TRY(ctx, hipMemcpyAsync(blk.p, mf.dev_copy, bytes, hipMemcpyDeviceToHost, ctx->post_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->post_stream));
    mf.ext = blk; mf.dev_only = false;
    return SX_OK;
}

int merge_drain(sx_ctx* ctx) {
    std::lock_guard<std::recursive_mutex> g(ctx->grow_mu);   // (a wave Mission's thread may get here through ensure_rp)
    if (ctx->post_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->post_stream));
    if (ctx->merge_copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->merge_copy_stream));
    ctx->merge_copy_pending[0] = ctx->merge_copy_pending[1] = false;
    ctx->interleave_pending = false;
    return SX_OK;
}

// The merger (src/main.rs:118-136) on the device: the missions' findings, each ordered by position and still in HBM, are
// interleaved there (sx_sort.hip: every finding's place follows from short searches in the other missions' lists) and arrive on
// the host as they will be printed.  str_off has 32 bits and the place table 32-bit indices: a large output is cut at slice
// boundaries into parts (each at most SX_MERGE_PART_MIB of strings and SX_MERGE_PART_FINDINGS findings) that become one segment
// of the result each; the copy of part j runs while part j+1 is interleaved.  If the conditions do not hold the host merges
// (merge_findings).
static int device_merge(sx_ctx* ctx, const ReplayJob& job, std::vector<MissionFindings>& per, Result* into) {
    const size_t nm = per.size();
    size_t with = 0, on_dev = 0, deferred = 0;
    uint64_t total = 0, bytes = 0;
    bool same_origin = true;
    for (size_t k = 0; k < nm; k++) {
        if (!per[k].count()) continue;
        with++;
        if ((per[k].ext.p || per[k].dev_only) && per[k].dev_copy) on_dev++;
        if (per[k].dev_only) deferred++;
        total += per[k].count(); bytes += per[k].strings_len();
        same_origin = same_origin && job.consumed0[k] == job.consumed0[0];
    }
    auto give_up = [&]() -> int {
        for (auto& mf : per) { if (mf.keep_on_device) continue; int rc = fetch_deferred(ctx, mf); if (rc != SX_OK) return rc; }   // (SX_OPT_RESULT_ON_DEVICE: it stays)
        return SX_OK;
    };
    if (!(with >= 2 && on_dev == with && same_origin && total >= 4096 && !getenv("SX_HOST_MERGE"))) return give_up();
    const double tm0 = now_ms();
    // The records cross PCIe as sx_finding16 (include/stringsext_amd.h): half the bytes of what bounds this path.  Not with lines of
    // more than 16 000 chars (str_len has 16 bits there), nor where the one-pass merger does not apply (SX_PACKED=0: tests).
    bool pack = !(getenv("SX_PACKED") && !atoi(getenv("SX_PACKED")));
    for (size_t k = 0; k < nm; k++) pack = pack && ctx->missions[k].q <= 16000;
    std::shared_ptr<SegInfo> seg_info;
    if (pack) {
        seg_info = std::make_shared<SegInfo>();
        seg_info->file_id = job.file_id; seg_info->slice_base = job.slice_base;
        for (auto& x : seg_info->pos0) x = 0;
        for (size_t k = 0; k < nm; k++) seg_info->pos0[ctx->missions[k].c.mission_id] = job.consumed0[k];
    }
    uint64_t part_bytes = 2048ull << 20, part_findings = 96ull << 20;
    if (const char* e = getenv("SX_MERGE_PART_MIB")) part_bytes = std::max<uint64_t>(1, (uint64_t)atoll(e)) << 20;
    if (const char* e = getenv("SX_MERGE_PART_FINDINGS")) part_findings = std::max<uint64_t>(1024, (uint64_t)atoll(e));
    if (part_bytes > (3584ull << 20)) part_bytes = 3584ull << 20;
    uint64_t K = std::max<uint64_t>(1, std::max((bytes + part_bytes - 1) / part_bytes, (total + part_findings - 1) / part_findings));
    hipStream_t s = ctx->post_stream;
    const uint64_t n_slices = (job.len + kInputBufLen - 1) / kInputBufLen;
    std::vector<std::vector<uint64_t>> idx(nm), off(nm);   // per mission: K + 1 finding indices and string offsets
    for (int attempt = 0;; attempt++) {
        if (K > n_slices) K = std::max<uint64_t>(1, n_slices);
        for (size_t k = 0; k < nm; k++) { idx[k].assign(K + 1, 0); off[k].assign(K + 1, 0); idx[k][K] = per[k].count(); off[k][K] = per[k].strings_len(); }
        if (K > 1) {
            std::vector<uint64_t> cuts(K - 1);
            for (uint64_t j = 1; j < K; j++) cuts[j - 1] = job.consumed0[0] + (n_slices * j / K) * kInputBufLen;
            const size_t row = (K - 1) * 8;
            int rc = ensure_scratch(ctx, row * (1 + 2 * nm) + 256);
            if (rc != SX_OK) return rc;
            rc = ensure_pinned2(ctx, row * 2 * nm + 256);
            if (rc != SX_OK) return rc;
            uint64_t* d_cuts = (uint64_t*)ctx->d_scratch;
            HIP_TRY(ctx, hipMemcpyAsync(d_cuts, cuts.data(), row, hipMemcpyHostToDevice, s));
            for (size_t k = 0; k < nm; k++)
                if (per[k].count())
                    HIP_TRY(ctx, launch_merge_cuts((const sx_finding*)per[k].dev_copy, per[k].count(), per[k].strings_len(), d_cuts,
                                                   (uint32_t)(K - 1), d_cuts + (K - 1) * (1 + 2 * k), d_cuts + (K - 1) * (2 + 2 * k), s));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pin2, d_cuts + (K - 1), row * 2 * nm, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            const uint64_t* h = (const uint64_t*)ctx->h_pin2;
            for (size_t k = 0; k < nm; k++)
                if (per[k].count())
                    for (uint64_t j = 1; j < K; j++) { idx[k][j] = h[(K - 1) * 2 * k + (j - 1)]; off[k][j] = h[(K - 1) * (2 * k + 1) + (j - 1)]; }
                else
                    for (uint64_t j = 1; j < K; j++) { idx[k][j] = 0; off[k][j] = 0; }
        }
        bool fits = true;
        for (uint64_t j = 0; j < K && fits; j++) {
            uint64_t pb = 0, pf = 0;
            for (size_t k = 0; k < nm; k++) { pb += off[k][j + 1] - off[k][j]; pf += idx[k][j + 1] - idx[k][j]; }
            fits = pb <= 0xFFFFFFF0ull && (pf <= 2 * part_findings || K >= n_slices);
        }
        if (fits) break;
        if (attempt >= 8 || K >= n_slices) return give_up();   // (one slice alone with more than 4 GiB of strings cannot be)
        K *= 2;
    }
    // the parts, two output buffers in turn: the copy of one runs while the next is sorted
    uint64_t max_out = 0, max_n = 0;
    for (uint64_t j = 0; j < K; j++) {
        uint64_t pb = 0, pf = 0;
        for (size_t k = 0; k < nm; k++) { pb += off[k][j + 1] - off[k][j]; pf += idx[k][j + 1] - idx[k][j]; }
        max_out = std::max(max_out, pf * sizeof(sx_finding) + pb); max_n = std::max(max_n, pf);
    }
    const size_t out_room = (max_out + 511) & ~(size_t)255;
    const size_t n_out = (K > 1 || ctx->merge_async) ? 2 : 1;
    const size_t tmp_need = merge_findings_scratch_bytes(max_n, (int)nm) + 512;
    { int rc = ensure_copy_stream(ctx); if (rc != SX_OK) return rc; }
    if (ctx->merge_out_room < out_room || ctx->merge_n_out < n_out || ctx->d_merge_cap < ctx->merge_n_out * ctx->merge_out_room + tmp_need) {
        // (the copy of an earlier call's last part may still read the buffers that are about to go)
        int rc = merge_drain(ctx);
        if (rc != SX_OK) return rc;
        if (ctx->d_merge) HIP_TRY(ctx, hipFree(ctx->d_merge));
        ctx->d_merge = nullptr; ctx->d_merge_cap = 0; ctx->merge_out_room = 0; ctx->merge_n_out = 0;
        // pieces of one buffer are alike but not equal: a quarter more than this one needs
        const size_t room = ctx->merge_async ? ((out_room + out_room / 4 + 511) & ~(size_t)255) : out_room;
        const size_t tmp = ctx->merge_async ? tmp_need + tmp_need / 4 : tmp_need;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_merge, n_out * room + tmp));
        ctx->d_merge_cap = n_out * room + tmp; ctx->merge_out_room = room; ctx->merge_n_out = n_out;
    }
    uint8_t* d_tmp = ctx->d_merge + ctx->merge_n_out * ctx->merge_out_room;
    const size_t tmp_bytes = ctx->d_merge_cap - ctx->merge_n_out * ctx->merge_out_room;
    hipStream_t cs = ctx->merge_copy_stream;
    hipEvent_t ev_sorted = ctx->merge_ev[0], ev_copied[2] = { ctx->merge_ev[1], ctx->merge_ev[2] };
    std::vector<const sx_finding*> fp(nm);
    std::vector<const uint8_t*> ap(nm);
    std::vector<uint64_t> pnf(nm), pnb(nm);
    std::vector<uint32_t> off0(nm);
    std::vector<MissionFindings> outs;
    // leaving early (a HIP error, no pinned memory): no copy may still be writing into a block that goes back to the pool
    struct Guard {
        sx_ctx* ctx; std::vector<MissionFindings>* outs; bool done = false;
        ~Guard() {
            if (done) return;
            (void)merge_drain(ctx);
            for (auto& o : *outs) if (o.ext.p) ctx->pool->give(o.ext);
            outs->clear();
        }
    } guard{ ctx, &outs };
    uint64_t rb = 0;
    for (size_t k = 0; k < nm; k++) rb += per[k].replay_bytes;
    for (uint64_t j = 0; j < K; j++) {
        uint64_t pb = 0, pf = 0;
        for (size_t k = 0; k < nm; k++) {
            const sx_finding* f0 = (const sx_finding*)per[k].dev_copy;
            const uint8_t* a0 = (const uint8_t*)per[k].dev_copy + per[k].count() * sizeof(sx_finding);
            fp[k] = f0 ? f0 + idx[k][j] : nullptr; ap[k] = f0 ? a0 + off[k][j] : nullptr;
            pnf[k] = idx[k][j + 1] - idx[k][j]; pnb[k] = off[k][j + 1] - off[k][j]; off0[k] = (uint32_t)off[k][j];
            pf += pnf[k]; pb += pnb[k];
        }
        if (!pf) continue;
        const unsigned ob = ctx->merge_n_out == 2 ? (unsigned)(ctx->merge_parts & 1) : 0u;   // (parts of all calls in turn: merge_async)
        ctx->merge_parts++;
        uint8_t* d_out = ctx->d_merge + ob * ctx->merge_out_room;
        if (ctx->merge_copy_pending[ob]) HIP_TRY(ctx, hipStreamWaitEvent(s, ev_copied[ob], 0));   // the buffer is free again
        const bool pack_part = pack && merge_part_can_pack(pf, (int)nm);
        HIP_TRY(ctx, merge_findings_device_part(fp.data(), ap.data(), pnf.data(), pnb.data(), off0.data(), (int)nm, d_out, d_tmp, tmp_bytes, s, pack_part ? 1 : 0));
        HIP_TRY(ctx, hipEventRecord(ev_sorted, s));
        const size_t out_bytes = pf * (pack_part ? sizeof(sx_finding16) : sizeof(sx_finding)) + pb;
        PinnedPool::Block blk = ctx->pool->take(out_bytes + 64);
        if (!blk.p) { ctx->err = "hipHostMalloc failed"; return SX_E_NOMEM; }
        HIP_TRY(ctx, hipStreamWaitEvent(cs, ev_sorted, 0));
        {
            // Next to the kernels of the following piece (merge_async) the copy is a kernel of two workgroups, each writing one
            // contiguous half into the pinned block: measured on the MI355X (BASELINE config 5, 2.3 GB per piece), hipMemcpyAsync
            // — a blit kernel that fills the chip with wavefronts waiting on PCIe — delays every kernel next to it until it is done
            // (a 6 ms scan takes 45), eight workgroups slow the count passes by half, one does not fill the link; two copy at
            // ~45 GB/s and cost the count passes 8 %.  With nothing next to it the runtime's copy is the faster one (53 GB/s).
            static const int copy_wgs_env = [] { const char* e = getenv("SX_MERGE_COPY_WGS"); return e ? atoi(e) : -1; }();
            const int copy_wgs = copy_wgs_env >= 0 ? copy_wgs_env : (ctx->merge_async ? 2 : 0);
            if (copy_wgs > 0) HIP_TRY(ctx, launch_copy_bytes(blk.p, d_out, out_bytes, (uint32_t)copy_wgs, cs));
            else HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_out, out_bytes, hipMemcpyDeviceToHost, cs));
        }
        HIP_TRY(ctx, hipEventRecord(ev_copied[ob], cs));
        ctx->merge_copy_pending[ob] = true;
        ctx->merged_out_bytes += out_bytes;
        outs.emplace_back();
        outs.back().ext = blk; outs.back().ext_nf = pf; outs.back().ext_na = pb;
        if (pack_part) { outs.back().packed = true; outs.back().info = seg_info; }
    }
    if (ctx->merge_async) {   // a Mission whose stage B writes on a stream of its own waits for this before it overwrites its findings
        if (!ctx->ev_interleaved) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_interleaved, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_interleaved, s));
        ctx->interleave_pending = true;
    }
    // (merge_async: the last copies run on while the caller scans the next piece of the buffer; the sorts are in order with
    // everything else stage B does — post_stream —, so the missions' findings and the scratch may be reused at once)
    if (!ctx->merge_async) { int rc = merge_drain(ctx); if (rc != SX_OK) return rc; }
    guard.done = true;
    for (size_t k = 0; k < nm; k++) {
        if (per[k].ext.p) ctx->pool->give(per[k].ext);
        per[k] = MissionFindings{};
    }
    if (!outs.empty()) outs[0].replay_bytes = rb; else per[0].replay_bytes = rb;
    into->pool = ctx->pool;
    for (auto& o : outs) { ctx->stats.replay_bytes += o.replay_bytes; into->segs.push_back(std::move(o)); o.ext = {}; }
    if (getenv("SX_TIMING"))
        fprintf(stderr, "[sx] device merge of %zu missions (%zu held back on the device): %llu findings, %llu parts, %.2f ms\n", with, deferred,
                (unsigned long long)total, (unsigned long long)K, now_ms() - tm0);
    return SX_OK;
}

// A Mission whose decoder state cannot be derived from the bytes near a position (ISO-2022-JP): FindingCollection::from over every
// window of the buffer, in order, from the carried state — no stage A, no regions.  Device-resident input comes to the host in pieces.
static int host_sequential_mission(sx_ctx* ctx, size_t k, ByteView& bytes, const ReplayJob& job, MissionFindings* out) {
    const Mission& m = ctx->missions[k];
    ScannerState st = ctx->states[k];
    if (bytes.all_on_host() || !job.d_bytes) {
        replay_exact_windows(m, st, job.consumed0[k], job.stream0[k], bytes, job.len, job.file_id, 0, job.len, out, job.is_last);
    } else {
        struct Piece : ByteView {
            std::vector<uint8_t> buf; uint64_t base = 0;
            const uint8_t* span(uint64_t off, size_t, size_t*) override { return buf.data() + (off - base); }
        } piece;
        const uint64_t step = 64ull << 20;
        for (uint64_t at = 0; at < job.len; at += step) {
            const uint64_t n = std::min(step, job.len - at);
            piece.buf.resize(n); piece.base = at;
            HIP_TRY(ctx, hipMemcpy(piece.buf.data(), job.d_bytes + at, n, hipMemcpyDeviceToHost));
            replay_exact_windows(m, st, job.consumed0[k], job.stream0[k], piece, job.len, job.file_id, at, at + n, out, job.is_last);
        }
    }
    out->replay_bytes += job.len;
    if (job.slice_base) for (auto& f : out->v) f.slice_index += job.slice_base;
    if (job.commit_state) {
        st.consumed_bytes = job.consumed0[k] + job.len;
        st.stream_bytes = job.stream0[k] + job.len;
        ctx->states[k] = st;
    }
    return SX_OK;
}

// Stage B for all missions: every (mission, part) pair is one task for a small thread pool;
// part 0 of a mission starts from its entry state, the others speculate, and the per-mission
// stitch verifies/repairs them serially.
int replay_all(sx_ctx* ctx, ByteView& bytes, const ReplayJob& job, const std::vector<RunList>& runs,
               Result* into, uint64_t* end_pos, PreReplayed* pre) {
    const double t0 = now_ms();
    const size_t nm = ctx->missions.size();
    const unsigned nthreads = replay_threads(ctx);
    std::vector<std::vector<uint64_t>> bounds(nm);
    std::vector<std::vector<ReplayPart>> parts(nm);
    std::vector<std::pair<size_t, size_t>> tasks;
    std::vector<char> on_device(nm, 0);
    uint64_t host_runs = 0;
    for (size_t k = 0; k < nm; k++)
        on_device[k] = (pre && pre->done[k]) ? 2 : ctx->missions[k].host_sequential() ? 3 : (device_replay_wanted(ctx, job, k, runs[k].size()) ? 1 : 0);
    for (size_t k = 0; k < nm; k++) {
        if (on_device[k]) continue;
        HIP_TRY(ctx, runs[k].wait());
        // parts are speculative restarts: worth a thread each only if they hold real work
        unsigned want_parts = (unsigned)std::min<uint64_t>(nthreads, std::max<uint64_t>(1, runs[k].size() / 512));
        if (ctx->missions[k].c.chars_min_nb == 0) want_parts = 1;  // no speculative restarts (see device_replay_wanted)
        host_runs += runs[k].size();
        replay_plan_range(std::min(job.lo[k], job.hi), job.hi, want_parts, &bounds[k]);
        parts[k].resize(bounds[k].size() - 1);
        for (size_t p = 0; p + 1 < bounds[k].size(); p++) tasks.emplace_back(k, p);
    }
    std::atomic<size_t> next{ 0 };
    std::vector<double> task_ms(tasks.size(), 0.0);
    auto worker = [&]() {
        for (;;) {
            const size_t t = next.fetch_add(1);
            if (t >= tasks.size()) break;
            const size_t k = tasks[t].first, p = tasks[t].second;
            const double tt0 = now_ms();
            replay_part(ctx->missions[k], ctx->states[k], job.consumed0[k], job.stream0[k], bytes, job.len, job.file_id,
                        job.is_last, runs[k].data(), runs[k].size(), bounds[k][p], bounds[k][p + 1],
                        p == 0 && job.entry_exact[k], &parts[k][p]);
            task_ms[t] = now_ms() - tt0;
        }
    };
    const size_t nw = host_runs < 2048 ? 1 : std::min<size_t>(nthreads, tasks.size());
    if (nw <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < nw; i++) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    const double t_parts = now_ms();
    std::vector<MissionFindings> per(nm);
    std::vector<uint64_t> ends(nm, 0);
    for (size_t k = 0; k < nm; k++) {
        if (on_device[k] == 2) { per[k] = std::move(pre->per[k]); pre->per[k].ext = {}; ends[k] = pre->ends[k]; }
        else if (on_device[k] == 3) {
            int rc = host_sequential_mission(ctx, k, bytes, job, &per[k]);
            if (rc != SX_OK) return rc;
            ends[k] = job.len;
        } else if (on_device[k]) {
            int rc = device_replay_mission(ctx, k, bytes, job, runs[k], &per[k], &ends[k], 0);
            if (rc != SX_OK) return rc;
        }
    }
    auto stitch = [&](size_t k) {
        if (on_device[k]) return;
        ScannerState st = ctx->states[k];
        replay_stitch(ctx->missions[k], st, job.consumed0[k], job.stream0[k], bytes, job.len, job.file_id, job.is_last,
                      runs[k].data(), runs[k].size(), parts[k], &per[k], nthreads, &ends[k]);
        if (job.commit_state) ctx->states[k] = st;
        if (job.slice_base) for (auto& f : per[k].v) f.slice_index += job.slice_base;
    };
    if (nm == 1) stitch(0);
    else {
        std::vector<std::thread> th;
        for (size_t k = 0; k < nm; k++) th.emplace_back(stitch, k);
        for (auto& t : th) t.join();
    }
    if (end_pos) for (size_t k = 0; k < nm; k++) end_pos[k] = ends[k];
    const double t_stitch = now_ms();
    SX_TL("replay_all: host parts + stitch done");
    const size_t count_before = into->count();
    {   // several missions with findings that are all still on the device: interleave them there
        // (a stable radix sort by position) instead of finding by finding on the host
        int rc = device_merge(ctx, job, per, into);
        if (rc != SX_OK) return rc;
    }
    SX_TL("replay_all: device_merge done");
    merge_findings(per, ctx->pool, into);
    SX_TL("replay_all: merge_findings done");
    if (getenv("SX_TIMING")) {
        double mx = 0, sum = 0;
        for (double v : task_ms) { sum += v; mx = std::max(mx, v); }
        fprintf(stderr, "[sx] replay: parts %.2f ms (%zu tasks, %zu workers; task sum %.1f max %.1f ms), stitch %.2f ms, merge %.2f ms, "
                        "on-demand fetches so far %llu\n", t_parts - t0, tasks.size(), nw, sum, mx, t_stitch - t_parts,
                now_ms() - t_stitch, (unsigned long long)ctx->ondemand_fetches);
    }
    for (auto& mf : per) ctx->stats.replay_bytes += mf.replay_bytes;
    ctx->stats.findings += into->count() - count_before;
    ctx->stats.replay_ms += now_ms() - t0;
    return SX_OK;
}

ReplayJob whole_chunk_job(sx_ctx* ctx, uint64_t len, int file_id, bool is_last) {
    ReplayJob j;
    const size_t nm = ctx->missions.size();
    j.len = len; j.file_id = file_id; j.is_last = is_last; j.hi = len;
    j.lo.assign(nm, 0); j.entry_exact.assign(nm, 1);
    for (size_t k = 0; k < nm; k++) { j.consumed0.push_back(ctx->states[k].consumed_bytes); j.stream0.push_back(ctx->states[k].stream_bytes); }
    return j;
}


// Downloads what the host part of stage B reads: the buffer's first and last 64 KiB ("base",
// entry and exit of every mission) and the replay ranges of the missions the host replays.
// runs == nullptr: the base only, copied into the view.  skip: missions not to plan for; if the
// plan then holds nothing beyond the base and `base_view` already has it, nothing is done and
// *used_base is set.
int download_for_replay(sx_ctx* ctx, const uint8_t* d_bytes, uint64_t len,
                               const std::vector<RunList>* runs_opt, SparseDeviceBytes* view,
                               const ReplayJob& job, const std::vector<char>* skip, bool* used_base) {
    const size_t nm = runs_opt ? ctx->missions.size() : 0;
    static const std::vector<RunList> no_runs;
    const std::vector<RunList>& runs = runs_opt ? *runs_opt : no_runs;
    if (used_base) *used_base = false;
        const double t0 = now_ms();
        std::vector<std::pair<uint64_t, uint64_t>> rg;
        // what the host always looks at: the chunk's first windows and its tail
        rg.emplace_back(0, std::min<uint64_t>(len, 64 * 1024));
        if (len > 64 * 1024) rg.emplace_back(len - 64 * 1024, len);
        for (size_t k = 0; k < nm; k++) {
            if ((skip && (*skip)[k]) || ctx->missions[k].host_sequential() || device_replay_wanted(ctx, job, k, runs[k].size())) continue;  // stage B of this mission runs on the device (or fetches its bytes itself)
            HIP_TRY(ctx, runs[k].wait());
            const size_t before = rg.size();
            // same partition count as replay_all will use
            unsigned want_parts = (unsigned)std::min<uint64_t>(replay_threads(ctx), std::max<uint64_t>(1, runs[k].size() / 512));
            if (ctx->missions[k].c.chars_min_nb == 0) want_parts = 1;
            replay_ranges(ctx->missions[k], ctx->states[k], len, runs[k].data(), runs[k].size(), want_parts, &rg);
            // a mission's ranges come out almost sorted (runs are); fix up, then merge the sorted lists
            if (!std::is_sorted(rg.begin() + before, rg.end())) std::sort(rg.begin() + before, rg.end());
            std::inplace_merge(rg.begin(), rg.begin() + before, rg.end());
        }
        const double t_rg = now_ms();
        std::vector<std::pair<uint64_t, uint64_t>> mg;
        mg.reserve(rg.size());
        for (auto& r : rg) {
            if (!mg.empty() && r.first <= mg.back().second) mg.back().second = std::max(mg.back().second, r.second);
            else mg.push_back(r);
        }
        if (used_base) {  // nothing beyond the base (already in the caller's view)?
            bool inside = true;
            for (auto& r : mg)
                inside = inside && (r.second <= std::min<uint64_t>(len, 64 * 1024) || (len > 64 * 1024 && r.first >= len - 64 * 1024));
            if (inside) { *used_base = true; return SX_OK; }
        }
        // split long ranges so that one gather wavefront never copies more than 64 KiB
        std::vector<uint64_t> seg_src, seg_dst;
        std::vector<uint32_t> seg_len;
        uint64_t total = 0;
        for (auto& r : mg)
            for (uint64_t a = r.first; a < r.second; a += 65536) {
                const uint64_t n = std::min<uint64_t>(65536, r.second - a);
                seg_src.push_back(a); seg_dst.push_back(total); seg_len.push_back((uint32_t)n);
                total += n;
            }
        const double t_seg = now_ms();
        if (total) {
            hipStream_t s = ctx->dev[0].stream_b;
            const size_t ns = seg_src.size();
            const uint64_t seg_bytes = ns * (8 + 8 + 4) + 64;
            int rc2 = ensure_pinned(ctx, total + 64);
            if (rc2 != SX_OK) return rc2;
            rc2 = ensure_scratch(ctx, total + seg_bytes + 256);
            if (rc2 != SX_OK) return rc2;
            uint8_t* d_out = ctx->d_scratch;
            uint64_t* d_src = (uint64_t*)(ctx->d_scratch + ((total + 255) & ~255ull));
            uint64_t* d_dst = d_src + ns;
            uint32_t* d_len = (uint32_t*)(d_dst + ns);
            HIP_TRY(ctx, hipMemcpyAsync(d_src, seg_src.data(), ns * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(d_dst, seg_dst.data(), ns * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, hipMemcpyAsync(d_len, seg_len.data(), ns * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(ctx, launch_gather(d_bytes, d_out, d_src, d_dst, d_len, (uint32_t)ns, s));
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_pin, d_out, total, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            uint64_t off = 0;
            for (auto& r : mg) {
                if (runs_opt) view->add(r.first, r.second, ctx->h_pin + off);
                else view->add_copy(r.first, r.second, ctx->h_pin + off);
                off += r.second - r.first;
            }
        }
        ctx->stats.d2h_ms += now_ms() - t0;
        if (getenv("SX_TIMING"))
            fprintf(stderr, "[sx] sparse download: ranges %.2f ms, sort+merge+segments %.2f ms (%zu ranges, %zu segs), gather+d2h %.2f ms (%.1f MB)\n",
                    t_rg - t0, t_seg - t_rg, mg.size(), seg_src.size(), now_ms() - t_seg, total / 1e6);
    return SX_OK;
}

}  // namespace sx
