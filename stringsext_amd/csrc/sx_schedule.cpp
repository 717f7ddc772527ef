// sx_schedule.cpp — one buffer from stage A to the findings: mission order, overlap of a finished
// mission's stage B with the scans still running, pieces of a large buffer, one shard of a
// sharded scan.
#include "sx_ctx.hpp"
#include <thread>

using namespace sx;

namespace sx {

// Bytes per piece of a large buffer (a multiple of the slice length), or `len` if the buffer is
// scanned in one go.  With pieces, stage A of piece p+1 and p+2 is queued while piece p is
// sorted, joined and replayed.  Rounds 1-5 (one scan launch per Mission, 35 ms per 64 GiB): no gain — a Mission's stage B already
// ran next to the other Missions' scans.  Round 6 (ONE fused launch, 13 ms): every Mission's stage B — 0.5 ms of joins, 1.3 ms of
// replay, 2.5 ms for 120 MB of findings over PCIe on the headline — would follow the scan; cut in two, the first half's stage B and
// copy run under the second half's scan: 18.2 -> 16.5 ms per 64 GiB (four pieces: 17.8, eight: 21.2 — a piece costs ~1 ms of fixed
// host waits; profiles/r06d_*).  So: two pieces from 16 GiB on when a fused launch is in play; SX_PIECE_MIB sets the size (0: one piece).
uint64_t piece_bytes(const sx_ctx* ctx, uint64_t len) {
    uint64_t piece = 0;
    bool halves = false;
    if (const char* e = getenv("SX_PIECE_MIB")) piece = (uint64_t)atoll(e) << 20;
    else if (len >= (16ull << 30) && ctx->missions.size() >= 2 && !(ctx->opt.flags & (SX_OPT_NO_FUSED_SCAN | SX_OPT_MISSION_STREAMS | SX_OPT_RESULT_ON_DEVICE))) {
        // (the first piece the larger one: its stage B must fit under the second piece's scan, and what follows the second piece's
        // scan — its own stage B and copy — is the step's tail; SX_PIECE_FRAC: percent of the buffer in the first piece)
        uint64_t pct = 50;
        if (const char* e = getenv("SX_PIECE_FRAC")) pct = (uint64_t)std::min(95, std::max(5, atoi(e)));
        piece = ((len / 100) * pct + kInputBufLen - 1) / kInputBufLen * kInputBufLen;
        halves = true;
    }
    if (piece == 0 || (!halves && len < 2 * piece)) return len;
    // a double-byte mission's token grid at a piece start follows from the bytes in front of it, which the
    // kernels of a piece queued ahead cannot be told: such buffers are scanned in one go
    for (const Mission& m : ctx->missions) if (m.is_dbcs()) return len;
    return piece / kInputBufLen * kInputBufLen;
}

// Bytes per piece of a buffer whose OUTPUT is large (several Missions on string-dense input: BASELINE config 5 yields 19 GB of
// findings per 64 GiB), or `len`.  Such a scan is bound by the copy of the interleaved findings to the host (53 GB/s), which
// device_merge can only start when every Mission has replayed the range: the buffer is cut into pieces that are scanned, replayed
// and merged one after the other — each piece starts when the one before has left its ScannerState, exactly like consecutive
// sx_scan calls, so the double-byte Missions' token grid is known — and the copy of a piece's findings runs next to the kernels
// of the following piece (sx_ctx::merge_async).  Sized from the last buffer's output so that a piece's findings are one part of
// the merger (2.25 GB); the first buffer of a stream is scanned in one go.  SX_SEQ_PIECE_MIB / SX_SEQ_PIECE_KIB set the size (0: never).
static uint64_t seq_piece_bytes(const sx_ctx* ctx, uint64_t len) {
    uint64_t piece = 0;
    if (const char* e = getenv("SX_SEQ_PIECE_KIB")) {   // (tests: small buffers)
        piece = (uint64_t)atoll(e) << 10;
        if (piece == 0) return len;
    } else if (const char* e = getenv("SX_SEQ_PIECE_MIB")) {
        piece = (uint64_t)atoll(e) << 20;
        if (piece == 0) return len;
    } else {
        if (ctx->missions.size() < 2 || ctx->out_density <= 0) return len;
        const double out = ctx->out_density * (double)len;
        if (out < 4.0 * (double)(1ull << 30)) return len;
        const uint64_t n = (uint64_t)(out / (2.25 * (double)(1ull << 30))) + 1;
        piece = std::max<uint64_t>((len + n - 1) / n, 1ull << 30);   // (n equal pieces: no sliver at the end)
    }
    piece = std::max<uint64_t>((piece + kInputBufLen - 1) / kInputBufLen, 2) * kInputBufLen;
    return len < 2 * piece ? len : piece;
}

// What the scan kernel needs to know about the state at buffer byte 0: UTF-16 the unit parity of the stream
// offset; Big5 / EUC-JP how many bytes finish the token pending in the carried decoder (0 without a state).
static int entry_param(sx_ctx* ctx, size_t k, const Decoder* carried, const uint8_t* host_bytes, const uint8_t* d_bytes,
                       uint64_t len, uint64_t stream_off, uint32_t* out, uint32_t* out_replay = nullptr) {
    const Mission& m = ctx->missions[k];
    *out = (uint32_t)(stream_off & 1);
    if (out_replay) *out_replay = 0;
    if (!m.is_dbcs()) return SX_OK;
    *out = 0;
    if (!carried || carried->idle() || len == 0) return SX_OK;
    uint8_t first[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    const uint64_t n = std::min<uint64_t>(8, len);
    if (host_bytes) memcpy(first, host_bytes, n);
    else HIP_TRY(ctx, hipMemcpy(first, d_bytes, n, hipMemcpyDeviceToHost));
    *out = carried->entry_skip_scan(first, n);                     // for the scan kernel (its own grammar for gb18030)
    if (out_replay) *out_replay = carried->entry_skip(first, n);   // for the replay: the true token grid
    if (getenv("SX_DEBUG_ENTRY")) { const DDecoder& dd = const_cast<Decoder*>(carried)->raw(); fprintf(stderr, "[sx] entry mission %zu: dlead %02x gb2 %02x gb3 %02x rq_n %u first %02x %02x -> scan %u replay %u\n", k, dd.dlead, dd.gb2, dd.gb3, dd.rq_n, first[0], first[1], *out, out_replay ? *out_replay : 0u); }
    return SX_OK;
}
int set_entry_params(sx_ctx* ctx, bool carried_state_is_entry, const uint8_t* host_bytes, const uint8_t* d_bytes, uint64_t len,
                     uint64_t stream_off, std::vector<uint32_t>* parity) {
    parity->clear();
    for (size_t k = 0; k < ctx->missions.size(); k++) {
        uint32_t ep = 0, ep_replay = 0;
        int rc = entry_param(ctx, k, carried_state_is_entry ? &ctx->states[k].decoder : nullptr, host_bytes, d_bytes, len,
                             stream_off, &ep, &ep_replay);
        if (rc != SX_OK) return rc;
        parity->push_back(ep);
        ctx->missions[k].buf_entry_skip = ctx->missions[k].is_dbcs() ? ep_replay : 0u;
    }
    return SX_OK;
}

// Missions in the order their kernels are queued (by their number of long runs in the previous buffer).
void mission_order(sx_ctx* ctx, std::vector<int>* out) {
    const size_t nm = ctx->missions.size();
    std::vector<int>& order = *out;
    order.resize(nm);
    for (size_t k = 0; k < nm; k++) order[k] = (int)k;
    if (ctx->last_runs.size() != nm) return;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ctx->last_runs[(size_t)x] > ctx->last_runs[(size_t)y]; });
    // Round 5: the busiest Mission goes FIRST whatever its size (SX_BUSIEST_LAST=1 / 2: last / second to last, the defaults of
    // rounds 2-3 / 4).  Its stage B — some forty small kernels, a dozen of them waited for by the host — then has two scan launches
    // to hide behind instead of one and is done when the last of them ends; measured in one session on the headline (three Missions,
    // 64 GiB, 2.8 M runs in the UTF-8 Mission; profiles/r05a_order_*.json): first 38.2 ms per step, second to last 39.9, last 40.5
    // (= the scans alone, 34.5 ms, plus the 6.0 ms stage B takes on an idle chip).  With round 4's scan kernels second to last was
    // the best (38.1): they were bound by VALU issue and lost to stage B what it won; the round-5 kernels issue a tenth fewer
    // instructions per tile and keep two tiles in flight.
    int mode = 0;
    if (const char* e = getenv("SX_BUSIEST_LAST")) mode = atoi(e);
    if (mode) std::reverse(order.begin(), order.end());
    if (mode == 2 && nm >= 3) std::swap(order[nm - 1], order[nm - 2]);
}

int sync_streams_and_return(sx_ctx* ctx, int rc) {  // do not leave kernels running on the caller's buffer
    ctx->merge_async = false;
    (void)merge_drain(ctx);   // ... nor copies into blocks of a result that is about to be freed
    for (auto& d : ctx->dev) { (void)hipStreamSynchronize(d.stream); (void)hipStreamSynchronize(d.stream_b); }
    return rc;
}

struct BufferScan {
    const uint8_t* host_bytes = nullptr;  // the same bytes on the host, or nullptr (device-resident input)
    const uint8_t* d_bytes = nullptr;
    uint64_t len = 0;
    std::vector<int> order;
    std::vector<uint32_t> parity;         // per mission: stream offset of byte 0, & 1
    std::vector<uint64_t> minc;           // per mission: long-run threshold
    int slot = 0;
    std::unique_ptr<SparseDeviceBytes> base_view;  // entry/exit bytes of a device-resident buffer
    bool whole = false;                   // a whole buffer of sx_scan* (not a shard)
    // A Mission whose last whole buffer was string-dense and went through the wave kernels (sx_wave.cpp) does without stage A
    // for the next one: its stage B replays every window anyway, and says afterwards whether the buffer was dense again.
    bool skip_scan(const sx_ctx* ctx, size_t k) const {
        return whole && len >= 2 * kInputBufLen && k < ctx->wave_pred.size() && ctx->wave_pred[k] && ctx->missions[k].wave_ok && !getenv("SX_WAVE_KEEP_SCAN")
               && !(getenv("SX_WAVE_REPLAY") && !atoi(getenv("SX_WAVE_REPLAY")));
    }
    bool none_scanned(const sx_ctx* ctx) const {
        if (not_scanned.size() != ctx->missions.size()) return false;
        for (char c : not_scanned) if (!c) return false;
        return true;
    }

    // what the host reads whatever the runs are (entry and exit of every mission): fetched
    // before the kernels start, so that the copy does not queue behind them
    int fetch_base(sx_ctx* ctx) {
        base_view.reset();
        if (host_bytes) return SX_OK;
        base_view.reset(new SparseDeviceBytes(ctx, d_bytes, len));
        if (none_scanned(ctx)) return SX_OK;   // the wave path reads two windows' worth: fetched when asked for (SparseDeviceBytes::span)
        ReplayJob none;
        return download_for_replay(ctx, d_bytes, len, nullptr, base_view.get(), none);
    }
    std::vector<char> not_scanned;        // per mission: launch() left its scan kernel out (skip_scan)
    int launch(sx_ctx* ctx) {
        not_scanned.assign(ctx->missions.size(), 0);
        // (one call for all of them: the Missions the fused kernel holds share a launch — sx_stage_a.cpp —, the others follow in `order`)
        std::vector<int> ks; std::vector<uint32_t> par; std::vector<uint64_t> mc;
        for (int k : order) {
            if (ctx->missions[(size_t)k].host_sequential() || skip_scan(ctx, (size_t)k)) { not_scanned[(size_t)k] = 1; continue; }
            ks.push_back(k); par.push_back(parity[(size_t)k]); mc.push_back(minc[(size_t)k]);
        }
        return ks.empty() ? SX_OK : stage_a_launch(ctx, ks, d_bytes, len, par, mc, slot);
    }
    // Collect stage A mission by mission (in launch order); a mission whose stage B runs on the
    // device is replayed at once, while the kernels of the missions behind it still scan; the
    // host's share of stage B follows when all kernels are done.  `after_last_finish` runs when
    // the record slot is free again (piece pipeline: queue the next piece).
    int finish_and_replay(sx_ctx* ctx, const ReplayJob& job, const std::function<int()>& after_last_finish,
                          std::vector<RunList>* runs, Result* into, uint64_t* ends) {
        const size_t nm = ctx->missions.size();
        runs->assign(nm, RunList{});
        HostBytes host_view(host_bytes ? host_bytes : (const uint8_t*)"");
        ByteView& early_view = host_bytes ? (ByteView&)host_view : (ByteView&)*base_view;
        PreReplayed pre(nm);
        if (ctx->last_runs.size() != nm) ctx->last_runs.assign(nm, 0);
        ctx->wave_off.assign(nm, 0);
        if (ctx->wave_pred.size() != nm) ctx->wave_pred.assign(nm, 0);
        // String-dense Missions that go without stage A (their last buffer went through the wave kernels) start their stage B at
        // once, a host thread and a stream each, NEXT TO each other and to the other Missions' scans and stage B: Big5's count pass
        // runs at two wavefronts per SIMD and waits on memory most of the time, KOI8-R's at four, the lane-per-region replay of a
        // third Mission is latency-bound as well — one after the other on one stream (as before, SX_WAVE_THREADS=0) they leave most
        // of the chip idle.  The threads share nothing but the context's lock for statistics (sx_wave.cpp, own_stream).
        std::vector<std::thread> wave_threads(nm);
        std::vector<int> wave_rc(nm, SX_OK);
        std::vector<char> threaded(nm, 0);
        struct Joiner {
            std::vector<std::thread>& t;
            ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); }
        } joiner{ wave_threads };
        const uint64_t defer_all = nm >= 2 ? [] { const char* e = getenv("SX_DEFER_MIN_BYTES"); return e ? (uint64_t)atoll(e) : (256ull << 20); }() : 0;
        if (nm >= 2 && !(getenv("SX_WAVE_THREADS") && !atoi(getenv("SX_WAVE_THREADS")))) {
            if (ctx->wave_density.size() != nm) ctx->wave_density.assign(nm, 0.0);
            for (size_t k = 0; k < nm; k++) {
                const bool unscanned = k < not_scanned.size() && not_scanned[k];
                if (ctx->missions[k].host_sequential() || !unscanned || !wave_replay_wanted(ctx, job, k, len)) continue;
                threaded[k] = 1;
                ctx->stats.bytes_scanned += len;
                wave_threads[k] = std::thread([ctx, k, &early_view, &job, &pre, &wave_rc, defer_all]() {
                    (void)hipSetDevice(ctx->device);
                    wave_rc[k] = wave_replay_mission(ctx, k, early_view, job, &pre.per[k], &pre.ends[k], defer_all, true);
                });
            }
        }
        std::vector<size_t> order2;
        for (size_t oi = 0; oi < nm; oi++) if (!threaded[(size_t)order[oi]]) order2.push_back((size_t)order[oi]);
        for (size_t oi = 0; oi < nm; oi++) if (threaded[(size_t)order[oi]]) order2.push_back((size_t)order[oi]);
        for (size_t oi = 0; oi < nm; oi++) {
            const size_t k = order2[oi];
            std::vector<RunList> one;
            int rc = SX_OK;
            const bool unscanned = k < not_scanned.size() && not_scanned[k];
            bool replayed = false;
            if (threaded[k]) {
                wave_threads[k].join();
                rc = wave_rc[k];
                if (rc == SX_WAVE_FALLBACK) {   // nothing was produced and nothing changed: the way device_replay_mission takes then, inline
                    ctx->wave_off[k] = 1; ctx->wave_pred[k] = 0;
                    ctx->stats.bytes_scanned -= len;
                    rc = SX_OK;
                } else if (rc != SX_OK) return rc;
                else replayed = true;
            }
            if (replayed) {
                one.assign(1, RunList{});
                one[0].n = len / 16; one[0].skipped = true; one[0].complete = true;
            } else if (ctx->missions[k].host_sequential()) {   // no stage A at all: every window is replayed on the host (sx_stage_b.cpp)
                one.assign(1, RunList{});
                one[0].complete = true;
            } else if (unscanned && wave_replay_wanted(ctx, job, k, len)) {   // no stage A: "as dense as the last buffer"
                one.assign(1, RunList{});
                one[0].n = len / 16; one[0].skipped = true; one[0].complete = true;
                ctx->stats.bytes_scanned += len;
            } else {
                if (unscanned) {   // (not launched with the others, but the job is not one for the wave kernels after all)
                    ctx->wave_pred[k] = 0;
                    rc = stage_a_launch(ctx, { (int)k }, d_bytes, len, { parity[k] }, { minc[k] }, slot);
                    if (rc != SX_OK) return rc;
                }
                rc = stage_a_finish(ctx, { (int)k }, d_bytes, len, { parity[k] }, { minc[k] }, slot, &one, true, &job);
            }
            if (rc != SX_OK) return rc;
            (*runs)[k] = std::move(one[0]);
            if (!(*runs)[k].own.empty()) (*runs)[k].use_own();  // the vector moved: point at it again
            ctx->last_runs[k] = (*runs)[k].size();
            // The record slot is free again once the last Mission's records are out of it — unless they were only COUNTED
            // (RunList::skipped: a string-dense buffer, the wave kernels replay every window): if those give the buffer back
            // (SX_NEED_RUNS below) stage A is launched again into this very slot, so the next piece may only be queued into it when
            // that is settled (piece p+2's scan shared the slot with the re-scan of piece p).
            const bool last_mission = oi + 1 == nm;
            bool next_queued = false;
            auto queue_next = [&]() -> int {
                if (!last_mission || !after_last_finish || next_queued) return SX_OK;
                next_queued = true;
                return after_last_finish();
            };
            if (!(*runs)[k].skipped && (rc = queue_next()) != SX_OK) return rc;
            if (replayed) { pre.done[k] = 1; if ((rc = queue_next()) != SX_OK) return rc; continue; }
            if (device_replay_wanted(ctx, job, k, (*runs)[k].size())) {
                // (with several missions a large output stays on the device: replay_all interleaves them there, one copy instead of two)
                uint64_t defer = nm >= 2 ? (256ull << 20) : 0;
                if (const char* e = getenv("SX_DEFER_MIN_BYTES")) defer = nm >= 2 ? (uint64_t)atoll(e) : 0;
                SX_TL("mission %zu: device replay begins (%zu runs)", k, (*runs)[k].size());
                rc = device_replay_mission(ctx, k, early_view, job, (*runs)[k], &pre.per[k], &pre.ends[k], defer);
                SX_TL("mission %zu: device replay done", k);
                if (rc == SX_NEED_RUNS) {   // the wave kernels gave up on a buffer whose runs were only counted: stage A in full, then the other stage B
                    ctx->wave_off[k] = 1;
                    rc = stage_a_launch(ctx, { (int)k }, d_bytes, len, { parity[k] }, { minc[k] }, slot);
                    if (rc == SX_OK) rc = stage_a_finish(ctx, { (int)k }, d_bytes, len, { parity[k] }, { minc[k] }, slot, &one, true, nullptr);
                    if (rc != SX_OK) return rc;
                    (*runs)[k] = std::move(one[0]);
                    if (!(*runs)[k].own.empty()) (*runs)[k].use_own();
                    ctx->last_runs[k] = (*runs)[k].size();
                    if ((rc = queue_next()) != SX_OK) return rc;   // (the slot's records are out: now the next piece may have it)
                    if (!device_replay_wanted(ctx, job, k, (*runs)[k].size())) continue;   // few runs after all: the host's share of stage B
                    rc = device_replay_mission(ctx, k, early_view, job, (*runs)[k], &pre.per[k], &pre.ends[k], defer);
                }
                if (rc != SX_OK) return rc;
                pre.done[k] = 1;
            }
            if ((rc = queue_next()) != SX_OK) return rc;
        }
        SX_TL("all missions finished / replayed on the device");
        if (host_bytes) return replay_all(ctx, host_view, job, *runs, into, ends, &pre);
        SparseDeviceBytes view(ctx, d_bytes, len);
        bool base_is_enough = false;
        int rc = download_for_replay(ctx, d_bytes, len, runs, &view, job, &pre.done, &base_is_enough);
        if (rc != SX_OK) return rc;
        return replay_all(ctx, base_is_enough ? (ByteView&)*base_view : (ByteView&)view, job, *runs, into, ends, &pre);
    }
};

// One buffer, start to end: stage A on the device, stage B on device and host, the findings in
// print order.
//  * The missions' scan kernels queue up in one stream, busiest mission (of the last buffer)
//    first; as soon as a mission's kernel is done its records are packed and joined and — if its
//    stage B runs on the device — replayed, in the second stream, while the kernels of the
//    remaining missions still scan.  What the host replays follows when all kernels are done.
//  * A large buffer can be cut into pieces (SX_PIECE_MIB) that behave exactly like consecutive
//    sx_scan calls (ScannerState carried from piece to piece) with their kernels queued two deep.
int scan_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes, uint64_t len, int file_id,
                       int is_last, sx_result** out, uint32_t slice_base0, sx_result* append_to) {
    const double t_begin = now_ms();
    g_tl_on = getenv("SX_TIMELINE") ? atoi(getenv("SX_TIMELINE")) : 0; g_tl_t0 = t_begin;
    SX_TL("scan_common: %llu bytes", (unsigned long long)len);
    const size_t nm = ctx->missions.size();
    // SX_OPT_RESULT_ON_DEVICE: a result left in HBM is the context's memory, and every BUFFER reuses it — the chunks of one sx_scan_stream /
    // sx_scan_file call too (the epoch only advanced per API call, so an earlier chunk's segment still passed the check
    // after the next chunk had overwritten it)
    ctx->dev_epoch->fetch_add(1);
    ctx->shard_runs_valid = false;   // the run lists a shard call left behind are about to be overwritten
    std::vector<uint64_t> stream0(nm);
    for (size_t k = 0; k < nm; k++) stream0[k] = ctx->states[k].stream_bytes;
    std::vector<int> order;
    mission_order(ctx, &order);
    const uint64_t seq_piece = ctx->host_only ? len : seq_piece_bytes(ctx, len);
    const bool seq = seq_piece < len;
    const uint64_t piece = seq ? seq_piece : piece_bytes(ctx, len);
    const uint64_t n_pieces = len ? (len + piece - 1) / piece : 1;
    ctx->single_piece = n_pieces == 1 && !append_to;   // (a result that accumulates several buffers lives in host memory: the device block is reused per buffer)
    const uint64_t merged0 = ctx->merged_out_bytes;
    int entry_rc = SX_OK;   // (a failed read of the buffer's first bytes must not go unnoticed: the token grid would be wrong)
    auto make = [&](uint64_t p) {
        BufferScan b;
        const uint64_t off = p * piece;
        b.host_bytes = host_bytes ? host_bytes + off : nullptr;
        b.d_bytes = d_bytes + off;
        b.len = std::min(piece, len - off);
        b.order = order;
        b.whole = true;
        b.slot = (int)(p & 1);
        for (size_t k = 0; k < nm; k++) {
            uint32_t ep = (uint32_t)((stream0[k] + off) & 1);
            if (ctx->missions[k].is_dbcs()) {  // one piece only (piece_bytes): the carried decoder describes byte 0
                uint32_t ep_replay = 0;
                entry_rc = std::min(entry_rc, entry_param(ctx, k, &ctx->states[k].decoder, b.host_bytes, b.d_bytes, b.len, 0, &ep, &ep_replay));
                ctx->missions[k].buf_entry_skip = ep_replay;
            }
            b.parity.push_back(ep);
            b.minc.push_back(ctx->missions[k].long_run);
        }
        return b;
    };
    ResultHolder res;
    int rc = SX_OK;
    if (seq) {   // one piece after the other, the copy of a piece's findings next to the following piece's kernels
        ctx->merge_async = true;
        ctx->stats.seq_pieces += n_pieces;
        for (uint64_t p = 0; p < n_pieces; p++) {
            BufferScan b = make(p);   // (the ScannerStates the piece in front left)
            if (entry_rc != SX_OK) return sync_streams_and_return(ctx, entry_rc);
            if ((rc = b.launch(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
            if ((rc = b.fetch_base(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
            ReplayJob job = whole_chunk_job(ctx, b.len, file_id, is_last != 0 && p + 1 == n_pieces);
            job.d_bytes = b.d_bytes;
            job.slice_base = slice_base0 + (uint32_t)(p * piece / kInputBufLen);
            std::vector<RunList> runs;
            rc = b.finish_and_replay(ctx, job, nullptr, &runs, append_to ? &append_to->r : &res.r->r, nullptr);
            if (rc != SX_OK) return sync_streams_and_return(ctx, rc);
        }
        ctx->merge_async = false;
        if ((rc = merge_drain(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
        if (len) ctx->out_density = (double)(ctx->merged_out_bytes - merged0) / (double)len;
        ctx->stats.total_ms = now_ms() - t_begin;
        if (!append_to) *out = res.release();
        return SX_OK;
    }
    std::vector<BufferScan> pieces;
    for (uint64_t p = 0; p < n_pieces; p++) pieces.push_back(make(p));
    if (entry_rc != SX_OK) return entry_rc;
    uint64_t launched = 0;
    for (; launched < std::min<uint64_t>(2, n_pieces); launched++)
        if ((rc = pieces[launched].launch(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
    // the few bytes the host always reads: a tiny gather in the second stream, it finds room
    // next to the scan kernels within ~0.1 ms
    SX_TL("scans queued");
    if ((rc = pieces[0].fetch_base(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
    SX_TL("base fetched");
    for (uint64_t p = 0; p < n_pieces; p++) {
        BufferScan& b = pieces[p];
        if (p > 0 && (rc = b.fetch_base(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
        ReplayJob job = whole_chunk_job(ctx, b.len, file_id, is_last != 0 && p + 1 == n_pieces);
        job.d_bytes = b.d_bytes;
        job.slice_base = slice_base0 + (uint32_t)(p * piece / kInputBufLen);
        std::vector<RunList> runs;
        rc = b.finish_and_replay(ctx, job,
                                 [&]() -> int { return launched < n_pieces ? pieces[launched++].launch(ctx) : SX_OK; },  // slot p&1 is free again
                                 &runs, append_to ? &append_to->r : &res.r->r, nullptr);
        if (rc != SX_OK) return sync_streams_and_return(ctx, rc);
    }
    if (len) ctx->out_density = (double)(ctx->merged_out_bytes - merged0) / (double)len;
    ctx->stats.total_ms = now_ms() - t_begin;
    SX_TL("scan_common done");
    if (!append_to) *out = res.release();
    return SX_OK;
}


int shard_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes,
                        const sx_run* const* given_runs, const uint64_t* given_n, uint64_t buf_off, uint64_t buf_len,
                        uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at, uint64_t file_stream_off, int file_id,
                        int reuse_runs, sx_result** out, uint64_t* end_pos) {
    if (!out || !end_pos || (buf_off % kInputBufLen) != 0 || own_lo < buf_off || own_hi < own_lo || own_hi > buf_off + buf_len) {
        ctx->err = "bad shard geometry"; return SX_E_INVALID;
    }
    ctx->sharded_call = true;   // (results stay host-side: the shard machinery reads them)
    const size_t nm = ctx->missions.size();
    const double t_begin = now_ms();
    for (const Mission& m : ctx->missions)
        if (m.host_sequential()) {
            ctx->err = "an ISO-2022-JP mission cannot be sharded: the character set in force at a shard start depends on every escape "
                       "sequence in front of it (its stage B is one sequential pass)";
            return SX_E_INVALID;
        }
    for (const Mission& m : ctx->missions)
        if (m.c.chars_min_nb == 0 && buf_off != 0) {
            ctx->err = "a mission with chars_min_nb 0 cannot be sharded: its state at a shard start cannot be derived from the bytes "
                       "in front of it (stage B runs as one sequential pass for it)";
            return SX_E_INVALID;
        }
    // Big5 / EUC-JP in a buffer that does not start the file: the token grid is known from the first byte outside the
    // lead range on (scan kernel and replay assume a token starts at byte 0, which only holds by then)
    if (buf_off > 0) {
        bool any = false;
        for (const Mission& m : ctx->missions) any = any || m.is_dbcs();
        if (any) {
            const uint64_t front = own_lo - buf_off;
            std::vector<char> found(nm, 0);   // per mission: a byte outside ITS lead range seen
            std::vector<uint8_t> tmp;
            auto all_found = [&]() { for (size_t k = 0; k < nm; k++) if (ctx->missions[k].is_dbcs() && !found[k]) return false; return true; };
            for (uint64_t at = 0; at < front && !all_found(); at += 65536) {
                const uint64_t n = std::min<uint64_t>(65536, front - at);
                const uint8_t* s = host_bytes ? host_bytes + at : nullptr;
                if (!s) {
                    tmp.resize(n);
                    HIP_TRY(ctx, hipMemcpy(tmp.data(), d_bytes + at, n, hipMemcpyDeviceToHost));
                    s = tmp.data();
                }
                for (size_t k = 0; k < nm; k++) {
                    const Mission& m = ctx->missions[k];
                    if (!m.is_dbcs() || found[k]) continue;
                    const int enc = m.c.encoding;
                    const bool two = enc_family((uint32_t)enc) == 4;
                    for (uint64_t i = 0; i < n; i++)
                        if (!(two ? dbcs_may_be_pending_after<4>(s[i], enc) : dbcs_may_be_pending_after<5>(s[i], enc))) { found[k] = 1; break; }
                }
            }
            if (!all_found()) {
                ctx->err = "no byte outside the lead range between the buffer start and own_lo: the token grid of a double-byte "
                           "mission is unknown there; repeat with a larger halo in front";
                return SX_E_HALO;
            }
        }
    }
    if (given_runs) {
        ctx->shard_runs.assign(nm, RunList{});
        for (size_t k = 0; k < nm; k++) ctx->shard_runs[k].assign(given_runs[k], given_runs[k] + given_n[k]);
    }
    // the runs of the last shard call may be reused for the same buffer only; runs the caller supplied are never "scanned" ones
    const void* buf_id = d_bytes ? (const void*)d_bytes : (const void*)host_bytes;
    const bool same_buffer = ctx->shard_runs_valid && ctx->shard_runs_off == buf_off && ctx->shard_runs_len == buf_len && ctx->shard_runs_ptr == buf_id;
    const bool scan_now = !given_runs && !(reuse_runs && same_buffer);
    if (given_runs || scan_now) ctx->shard_runs_valid = false;
    BufferScan b;
    {   // only the shard that starts the file knows the state at its byte 0 (the context's carried state)
        int rc = set_entry_params(ctx, buf_off == 0, host_bytes, d_bytes, buf_len, file_stream_off + buf_off, &b.parity);
        if (rc != SX_OK) return rc;
    }
    if (scan_now) {
        b.host_bytes = host_bytes; b.d_bytes = d_bytes; b.len = buf_len; mission_order(ctx, &b.order); b.slot = 0;
        for (size_t k = 0; k < nm; k++) b.minc.push_back(ctx->missions[k].long_run);
        int rc = b.launch(ctx);
        if (rc == SX_OK) rc = b.fetch_base(ctx);
        if (rc != SX_OK) return sync_streams_and_return(ctx, rc);
    }

    ReplayJob job;
    job.len = buf_len; job.file_id = file_id; job.is_last = false;
    job.hi = own_hi - buf_off;
    job.commit_state = job.hi >= buf_len;
    job.slice_base = (uint32_t)(buf_off / kInputBufLen);
    for (size_t k = 0; k < nm; k++) {
        uint64_t lo = own_lo;
        if (start_at && start_at[k] > lo) lo = start_at[k];
        if (lo > own_hi) lo = own_hi;
        job.lo.push_back(lo - buf_off);
        job.entry_exact.push_back(buf_off == 0 && lo == 0);
        job.consumed0.push_back(ctx->missions[k].c.counter_offset + file_stream_off + buf_off);
        job.stream0.push_back(file_stream_off + buf_off);
    }
    job.d_bytes = d_bytes;
    int rc;
    ResultHolder res;
    std::vector<uint64_t> ends(nm, 0);
    if (scan_now) {
        ctx->shard_runs_valid = false;
        rc = b.finish_and_replay(ctx, job, nullptr, &ctx->shard_runs, &res.r->r, ends.data());
        if (rc != SX_OK) return sync_streams_and_return(ctx, rc);
        ctx->shard_runs_valid = true; ctx->shard_runs_off = buf_off; ctx->shard_runs_len = buf_len; ctx->shard_runs_ptr = buf_id;
    } else if (host_bytes) {
        HostBytes view(host_bytes);
        rc = replay_all(ctx, view, job, ctx->shard_runs, &res.r->r, ends.data());
    } else {
        SparseDeviceBytes view(ctx, d_bytes, buf_len);
        rc = download_for_replay(ctx, d_bytes, buf_len, &ctx->shard_runs, &view, job);
        if (rc != SX_OK) return rc;
        rc = replay_all(ctx, view, job, ctx->shard_runs, &res.r->r, ends.data());
    }
    if (rc == SX_OK) *out = res.release();
    for (size_t k = 0; k < nm; k++) end_pos[k] = buf_off + ends[k];
    ctx->stats.total_ms = now_ms() - t_begin;
    return rc;
}


}  // namespace sx
