// sx_wave.cpp — stage B of a string-dense Mission through the wave-cooperative kernels (sx_wave_dev.hip): the host's share.
//
// The lane-per-region replay costs time per long run, the wave kernels cost time per input byte (every window of the
// buffer is replayed, a lane each, whether anything is found in it or not): they take over when a Mission has more
// than a run per ~500 bytes — `-e ascii -n 4` on binaries, text, a legacy code page on random bytes — and cover the
// Missions sx_wave_core.hpp names (no -g, no -r, 1 <= n <= q <= 64; a single-byte decoder, UTF-8, UTF-16LE / BE, Big5 / Shift_JIS / EUC-KR, EUC-JP).
//   host:   the buffer's first window(s) from the exact carried ScannerState (its leftover's bytes lie in the previous
//           buffer) — FindingCollection::from as ever, replay_exact_windows;
//   device: every other window: count pass -> exclusive sums + verification of the wavefronts' assumed entry states ->
//           write pass straight into the result's layout [findings][strings], in slabs of wavefronts whose copy to the
//           host runs while the next slab is written;
//   host:   the state handed to the next buffer, rebuilt from the (leftover chars, cut flag) the last window left.
#include "sx_ctx.hpp"
#include "sx_wave_core.hpp"

using namespace sx;

namespace sx {

// bytes per run below which the wave kernels are the cheaper stage B.  Measured on the MI355X: the lane-per-region path costs
// ~1.2 ns per run (single byte, UTF-8) / ~3.6 ns (two-byte family: sort + join, both passes); the wave kernels, since their writer
// works a lane per finding, 2.0 ps per input byte (single byte: 5.5 + 2.9 ms per 4 GiB with 15 M findings) / 6.3 ps (two-byte
// family: 24.7 + 2.2 ms, Big5 on random bytes with a run per 490 bytes) / 5.5 ps (UTF-8 on random bytes, where a decoder call starts
// every other byte; on text it is the single-byte figure).
static uint64_t wave_min_density_bytes(uint32_t family = 0) {
    static const uint64_t v = [] { const char* e = getenv("SX_WAVE_BYTES_PER_RUN"); return e ? (uint64_t)atoll(e) : 0ull; }();
    // (round 4: the count passes are 2.5 to 4 times faster — single byte 0.7 ps per byte, two-byte family 1.4, EUC-JP 2.3 — and a Mission on
    // the wave path leaves the shared stage-B stream alone: EUC-JP + Asian on random bytes, a run per 3.4 KB, C5: 444 -> 415 ms per step)
    return v ? v : (family == 5 ? 4000ull : family == 4 ? 1600ull : family == 1 ? 480ull : family == 2 ? 960ull : 1000ull);
}

// n_runs: the long runs (or records) of the buffer; heavy_tiles: 1 KiB tiles of it that took the scan kernel's general path.  Dense =
// a run per wave_min_density_bytes or more — or hardly any runs but half of the tiles on the general path: GIANT runs (a fill of
// printable bytes: gigabytes of spaces or '0' in a disk image), which the lane-per-region path hands to the host as one region.
bool wave_replay_wanted(const sx_ctx* ctx, const ReplayJob& job, size_t k, size_t n_runs, uint64_t heavy_tiles) {
    const Mission& m = ctx->missions[k];
    if (!m.wave_ok || !job.d_bytes || ctx->host_only || job.is_last || !job.commit_state) return false;
    if (k < ctx->wave_off.size() && ctx->wave_off[k]) return false;
    if (job.lo[k] != 0 || job.hi != job.len || !job.entry_exact[k] || job.len < 2 * kInputBufLen) return false;   // whole buffers only
    if (ctx->opt.flags & SX_OPT_HOST_REPLAY) return false;
    if (const char* e = getenv("SX_WAVE_REPLAY")) return atoi(e) != 0;
    if (getenv("SX_HOST_REPLAY") || getenv("SX_HOST_STITCH") || getenv("SX_NO_REPLAY_CACHE")) return false;   // tests of the other path
    if ((uint64_t)n_runs * wave_min_density_bytes(m.wave_family) > job.len) return true;
    // (round 4: the two-byte family too — inside a fill of lead-range bytes its wave kernels take the token grid from the wavefront in
    // front by parity; EUC-JP, whose tokens have two or three bytes, still gives up after 64 KiB of look-back)
    return m.wave_family <= 4 && heavy_tiles * 2048 > job.len && (uint64_t)n_runs * 4096 < job.len;
}

static uint32_t utf8_chars(const std::string& s) {
    uint32_t n = 0;
    for (unsigned char c : s) n += (c & 0xC0) != 0x80;
    return n;
}

// SX_OK, an error, or SX_WAVE_FALLBACK: nothing was produced and nothing changed — use the lane-per-region path.
// own_stream: the call runs on a host thread next to the other Missions' stage B (sx_schedule.cpp): its kernels go to the Mission's own
// stream, its scratch, totals and events are the Mission's own, statistics are added under the context's lock.
int wave_replay_mission(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, MissionFindings* out, uint64_t* end_pos,
                        uint64_t defer_min_bytes, bool own_stream) {
    const Mission& m = ctx->missions[k];
    MissionDev& d = ctx->dev[k];
    if (own_stream && !d.stream_w) {
        int lo = 0, hi = 0;   // (lo: the numerically greatest = lowest priority, hi: the highest)
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        int prio = hi;
        if (const char* e = getenv("SX_WAVE_STREAM_PRIO")) prio = atoi(e) == 0 ? lo : atoi(e) == 1 ? (lo + hi) / 2 : hi;
        HIP_TRY(ctx, hipStreamCreateWithPriority(&d.stream_w, hipStreamNonBlocking, prio));
    }
    const hipStream_t sb = own_stream ? d.stream_w : d.stream_b;
    const uint32_t W = (uint32_t)m.window, wps = wv_wps(W);
    const uint64_t len = job.len;
    const double t0 = now_ms();

    // ---- host: the first window(s), until the leftover's bytes lie inside this buffer
    ScannerState st = ctx->states[k];
    MissionFindings hf;
    uint64_t E = 0;
    const size_t src_per_out = m.wave_family == 2 ? 2 : 1;   // source bytes per byte of leftover text, at most (UTF-16: 'A' is two)
    for (int guard = 0; guard < 64 && E < len; guard++) {
        uint64_t ws; uint32_t wn;
        wv_window_at(wv_window_no(E, W, wps), W, wps, len, &ws, &wn);
        const uint64_t next = ws + wn;
        replay_exact_windows(m, st, job.consumed0[k], job.stream0[k], view, len, job.file_id, E, next, &hf);
        E = next;
        if (st.last_scan_run_leftover.size() * src_per_out + 4 <= E) break;
    }
    if (st.last_scan_run_leftover.size() * src_per_out + 4 > E && E < len) return SX_WAVE_FALLBACK;
    if (m.wave_family == 2 && E < len) {
        // UTF-16 (sx_wave_core.hpp): the device's masks assume units that begin on the buffer's even offsets, no window that ends in half a unit,
        // and no character kept from the call before (`pending_bmp`) where its windows begin
        const DDecoder& d0 = st.decoder.raw();
        if ((len & 1) || d0.lead_byte >= 0 || d0.pending_bmp) return SX_WAVE_FALLBACK;
    }
    for (sx_finding& f : hf.v) f.slice_index += job.slice_base;
    const uint64_t nfh = hf.v.size(), nbh = hf.arena.size();

    const uint64_t g_all = wv_window_count(len, W);
    const uint64_t g_lo = E < len ? wv_window_no(E, W, wps) : g_all;
    uint32_t final_state = 0;
    uint64_t n_waves = 0, nf_all = 0, nb_all = 0;
    bool deferred = false;
    std::vector<MissionFindings> segs;   // one per slab, in order
    // leaving early: no copy may still be writing into a block that goes back to the pool
    auto abandon = [&](int rc) -> int {
        if (ctx->merge_copy_stream) (void)hipStreamSynchronize(ctx->merge_copy_stream);
        (void)hipStreamSynchronize(sb);
        for (auto& g : segs) if (g.ext.p) ctx->pool->give(g.ext);
        return rc;
    };
    if (g_lo < g_all) {
        const uint32_t lc = utf8_chars(st.last_scan_run_leftover), lb = (uint32_t)st.last_scan_run_leftover.size();
        // source bytes from the leftover's first byte to E: single byte: one per char; UTF-8: its bytes + what the decoder holds of the next char
        const DDecoder& dd = st.decoder.raw();
        uint32_t lback = m.wave_family == 0 ? lc : lb + (dd.needed ? dd.seen + 1u : 0u);
        if (m.wave_family == 2) {   // UTF-16: two source bytes per character, four for those with four bytes of UTF-8; + a pending high surrogate
            lback = 0;
            for (unsigned char c : st.last_scan_run_leftover) if ((c & 0xC0) != 0x80) lback += c >= 0xF0 ? 4u : 2u;
            if (dd.lead_surrogate) lback += 2;
        }
        if (m.wave_family >= 4) {
            // two-byte family / EUC-JP: one to three source bytes per char — which, the text does not say: the bytes in front of E (less
            // the byte(s) of the token the decoder holds) that decode to exactly the leftover
            const uint32_t pend = dd.dlead ? (m.wave_family == 5 && dd.dflag ? 2u : 1u) : 0u;
            lback = 0;
            for (uint32_t cand = lc; lc && cand <= 3 * lc && cand + pend <= E; cand++) {
                size_t hint = 0;
                const uint8_t* src = view.span(E - pend - cand, cand, &hint);
                Decoder probe(m.c.encoding);
                uint8_t buf[64 * 4 + 16];
                const DecodeStep r = probe.decode_to_str_without_replacement(src, cand, buf, sizeof buf, false);
                if (r.result == DecoderResult::InputEmpty && r.read == cand && probe.idle() && r.written == lb
                    && memcmp(buf, st.last_scan_run_leftover.data(), lb) == 0) { lback = cand + pend; break; }
            }
            if (lc && !lback) return SX_WAVE_FALLBACK;
        }
        // (-g: does the leftover hold the grep char?  SplitStr walks its chars again when it is prepended, helper.rs:252-254)
        const uint32_t lg = m.c.grep_char >= 0 && st.last_scan_run_leftover.find((char)m.c.grep_char) != std::string::npos ? 1u : 0u;
        // (-r in the kernels: the lead byte of the leftover's last multi-byte character, as a code — helper.rs:279-296 meets it again first)
        uint32_t lm = 0;
        if (m.wave_same)
            for (unsigned char c : st.last_scan_run_leftover) if (c >= 0xC2) lm = wv_lead_code(m.c.ubf, c);
        const WvState in{ lc, lb, lc ? lback : 0u, st.last_run_str_was_printed_and_is_maybe_cut_str ? 1u : 0u, lc ? lg : 0u, lc ? lm : 0u };
        const uint64_t n_windows = g_all - g_lo;
        uint64_t batches = (n_windows + (8192ull * 64) - 1) / (8192ull * 64);
        batches = std::max<uint64_t>(1, std::min<uint64_t>(8, batches));
        if (m.wave_same) batches = 1;   // (-r in the kernels: a window costs 5 to 20 times the usual, one wavefront per SIMD does not hide it — Russian text 51 -> 26 ms per 256 MiB)
        if (const char* e = getenv("SX_WAVE_BATCHES")) batches = (uint64_t)std::max(1, std::min(64, atoi(e)));
        const uint32_t nwin = (uint32_t)(batches * kWvBatch - kWvWarm);
        n_waves = (n_windows + nwin - 1) / nwin;
        // Slabs of wavefronts: count -> write -> copy to the host, the copy of a slab next to the kernels of the following one.
        // (Several Missions: one slab; their outputs stay in HBM and are interleaved there, sx_stage_b.cpp device_merge.)
        uint64_t K = 1;
        if (ctx->missions.size() == 1 && defer_min_bytes == 0) K = std::min<uint64_t>(8, std::max<uint64_t>(1, len >> 25));   // >= 32 MiB of input each, eight at most (measured: 256 MiB of text in 4 / 8 / 16 slabs 7.9 / 7.2 / 8.0 ms; `-e ascii -n 4` on 1 GiB in 8 / 16 / 32: 5.15 / 5.27 / 7.0 ms)
        if (const char* e = getenv("SX_WAVE_SLABS")) K = (uint64_t)std::max(1, std::min(64, atoi(e)));
        if (ctx->missions.size() != 1 || defer_min_bytes != 0) K = 1;
        // SX_OPT_RESULT_ON_DEVICE (round 5): one Mission, no pieces — the result stays where the writer put it (one slab: one block of the context's)
        const bool keep_dev = (ctx->opt.flags & SX_OPT_RESULT_ON_DEVICE) && ctx->missions.size() == 1 && defer_min_bytes == 0 && job.commit_state && !ctx->sharded_call && ctx->single_piece;
        if (keep_dev) K = 1;
        K = std::min<uint64_t>(K, n_waves);

        if (!d.d_wave_lut) {
            HIP_TRY(ctx, hipMalloc((void**)&d.d_wave_lut, m.wave_lut.size()));   // (256 bytes; UTF-16: 512)
            HIP_TRY(ctx, hipMemcpy(d.d_wave_lut, m.wave_lut.data(), m.wave_lut.size(), hipMemcpyHostToDevice));
        }
        if (m.wave_family == 4 && !d.d_wave_pairs) {   // (family 5 has no 4-bit table)
            HIP_TRY(ctx, hipMalloc((void**)&d.d_wave_pairs, 8192 * 4));
            HIP_TRY(ctx, hipMemcpy(d.d_wave_pairs, m.wave_pairs.data(), 8192 * 4, hipMemcpyHostToDevice));
        }
        if (m.wave_family >= 4 && m.wave_pairs2.size() == 4096 && !d.d_wave_pairs2) {
            HIP_TRY(ctx, hipMalloc((void**)&d.d_wave_pairs2, 4096 * 4));
            HIP_TRY(ctx, hipMemcpy(d.d_wave_pairs2, m.wave_pairs2.data(), 4096 * 4, hipMemcpyHostToDevice));
        }
        // per wavefront: 4 x u32 (pass 1 out) + 2 x u64 (offsets); + totals per slab
        const uint64_t per = 5 * 4 + 2 * 8;   // (+ the two-byte family's grid word)
        int rc = ensure_rp(ctx, d, 1, n_waves * per + 4096); if (rc) return rc;
        uint8_t* w_scratch; uint64_t w_scratch_cap;
        if (own_stream) {
            rc = ensure_rp(ctx, d, 10, wave_scratch_bytes(n_waves)); if (rc) return rc;
            w_scratch = (uint8_t*)d.d_rp[10]; w_scratch_cap = d.d_rp_cap[10];
            if (!d.h_tot) HIP_TRY(ctx, hipHostMalloc((void**)&d.h_tot, 4096, hipHostMallocNonCoherent));
        } else {
            rc = ensure_scratch(ctx, wave_scratch_bytes(n_waves)); if (rc) return rc;
            rc = ensure_pinned2(ctx, 4096); if (rc) return rc;
            w_scratch = ctx->d_scratch; w_scratch_cap = ctx->d_scratch_cap;
        }
        uint8_t* base = (uint8_t*)d.d_rp[1];
        uint64_t* d_tot = (uint64_t*)base;            // 4 x u64 per slab
        uint64_t* d_fb = (uint64_t*)(base + 2048);
        uint64_t* d_ab = d_fb + n_waves;
        uint32_t* d_u = (uint32_t*)(d_ab + n_waves);
        WaveParams P{};
        P.data = job.d_bytes; P.len = len; P.consumed0 = job.consumed0[k]; P.slice_base = job.slice_base;
        P.W = W; P.wps = wps; P.q = (uint32_t)m.q; P.n_min = m.c.chars_min_nb;
        P.g_lo = g_lo; P.g_hi = g_all; P.nwin = nwin; P.inject = wv_pack(in);
        P.mission_id = m.c.mission_id; P.file_id = job.file_id; P.family = m.wave_family; P.lut = d.d_wave_lut; P.table = d.d_table;
        P.pairs = d.d_wave_pairs; P.encoding = m.c.encoding; P.entry_skip = m.buf_entry_skip;
        P.swar = m.wave_swar; P.pairs2 = d.d_wave_pairs2;
        if (m.wave_family == 4 && !d.d_wave_pairs2) P.swar.cls = 0;
        if (m.wave_family == 5 && !d.d_wave_pairs2) return SX_WAVE_FALLBACK;
        if (const char* e = getenv("SX_WAVE_LUT")) if (atoi(e) && m.wave_family != 5) P.swar.cls = 0;   // tests: the class table also where ranges would do
        P.wave_nf = d_u; P.wave_nb = d_u + n_waves; P.wave_in = d_u + 2 * n_waves; P.wave_out = d_u + 3 * n_waves;
        P.wave_grid = m.wave_family == 4 || m.wave_family == 5 ? d_u + 4 * n_waves : nullptr;   // (EUC-JP since round 5: fills without 8E / 8F)
        // -r on a UTF-8 Mission: the kernels collect the lead bytes that pass ubf; two kinds (the leftover's included) and the buffer goes back
        uint64_t* d_leads = nullptr;
        uint64_t left_leads = 0;
        if (m.wave_lead_check) {
            d_leads = (uint64_t*)(((uintptr_t)(d_u + 5 * n_waves) + 7) & ~(uintptr_t)7);   // (inside the 4096 spare bytes of d_rp[1])
            HIP_TRY(ctx, hipMemsetAsync(d_leads, 0, 8, sb));
            for (unsigned char c : st.last_scan_run_leftover) if (c >= 0xC2 && c <= 0xF4 && ((m.c.ubf >> (c & 0x3F)) & 1)) left_leads |= 1ull << (c & 0x3F);
            if (__builtin_popcountll(left_leads) > 1) return SX_WAVE_FALLBACK;
        }
        P.lead_set = d_leads; P.ubf = m.c.ubf;
        P.grep_char = m.c.grep_char;
        P.same = m.wave_same ? 1u : 0u;
        P.wave_fbase = d_fb; P.wave_abase = d_ab;
        // Descriptors for the lane-per-finding writer: room for twice the findings a wavefront is expected to hold (the last buffer's
        // density; stage A's record count; else one per window), at most two per window and a third of the input's size in all.  A
        // wavefront that finds more is only counted, and the launch takes the window-parallel writer.
        P.desc = nullptr; P.desc_cap = 0;
        if (nwin <= kWvDescMaxWin && !(getenv("SX_WAVE_DESC") && !atoi(getenv("SX_WAVE_DESC")))) {
            if (ctx->wave_density.size() != ctx->missions.size()) ctx->wave_density.assign(ctx->missions.size(), 0.0);
            double per_byte = ctx->wave_density[k];
            if (per_byte <= 0 && k < ctx->last_runs.size() && ctx->last_runs[k] && ctx->last_runs[k] < len / 16) per_byte = (double)ctx->last_runs[k] / (double)len;
            const double bytes_per_wave = (double)len / (double)n_waves;
            uint64_t cap = per_byte > 0 ? (uint64_t)(2.0 * per_byte * bytes_per_wave) + 128 : (uint64_t)nwin + 64;
            // (-r in the kernels: text whose lead bytes change every few characters holds 5 to 10 findings per window, and the writer that
            // works a lane per window pays for every one of them in turn — 1.5 / 3.5 ms per 32 MiB against 0.8 / 1.4 for the count pass:
            // room for 16 per window, descriptors up to the input's size)
            const uint64_t per_win = m.wave_same ? 16 : 2, share = m.wave_same ? 1 : 3;
            cap = std::min<uint64_t>(cap, per_win * nwin + 64);
            cap = std::min<uint64_t>(cap, std::max<uint64_t>(64, len / share / 12 / n_waves));
            // (more findings expected than descriptors may be kept: the count pass would leave them for nothing — the window-parallel writer at once)
            // (clearly more: `-e ascii -n 4` on random bytes expects 1.0 to 1.15 times the room and its wavefronts mostly fit — with the lane-per-finding
            // writer 12.4 -> 6 ms per GiB; Russian text with -r: 1.65 times)
            const bool too_dense = per_byte > 0 && per_byte * bytes_per_wave > 1.4 * (double)(per_win * nwin + 64) && !getenv("SX_WAVE_DESC_CAP");
            if (const char* e = getenv("SX_WAVE_DESC_CAP")) cap = (uint64_t)std::max(1, atoi(e));
            if (too_dense) cap = 0;
            if (cap == 0) { }
            else if (ensure_rp(ctx, d, 2, n_waves * cap * 12 + 64) == SX_OK) { P.desc = (uint32_t*)d.d_rp[2]; P.desc_cap = (uint32_t)cap; }
            else ctx->set_err(std::string());   // (no room: the other writer)
        }
        if (K > 1) {
            { const int rc = ensure_copy_stream(ctx); if (rc != SX_OK) return rc; }
        }
        while (d.wave_ev.size() < 4 * K) {
            hipEvent_t e;
            HIP_TRY(ctx, hipEventCreate(&e));
            d.wave_ev.push_back(e);
        }
        // A single Mission's slabs go to the host as they are written: as sx_finding16 (include/stringsext_amd.h), half the bytes of what
        // bounds this path.  (Several Missions: the outputs are interleaved first, 32-byte records; device_merge packs then.)
        const bool pack = ctx->missions.size() == 1 && defer_min_bytes == 0 && !(getenv("SX_PACKED") && !atoi(getenv("SX_PACKED")));
        const size_t rec = pack ? sizeof(sx_finding16) : sizeof(sx_finding);
        std::shared_ptr<SegInfo> seg_info;
        std::vector<sx_finding16> hf16;
        if (pack) {
            seg_info = std::make_shared<SegInfo>();
            seg_info->file_id = job.file_id; seg_info->slice_base = job.slice_base;
            for (auto& x : seg_info->pos0) x = 0;
            seg_info->pos0[m.c.mission_id] = job.consumed0[k];
            hf16.reserve(hf.v.size());
            for (const sx_finding& f : hf.v) hf16.push_back(pack_finding(f));
        }
        P.packed = pack ? 1u : 0u;
        std::vector<char> wrote(K, 0);
        uint64_t* h_tot = own_stream ? d.h_tot : (uint64_t*)ctx->h_pin2;
        hipEvent_t copied[2] = { ctx->merge_ev[1], ctx->merge_ev[2] };
        bool pending[2] = { false, false };
        for (uint64_t j = 0; j < K; j++) {
            const uint64_t v0 = n_waves * j / K, v1 = n_waves * (j + 1) / K;
            if (getenv("SX_TIMING2")) fprintf(stderr, "[sx]   wave mission %zu slab %llu: waves [%llu, %llu) of %llu, E %llu, count...\n", k, (unsigned long long)j, (unsigned long long)v0, (unsigned long long)v1, (unsigned long long)n_waves, (unsigned long long)E);
            HIP_TRY(ctx, hipEventRecord(d.wave_ev[4 * j], sb));
            HIP_TRY(ctx, launch_wave_count(P, v0, v1, d_fb, d_ab, d_tot + 4 * j, w_scratch, w_scratch_cap, sb));
            HIP_TRY(ctx, hipEventRecord(d.wave_ev[4 * j + 1], sb));
            HIP_TRY(ctx, hipMemcpyAsync(h_tot + 4 * j, d_tot + 4 * j, 4 * 8, hipMemcpyDeviceToHost, sb));
            if (d_leads) HIP_TRY(ctx, hipMemcpyAsync(h_tot + 4 * K, d_leads, 8, hipMemcpyDeviceToHost, sb));   // (pinned: 4096 bytes, K <= 64 slabs use 2048)
            HIP_TRY(ctx, hipStreamSynchronize(sb));
            if (d_leads && __builtin_popcountll(h_tot[4 * K] | left_leads) > 1) {
                if (getenv("SX_TIMING")) fprintf(stderr, "[sx] wave replay mission %zu: -r and two kinds of lead bytes in this buffer: lane-per-region path\n", k);
                return abandon(SX_WAVE_FALLBACK);
            }
            // Round 5: wavefronts whose warm-up windows led them to a wrong entry state run again from what their predecessor really left
            // (WaveParams::redo) — with -g a stretch without the grep char hands on "q chars carried" and "nothing" in turns, as far back as
            // it began — until the verification passes; a chain of wrong wavefronts takes a launch per link.  (Before: any wrong
            // assumption sent the whole buffer to the lane-per-region path.)
            bool repaired = false;
            if ((h_tot[4 * j + 2] & 0xFFFFFFFFull) != 0 && (h_tot[4 * j + 3] >> 32) == 0 && !getenv("SX_WAVE_FAIL") && !(getenv("SX_WAVE_REPAIR") && !atoi(getenv("SX_WAVE_REPAIR")))) {
                const int max_rounds = getenv("SX_WAVE_REPAIR") ? std::max(1, atoi(getenv("SX_WAVE_REPAIR"))) : 48;
                P.redo = 1;
                for (int round = 0; round < max_rounds && (h_tot[4 * j + 2] & 0xFFFFFFFFull) != 0; round++) {
                    HIP_TRY(ctx, launch_wave_count(P, v0, v1, d_fb, d_ab, d_tot + 4 * j, w_scratch, w_scratch_cap, sb));
                    HIP_TRY(ctx, hipMemcpyAsync(h_tot + 4 * j, d_tot + 4 * j, 4 * 8, hipMemcpyDeviceToHost, sb));
                    HIP_TRY(ctx, hipStreamSynchronize(sb));
                    std::lock_guard<std::mutex> g(ctx->mu);
                    ctx->stats.wave_repairs++;
                }
                P.redo = 0;
                repaired = true;
                HIP_TRY(ctx, hipEventRecord(d.wave_ev[4 * j + 1], sb));   // (the count pass' time includes its repairs)
            }
            if ((h_tot[4 * j + 2] & 0xFFFFFFFFull) != 0 || getenv("SX_WAVE_FAIL")) {   // (SX_WAVE_FAIL: tests of the way back)
                if (getenv("SX_TIMING")) fprintf(stderr, "[sx] wave replay mission %zu: %llu wavefronts assumed a wrong entry state: lane-per-region path\n", k, (unsigned long long)(h_tot[4 * j + 2] & 0xFFFFFFFFull));
                return abandon(SX_WAVE_FALLBACK);
            }
            const bool by_desc = P.desc && (h_tot[4 * j + 2] >> 32) == 0;   // every wavefront of the slab left all its descriptors
            // (after repairs the window-parallel writer must start every wavefront from the state its predecessor left, not from its own
            // warm-up windows; the two-byte family's token grid is published per launch geometry: that combination goes back)
            P.use_entry = 0;
            if (repaired && !by_desc) {
                // (UTF-16 too, round 5: a wavefront that starts its batch at its own first window can meet a case its masks cannot say — a
                // unit pair across the batch's first tile edge — which the count pass, with its warm-up windows in front, did not; found by
                // tools/wave_fuzz.py after 20 minutes)
                if (m.wave_family >= 2) return abandon(SX_WAVE_FALLBACK);
                P.use_entry = 1;
            }
            if (P.desc && !by_desc) { std::lock_guard<std::mutex> g(ctx->mu); ctx->stats.wave_desc_overflows++; }
            const uint64_t nf = h_tot[4 * j], nb = h_tot[4 * j + 1];
            if (getenv("SX_TIMING2")) fprintf(stderr, "[sx]   ... counted %llu findings, %llu bytes at +%.2f ms\n", (unsigned long long)nf, (unsigned long long)nb, now_ms() - t0);
            final_state = (uint32_t)h_tot[4 * j + 3];
            const uint64_t nfh_j = j == 0 ? nfh : 0, nbh_j = j == 0 ? nbh : 0;   // the host's entry windows go in front of the first slab
            if (nb + nbh_j > 0xFFFFFFFFull) { ctx->set_err("more than 4 GiB of strings in one chunk"); return abandon(SX_E_NOMEM); }
            nf_all += nf; nb_all += nb;
            if (nf + nfh_j == 0) continue;
            // ---- pass 2 straight into the result's layout: [host findings][device findings][host strings][device strings]
            const uint64_t out_bytes = (nfh_j + nf) * rec + nbh_j + nb;
            const int slot = (j & 1) ? 8 : 5;
            if (pending[j & 1]) {   // the copy of slab j - 2 may still read this buffer
                if (d.d_rp_cap[slot] < out_bytes + 64) HIP_TRY(ctx, hipEventSynchronize(copied[j & 1]));
                else HIP_TRY(ctx, hipStreamWaitEvent(sb, copied[j & 1], 0));
            }
            rc = ensure_rp(ctx, d, slot, out_bytes + 64); if (rc) return abandon(rc);
            uint8_t* d_all = (uint8_t*)d.d_rp[slot];
            if (nf) {
                P.findings = (sx_finding*)(d_all + nfh_j * rec);
                P.arena = d_all + (nfh_j + nf) * rec + nbh_j;
                P.str_off_base = (uint32_t)nbh_j; P.f_sub = 0; P.a_sub = 0;
                // (pieces: the interleave of the piece before may still read this Mission's findings — on post_stream, not this stream)
                if (own_stream && ctx->interleave_pending) HIP_TRY(ctx, hipStreamWaitEvent(sb, ctx->ev_interleaved, 0));
                HIP_TRY(ctx, hipEventRecord(d.wave_ev[4 * j + 2], sb));
                HIP_TRY(ctx, by_desc ? launch_wave_emit(P, v0, v1, sb) : launch_wave_write(P, v0, v1, sb));
                HIP_TRY(ctx, hipEventRecord(d.wave_ev[4 * j + 3], sb));
                wrote[j] = 1;
            }
            if (nfh_j) {
                HIP_TRY(ctx, hipMemcpyAsync(d_all, pack ? (const void*)hf16.data() : (const void*)hf.v.data(), nfh_j * rec, hipMemcpyHostToDevice, sb));
                if (nbh_j) HIP_TRY(ctx, hipMemcpyAsync(d_all + (nfh_j + nf) * rec, hf.arena.data(), nbh_j, hipMemcpyHostToDevice, sb));
            }
            MissionFindings seg;
            if (pack) { seg.packed = true; seg.info = seg_info; }
            deferred = K == 1 && defer_min_bytes && nf * sizeof(sx_finding) + nb >= defer_min_bytes;
            if (keep_dev) {
                HIP_TRY(ctx, hipStreamSynchronize(sb));
                seg.dev_only = true; seg.keep_on_device = true; seg.dev_epoch_ref = ctx->dev_epoch; seg.dev_epoch = ctx->dev_epoch->load();
                seg.ext_nf = nfh_j + nf; seg.ext_na = nbh_j + nb; seg.dev_copy = d_all;
            } else if (deferred) {
                HIP_TRY(ctx, hipStreamSynchronize(sb));
                if (getenv("SX_TIMING2")) fprintf(stderr, "[sx]   ... written (left on the device) at +%.2f ms\n", now_ms() - t0);
                seg.dev_only = true; seg.ext_nf = nfh_j + nf; seg.ext_na = nbh_j + nb; seg.dev_copy = d_all;
            } else {
                PinnedPool::Block blk = ctx->pool->take(out_bytes + 64);
                if (!blk.p) { ctx->set_err("hipHostMalloc failed"); return abandon(SX_E_NOMEM); }
                seg.ext = blk; seg.ext_nf = nfh_j + nf; seg.ext_na = nbh_j + nb;
                if (K > 1) {   // on the copy stream: the next slab's kernels run meanwhile
                    segs.push_back(std::move(seg));
                    HIP_TRY(ctx, hipEventRecord(ctx->merge_ev[0], sb));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->merge_copy_stream, ctx->merge_ev[0], 0));
                    HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_all, out_bytes, hipMemcpyDeviceToHost, ctx->merge_copy_stream));
                    HIP_TRY(ctx, hipEventRecord(copied[j & 1], ctx->merge_copy_stream));
                    pending[j & 1] = true;
                    if (nfh_j) HIP_TRY(ctx, hipStreamSynchronize(sb));   // (the entry part's upload reads host vectors)
                    continue;
                }
                HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_all, out_bytes, hipMemcpyDeviceToHost, sb));
                HIP_TRY(ctx, hipStreamSynchronize(sb));
                seg.dev_copy = d_all;
            }
            segs.push_back(std::move(seg));
        }
        if (K > 1) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->merge_copy_stream));
            HIP_TRY(ctx, hipStreamSynchronize(sb));
        }
        {
            std::lock_guard<std::mutex> g(ctx->mu);
            for (uint64_t j = 0; j < K; j++) {
                float ms = 0;
                if (hipEventElapsedTime(&ms, d.wave_ev[4 * j], d.wave_ev[4 * j + 1]) == hipSuccess) ctx->stats.wave_count_ms += ms;
                if (wrote[j] && hipEventElapsedTime(&ms, d.wave_ev[4 * j + 2], d.wave_ev[4 * j + 3]) == hipSuccess) ctx->stats.wave_write_ms += ms;
            }
        }
        (void)hipGetLastError();
    } else if (nfh) {   // the host's windows were the whole buffer
        out->v = std::move(hf.v); out->arena = std::move(hf.arena);
    }
    if (!segs.empty()) {
        const uint64_t rb = out->replay_bytes;
        *out = std::move(segs[0]);
        for (size_t j = 1; j < segs.size(); j++) out->more.push_back(std::move(segs[j]));
        out->replay_bytes = rb;
    } else if (g_lo < g_all && nfh) { out->v = std::move(hf.v); out->arena = std::move(hf.arena); }
    out->replay_bytes += len;
    { std::lock_guard<std::mutex> g(ctx->mu); ctx->stats.wave_windows += g_all - g_lo; }
    // the next whole buffer of this Mission does without stage A (sx_schedule.cpp) as long as this one was string-dense
    if (k < ctx->wave_pred.size()) ctx->wave_pred[k] = (nf_all + nfh) * wave_min_density_bytes(m.wave_family) * 2 > len ? 1 : 0;
    if (k < ctx->wave_density.size() && len) ctx->wave_density[k] = (double)(nf_all + nfh) / (double)len;   // sizes the next buffer's descriptors
    const double t1 = now_ms();

    // ---- the state handed to the next buffer
    if (job.commit_state) {
        ScannerState fin = st;   // (the host's own exit state if the device had nothing to do)
        if (g_lo < g_all) {
            const WvState fs = wv_unpack(final_state);
            // decode the bytes from the leftover's first one to the buffer end once more: the leftover's text and what the decoder
            // holds of a character that is still incomplete (without leftover: the last bytes, for the decoder alone)
            fin.decoder.reset(m.c.encoding);
            fin.last_scan_run_leftover.clear();
            // (two-byte family without leftover: a fresh decoder cannot find the token grid in the last bytes; the device says whether
            // the buffer ends inside a token — its last byte is then the lead byte the decoder holds)
            const bool dbcs_tail = m.wave_family >= 4 && !fs.lc;
            if (const uint32_t pend = (final_state >> 27) & 3u; dbcs_tail && pend) {
                size_t hint1 = 0;
                fin.decoder.raw().dlead = *view.span(len - 1, 1, &hint1);
                if (pend == 2) fin.decoder.raw().dflag = 1;   // EUC-JP: 8F and the byte behind it are in (sx_codec_core.hpp ddec_eucjp)
            }
            const uint64_t back = dbcs_tail ? 0 : fs.lc ? fs.lback : std::min<uint64_t>(8, len);
            size_t hint = 0;
            const uint8_t* src = back ? view.span(len - back, back, &hint) : (const uint8_t*)"";
            uint8_t buf[64 * 4 + 16];
            size_t at = 0, written = 0;
            for (; back;) {
                const DecodeStep r = fin.decoder.decode_to_str_without_replacement(src + at, back - at, buf + written, sizeof buf - written, false);
                at += r.read; written += r.written;
                if (r.result != DecoderResult::Malformed) break;
                written = 0;   // (only without leftover: the bytes of a leftover decode without error)
            }
            if (fs.lc) {
                if (written != fs.lb) { ctx->set_err("wave replay: the exit leftover does not decode to what the device counted"); return SX_E_STATE; }
                fin.last_scan_run_leftover.assign((const char*)buf, written);
            }
            fin.last_run_str_was_printed_and_is_maybe_cut_str = fs.cut != 0;
        }
        fin.consumed_bytes = job.consumed0[k] + len;
        fin.stream_bytes = job.stream0[k] + len;
        ctx->states[k] = fin;
    }
    if (end_pos) *end_pos = len;
    if (getenv("SX_TIMING"))
        fprintf(stderr, "[sx] wave replay mission %zu: %llu windows in %llu wavefronts, %zu slab(s): entry + count + write + d2h %.2f ms (%llu findings, %llu string bytes%s), state %.2f ms\n",
                k, (unsigned long long)(g_all - g_lo), (unsigned long long)n_waves, segs.size(), t1 - t0, (unsigned long long)(nf_all + nfh),
                (unsigned long long)(nb_all + nbh), deferred ? ", left on the device" : "", now_ms() - t1);
    return SX_OK;
}

}  // namespace sx
