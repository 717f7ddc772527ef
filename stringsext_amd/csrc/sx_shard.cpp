// sx_shard.cpp — byte-range sharding of one input file over the ranks of a job (one process per GPU), behind
// the C-ABI: the "where did you stop" exchange, the repeat rule, the widening of halos, and the splice of the
// ranks' findings into the reference's order.  The transport is the caller's: an all-gather callback (RCCL /
// torch.distributed / MPI / anything) — the library never opens a connection.  The reference has nothing
// comparable (one thread per Mission over one sequential stream, src/main.rs:97-168); the splice reproduces
// what that stream would have printed.
#include "sx_ctx.hpp"

using namespace sx;

extern "C" {

void sx_shard_bounds(uint64_t file_len, int world, int rank, uint64_t* own_lo, uint64_t* own_hi) {
    // contiguous, on the 4096-byte slice grid (src/input.rs:22)
    const uint64_t per = (file_len / (uint64_t)world + kInputBufLen - 1) / kInputBufLen * kInputBufLen;
    const uint64_t lo = std::min<uint64_t>(file_len, (uint64_t)rank * per);
    const uint64_t hi = rank == world - 1 ? file_len : std::min<uint64_t>(file_len, (uint64_t)(rank + 1) * per);
    if (own_lo) *own_lo = lo;
    if (own_hi) *own_hi = hi;
}

int sx_scan_sharded(sx_ctx* ctx, int rank, int world, uint64_t file_len, uint64_t file_stream_off, int input_file_id,
                    uint64_t halo, sx_shard_buffer_fn get_buffer, void* buffer_user, sx_shard_runs_fn get_runs, void* runs_user,
                    sx_allgather_fn allgather, void* allgather_user, sx_result** out, uint64_t* counts, uint64_t* overflow) {
    if (!ctx || !get_buffer || !allgather || !out || world < 1 || rank < 0 || rank >= world) return SX_E_INVALID;
    const size_t nm = ctx->missions.size();
    // What no rank can do is refused by EVERY rank, before anybody waits for anybody (shard_common would refuse it only on the
    // ranks whose buffer does not start the file: rank 0 would then wait in the all-gather for ever — ADVICE, round 2).
    if (world > 1)
        for (const Mission& m : ctx->missions) {
            if (m.host_sequential()) { ctx->err = "an ISO-2022-JP mission cannot be sharded (its stage B is one sequential pass)"; return SX_E_INVALID; }
            if (m.c.chars_min_nb == 0) { ctx->err = "a mission with chars_min_nb 0 cannot be sharded (its stage B is one sequential pass)"; return SX_E_INVALID; }
        }
    uint64_t own_lo, own_hi;
    sx_shard_bounds(file_len, world, rank, &own_lo, &own_hi);
    if (halo == 0) halo = 1u << 20;

    sx_result* res = nullptr;
    std::vector<uint64_t> ends(nm, 0), start(nm, own_lo);
    uint64_t h = halo;
    // one attempt: this rank's range plus the halo; truncated = the buffer was too short (a run crosses the whole
    // halo behind the range, or a Big5 / EUC-JP mission found no token boundary in the halo in front of it)
    auto attempt = [&](const uint64_t* start_at, int reuse, bool* truncated) -> int {
        const uint64_t buf_lo = (own_lo > h ? own_lo - h : 0) / kInputBufLen * kInputBufLen;
        const uint64_t buf_hi = std::min(file_len, own_hi + h);
        const void* ptr = nullptr;
        int is_device = 0;
        int rc = get_buffer(buffer_user, buf_lo, buf_hi, &ptr, &is_device);
        if (rc != 0) { ctx->err = "the buffer callback failed"; return SX_E_INVALID; }
        if (res) { sx_result_free(res); res = nullptr; }
        if (get_runs) {
            const sx_run* const* runs = nullptr;
            const uint64_t* n_runs = nullptr;
            if (get_runs(runs_user, (const uint8_t*)ptr, buf_lo, buf_hi - buf_lo, &runs, &n_runs) != 0) { ctx->err = "the runs callback failed"; return SX_E_INVALID; }
            rc = sx_replay_shard_runs(ctx, (const uint8_t*)ptr, buf_lo, buf_hi - buf_lo, own_lo, own_hi, start_at, file_stream_off,
                                      input_file_id, runs, n_runs, &res, ends.data());
        } else if (is_device)
            rc = sx_scan_shard_device(ctx, ptr, buf_lo, buf_hi - buf_lo, own_lo, own_hi, start_at, file_stream_off, input_file_id,
                                      reuse, &res, ends.data());
        else
            rc = sx_scan_shard(ctx, (const uint8_t*)ptr, buf_lo, buf_hi - buf_lo, own_lo, own_hi, start_at, file_stream_off,
                               input_file_id, reuse, &res, ends.data());
        if (rc == SX_E_HALO && buf_lo > 0) { *truncated = true; return SX_OK; }
        if (rc != SX_OK) return rc;
        *truncated = false;
        if (buf_hi < file_len) for (uint64_t e : ends) if (e >= buf_hi) *truncated = true;
        return SX_OK;
    };
    auto attempt_until_it_fits = [&](const uint64_t* start_at, int reuse) -> int {
        for (;;) {
            bool truncated = false;
            int rc = attempt(start_at, reuse, &truncated);
            if (rc != SX_OK) return rc;
            if (!truncated) return SX_OK;
            h *= 8;
            reuse = 0;
        }
    };
    // first attempt: everybody assumes the previous rank stops at the shard boundary
    int rc = attempt_until_it_fits(nullptr, 0);

    // Where did everybody stop?  One all-gather of (start used, end reached) per mission + the finding count + this rank's status;
    // rank k repeats its replay if rank k-1 ran past the point rank k started from (a region across the shard boundary).
    // Every rank evaluates the same table, so all agree on who repeats; a repeat can move that rank's own end: loop.
    // A rank whose own work failed (no memory, a HIP error, its buffer callback) still joins the exchange, with its error code in
    // the row: every rank then leaves with an error in the same round, and nobody waits for a rank that is gone.
    const size_t row = 2 * nm + 2;
    std::vector<uint64_t> mine(row), table((size_t)world * row);
    for (;;) {
        for (size_t m = 0; m < nm; m++) { mine[m] = start[m]; mine[nm + m] = ends[m]; }
        mine[2 * nm] = res ? sx_result_count(res) : 0;
        mine[2 * nm + 1] = (uint64_t)(int64_t)rc;
        if (allgather(allgather_user, mine.data(), row * 8, table.data()) != 0) { ctx->err = "the all-gather callback failed"; if (res) sx_result_free(res); return SX_E_INVALID; }
        for (int k = 0; k < world; k++) {
            const int krc = (int)(int64_t)table[(size_t)k * row + 2 * nm + 1];
            if (krc == SX_OK) continue;
            if (res) sx_result_free(res);
            if (rc != SX_OK) return rc;   // this rank's own failure: its message stands
            ctx->err = "rank " + std::to_string(k) + " of the sharded scan failed (error " + std::to_string(krc) + "): nothing of this file's scan can be used";
            return SX_E_STATE;
        }
        std::vector<char> redo((size_t)world, 0);
        bool any = false;
        for (int k = 1; k < world; k++) {
            uint64_t k_lo;
            sx_shard_bounds(file_len, world, k, &k_lo, nullptr);
            for (size_t m = 0; m < nm; m++)
                if (std::max(k_lo, table[(size_t)(k - 1) * row + nm + m]) > table[(size_t)k * row + m]) redo[(size_t)k] = 1;
            any = any || redo[(size_t)k];
        }
        if (!any) break;
        if (redo[(size_t)rank]) {
            std::vector<uint64_t> prev_end(nm);
            for (size_t m = 0; m < nm; m++) {
                prev_end[m] = table[(size_t)(rank - 1) * row + nm + m];
                start[m] = std::max(std::max(own_lo, prev_end[m]), start[m]);
            }
            rc = attempt_until_it_fits(start.data(), 1);   // (a failure travels in the next round's row)
            if (rc == SX_OK) for (size_t m = 0; m < nm; m++) ends[m] = std::max(ends[m], prev_end[m]);
        }
    }
    // A region that crosses the shard boundary is finished by the rank it began on: the tail of rank k's findings can
    // lie in slices of rank k+1, where it interleaves with that rank's findings of other Missions (the reference
    // prints slice by slice).  How many such findings each rank holds goes along; sx_shard_splice puts them in place.
    uint64_t over = 0;
    if (rank + 1 < world) {
        const uint32_t boundary = (uint32_t)(own_hi / kInputBufLen);
        for (size_t s = res->r.segs.size(); s-- > 0;) {
            const MissionFindings& seg = res->r.segs[s];
            size_t i = seg.count();
            while (i > 0 && seg.data()[i - 1].slice_index >= boundary) i--;
            over += seg.count() - i;
            if (i > 0) break;
        }
    }
    std::vector<uint64_t> overs((size_t)world, 0);
    if (allgather(allgather_user, &over, 8, overs.data()) != 0) { ctx->err = "the all-gather callback failed"; sx_result_free(res); return SX_E_INVALID; }
    for (int k = 0; k < world; k++) {
        if (counts) counts[k] = table[(size_t)k * row + 2 * nm];
        if (overflow) overflow[k] = overs[(size_t)k];
    }
    // The state at the file's end — decoder, leftover, cut flag: what the reference carries into the next file (src/main.rs:153-168,
    // src/input.rs:116-132: one ScannerState per Mission for the whole stream) — is known to the LAST rank only (its range ends the
    // file: shard_common commits it).  It goes to every rank, so that the next file's first shard (rank 0, buffer offset 0) starts
    // from it: a multi-file stream scanned file by file with sx_scan_sharded gives what one process would.
    if (world > 1) {
        std::vector<size_t> off(nm + 1, 0);
        for (size_t m = 0; m < nm; m++) off[m + 1] = off[m] + ((sizeof(DDecoder) + 24 + 4 * ctx->missions[m].q + 8 + 7) & ~(size_t)7);
        std::vector<uint8_t> blob(off[nm], 0), all((size_t)world * off[nm], 0);
        for (size_t m = 0; m < nm; m++) {
            uint8_t* b = blob.data() + off[m];
            ScannerState& st = ctx->states[m];
            DDecoder dd = st.decoder.raw();
            dd.table = nullptr;   // (a pointer of this process: the receiver puts its own back)
            memcpy(b, &dd, sizeof dd);
            uint64_t head[3] = { st.last_run_str_was_printed_and_is_maybe_cut_str ? 1ull : 0ull, st.last_scan_run_leftover.size(), st.consumed_bytes };
            memcpy(b + sizeof dd, head, sizeof head);
            memcpy(b + sizeof dd + sizeof head, st.last_scan_run_leftover.data(), std::min<size_t>(st.last_scan_run_leftover.size(), 4 * ctx->missions[m].q + 8));
        }
        if (allgather(allgather_user, blob.data(), blob.size(), all.data()) != 0) { ctx->err = "the all-gather callback failed"; sx_result_free(res); return SX_E_INVALID; }
        const uint8_t* last = all.data() + (size_t)(world - 1) * off[nm];
        for (size_t m = 0; m < nm; m++) {
            const uint8_t* b = last + off[m];
            ScannerState& st = ctx->states[m];
            DDecoder dd;
            memcpy(&dd, b, sizeof dd);
            dd.table = st.decoder.raw().table;
            st.decoder.raw() = dd;
            uint64_t head[3];
            memcpy(head, b + sizeof dd, sizeof head);
            st.last_run_str_was_printed_and_is_maybe_cut_str = head[0] != 0;
            st.last_scan_run_leftover.assign((const char*)(b + sizeof dd + sizeof head), (size_t)std::min<uint64_t>(head[1], 4 * ctx->missions[m].q + 8));
            st.consumed_bytes = head[2];
            st.stream_bytes = file_stream_off + file_len;
        }
    }
    *out = res;
    return SX_OK;
}

// The ranks' findings (rank k: findings[k][0..n_findings[k]) with strings in arenas[k]) -> one result in the
// reference's order: a rank's findings that lie behind its range end are merged into the head of the next rank's
// (slice, position, then Mission: the library's own merge key; both sides are sorted already).
// The general form (round 5): rank k's findings arrive in n_segs_of_rank[k] segments in a row — a rank with more than 4 GiB of strings ships
// its result segment by segment, every segment with its own str_off space (BASELINE config 5 at 8 x 32 GiB: 2.4 GB of strings per GiB-eighth)
// — and the result has as many segments as its strings need (each < 2 GiB), cut between findings.  seg s = (findings[s], n_findings[s],
// arenas[s], arena_lens[s]); the segments of rank 0 come first, then rank 1's, ...
static int splice_segs(const sx_finding* const* findings, const uint64_t* n_findings, const uint8_t* const* arenas,
                       const uint64_t* arena_lens, const uint32_t* n_segs_of_rank, int world, uint64_t file_len, uint64_t seg_cap, sx_result** out) {
    if (!findings || !n_findings || !arenas || !arena_lens || !n_segs_of_rank || !out || world < 1) return SX_E_INVALID;
    struct Ref { uint32_t seg; int rank; const sx_finding* f; };
    auto less_eq = [](const Ref& a, const Ref& b) {   // a goes first on a tie if it is of the lower Mission (then the earlier rank)
        if (a.f->slice_index != b.f->slice_index) return a.f->slice_index < b.f->slice_index;
        if (a.f->position != b.f->position) return a.f->position < b.f->position;
        return a.f->mission_id <= b.f->mission_id;
    };
    std::vector<Ref> order, carry;
    uint64_t total = 0;
    uint32_t n_all = 0;
    for (int k = 0; k < world; k++) n_all += n_segs_of_rank[k];
    for (uint32_t s = 0; s < n_all; s++) { total += n_findings[s]; if (arena_lens[s] > 0xFFFFFFFFull) return SX_E_INVALID; }
    order.reserve(total);
    uint32_t s0 = 0;
    for (int k = 0; k < world; k++) {
        uint64_t hi;
        sx_shard_bounds(file_len, world, k, nullptr, &hi);
        const uint32_t b = (uint32_t)(hi / kInputBufLen);
        const bool last = k + 1 == world;
        std::vector<Ref> own, nxt;
        for (uint32_t s = s0; s < s0 + n_segs_of_rank[k]; s++)
            for (uint64_t i = 0; i < n_findings[s]; i++) own.push_back({ s, k, &findings[s][i] });
        s0 += n_segs_of_rank[k];
        size_t cut = own.size();
        while (!last && cut > 0 && own[cut - 1].f->slice_index >= b) cut--;
        nxt.assign(own.begin() + (long)cut, own.end());
        own.resize(cut);
        if (!carry.empty()) {
            std::vector<Ref> merged;
            merged.reserve(own.size() + carry.size());
            size_t i = 0, j = 0;
            while (i < carry.size() && j < own.size()) {
                if (less_eq(carry[i], own[j])) merged.push_back(carry[i++]); else merged.push_back(own[j++]);
            }
            while (i < carry.size()) merged.push_back(carry[i++]);
            while (j < own.size()) merged.push_back(own[j++]);
            // what was carried may itself lie behind this rank's end (a region across a whole shard)
            size_t cut2 = merged.size();
            while (!last && cut2 > 0 && merged[cut2 - 1].f->slice_index >= b) cut2--;
            std::vector<Ref> tail(merged.begin() + (long)cut2, merged.end());
            tail.insert(tail.end(), nxt.begin(), nxt.end());
            std::stable_sort(tail.begin(), tail.end(), [](const Ref& a, const Ref& c) {
                if (a.f->slice_index != c.f->slice_index) return a.f->slice_index < c.f->slice_index;
                if (a.f->position != c.f->position) return a.f->position < c.f->position;
                return a.f->mission_id < c.f->mission_id;
            });
            merged.resize(cut2);
            own.swap(merged);
            nxt.swap(tail);
        }
        order.insert(order.end(), own.begin(), own.end());
        carry.swap(nxt);
    }
    order.insert(order.end(), carry.begin(), carry.end());
    ResultHolder res;
    // the strings are copied finding by finding: an output segment ends where its arena would pass seg_cap
    res.r->r.segs.emplace_back();
    for (const Ref& r : order) {
        MissionFindings* m = &res.r->r.segs.back();
        if (!m->v.empty() && m->arena.size() + r.f->str_len > seg_cap) { res.r->r.segs.emplace_back(); m = &res.r->r.segs.back(); }
        if ((uint64_t)r.f->str_off + r.f->str_len > arena_lens[r.seg]) return SX_E_INVALID;
        sx_finding f = *r.f;
        f.str_off = (uint32_t)m->arena.size();
        m->arena.append((const char*)arenas[r.seg] + r.f->str_off, r.f->str_len);
        m->v.push_back(f);
    }
    *out = res.release();
    return SX_OK;
}

// (an output segment ends where its arena would pass 2 GiB; SX_SPLICE_SEG_BYTES: tests)
int sx_shard_splice_segs(const sx_finding* const* findings, const uint64_t* n_findings, const uint8_t* const* arenas,
                         const uint64_t* arena_lens, const uint32_t* n_segs_of_rank, int world, uint64_t file_len, sx_result** out) {
    uint64_t seg_cap = 2048ull << 20;
    if (const char* e = getenv("SX_SPLICE_SEG_BYTES")) seg_cap = std::max<uint64_t>(1, (uint64_t)atoll(e));
    return splice_segs(findings, n_findings, arenas, arena_lens, n_segs_of_rank, world, file_len, seg_cap, out);
}

int sx_shard_splice(const sx_finding* const* findings, const uint64_t* n_findings, const uint8_t* const* arenas,
                    const uint64_t* arena_lens, int world, uint64_t file_len, sx_result** out) {
    if (world < 1) return SX_E_INVALID;
    uint64_t bytes = 0;
    if (arena_lens) for (int k = 0; k < world; k++) bytes += arena_lens[k];
    if (bytes > 0xFFFFFFFFull) return SX_E_NOMEM;   // (ONE result segment — str_off has 32 bits —: callers with more take sx_shard_splice_segs)
    std::vector<uint32_t> one((size_t)world, 1u);
    return splice_segs(findings, n_findings, arenas, arena_lens, one.data(), world, file_len, 0xFFFFFFFFull, out);
}

}  // extern "C"
