// sx_api.cpp — the C-ABI of include/stringsext_amd.h: context, HIP resources (one stream
// per Mission, the reference's one thread per Mission: src/main.rs:97,151), the two scan
// entry points that replace the loop src/main.rs:153-168, and the lower-level stages.
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sx_host.hpp"

using namespace sx;

namespace sx {
PinnedPool::Block PinnedPool::take(size_t bytes) {
    {
        std::lock_guard<std::mutex> g(mu);
        size_t best = free_blocks.size();
        for (size_t i = 0; i < free_blocks.size(); i++)
            if (free_blocks[i].cap >= bytes && (best == free_blocks.size() || free_blocks[i].cap < free_blocks[best].cap)) best = i;
        if (best < free_blocks.size() && free_blocks[best].cap <= 4 * bytes + (4u << 20)) {
            Block b = free_blocks[best];
            free_blocks.erase(free_blocks.begin() + (long)best);
            return b;
        }
    }
    Block b;
    const size_t cap = bytes + bytes / 8 + (1u << 20);
    if (hipHostMalloc(&b.p, cap, hipHostMallocNonCoherent) != hipSuccess) { b.p = nullptr; return b; }
    b.cap = cap;
    return b;
}
void PinnedPool::give(Block b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> g(mu);
    free_blocks.push_back(b);
    if (free_blocks.size() > 40) {  // a result of a large buffer holds one block per piece; beyond that, drop the smallest
        size_t m = 0;
        for (size_t i = 1; i < free_blocks.size(); i++) if (free_blocks[i].cap < free_blocks[m].cap) m = i;
        (void)hipHostFree(free_blocks[m].p);
        free_blocks.erase(free_blocks.begin() + (long)m);
    }
}
PinnedPool::~PinnedPool() { for (Block& b : free_blocks) (void)hipHostFree(b.p); }
}  // namespace sx

namespace {

std::string g_create_error;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// The long runs of one mission over one buffer: owned (host merge, caller-supplied) or a view
// of the mission's pinned download buffer; `on_device` says MissionDev::d_rp[0] holds the same list.
struct RunList {
    std::vector<sx_run> own;
    const sx_run* p = nullptr;
    size_t n = 0;
    bool on_device = false;
    void use_own() { p = own.data(); n = own.size(); on_device = false; }
    void assign(const sx_run* b, const sx_run* e) { own.assign(b, e); use_own(); }
    const sx_run* data() const { return p; }
    size_t size() const { return n; }
    const sx_run& operator[](size_t i) const { return p[i]; }
};

// Stage A writes its run records into one of two slots, so that the kernel of the next piece
// of a large buffer can run while the previous piece's records are sorted, joined and replayed.
struct ScanSlot {
    DevRun* d_recs = nullptr;
    uint32_t capacity = 0;
    uint32_t* d_counters = nullptr;   // 4 x u32: records, heavy tiles, joined runs, -
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the scan kernel
    hipEvent_t ev_free = nullptr;     // the slot's records have been consumed (recorded on stream_b)
    bool free_pending = false;
    // region mode (ScanParams::region_cap): per-sub-chunk counts and the packed, ordered records
    uint32_t* d_cnt = nullptr;  uint64_t cnt_cap = 0;
    DevRun* d_packed = nullptr; uint64_t packed_cap = 0;
    uint32_t region_cap = 0;    // of the launch in flight (0: shared pool)
    uint64_t n_regions = 0;
};
struct MissionDev {
    hipStream_t stream = nullptr;     // scan kernels only
    hipStream_t stream_b = nullptr;   // everything after them (sort/join, stage B, copies); higher priority
    ScanSlot slot[2];
    // stage B on the device: grow-only buffers
    uint16_t* d_table = nullptr;                        // single-byte decoder table
    sx_run* h_runs = nullptr; uint64_t h_runs_cap = 0;   // pinned: runs joined on the device
    void* d_rp[9] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };  // runs, region outs, idx, fbase, abase, findings+arena
    uint64_t d_rp_cap[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };  // + stitch blocks, totals, pass-1 output cache
};

}  // namespace

struct sx_result {
    Result r;
};

struct sx_ctx {
    std::shared_ptr<PinnedPool> pool = std::make_shared<PinnedPool>();
    std::vector<Mission> missions;
    std::vector<ScannerState> states;
    std::vector<MissionDev> dev;
    bool host_only = false;
    int device = -1;
    sx_options opt{};
    std::string err;
    sx_stats stats{};
    hipStream_t scan_stream = nullptr, post_stream = nullptr;
    unsigned n_cus = 256, scan_blocks_per_cu = 8;
    // sx_scan_stream: two pinned host buffers and two device buffers, filled by a reader thread
    hipStream_t copy_stream = nullptr;
    uint8_t* ing_pin[2] = { nullptr, nullptr };
    uint8_t* ing_dev[2] = { nullptr, nullptr };
    uint64_t ing_cap = 0, ing_dev_cap = 0;
    uint32_t region_cap = 32;         // record slots per sub-chunk in region mode (0: never use it)
    std::vector<char> dense;          // per mission: the last buffer overflowed its regions -> shared pool + sort
    uint8_t* d_input = nullptr;  // staging for host input
    uint64_t d_input_cap = 0;
    uint64_t ondemand_fetches = 0;
    // grow-only scratch reused by every call (pinned host memory: D2H at full PCIe rate)
    std::vector<uint64_t> last_runs;  // long runs per mission of the last scanned buffer: busiest mission scans first
    std::vector<RunList> shard_runs;  // device runs of the last sx_scan_shard* buffer (reuse_runs)
    bool shard_runs_valid = false;
    uint8_t* h_pin = nullptr;   uint64_t h_pin_cap = 0;
    uint8_t* h_pin2 = nullptr;  uint64_t h_pin2_cap = 0;   // device replay traffic (h_pin may back a live byte view)
    uint8_t* d_scratch = nullptr; uint64_t d_scratch_cap = 0;
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return SX_E_HIP;                                                                   \
        }                                                                                      \
    } while (0)

namespace {

// Device-resident chunk of which only some byte ranges were downloaded.
class SparseDeviceBytes : public ByteView {
public:
    SparseDeviceBytes(sx_ctx* ctx, const uint8_t* d_base) : ctx_(ctx), d_base_(d_base) {}
    void add(uint64_t lo, uint64_t hi, const uint8_t* p) { segs_.push_back({ lo, hi, p }); }
    // same, but the bytes are copied (the caller's buffer may be reused while the view lives)
    void add_copy(uint64_t lo, uint64_t hi, const uint8_t* p) {
        owned_.emplace_back(p, p + (hi - lo));
        segs_.push_back({ lo, hi, owned_.back().data() });
    }
    bool empty() const { return segs_.empty(); }
    const uint8_t* span(uint64_t off, size_t n, size_t* hint) override {
        // segments are sorted and disjoint; the caller moves forward, so look near its cursor first
        size_t a = *hint < segs_.size() ? *hint : 0;
        if (!(a < segs_.size() && segs_[a].lo <= off)) a = 0;
        size_t steps = 0;
        while (a < segs_.size() && segs_[a].hi <= off && steps < 8) { a++; steps++; }
        if (!(a < segs_.size() && segs_[a].lo <= off && off < segs_[a].hi)) {
            size_t lo = 0, hi = segs_.size();
            while (lo < hi) {
                size_t mid = (lo + hi) / 2;
                if (segs_[mid].hi <= off) lo = mid + 1; else hi = mid;
            }
            a = lo;
        }
        if (a < segs_.size() && segs_[a].lo <= off && off + n <= segs_[a].hi) { *hint = a; return segs_[a].p + (off - segs_[a].lo); }
        // rare: the replay ran further than planned — fetch exactly what is asked for
        std::lock_guard<std::mutex> g(mu_);
        extra_.emplace_back(n);
        if (hipMemcpy(extra_.back().data(), d_base_ + off, n, hipMemcpyDeviceToHost) != hipSuccess)
            memset(extra_.back().data(), 0, n);
        ctx_->ondemand_fetches++;
        return extra_.back().data();
    }

private:
    struct Seg { uint64_t lo, hi; const uint8_t* p; };
    sx_ctx* ctx_;
    const uint8_t* d_base_;
    std::vector<Seg> segs_;
    std::deque<std::vector<uint8_t>> extra_, owned_;
    std::mutex mu_;
};

int ensure_pinned(sx_ctx* ctx, uint64_t bytes) {
    if (ctx->h_pin_cap >= bytes) return SX_OK;
    if (ctx->h_pin) HIP_TRY(ctx, hipHostFree(ctx->h_pin));
    ctx->h_pin = nullptr; ctx->h_pin_cap = 0;
    bytes += bytes / 4 + (1u << 20);
    unsigned flags = hipHostMallocNonCoherent;  // CPU-cached: it is only read by the host after a stream sync
    if (const char* e = getenv("SX_PIN_FLAGS")) flags = (unsigned)strtoul(e, nullptr, 0);
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_pin, bytes, flags));
    ctx->h_pin_cap = bytes;
    return SX_OK;
}
int ensure_pinned2(sx_ctx* ctx, uint64_t bytes) {
    if (ctx->h_pin2_cap >= bytes) return SX_OK;
    if (ctx->h_pin2) HIP_TRY(ctx, hipHostFree(ctx->h_pin2));
    ctx->h_pin2 = nullptr; ctx->h_pin2_cap = 0;
    bytes += bytes / 4 + (1u << 20);
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_pin2, bytes, hipHostMallocNonCoherent));
    ctx->h_pin2_cap = bytes;
    return SX_OK;
}
int ensure_scratch(sx_ctx* ctx, uint64_t bytes) {
    if (ctx->d_scratch_cap >= bytes) return SX_OK;
    if (ctx->d_scratch) HIP_TRY(ctx, hipFree(ctx->d_scratch));
    ctx->d_scratch = nullptr; ctx->d_scratch_cap = 0;
    bytes += bytes / 4 + (1u << 20);
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_scratch, bytes));
    ctx->d_scratch_cap = bytes;
    return SX_OK;
}

int ensure_capacity(sx_ctx* ctx, ScanSlot& s, uint32_t cap) {
    if (s.capacity >= cap) return SX_OK;
    if (s.d_recs) HIP_TRY(ctx, hipFree(s.d_recs));
    s.d_recs = nullptr; s.capacity = 0;
    HIP_TRY(ctx, hipMalloc((void**)&s.d_recs, (size_t)cap * sizeof(DevRun)));
    s.capacity = cap;
    return SX_OK;
}

// Stage A for a set of missions: launch every mission's kernel on its own stream, then
// collect, growing a record buffer and re-running that mission if it overflowed.
int ensure_rp(sx_ctx* ctx, MissionDev& d, int slot, uint64_t bytes) {
    if (d.d_rp_cap[slot] >= bytes) return SX_OK;
    if (d.d_rp[slot]) HIP_TRY(ctx, hipFree(d.d_rp[slot]));
    d.d_rp[slot] = nullptr; d.d_rp_cap[slot] = 0;
    bytes += bytes / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(&d.d_rp[slot], bytes));
    d.d_rp_cap[slot] = bytes;
    return SX_OK;
}

ScanParams scan_params(const sx_ctx* ctx, int mission, const ScanSlot& s, const uint8_t* d_bytes, uint64_t len,
                       uint32_t parity, uint64_t min_chars) {
    const Mission& m = ctx->missions[(size_t)mission];
    uint32_t sub = ctx->opt.subchunk_bytes ? ctx->opt.subchunk_bytes : 256u * 1024u;
    sub = std::max<uint32_t>(kTileBytes, sub / kTileBytes * kTileBytes);
    ScanParams p = m.proto;
    p.data = d_bytes; p.len = len; p.subchunk = sub; p.parity = parity;
    p.min_chars = (uint32_t)std::min<uint64_t>(min_chars, kRecCharsMask);
    if (p.min_chars == 0) p.min_chars = 1;
    p.cand_bytes = std::min<uint32_t>(p.min_chars * (m.is_utf16() ? 2u : 1u), 17u);
    {   // r &= r << sh, doubling the proven run length until it reaches cand_bytes
        uint32_t have = 1;
        for (int i = 0; i < 5; i++) {
            const uint32_t sh = have < p.cand_bytes ? std::min(have, p.cand_bytes - have) : 0;
            p.cand_sh[i] = sh;
            have += sh;
        }
    }
    p.capacity = s.capacity; p.recs = s.d_recs; p.counters = s.d_counters;
    p.region_cap = s.region_cap; p.region_counts = s.d_cnt;
    p.traversal = (ctx->opt.flags & SX_OPT_TILE_TRAVERSAL) ? 1u : 0u;
    if (const char* e = getenv("SX_TRAVERSAL")) p.traversal = (uint32_t)atoi(e);
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
            const bool tile_traversal = (ctx->opt.flags & SX_OPT_TILE_TRAVERSAL) || (getenv("SX_TRAVERSAL") && atoi(getenv("SX_TRAVERSAL")));
            if (ctx->dense.size() != ctx->missions.size()) ctx->dense.assign(ctx->missions.size(), 0);
            s.region_cap = 0; s.n_regions = n_regions;
            if (ctx->region_cap && !ctx->dense[(size_t)which[k]] && !tile_traversal && n_regions * ctx->region_cap < (1ull << 28)) {
                s.region_cap = ctx->region_cap;
                int rc = ensure_capacity(ctx, s, (uint32_t)(n_regions * s.region_cap));
                if (rc != SX_OK) return rc;
                if (s.cnt_cap < n_regions) {
                    if (s.d_cnt) HIP_TRY(ctx, hipFree(s.d_cnt));
                    s.d_cnt = nullptr; s.cnt_cap = 0;
                    HIP_TRY(ctx, hipMalloc((void**)&s.d_cnt, (n_regions + n_regions / 4 + 64) * 4));
                    s.cnt_cap = n_regions + n_regions / 4 + 64;
                }
            }
        }
        const ScanParams p = scan_params(ctx, which[k], s, d_bytes, len, parity[k], min_chars[k]);
        HIP_TRY(ctx, hipMemsetAsync(s.d_counters, 0, 4 * sizeof(uint32_t), d.stream));
        HIP_TRY(ctx, hipEventRecord(s.ev0, d.stream));
        HIP_TRY(ctx, launch_scan(ctx->missions[(size_t)which[k]].kind, p, d.stream));
        HIP_TRY(ctx, hipEventRecord(s.ev1, d.stream));
    }
    return SX_OK;
}

// Stage A, second half: wait for slot `si`, re-run a mission whose record buffer overflowed,
// and turn the records into the mission's sorted long runs (joined on the device when there
// are many).  Only stream_b is used from here on: the scan streams may already hold the next piece.
int stage_a_finish(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                   const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars, int si,
                   std::vector<RunList>* out) {
    out->assign(which.size(), RunList{});
    if (len == 0) return SX_OK;
    const double t0 = now_ms();
    for (size_t k = 0; k < which.size(); k++) {
        MissionDev& d = ctx->dev[(size_t)which[k]];
        ScanSlot& s = d.slot[si];
        HIP_TRY(ctx, hipEventSynchronize(s.ev1));
        const double t_ev = now_ms();
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, s.ev0, s.ev1));
        if (which[k] < 16) ctx->stats.kernel_ms[which[k]] += ms;
        uint32_t counters[4] = { 0, 0, 0, 0 };
        for (int round = 0;; round++) {
            HIP_TRY(ctx, hipMemcpyAsync(counters, s.d_counters, sizeof counters, hipMemcpyDeviceToHost, d.stream_b));
            HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            if (s.region_cap) {
                if (counters[0] == 0) break;  // every sub-chunk's records fit its region
                // too dense for regions: this mission uses the shared pool (and a sort) from now on
                ctx->dense[(size_t)which[k]] = 1;
                s.region_cap = 0;
            } else {
                if (counters[0] <= s.capacity) break;
                if (round >= 8) { ctx->err = "device run-record buffer kept overflowing"; return SX_E_NOMEM; }
                // overflow: grow the slot and scan this piece again for this mission
                int rc = ensure_capacity(ctx, s, counters[0] + counters[0] / 8 + 1024);
                if (rc != SX_OK) return rc;
            }
            const ScanParams p = scan_params(ctx, which[k], s, d_bytes, len, parity[k], min_chars[k]);
            HIP_TRY(ctx, hipMemsetAsync(s.d_counters, 0, 4 * sizeof(uint32_t), d.stream_b));
            HIP_TRY(ctx, launch_scan(ctx->missions[(size_t)which[k]].kind, p, d.stream_b));
        }
        const double tc0 = now_ms();
        if (getenv("SX_TIMING2")) fprintf(stderr, "[sx]   mission %d: kernel done at +%.2f ms, counters at +%.2f ms\n", which[k], t_ev - t0, tc0 - t0);
        uint32_t nrec = counters[0];
        const DevRun* d_records = s.d_recs;   // sorted already in region mode
        const bool regions = s.region_cap != 0;
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
            HIP_TRY(ctx, hipMemcpyAsync(&nrec, s.d_counters + 1, 4, hipMemcpyDeviceToHost, d.stream_b));
            HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            d_records = s.d_packed;
        } else if (ctx->region_cap && nrec < s.n_regions * ctx->region_cap / 4)
            ctx->dense[(size_t)which[k]] = 0;  // sparse again: regions next time
        const uint32_t join_min = getenv("SX_DEVICE_JOIN_MIN") ? (uint32_t)atoi(getenv("SX_DEVICE_JOIN_MIN")) : 65536u;
        const bool dev_sorted = nrec >= join_min && nrec > 0;  // worth a handful of small kernels
        double tc1 = tc0;
        RunList& rl = (*out)[k];
        if (dev_sorted) {
            // sort the records and join them into runs on the device; only the runs travel
            const size_t sb = std::max(sort_scratch_bytes(nrec), merge_scratch_bytes(nrec));
            int rc = ensure_scratch(ctx, sb); if (rc != SX_OK) return rc;
            rc = ensure_rp(ctx, d, 0, (uint64_t)nrec * sizeof(sx_run)); if (rc != SX_OK) return rc;
            if (!regions) HIP_TRY(ctx, sort_records(s.d_recs, nrec, ctx->d_scratch, ctx->d_scratch_cap, d.stream_b));
            if (getenv("SX_TIMING2")) { HIP_TRY(ctx, hipStreamSynchronize(d.stream_b)); fprintf(stderr, "[sx]   sort done +%.2f ms\n", now_ms() - tc0); }
            HIP_TRY(ctx, merge_sorted_records(d_records, nrec, min_chars[k], ctx->d_scratch, ctx->d_scratch_cap,
                                              (sx_run*)d.d_rp[0], s.d_counters + 2, d.stream_b));
            HIP_TRY(ctx, hipEventRecord(s.ev_free, d.stream_b));
            s.free_pending = true;
            if (getenv("SX_TIMING2")) { HIP_TRY(ctx, hipStreamSynchronize(d.stream_b)); fprintf(stderr, "[sx]   join done +%.2f ms\n", now_ms() - tc0); }
            uint32_t nruns = 0;
            HIP_TRY(ctx, hipMemcpyAsync(&nruns, s.d_counters + 2, 4, hipMemcpyDeviceToHost, d.stream_b));
            HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            tc1 = now_ms();
            if ((uint64_t)nruns * sizeof(sx_run) > d.h_runs_cap) {
                if (d.h_runs) HIP_TRY(ctx, hipHostFree(d.h_runs));
                d.h_runs = nullptr; d.h_runs_cap = 0;
                const uint64_t cap = (uint64_t)nruns * sizeof(sx_run) * 5 / 4 + 4096;
                HIP_TRY(ctx, hipHostMalloc((void**)&d.h_runs, cap, hipHostMallocNonCoherent));
                d.h_runs_cap = cap;
            }
            if (nruns) {
                HIP_TRY(ctx, hipMemcpyAsync(d.h_runs, d.d_rp[0], (size_t)nruns * sizeof(sx_run), hipMemcpyDeviceToHost, d.stream_b));
                HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
            }
            rl.p = d.h_runs; rl.n = nruns; rl.on_device = true;
        } else {
            int rc = ensure_pinned(ctx, (uint64_t)nrec * sizeof(DevRun) + 16);
            if (rc != SX_OK) return rc;
            DevRun* recs_p = (DevRun*)ctx->h_pin;
            if (nrec) {
                HIP_TRY(ctx, hipMemcpyAsync(recs_p, d_records, (size_t)nrec * sizeof(DevRun), hipMemcpyDeviceToHost, d.stream_b));
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
                    regions ? "records (regions)" : "record slots (pool)", dev_sorted ? (regions ? "device pack+join" : "device sort+join") : "d2h",
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

// CPUs this process may really use: the cgroup quota can be far below the visible cores.
unsigned usable_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = { 0 };
        unsigned long long period = 0;
        if (fscanf(f, "%63s %llu", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) {
            const unsigned long long quota = strtoull(a, nullptr, 10);
            const unsigned q = (unsigned)((quota + period - 1) / period);
            if (q >= 1 && q < n) n = q;
        }
        fclose(f);
    }
    return n;
}

unsigned replay_threads(const sx_ctx* ctx) {
    // a few threads per usable CPU smooth out the quota's time slicing (measured: 4x is best)
    unsigned n = ctx->opt.replay_threads ? ctx->opt.replay_threads
                                         : std::min(std::thread::hardware_concurrency(), 4 * usable_cpus());
    if (const char* e = getenv("SX_REPLAY_THREADS")) n = (unsigned)atoi(e);
    if (n < 1) n = 1;
    return n > 256 ? 256 : n;
}

// What stage B is asked to do for one buffer ("chunk" of sx_scan, or a shard's buffer).
struct ReplayJob {
    uint64_t len = 0;                   // buffer bytes (its byte 0 lies on the slice grid)
    int file_id = -1;
    bool is_last = false;
    std::vector<uint64_t> lo;           // per mission: replay regions that begin in [lo, hi)
    uint64_t hi = 0;
    std::vector<char> entry_exact;      // per mission: ctx->states[m] is the exact state at lo
    std::vector<uint64_t> consumed0, stream0;  // per mission: ScannerState counters at buffer byte 0
    bool commit_state = true;           // store the final state in the context
    uint32_t slice_base = 0;            // added to slice_index of the findings
    const uint8_t* d_bytes = nullptr;   // the buffer in HBM, if stage B may run on the device
};



static inline uint64_t win_start_h(uint64_t p, size_t W) {
    const uint64_t s0 = p / kInputBufLen * kInputBufLen;
    return s0 + (p - s0) / W * W;
}

bool device_replay_wanted(const sx_ctx* ctx, const ReplayJob& job, size_t k, size_t n_runs) {
    if (!job.d_bytes || ctx->host_only || job.is_last) return false;
    if (ctx->opt.flags & SX_OPT_HOST_REPLAY) return false;
    if (ctx->missions[k].q > 64) return false;
    if (getenv("SX_HOST_REPLAY")) return false;
    return (ctx->opt.flags & SX_OPT_DEVICE_REPLAY) || getenv("SX_DEVICE_REPLAY") || n_runs >= 4096;
}

// Stage B of one mission on the device (sx_replay_dev.hip) + the little the host keeps:
// the chunk's strict entry region, regions the device gave back, the exact exit state.
int device_replay_mission(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, const RunList& runs,
                          MissionFindings* out, uint64_t* end_pos) {
    const Mission& m = ctx->missions[k];
    MissionDev& d = ctx->dev[k];
    const size_t n = runs.size();
    const size_t W = m.window;
    const double t0 = now_ms();

    // ---- pass 1 on the device: every region's extent and output size; then (still on the
    // device) which regions stand and where each writes.  The host keeps its own version of
    // that step for buffers with regions the device gave back (kRegionTooLong).
    ReplayParams P{};
    bool dev_stitch = n > 0 && !getenv("SX_HOST_STITCH");
    uint64_t* h_tot = nullptr;
    ReplayRegionOut* ro = nullptr;
    void* cache_used = nullptr;
    {
        int rc = ensure_pinned2(ctx, n * sizeof(ReplayRegionOut) + 256);
        if (rc != SX_OK) return rc;
        h_tot = (uint64_t*)ctx->h_pin2;
        ro = (ReplayRegionOut*)(ctx->h_pin2 + 128);
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
        if (!runs.on_device)
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[0], runs.data(), n * sizeof(sx_run), hipMemcpyHostToDevice, d.stream_b));
        P.data = job.d_bytes; P.len = job.len; P.runs = (const sx_run*)d.d_rp[0]; P.n_runs = n;
        P.lo = job.lo[k]; P.hi = job.hi; P.consumed0 = job.consumed0[k]; P.stream0 = job.stream0[k];
        P.slice_base = job.slice_base; P.encoding = m.c.encoding; P.table = d.d_table;
        P.chars_min_nb = m.c.chars_min_nb; P.same_block = m.c.require_same_unicode_block; P.q = (uint32_t)m.q;
        P.W = (uint32_t)W; P.long_run = m.long_run; P.skip = getenv("SX_NO_REPLAY_SKIP") ? 0u : 1u; P.grep_char = m.c.grep_char; P.mission_id = m.c.mission_id;
        P.file_id = job.file_id; P.af_lo = m.c.af_lo; P.af_hi = m.c.af_hi; P.ubf = m.c.ubf;
        void* cache = nullptr;
        if (dev_stitch && n <= (32u << 20) && !getenv("SX_NO_REPLAY_CACHE")) {
            rc = ensure_rp(ctx, d, 8, replay_cache_bytes(n)); if (rc) return rc;
            cache = d.d_rp[8];
        }
        cache_used = cache;
        HIP_TRY(ctx, launch_replay_count(P, (ReplayRegionOut*)d.d_rp[1], cache, d.stream_b));
        if (dev_stitch) {
            HIP_TRY(ctx, hipMemsetAsync(d.d_rp[7], 0, kTotCount * 8, d.stream_b));
            HIP_TRY(ctx, launch_stitch_blocks(P, (const ReplayRegionOut*)d.d_rp[1], (uint8_t*)d.d_rp[2], d.d_rp[6],
                                              (uint64_t*)d.d_rp[7], d.stream_b));
        } else
            HIP_TRY(ctx, hipMemcpyAsync(ro, d.d_rp[1], n * sizeof(ReplayRegionOut), hipMemcpyDeviceToHost, d.stream_b));
    }

    // ---- meanwhile on the host: the strict entry region (exact carried state), if any
    std::deque<ReplayPart> host_parts;
    struct Seg { int host_part; size_t v0, v1; };  // host_part >= 0, or device regions [v0, v1) of `valid`
    std::vector<Seg> segs;
    uint64_t E = std::min(job.lo[k], job.hi);
    uint64_t last_start = E;   // start of the last region of any kind (for the exit state)
    bool last_is_entry = false;
    if (job.entry_exact[k]) {
        // The chunk's first window belongs to the host: only it has the exact carried state
        // (leftover, cut flag, and the decoder's pending bytes, which cannot be re-derived here).
        host_parts.emplace_back();
        replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, runs.data(), n,
                    job.lo[k], job.lo[k] + 1, true, &host_parts.back());
        if (host_parts.back().regions.empty()) host_parts.pop_back();
        else { segs.push_back({ (int)host_parts.size() - 1, 0, 0 }); last_is_entry = true; E = std::max(E, host_parts.back().end_pos); }
        E = std::max(E, job.lo[k] + 1);
    }
    if (dev_stitch) {
        HIP_TRY(ctx, launch_stitch_finish(P, (const ReplayRegionOut*)d.d_rp[1], (uint8_t*)d.d_rp[2], d.d_rp[6], E,
                                          (uint64_t*)d.d_rp[3], (uint64_t*)d.d_rp[4], (uint64_t*)d.d_rp[7], ctx->d_scratch,
                                          ctx->d_scratch_cap, d.stream_b));
        HIP_TRY(ctx, hipMemcpyAsync(h_tot, d.d_rp[7], kTotCount * 8, hipMemcpyDeviceToHost, d.stream_b));
    }
    if (n) HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
    if (dev_stitch && h_tot[kTotTooLong]) {  // regions for the host: it also decides what stands
        dev_stitch = false;
        HIP_TRY(ctx, hipMemcpyAsync(ro, d.d_rp[1], n * sizeof(ReplayRegionOut), hipMemcpyDeviceToHost, d.stream_b));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
    }
    const double t1 = now_ms();

    std::vector<uint64_t> valid, fbase, abase;
    uint64_t nf = 0, nb = 0, n_standing = 0;
    if (dev_stitch) {
        nf = h_tot[kTotFindings]; nb = h_tot[kTotBytes]; n_standing = h_tot[kTotStanding];
        out->replay_bytes += h_tot[kTotReplayBytes];
        if (h_tot[kTotLast] != ~0ull) {
            E = std::max(E, h_tot[kTotEnd]);
            last_start = win_start_h(runs[(size_t)h_tot[kTotLast]].start, W);
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
    // pinned block that becomes the result's storage (no copy) unless host parts must be spliced in
    PinnedPool::Block blk{};
    if (n_standing) {
        int rc = ensure_rp(ctx, d, 5, nf * sizeof(sx_finding) + nb + 64); if (rc) return rc;
        sx_finding* d_f = (sx_finding*)d.d_rp[5];
        uint8_t* d_a = (uint8_t*)d.d_rp[5] + nf * sizeof(sx_finding);
        if (dev_stitch) {
            HIP_TRY(ctx, launch_replay_write_flagged(P, (const ReplayRegionOut*)d.d_rp[1], (const uint8_t*)d.d_rp[2],
                                                     (const uint64_t*)d.d_rp[3], (const uint64_t*)d.d_rp[4], cache_used, d_f,
                                                     d_a, d.stream_b));
        } else {
            const size_t nv = valid.size();
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[2], valid.data(), nv * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[3], fbase.data(), (nv + 1) * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, hipMemcpyAsync(d.d_rp[4], abase.data(), (nv + 1) * 8, hipMemcpyHostToDevice, d.stream_b));
            HIP_TRY(ctx, launch_replay_write(P, (const uint64_t*)d.d_rp[2], (const uint64_t*)d.d_rp[3],
                                             (const uint64_t*)d.d_rp[4], nv, d_f, d_a, d.stream_b));
        }
        blk = ctx->pool->take(nf * sizeof(sx_finding) + nb + 64);
        if (!blk.p) { ctx->err = "hipHostMalloc failed"; return SX_E_NOMEM; }
        HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_f, nf * sizeof(sx_finding) + nb, hipMemcpyDeviceToHost, d.stream_b));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream_b));
    }
    const double t3 = now_ms();

    // ---- splice (almost always: device findings only)
    if (host_parts.empty()) {
        if (blk.p) { out->ext = blk; out->ext_nf = nf; out->ext_na = nb; out->dev_copy = d.d_rp[5]; }
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
        uint64_t from = last_is_entry ? job.lo[k] : (E > ts ? last_start : ts);
        if (from < job.lo[k]) from = job.lo[k];
        ReplayPart fin;
        replay_part(m, ctx->states[k], job.consumed0[k], job.stream0[k], view, job.len, job.file_id, false, runs.data(), n,
                    from, job.len, job.entry_exact[k] && from == job.lo[k], &fin);
        ctx->states[k] = fin.state;
        ctx->states[k].consumed_bytes = job.consumed0[k] + job.len;
        ctx->states[k].stream_bytes = job.stream0[k] + job.len;
        E = job.len;
    }
    if (end_pos) *end_pos = std::max(E, std::min(job.hi, job.len));
    if (getenv("SX_TIMING"))
        fprintf(stderr, "[sx] device replay mission %zu: %zu runs, pass1+entry %.2f ms, validity %.2f ms (%zu standing, %zu host parts), "
                        "pass2+d2h %.2f ms (%llu findings), splice+state %.2f ms\n", k, n, t1 - t0, t2 - t1, (size_t)n_standing,
                host_parts.size(), t3 - t2, (unsigned long long)nf, now_ms() - t3);
    return SX_OK;
}

// Stage B for all missions: every (mission, part) pair is one task for a small thread pool;
// part 0 of a mission starts from its entry state, the others speculate, and the per-mission
// stitch verifies/repairs them serially.
// Missions whose stage B already ran (on the device, while later missions were still being scanned).
struct PreReplayed {
    std::vector<char> done;
    std::vector<MissionFindings> per;
    std::vector<uint64_t> ends;
    explicit PreReplayed(size_t nm) : done(nm, 0), per(nm), ends(nm, 0) {}
};

int replay_all(sx_ctx* ctx, ByteView& bytes, const ReplayJob& job, const std::vector<RunList>& runs,
               Result* into, uint64_t* end_pos, PreReplayed* pre = nullptr) {
    const double t0 = now_ms();
    const size_t nm = ctx->missions.size();
    const unsigned nthreads = replay_threads(ctx);
    std::vector<std::vector<uint64_t>> bounds(nm);
    std::vector<std::vector<ReplayPart>> parts(nm);
    std::vector<std::pair<size_t, size_t>> tasks;
    std::vector<char> on_device(nm, 0);
    uint64_t host_runs = 0;
    for (size_t k = 0; k < nm; k++)
        on_device[k] = (pre && pre->done[k]) ? 2 : (device_replay_wanted(ctx, job, k, runs[k].size()) ? 1 : 0);
    for (size_t k = 0; k < nm; k++) {
        if (on_device[k]) continue;
        // parts are speculative restarts: worth a thread each only if they hold real work
        const unsigned want_parts = (unsigned)std::min<uint64_t>(nthreads, std::max<uint64_t>(1, runs[k].size() / 512));
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
        else if (on_device[k]) {
            int rc = device_replay_mission(ctx, k, bytes, job, runs[k], &per[k], &ends[k]);
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
    const size_t count_before = into->count();
    {   // several missions with findings that are all still on the device: interleave them there
        // (a stable radix sort by position) instead of finding by finding on the host
        size_t with = 0, on_dev = 0, total = 0, bytes = 0;
        bool same_origin = true;
        for (size_t k = 0; k < nm; k++) {
            if (!per[k].count()) continue;
            with++;
            if (per[k].ext.p && per[k].dev_copy) on_dev++;
            total += per[k].count(); bytes += per[k].strings_len();
            same_origin = same_origin && ctx->missions[k].c.counter_offset == ctx->missions[0].c.counter_offset;
        }
        if (with >= 2 && on_dev == with && same_origin && bytes <= 0xFFFFFFFFull && total >= 4096 && !getenv("SX_HOST_MERGE")) {
            const double tm0 = now_ms();
            std::vector<const void*> srcs(nm, nullptr);
            std::vector<uint64_t> nfs(nm, 0), nbs(nm, 0);
            for (size_t k = 0; k < nm; k++)
                if (per[k].count()) { srcs[k] = per[k].dev_copy; nfs[k] = per[k].ext_nf; nbs[k] = per[k].ext_na; }
            int rc = ensure_scratch(ctx, merge_findings_scratch_bytes(total) + total * sizeof(sx_finding) + bytes + 512);
            if (rc != SX_OK) return rc;
            uint8_t* d_out = ctx->d_scratch;
            const size_t out_bytes = total * sizeof(sx_finding) + bytes;
            uint8_t* d_tmp = d_out + ((out_bytes + 255) & ~(size_t)255);
            hipStream_t s = ctx->post_stream;
            HIP_TRY(ctx, merge_findings_device(srcs.data(), nfs.data(), nbs.data(), (int)nm, d_out, d_tmp,
                                               ctx->d_scratch_cap - (size_t)(d_tmp - ctx->d_scratch), s));
            PinnedPool::Block blk = ctx->pool->take(out_bytes + 64);
            if (!blk.p) { ctx->err = "hipHostMalloc failed"; return SX_E_NOMEM; }
            HIP_TRY(ctx, hipMemcpyAsync(blk.p, d_out, out_bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipStreamSynchronize(s));
            uint64_t rb = 0;
            for (size_t k = 0; k < nm; k++) {
                rb += per[k].replay_bytes;
                if (per[k].ext.p) ctx->pool->give(per[k].ext);
                per[k] = MissionFindings{};
            }
            per[0].ext = blk; per[0].ext_nf = total; per[0].ext_na = bytes; per[0].replay_bytes = rb;
            if (getenv("SX_TIMING")) fprintf(stderr, "[sx] device merge of %zu missions: %zu findings, %.2f ms\n", with, total, now_ms() - tm0);
        }
    }
    merge_findings(per, ctx->pool, into);
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

void begin_call(sx_ctx* ctx) {
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->err.clear();
}

}  // namespace

extern "C" {

int sx_abi_version(void) { return SX_ABI_VERSION; }

int sx_create(sx_ctx** out, const sx_mission* missions, int n_missions, int hip_device, const sx_options* opt) {
    if (!out || !missions || n_missions <= 0 || n_missions > 26) { g_create_error = "bad arguments"; return SX_E_INVALID; }
    sx_ctx* ctx = new sx_ctx();
    if (opt) ctx->opt = *opt;
    ctx->missions.resize((size_t)n_missions);
    ctx->states.resize((size_t)n_missions);
    for (int k = 0; k < n_missions; k++) {
        std::string err;
        int rc = Mission::from_c(missions[k], (ctx->opt.flags & SX_OPT_GENERIC_KERNELS) != 0, &ctx->missions[(size_t)k], &err);
        if (rc != SX_OK) { g_create_error = "mission " + std::to_string(k) + ": " + err; delete ctx; return rc; }
        ctx->states[(size_t)k].reset(ctx->missions[(size_t)k]);
    }
    if (hip_device == SX_HOST_ONLY) { ctx->host_only = true; *out = ctx; return SX_OK; }

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0 || hip_device < 0 || hip_device >= n_dev) {
        g_create_error = std::string("no usable HIP device (hipGetDeviceCount: ")
                         + (e == hipSuccess ? "ok" : hipGetErrorString(e)) + ", devices=" + std::to_string(n_dev)
                         + ", requested=" + std::to_string(hip_device) + "); the scan only runs on the GPU";
        delete ctx;
        return SX_E_NO_DEVICE;
    }
    ctx->device = hip_device;
    auto fail = [&](const char* what, hipError_t err) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        sx_destroy(ctx);
        return SX_E_HIP;
    };
    if ((e = hipSetDevice(hip_device)) != hipSuccess) return fail("hipSetDevice", e);
    ctx->dev.resize((size_t)n_missions);
    const uint32_t cap = ctx->opt.record_capacity ? ctx->opt.record_capacity : (1u << 20);
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hip_device) == hipSuccess && cus > 0) ctx->n_cus = (unsigned)cus;
        if (const char* e2 = getenv("SX_SCAN_BLOCKS_PER_CU")) ctx->scan_blocks_per_cu = (unsigned)atoi(e2);
        if (const char* e2 = getenv("SX_REGION_CAP")) ctx->region_cap = (uint32_t)atoi(e2);
        if (const char* e2 = getenv("SX_SCAN_CUS")) ctx->n_cus = (unsigned)std::max(1, atoi(e2));  // tests: persistent grid on small inputs
    }
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);  // numerically lower = higher priority
    if (const char* e = getenv("SX_PRIO")) { (void)sscanf(e, "%d,%d", &prio_lo, &prio_hi); fprintf(stderr, "[sx] stream priorities: scan %d, post %d\n", prio_lo, prio_hi); }
    // Two streams for the whole context: the scan kernels of all missions queue up in one (a
    // kernel alone already fills the device, and HIP multiplexes streams onto few hardware
    // queues: more streams only alias), everything else runs in a second, higher-priority one
    // so that it overlaps the scans of the next piece.  SX_OPT_MISSION_STREAMS: a scan stream per mission.
    const bool per_mission = (ctx->opt.flags & SX_OPT_MISSION_STREAMS) || getenv("SX_MISSION_STREAMS");
    if ((e = hipStreamCreateWithPriority(&ctx->post_stream, hipStreamNonBlocking, prio_hi)) != hipSuccess) return fail("hipStreamCreate", e);
    if (!per_mission && (e = hipStreamCreateWithPriority(&ctx->scan_stream, hipStreamNonBlocking, prio_lo)) != hipSuccess)
        return fail("hipStreamCreate", e);
    for (auto& d : ctx->dev) {
        d.stream_b = ctx->post_stream;
        if (per_mission) {
            if ((e = hipStreamCreateWithPriority(&d.stream, hipStreamNonBlocking, prio_lo)) != hipSuccess) return fail("hipStreamCreate", e);
        } else d.stream = ctx->scan_stream;
        for (ScanSlot& s : d.slot) {
            if ((e = hipEventCreate(&s.ev0)) != hipSuccess) return fail("hipEventCreate", e);
            if ((e = hipEventCreate(&s.ev1)) != hipSuccess) return fail("hipEventCreate", e);
            if ((e = hipEventCreateWithFlags(&s.ev_free, hipEventDisableTiming)) != hipSuccess) return fail("hipEventCreate", e);
            if ((e = hipMalloc((void**)&s.d_counters, 4 * sizeof(uint32_t))) != hipSuccess) return fail("hipMalloc", e);
            if ((e = hipMalloc((void**)&s.d_recs, (size_t)cap * sizeof(DevRun))) != hipSuccess) return fail("hipMalloc", e);
            s.capacity = cap;
        }
    }
    for (size_t k = 0; k < ctx->dev.size(); k++)
        if (const uint16_t* t = single_byte_table(ctx->missions[k].c.encoding)) {
            if ((e = hipMalloc((void**)&ctx->dev[k].d_table, 256)) != hipSuccess) return fail("hipMalloc", e);
            if ((e = hipMemcpy(ctx->dev[k].d_table, t, 256, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
        }
    *out = ctx;
    return SX_OK;
}

void sx_destroy(sx_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->host_only && ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        for (auto& d : ctx->dev) {
            if (d.stream) (void)hipStreamSynchronize(d.stream);
            if (d.stream_b) (void)hipStreamSynchronize(d.stream_b);
            for (ScanSlot& s : d.slot) {
                if (s.d_recs) (void)hipFree(s.d_recs);
                if (s.d_cnt) (void)hipFree(s.d_cnt);
                if (s.d_packed) (void)hipFree(s.d_packed);
                if (s.d_counters) (void)hipFree(s.d_counters);
                if (s.ev0) (void)hipEventDestroy(s.ev0);
                if (s.ev1) (void)hipEventDestroy(s.ev1);
                if (s.ev_free) (void)hipEventDestroy(s.ev_free);
            }
            if (d.d_table) (void)hipFree(d.d_table);
            for (void* q : d.d_rp) if (q) (void)hipFree(q);
            if (d.h_runs) (void)hipHostFree(d.h_runs);
            if (d.stream && d.stream != ctx->scan_stream) (void)hipStreamDestroy(d.stream);
        }
        for (int i = 0; i < 2; i++) {
            if (ctx->ing_pin[i]) (void)hipHostFree(ctx->ing_pin[i]);
            if (ctx->ing_dev[i]) (void)hipFree(ctx->ing_dev[i]);
        }
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->scan_stream) (void)hipStreamDestroy(ctx->scan_stream);
        if (ctx->post_stream) (void)hipStreamDestroy(ctx->post_stream);
        if (ctx->d_input) (void)hipFree(ctx->d_input);
        if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
        if (ctx->h_pin) (void)hipHostFree(ctx->h_pin);
        if (ctx->h_pin2) (void)hipHostFree(ctx->h_pin2);
    }
    delete ctx;
}

const char* sx_last_error(const sx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int sx_reset(sx_ctx* ctx) {
    if (!ctx) return SX_E_INVALID;
    for (size_t k = 0; k < ctx->missions.size(); k++) ctx->states[k].reset(ctx->missions[k]);
    ctx->shard_runs_valid = false;
    return SX_OK;
}

// Device-resident input: download only the byte ranges the replay will look at.
// Downloads what the host part of stage B reads: the buffer's first and last 64 KiB ("base",
// entry and exit of every mission) and the replay ranges of the missions the host replays.
// runs == nullptr: the base only, copied into the view.  skip: missions not to plan for; if the
// plan then holds nothing beyond the base and `base_view` already has it, nothing is done and
// *used_base is set.
static int download_for_replay(sx_ctx* ctx, const uint8_t* d_bytes, uint64_t len,
                               const std::vector<RunList>* runs_opt, SparseDeviceBytes* view,
                               const ReplayJob& job, const std::vector<char>* skip = nullptr,
                               bool* used_base = nullptr) {
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
            if ((skip && (*skip)[k]) || device_replay_wanted(ctx, job, k, runs[k].size())) continue;  // stage B of this mission runs on the device
            const size_t before = rg.size();
            // same partition count as replay_all will use
            const unsigned want_parts = (unsigned)std::min<uint64_t>(replay_threads(ctx), std::max<uint64_t>(1, runs[k].size() / 512));
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

namespace {
struct ResultHolder {
    sx_result* r = new sx_result();
    ~ResultHolder() { delete r; }
    sx_result* release() { sx_result* x = r; r = nullptr; return x; }
};

// Bytes per piece of a large buffer (a multiple of the slice length), or `len` if the buffer is
// scanned in one go.  With pieces, stage A of piece p+1 and p+2 is queued while piece p is
// sorted, joined and replayed.  Measured on MI355X (C3(i), 64 GiB): no gain — the replay
// kernels are latency-bound and run ~3x slower next to a scan kernel that saturates HBM, and
// the host waits on them six times per piece — so the default is one piece; SX_PIECE_MIB
// turns the pipeline on (tests do, to keep it correct for an asynchronous stage B later).
uint64_t piece_bytes(const sx_ctx* ctx, uint64_t len) {
    uint64_t piece = 0;
    if (const char* e = getenv("SX_PIECE_MIB")) piece = (uint64_t)atoll(e) << 20;
    if (piece == 0 || len < 2 * piece) return len;
    return piece / kInputBufLen * kInputBufLen;
}
}  // namespace

namespace {
// Missions in the order their kernels are queued: busiest of the previous buffer first, so that
// its stage B (the longest) overlaps the scans of the others.
void mission_order(sx_ctx* ctx, std::vector<int>* out) {
    const size_t nm = ctx->missions.size();
    std::vector<int>& order = *out;
    order.resize(nm);
    for (size_t k = 0; k < nm; k++) order[k] = (int)k;
    if (ctx->last_runs.size() == nm)
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return ctx->last_runs[(size_t)x] > ctx->last_runs[(size_t)y]; });
}

int sync_streams_and_return(sx_ctx* ctx, int rc) {  // do not leave kernels running on the caller's buffer
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

    // what the host reads whatever the runs are (entry and exit of every mission): fetched
    // before the kernels start, so that the copy does not queue behind them
    int fetch_base(sx_ctx* ctx) {
        base_view.reset();
        if (host_bytes) return SX_OK;
        base_view.reset(new SparseDeviceBytes(ctx, d_bytes));
        ReplayJob none;
        return download_for_replay(ctx, d_bytes, len, nullptr, base_view.get(), none);
    }
    int launch(sx_ctx* ctx) {
        for (int k : order) {
            int rc = stage_a_launch(ctx, { k }, d_bytes, len, { parity[(size_t)k] }, { minc[(size_t)k] }, slot);
            if (rc != SX_OK) return rc;
        }
        return SX_OK;
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
        for (size_t oi = 0; oi < nm; oi++) {
            const size_t k = (size_t)order[oi];
            std::vector<RunList> one;
            int rc = stage_a_finish(ctx, { (int)k }, d_bytes, len, { parity[k] }, { minc[k] }, slot, &one);
            if (rc != SX_OK) return rc;
            (*runs)[k] = std::move(one[0]);
            if (!(*runs)[k].own.empty()) (*runs)[k].use_own();  // the vector moved: point at it again
            ctx->last_runs[k] = (*runs)[k].size();
            if (oi + 1 == nm && after_last_finish && (rc = after_last_finish()) != SX_OK) return rc;
            if (device_replay_wanted(ctx, job, k, (*runs)[k].size())) {
                rc = device_replay_mission(ctx, k, early_view, job, (*runs)[k], &pre.per[k], &pre.ends[k]);
                if (rc != SX_OK) return rc;
                pre.done[k] = 1;
            }
        }
        if (host_bytes) return replay_all(ctx, host_view, job, *runs, into, ends, &pre);
        SparseDeviceBytes view(ctx, d_bytes);
        bool base_is_enough = false;
        int rc = download_for_replay(ctx, d_bytes, len, runs, &view, job, &pre.done, &base_is_enough);
        if (rc != SX_OK) return rc;
        return replay_all(ctx, base_is_enough ? (ByteView&)*base_view : (ByteView&)view, job, *runs, into, ends, &pre);
    }
};
}  // namespace

// One buffer, start to end: stage A on the device, stage B on device and host, the findings in
// print order.
//  * The missions' scan kernels queue up in one stream, busiest mission (of the last buffer)
//    first; as soon as a mission's kernel is done its records are packed and joined and — if its
//    stage B runs on the device — replayed, in the second stream, while the kernels of the
//    remaining missions still scan.  What the host replays follows when all kernels are done.
//  * A large buffer can be cut into pieces (SX_PIECE_MIB) that behave exactly like consecutive
//    sx_scan calls (ScannerState carried from piece to piece) with their kernels queued two deep.
static int scan_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes, uint64_t len, int file_id,
                       int is_last, sx_result** out, uint32_t slice_base0 = 0, sx_result* append_to = nullptr) {
    const double t_begin = now_ms();
    const size_t nm = ctx->missions.size();
    std::vector<uint64_t> stream0(nm);
    for (size_t k = 0; k < nm; k++) stream0[k] = ctx->states[k].stream_bytes;
    std::vector<int> order;
    mission_order(ctx, &order);
    const uint64_t piece = piece_bytes(ctx, len);
    const uint64_t n_pieces = len ? (len + piece - 1) / piece : 1;
    auto make = [&](uint64_t p) {
        BufferScan b;
        const uint64_t off = p * piece;
        b.host_bytes = host_bytes ? host_bytes + off : nullptr;
        b.d_bytes = d_bytes + off;
        b.len = std::min(piece, len - off);
        b.order = order;
        b.slot = (int)(p & 1);
        for (size_t k = 0; k < nm; k++) {
            b.parity.push_back((uint32_t)((stream0[k] + off) & 1));
            b.minc.push_back(ctx->missions[k].long_run);
        }
        return b;
    };
    ResultHolder res;
    std::vector<BufferScan> pieces;
    for (uint64_t p = 0; p < n_pieces; p++) pieces.push_back(make(p));
    int rc = SX_OK;
    uint64_t launched = 0;
    for (; launched < std::min<uint64_t>(2, n_pieces); launched++)
        if ((rc = pieces[launched].launch(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
    // the few bytes the host always reads: a tiny gather in the second stream, it finds room
    // next to the scan kernels within ~0.1 ms
    if ((rc = pieces[0].fetch_base(ctx)) != SX_OK) return sync_streams_and_return(ctx, rc);
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
    ctx->stats.total_ms = now_ms() - t_begin;
    if (!append_to) *out = res.release();
    return SX_OK;
}

static int stream_core(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                       sx_result_fn sink, void* sink_user, sx_result* accumulate, int is_last_at_eof,
                       const uint8_t* direct = nullptr, uint64_t direct_len = 0);
namespace {
struct MemReader { const uint8_t* p; uint64_t len, off; unsigned threads; };
// several threads: one memcpy into pinned memory moves ~10 GB/s, PCIe takes five times that
int64_t read_mem(void* user, uint8_t* dst, uint64_t max_bytes) {
    MemReader& mr = *(MemReader*)user;
    const uint64_t want = std::min<uint64_t>(max_bytes, mr.len - mr.off);
    if (want == 0) return 0;
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(mr.threads, want / (4u << 20)));
    if (nt == 1) memcpy(dst, mr.p + mr.off, want);
    else {
        const uint64_t per = (want / nt + 4095) / 4096 * 4096;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) {
            const uint64_t a0 = std::min<uint64_t>(want, (uint64_t)t * per), b0 = std::min<uint64_t>(want, a0 + per);
            th.emplace_back([=, &mr]() { memcpy(dst + a0, mr.p + mr.off + a0, b0 - a0); });
        }
        for (auto& t : th) t.join();
    }
    mr.off += want;
    return (int64_t)want;
}
}  // namespace

int sx_scan(sx_ctx* ctx, const uint8_t* bytes, uint64_t len, int input_file_id, int is_last_input_buffer,
            sx_result** out) {
    if (!ctx || !out || (!bytes && len)) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Measured (4 GiB, MI355X box): one hipMemcpy from pageable memory + one scan moves 50 GiB/s, the
    // chunked pipeline below 34 GiB/s (its staging memcpy is the bottleneck) — so it is opt-in.
    const uint64_t stream_from = getenv("SX_SCAN_STREAM_MIB") ? (uint64_t)atoll(getenv("SX_SCAN_STREAM_MIB")) << 20 : 0;
    if (stream_from >= kInputBufLen && len >= 2 * stream_from) {
        // the ingest pipeline: pinned staging, the copy of one chunk overlapped with the scan of
        // the chunk before; one result with a segment per chunk
        MemReader mr{ bytes, len, 0, std::max(1u, std::min(8u, usable_cpus() / 2)) };
        ResultHolder res;
        int rc = stream_core(ctx, read_mem, &mr, stream_from, input_file_id, nullptr, nullptr, res.r, is_last_input_buffer);
        if (rc == SX_OK) *out = res.release();
        return rc;
    }
    const double t0 = now_ms();
    if (len > ctx->d_input_cap) {
        if (ctx->d_input) HIP_TRY(ctx, hipFree(ctx->d_input));
        ctx->d_input = nullptr; ctx->d_input_cap = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_input, len));
        ctx->d_input_cap = len;
    }
    if (len) HIP_TRY(ctx, hipMemcpy(ctx->d_input, bytes, len, hipMemcpyHostToDevice));
    const double h2d = now_ms() - t0;
    int rc = scan_common(ctx, bytes ? bytes : (const uint8_t*)"", ctx->d_input, len, input_file_id, is_last_input_buffer, out);
    ctx->stats.h2d_ms = h2d;
    ctx->stats.total_ms += h2d;
    return rc;
}

int sx_scan_device(sx_ctx* ctx, const void* device_bytes, uint64_t len, int input_file_id, int is_last_input_buffer,
                   sx_result** out) {
    if (!ctx || !out || (!device_bytes && len)) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    if ((uintptr_t)device_bytes & 15) { ctx->err = "device_bytes must be 16-byte aligned"; return SX_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return scan_common(ctx, nullptr, (const uint8_t*)device_bytes, len, input_file_id, is_last_input_buffer, out);
}

// Ingest pipeline (reference: the Slicer, src/input.rs:57-167, feeding FindingCollection::from).
// A reader thread fills one of two pinned buffers from the caller's read function and copies
// it to HBM on its own stream while the main thread scans the buffer before; every chunk
// behaves exactly like one sx_scan call (ScannerState carried), its result goes to `sink`,
// which owns it (sx_result_free).  Throughput is what the slowest of read / PCIe / scan allows.
// `accumulate`: instead of handing every chunk's result to the sink, append them all to this one
// (slice indices running on), the last chunk with `is_last_at_eof` — that is sx_scan for a large
// host buffer.
// `direct`: the whole input is addressable host memory (a mapped file): no staging buffer, the
// chunks are copied to HBM straight from there (HIP's pageable-memory path) and the host part of
// stage B reads them in place.
static int stream_core(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                       sx_result_fn sink, void* sink_user, sx_result* accumulate, int is_last_at_eof,
                       const uint8_t* direct, uint64_t direct_len) {
    const double t_begin = now_ms();
    if (chunk_bytes == 0) chunk_bytes = 256ull << 20;
    chunk_bytes = std::max<uint64_t>(kInputBufLen, chunk_bytes / kInputBufLen * kInputBufLen);
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    if (ctx->ing_dev_cap < chunk_bytes) {
        for (int i = 0; i < 2; i++) {
            if (ctx->ing_dev[i]) HIP_TRY(ctx, hipFree(ctx->ing_dev[i]));
            ctx->ing_dev[i] = nullptr;
        }
        ctx->ing_dev_cap = 0;
        for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipMalloc((void**)&ctx->ing_dev[i], chunk_bytes));
        ctx->ing_dev_cap = chunk_bytes;
    }
    if (!direct && ctx->ing_cap < chunk_bytes) {
        for (int i = 0; i < 2; i++) {
            if (ctx->ing_pin[i]) HIP_TRY(ctx, hipHostFree(ctx->ing_pin[i]));
            ctx->ing_pin[i] = nullptr;
        }
        ctx->ing_cap = 0;
        for (int i = 0; i < 2; i++) HIP_TRY(ctx, hipHostMalloc((void**)&ctx->ing_pin[i], chunk_bytes, hipHostMallocDefault));
        ctx->ing_cap = chunk_bytes;
    }
    struct Slot { uint64_t n = 0; bool ready = false, eof = false; int error = 0; const uint8_t* host = nullptr; };
    uint64_t direct_off = 0;
    Slot slots[2];
    std::mutex mu;
    std::condition_variable cv;
    bool stop = false;
    std::string reader_err;
    uint8_t carry = 0;
    bool carry_valid = false;
    std::thread reader([&]() {
        (void)hipSetDevice(ctx->device);
        for (uint64_t k = 0;; k++) {
            Slot& s = slots[k & 1];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !s.ready || stop; });
                if (stop) return;
            }
            uint64_t n = 0;
            bool eof = false;
            int error = 0;
            const uint8_t* host = ctx->ing_pin[k & 1];
            if (direct) {
                host = direct + direct_off;
                n = std::min<uint64_t>(chunk_bytes, direct_len - direct_off);
                direct_off += n;
                eof = direct_off >= direct_len;
            } else {
            if (carry_valid) { ctx->ing_pin[k & 1][0] = carry; n = 1; carry_valid = false; }
            while (n < chunk_bytes) {
                const int64_t got = read(read_user, ctx->ing_pin[k & 1] + n, chunk_bytes - n);
                if (got < 0) { error = (int)got; break; }
                if (got == 0) { eof = true; break; }
                n += (uint64_t)got;
            }
            if (!error && !eof && accumulate) {  // is this the last chunk?  (only then may it carry is_last)
                const int64_t got = read(read_user, &carry, 1);
                if (got < 0) error = (int)got;
                else if (got == 0) eof = true;
                else carry_valid = true;
            }
            }
            if (!error && n) {
                hipError_t e = hipMemcpyAsync(ctx->ing_dev[k & 1], host, n, hipMemcpyHostToDevice, ctx->copy_stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
                if (e != hipSuccess) { error = SX_E_HIP; reader_err = std::string("H2D copy: ") + hipGetErrorString(e); }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                s.n = n; s.eof = eof || error; s.error = error; s.host = host; s.ready = true;
            }
            cv.notify_all();
            if (eof || error) return;
        }
    });
    int rc = SX_OK;
    uint64_t done_bytes = 0;
    for (uint64_t k = 0;; k++) {
        Slot& s = slots[k & 1];
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return s.ready; });
        }
        if (s.error) { rc = s.error < 0 && s.error >= SX_E_STATE ? s.error : SX_E_INVALID; ctx->err = reader_err.empty() ? "read function failed" : reader_err; break; }
        if (accumulate) {
            if (s.n || (s.eof && is_last_at_eof)) {
                rc = scan_common(ctx, s.host, ctx->ing_dev[k & 1], s.n, input_file_id, s.eof ? is_last_at_eof : 0, nullptr,
                                 (uint32_t)(done_bytes / kInputBufLen), accumulate);
                if (rc != SX_OK) break;
            }
        } else if (s.n) {
            sx_result* r = nullptr;
            rc = scan_common(ctx, s.host, ctx->ing_dev[k & 1], s.n, input_file_id, 0, &r);
            if (rc != SX_OK) break;
            const int src = sink(sink_user, r);
            if (src != 0) { rc = SX_E_INVALID; ctx->err = "the result sink asked to stop"; break; }
        }
        done_bytes += s.n;
        const bool last = s.eof;
        {
            std::lock_guard<std::mutex> lk(mu);
            s.ready = false;
        }
        cv.notify_all();
        if (last) break;
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
    }
    cv.notify_all();
    reader.join();
    ctx->stats.total_ms = now_ms() - t_begin;
    return rc;
}

int sx_scan_stream(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                   sx_result_fn sink, void* sink_user) {
    if (!ctx || !read || !sink) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return stream_core(ctx, read, read_user, chunk_bytes, input_file_id, sink, sink_user, nullptr, 0);
}

namespace {
struct FileReader {
    int fd;
    bool seekable;
    uint64_t off, size;
    unsigned threads;
};
// A regular file is read with several pread(2) threads (one thread copies ~10 GB/s from the page
// cache, PCIe takes five times that); pipes and stdin with plain read(2).
int64_t read_fd(void* user, uint8_t* dst, uint64_t max_bytes) {
    FileReader& fr = *(FileReader*)user;
    if (!fr.seekable) return (int64_t)::read(fr.fd, dst, (size_t)std::min<uint64_t>(max_bytes, 1ull << 30));
    if (fr.off >= fr.size) return 0;
    const uint64_t want = std::min<uint64_t>(max_bytes, fr.size - fr.off);
    const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(fr.threads, want / (4u << 20)));
    const uint64_t per = (want / nt + 4095) / 4096 * 4096;
    std::vector<int64_t> got(nt, 0);
    auto work = [&](unsigned t) {
        const uint64_t a = std::min<uint64_t>(want, (uint64_t)t * per), b = std::min<uint64_t>(want, a + per);
        uint64_t done = 0;
        while (a + done < b) {
            const ssize_t n = ::pread(fr.fd, dst + a + done, (size_t)std::min<uint64_t>(b - a - done, 1ull << 30), (off_t)(fr.off + a + done));
            if (n < 0) { got[t] = -1; return; }
            if (n == 0) break;
            done += (uint64_t)n;
        }
        got[t] = (int64_t)done;
    };
    if (nt == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
        for (auto& t : th) t.join();
    }
    uint64_t total = 0;
    for (unsigned t = 0; t < nt; t++) {
        if (got[t] < 0) return -1;
        total += (uint64_t)got[t];
        if ((uint64_t)got[t] < std::min<uint64_t>(want, (uint64_t)(t + 1) * per) - std::min<uint64_t>(want, (uint64_t)t * per)) break;  // the file shrank
    }
    fr.off += total;
    return (int64_t)total;
}
}  // namespace

// The same for one file (path "-" = stdin).
int sx_scan_file(sx_ctx* ctx, const char* path, uint64_t chunk_bytes, int input_file_id, sx_result_fn sink, void* sink_user) {
    if (!ctx || !path || !sink) return SX_E_INVALID;
    FileReader fr{ strcmp(path, "-") == 0 ? 0 : ::open(path, O_RDONLY), false, 0, 0, std::max(1u, std::min(8u, usable_cpus() / 2)) };
    if (fr.fd < 0) { ctx->err = std::string("cannot open `") + path + "`: " + strerror(errno); return SX_E_INVALID; }
    struct stat st;
    if (fr.fd > 0 && fstat(fr.fd, &st) == 0 && S_ISREG(st.st_mode)) { fr.seekable = true; fr.size = (uint64_t)st.st_size; }
    if (fr.seekable && fr.size > 0 && getenv("SX_INGEST_MMAP")) {
        // opt-in: map the file and copy to HBM straight from the page cache.  Measured slower than the
        // pread threads + pinned staging (19 vs 27 GiB/s on 16 GiB): the page faults of the mapping cost more
        // than the staging copy.
        void* map = mmap(nullptr, fr.size, PROT_READ, MAP_PRIVATE, fr.fd, 0);
        if (map != MAP_FAILED) {
            (void)madvise(map, fr.size, MADV_SEQUENTIAL);
            begin_call(ctx);
            int rc = SX_E_STATE;
            if (ctx->host_only) ctx->err = "host-only context: no device scan";
            else if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; rc = SX_E_HIP; }
            else rc = stream_core(ctx, nullptr, nullptr, chunk_bytes, input_file_id, sink, sink_user, nullptr, 0, (const uint8_t*)map, fr.size);
            munmap(map, fr.size);
            ::close(fr.fd);
            return rc;
        }
    }
    const int rc = sx_scan_stream(ctx, read_fd, &fr, chunk_bytes, input_file_id, sink, sink_user);
    if (fr.fd > 0) ::close(fr.fd);
    return rc;
}

int sx_device_runs(sx_ctx* ctx, int mission_index, const void* device_bytes, uint64_t len, int stream_parity,
                   uint64_t min_chars, sx_run** runs, uint64_t* n_runs) {
    if (!ctx || !runs || !n_runs || mission_index < 0 || (size_t)mission_index >= ctx->missions.size()) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    if ((uintptr_t)device_bytes & 15) { ctx->err = "device_bytes must be 16-byte aligned"; return SX_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<RunList> out;
    ctx->shard_runs_valid = false;  // the mission's device-side run list is about to be replaced
    int rc = device_runs(ctx, { mission_index }, (const uint8_t*)device_bytes, len, { (uint32_t)(stream_parity & 1) },
                         { min_chars }, &out);
    if (rc != SX_OK) return rc;
    *n_runs = out[0].size();
    *runs = (sx_run*)malloc(sizeof(sx_run) * (out[0].size() ? out[0].size() : 1));
    if (!*runs) return SX_E_NOMEM;
    memcpy(*runs, out[0].data(), sizeof(sx_run) * out[0].size());
    return SX_OK;
}

int sx_replay_runs(sx_ctx* ctx, const uint8_t* bytes, uint64_t len, int input_file_id, int is_last_input_buffer,
                   const sx_run* const* runs, const uint64_t* n_runs, sx_result** out) {
    if (!ctx || !out || (!bytes && len) || !runs || !n_runs) return SX_E_INVALID;
    begin_call(ctx);
    std::vector<RunList> r(ctx->missions.size());
    for (size_t k = 0; k < r.size(); k++) r[k].assign(runs[k], runs[k] + n_runs[k]);
    HostBytes view(bytes ? bytes : (const uint8_t*)"");
    ResultHolder res;
    int rc = replay_all(ctx, view, whole_chunk_job(ctx, len, input_file_id, is_last_input_buffer != 0), r, &res.r->r, nullptr);
    if (rc == SX_OK) *out = res.release();
    return rc;
}

static int shard_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes,
                        const sx_run* const* given_runs, const uint64_t* given_n, uint64_t buf_off, uint64_t buf_len,
                        uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at, uint64_t file_stream_off, int file_id,
                        int reuse_runs, sx_result** out, uint64_t* end_pos) {
    if (!out || !end_pos || (buf_off % kInputBufLen) != 0 || own_lo < buf_off || own_hi < own_lo || own_hi > buf_off + buf_len) {
        ctx->err = "bad shard geometry"; return SX_E_INVALID;
    }
    const size_t nm = ctx->missions.size();
    const double t_begin = now_ms();
    if (given_runs) {
        ctx->shard_runs.assign(nm, RunList{});
        for (size_t k = 0; k < nm; k++) ctx->shard_runs[k].assign(given_runs[k], given_runs[k] + given_n[k]);
    }
    const bool scan_now = !given_runs && !(reuse_runs && ctx->shard_runs_valid);
    BufferScan b;
    if (scan_now) {
        b.host_bytes = host_bytes; b.d_bytes = d_bytes; b.len = buf_len; mission_order(ctx, &b.order); b.slot = 0;
        b.parity.assign(nm, (uint32_t)((file_stream_off + buf_off) & 1));
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
        ctx->shard_runs_valid = true;
    } else if (host_bytes) {
        ctx->shard_runs_valid = true;
        HostBytes view(host_bytes);
        rc = replay_all(ctx, view, job, ctx->shard_runs, &res.r->r, ends.data());
    } else {
        ctx->shard_runs_valid = true;
        SparseDeviceBytes view(ctx, d_bytes);
        rc = download_for_replay(ctx, d_bytes, buf_len, &ctx->shard_runs, &view, job);
        if (rc != SX_OK) return rc;
        rc = replay_all(ctx, view, job, ctx->shard_runs, &res.r->r, ends.data());
    }
    if (rc == SX_OK) *out = res.release();
    for (size_t k = 0; k < nm; k++) end_pos[k] = buf_off + ends[k];
    ctx->stats.total_ms = now_ms() - t_begin;
    return rc;
}

int sx_scan_shard_device(sx_ctx* ctx, const void* device_bytes, uint64_t buf_off, uint64_t buf_len, uint64_t own_lo,
                         uint64_t own_hi, const uint64_t* start_at, uint64_t file_stream_off, int input_file_id,
                         int reuse_runs, sx_result** out, uint64_t* end_pos) {
    if (!ctx || (!device_bytes && buf_len)) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    if ((uintptr_t)device_bytes & 15) { ctx->err = "device_bytes must be 16-byte aligned"; return SX_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return shard_common(ctx, nullptr, (const uint8_t*)device_bytes, nullptr, nullptr, buf_off, buf_len, own_lo, own_hi, start_at,
                        file_stream_off, input_file_id, reuse_runs, out, end_pos);
}

int sx_scan_shard(sx_ctx* ctx, const uint8_t* bytes, uint64_t buf_off, uint64_t buf_len, uint64_t own_lo, uint64_t own_hi,
                  const uint64_t* start_at, uint64_t file_stream_off, int input_file_id, int reuse_runs, sx_result** out,
                  uint64_t* end_pos) {
    if (!ctx || (!bytes && buf_len)) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!(reuse_runs && ctx->shard_runs_valid)) {
        if (buf_len > ctx->d_input_cap) {
            if (ctx->d_input) HIP_TRY(ctx, hipFree(ctx->d_input));
            ctx->d_input = nullptr; ctx->d_input_cap = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_input, buf_len));
            ctx->d_input_cap = buf_len;
        }
        if (buf_len) HIP_TRY(ctx, hipMemcpy(ctx->d_input, bytes, buf_len, hipMemcpyHostToDevice));
    }
    return shard_common(ctx, bytes ? bytes : (const uint8_t*)"", ctx->d_input, nullptr, nullptr, buf_off, buf_len, own_lo, own_hi,
                        start_at, file_stream_off, input_file_id, reuse_runs, out, end_pos);
}

int sx_replay_shard_runs(sx_ctx* ctx, const uint8_t* bytes, uint64_t buf_off, uint64_t buf_len, uint64_t own_lo,
                         uint64_t own_hi, const uint64_t* start_at, uint64_t file_stream_off, int input_file_id,
                         const sx_run* const* runs, const uint64_t* n_runs, sx_result** out, uint64_t* end_pos) {
    if (!ctx || (!bytes && buf_len) || !runs || !n_runs) return SX_E_INVALID;
    begin_call(ctx);
    return shard_common(ctx, bytes ? bytes : (const uint8_t*)"", nullptr, runs, n_runs, buf_off, buf_len, own_lo, own_hi, start_at,
                        file_stream_off, input_file_id, 0, out, end_pos);
}

uint64_t sx_result_count(const sx_result* r) { return r ? r->r.count() : 0; }
uint64_t sx_result_segments(const sx_result* r) { return r ? r->r.segs.size() : 0; }
int sx_result_segment(const sx_result* r, uint64_t i, const sx_finding** findings, uint64_t* n_findings,
                      const uint8_t** arena, uint64_t* arena_len) {
    if (!r || i >= r->r.segs.size()) return SX_E_INVALID;
    const MissionFindings& s = r->r.segs[(size_t)i];
    if (findings) *findings = s.data();
    if (n_findings) *n_findings = s.count();
    if (arena) *arena = (const uint8_t*)s.strings();
    if (arena_len) *arena_len = s.strings_len();
    return SX_OK;
}
// The contiguous view: joins the segments on first use (a copy; none if there is one segment).
const sx_finding* sx_result_findings(const sx_result* r) {
    if (!r || !const_cast<sx_result*>(r)->r.flatten(nullptr) || r->r.segs.empty()) return nullptr;
    return r->r.segs[0].data();
}
const uint8_t* sx_result_arena(const sx_result* r, uint64_t* len) {
    if (len) *len = 0;
    if (!r || !const_cast<sx_result*>(r)->r.flatten(nullptr) || r->r.segs.empty()) return nullptr;
    if (len) *len = r->r.segs[0].strings_len();
    return (const uint8_t*)r->r.segs[0].strings();
}
void sx_result_free(sx_result* r) { delete r; }

int sx_print_findings(const sx_ctx* ctx, const sx_result* r, int n_inputs, int radix, int no_metadata, uint8_t** out,
                      uint64_t* out_len) {
    if (!ctx || !r || !out || !out_len) return SX_E_INVALID;
    if (radix != 0 && radix != 'x' && radix != 'd' && radix != 'o') return SX_E_INVALID;
    std::string s;
    print_findings(ctx->missions, r->r, n_inputs, radix, no_metadata != 0, &s);
    *out = (uint8_t*)malloc(s.size() ? s.size() : 1);
    if (!*out) return SX_E_NOMEM;
    memcpy(*out, s.data(), s.size());
    *out_len = s.size();
    return SX_OK;
}

int sx_get_stats(const sx_ctx* ctx, sx_stats* out) {
    if (!ctx || !out) return SX_E_INVALID;
    *out = ctx->stats;
    return SX_OK;
}

void sx_free(void* p) { free(p); }

int sx_fill_background_device(sx_ctx* ctx, void* device_bytes, uint64_t first_byte_index, uint64_t len, uint64_t seed) {
    if (!ctx || ctx->host_only) return SX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_fill_background((uint8_t*)device_bytes, first_byte_index, len, seed, ctx->dev[0].stream_b));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->dev[0].stream_b));
    return SX_OK;
}

int sx_device_alloc(sx_ctx* ctx, uint64_t bytes, void** device_ptr) {
    if (!ctx || ctx->host_only || !device_ptr) return SX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(device_ptr, bytes ? bytes : 1));
    return SX_OK;
}
int sx_device_free(sx_ctx* ctx, void* device_ptr) {
    if (!ctx || ctx->host_only) return SX_E_STATE;
    HIP_TRY(ctx, hipFree(device_ptr));
    return SX_OK;
}
int sx_device_upload(sx_ctx* ctx, void* device_dst, const void* host_src, uint64_t bytes) {
    if (!ctx || ctx->host_only) return SX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(device_dst, host_src, bytes, hipMemcpyHostToDevice));
    return SX_OK;
}
int sx_device_download(sx_ctx* ctx, void* host_dst, const void* device_src, uint64_t bytes) {
    if (!ctx || ctx->host_only) return SX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
    return SX_OK;
}

int sx_device_read_bandwidth(sx_ctx* ctx, const void* device_bytes, uint64_t len, int repeats, double* gbytes_per_s) {
    if (!ctx || ctx->host_only || !gbytes_per_s || repeats == 0) return SX_E_STATE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    MissionDev& d = ctx->dev[0];
    uint64_t* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_out, 8));
    HIP_TRY(ctx, hipMemsetAsync(d_out, 0, 8, d.stream));
    // repeats < 0: probe with the scan kernels' traversal (private sub-chunk per wavefront) instead of grid-stride
    const uint32_t sub = repeats < 0 ? (ctx->opt.subchunk_bytes ? ctx->opt.subchunk_bytes : 256u * 1024u) : 0u;
    if (repeats < 0) repeats = -repeats;
    HIP_TRY(ctx, launch_read_sum((const uint8_t*)device_bytes, len, d_out, d.stream, sub));  // warm-up
    float best = 1e30f;
    for (int i = 0; i < repeats; i++) {
        HIP_TRY(ctx, hipEventRecord(d.slot[0].ev0, d.stream));
        HIP_TRY(ctx, launch_read_sum((const uint8_t*)device_bytes, len, d_out, d.stream, sub));
        HIP_TRY(ctx, hipEventRecord(d.slot[0].ev1, d.stream));
        HIP_TRY(ctx, hipStreamSynchronize(d.stream));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, d.slot[0].ev0, d.slot[0].ev1));
        if (ms < best) best = ms;
    }
    (void)hipFree(d_out);
    *gbytes_per_s = (double)(len / 16 * 16) / (best * 1e-3) / 1e9;
    return SX_OK;
}

}  // extern "C"
