// sx_ctx.hpp — what the translation units behind the C-ABI share: the context (HIP resources,
// carried ScannerStates, grow-only buffers), the per-mission device state, and the internal
// interfaces of stage A (sx_stage_a.cpp), stage B (sx_stage_b.cpp), the schedule that ties them
// together (sx_schedule.cpp) and the ingest pipeline (sx_ingest.cpp).  Nothing here is public.
#pragma once
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
#include <unordered_map>
#include <vector>

#include "sx_host.hpp"

namespace sx {

double now_ms();
constexpr size_t kSmallReadBytes = 4096;
// SX_TIMELINE=1: host-side marks on stderr, milliseconds since the current scan call began (round 5: where a step's time goes)
extern double g_tl_t0;
extern int g_tl_on;
#define SX_TL(...) do { if (sx::g_tl_on) { fprintf(stderr, "[tl %+8.3f] ", sx::now_ms() - sx::g_tl_t0); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)

// The long runs of one mission over one buffer: owned (host merge, caller-supplied) or a view
// of the mission's pinned download buffer; `on_device` says MissionDev::d_rp[0] holds the same list.
struct RunList {
    std::vector<sx_run> own;
    const sx_run* p = nullptr;
    size_t n = 0;
    bool on_device = false;
    // the records were only counted (n = their number): the buffer is string-dense and the Mission's stage B replays every
    // window (sx_wave.cpp) — nobody reads the runs.  If that stage B gives up, stage A is finished again, in full.
    bool skipped = false;
    const sx_run* dev_ptr = nullptr;   // where the list lies on the device (on_device)
    // A list joined on the device: p[] (pinned) is filled by a copy that is only started when somebody
    // asks for it — the device replay asks after its first pass is launched, so that the copy (a blit
    // kernel on this stack) does not run next to the short kernels in front of that pass.
    const void* dev_src = nullptr; size_t copy_bytes = 0;
    hipStream_t copy_stream = nullptr; hipEvent_t ready = nullptr;
    mutable bool issued = false;
    hipError_t start_copy() const {
        if (!dev_src || issued || !copy_bytes) return hipSuccess;
        hipError_t e = hipMemcpyAsync(const_cast<sx_run*>(p), dev_src, copy_bytes, hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipEventRecord(ready, copy_stream);
        issued = true;
        return e;
    }
    // Only the first and last `edge` runs (what the host's share of a device replay reads: the buffer's entry region
    // and its exit state) — the middle of p[] stays unwritten.  full() says whether everything is there.
    mutable bool complete = false, edges_done = false;
    bool full() const { return !dev_src || !copy_bytes || complete; }
    hipError_t fetch_edges(size_t edge) const {
        if (full() || issued) return wait();
        if (n <= 2 * edge) return wait();
        if (edges_done) return hipSuccess;
        edges_done = true;
        hipError_t e = hipMemcpyAsync(const_cast<sx_run*>(p), dev_src, edge * sizeof(sx_run), hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(const_cast<sx_run*>(p) + (n - edge), (const sx_run*)dev_src + (n - edge), edge * sizeof(sx_run),
                               hipMemcpyDeviceToHost, copy_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(copy_stream);
        return e;
    }
    // the list's contents are on the host from here on (its size always is)
    hipError_t wait() const {
        hipError_t e = start_copy();
        if (e == hipSuccess && dev_src && copy_bytes) e = hipEventSynchronize(ready);
        if (e == hipSuccess) complete = true;
        return e;
    }
    void use_own() { p = own.data(); n = own.size(); on_device = false; dev_src = nullptr; copy_bytes = 0; complete = true; }
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
    uint32_t* d_counters = nullptr;   // kCounterWords x u32 (sx_device.hpp): records, (heavy tiles), (joined runs), -, then the statistics' shards
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // around the scan kernel
    hipEvent_t ev_free = nullptr;     // the slot's records have been consumed (recorded on stream_b)
    bool free_pending = false;
    // region mode (ScanParams::region_cap): per-sub-chunk counts and the packed, ordered records
    uint32_t* d_cnt = nullptr;  uint64_t cnt_cap = 0;
    uint32_t* d_grid = nullptr; uint64_t grid_cap = 0;   // ScanParams::grid_flags (double-byte Missions)
    DevRun* d_packed = nullptr; uint64_t packed_cap = 0;
    uint32_t region_cap = 0;    // of the launch in flight (0: shared pool)
    uint64_t n_regions = 0;
    bool fused = false, fused_first = false;   // the launch in flight is a fused one (sx_fused.hip) shared with other Missions; ... and this is its first slot
};
struct MissionDev {
    hipStream_t stream = nullptr;     // scan kernels only
    hipStream_t stream_b = nullptr;   // everything after them (sort/join, stage B, copies); higher priority
    ScanSlot slot[2];
    // stage B on the device: grow-only buffers
    uint16_t* d_table = nullptr;                        // decoder table: single byte (128 entries) or the Big5 / EUC-JP blob
    uint32_t* d_pair_lut = nullptr;                     // Big5 / EUC-JP: Mission::pair_lut for the scan kernel
    uint32_t* d_wave_pairs = nullptr;                   // ... two-byte family: Mission::wave_pairs
    uint32_t* d_wave_pairs2 = nullptr;                  // ... Mission::wave_pairs2
    uint8_t* d_wave_lut = nullptr;                      // wave-cooperative stage B: Mission::wave_lut (uploaded at its first use)
    sx_run* h_runs = nullptr; uint64_t h_runs_cap = 0;   // pinned: runs joined on the device
    hipEvent_t ev_runs = nullptr;                         // their copy (on sx_ctx::d2h_stream) is done
    void* d_rp[12] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };  // runs, region outs, idx, fbase, abase, findings+arena
    uint64_t d_rp_cap[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };  // + stitch blocks, totals, pass-1 output cache, runs cut into pieces, the wave kernels' own scratch (10)
    // the wave-cooperative stage B of a Mission that runs next to the other Missions' stage B (sx_schedule.cpp: a host thread each):
    // its own stream, totals in its own pinned words, its own timing events (4 per slab: count begin / end, write begin / end)
    hipStream_t stream_w = nullptr;
    uint64_t* h_tot = nullptr;
    // the token grid the last scan kernel of this Mission published (two-byte families): stage B's walk back to a token boundary ends at a
    // sub-chunk start (round 5).  Valid for the buffer (data pointer, length) it was taken from
    const uint32_t* grid_of_scan = nullptr; uint32_t grid_sub = 0; const uint8_t* grid_data = nullptr; uint64_t grid_len = 0;
    uint8_t* h_small = nullptr;        // pinned, kSmallReadBytes: the target of read_back_sync (sx_api.cpp)
    std::vector<hipEvent_t> wave_ev;
};


}  // namespace sx

struct sx_result {
    sx::Result r;
};

struct sx_ctx {
    std::shared_ptr<sx::PinnedPool> pool = std::make_shared<sx::PinnedPool>();
    std::shared_ptr<std::atomic<uint64_t>> dev_epoch = std::make_shared<std::atomic<uint64_t>>(0);   // SX_OPT_RESULT_ON_DEVICE: advanced by every scan call
    std::vector<sx::Mission> missions;
    std::vector<sx::ScannerState> states;
    std::vector<sx::MissionDev> dev;
    bool host_only = false;
    bool sharded_call = false, single_piece = true;   // (SX_OPT_RESULT_ON_DEVICE applies to plain scans of one piece only)
    int device = -1;
    sx_options opt{};
    std::string err;
    sx_stats stats{};
    hipStream_t scan_stream = nullptr, post_stream = nullptr;
    uint8_t* d_cache = nullptr;   // stage B, pass 1's output cache (sx_stage_b.cpp)
    uint64_t d_cache_cap = 0;
    hipStream_t merge_copy_stream = nullptr;   // device_merge: the copy of one part next to the sort of the following one
    hipEvent_t merge_ev[3] = { nullptr, nullptr, nullptr };
    // device_merge's own memory (two output buffers of merge_out_room bytes each, then the sort's scratch): the copy of a part may
    // still read it while the next piece of the buffer is scanned and replayed (merge_async, scan_common's sequential pieces)
    uint8_t* d_merge = nullptr; uint64_t d_merge_cap = 0, merge_out_room = 0, merge_n_out = 0;
    bool merge_async = false;                           // device_merge returns with its last copies in flight; merge_drain() waits
    bool merge_copy_pending[2] = { false, false };      // a copy out of output buffer 0 / 1 was queued and not waited for
    uint64_t merge_parts = 0;                           // parts merged so far (their parity picks the output buffer)
    uint64_t merged_out_bytes = 0;                      // bytes device_merge sent to the host (all calls)
    double out_density = 0;                             // ... per input byte of the last whole buffer: sizes the next one's pieces
    hipEvent_t ev_interleaved = nullptr;                // merge_async: the last interleave that read the Missions' findings (on post_stream)
    std::atomic<bool> interleave_pending{ false };      // ... recorded and not yet waited for by merge_drain: writers on other streams wait for it
    std::mutex mu;                                      // statistics and the like, when Missions' stage B run on host threads next to each other
    // ... and what those threads share beyond statistics: the error text (HIP_TRY on any thread -> set_err) and the
    // merger's in-flight state, which ensure_rp / ensure_scratch -> merge_drain settle before memory a queued copy may read is freed
    std::mutex err_mu;
    std::recursive_mutex grow_mu;
    void set_err(std::string text) { std::lock_guard<std::mutex> g(err_mu); err = std::move(text); }
    unsigned n_cus = 256, scan_blocks_per_cu = 8;
    // sx_scan_stream: two pinned host buffers and two device buffers, filled by a reader thread
    hipStream_t copy_stream = nullptr;
    hipStream_t d2h_stream = nullptr;   // run lists travel to the host while stage B's first pass runs
    uint8_t* ing_pin[2] = { nullptr, nullptr };
    uint8_t* ing_dev[2] = { nullptr, nullptr };
    uint64_t ing_cap = 0, ing_dev_cap = 0;
    uint32_t region_cap = 64;         // record slots per sub-chunk in region mode (0: never use it).  Round 4: 32 -> 64 — C2 (17 records per sub-chunk on average) overflowed some regions and fell into the large-region mode with its radix sort: 2.11 -> 1.95 ms per step
    // per mission, from the last buffer: 0 = few records (small regions, packed in order); 1 = shared pool + sort;
    // > 1 = string-dense: regions of this many slots (no atomics in the scan kernel), sorted like the pool
    std::vector<uint32_t> dense;
    uint8_t* d_input = nullptr;  // staging for host input
    uint64_t d_input_cap = 0;
    uint64_t ondemand_fetches = 0;
    // grow-only scratch reused by every call (pinned host memory: D2H at full PCIe rate)
    std::vector<uint64_t> last_runs;  // long runs per mission of the last scanned buffer: busiest mission scans first
    std::vector<char> wave_pred;      // per mission: its last whole buffer was string-dense and went through the wave kernels: the next one does without stage A
    std::vector<double> wave_density; // per mission: findings per input byte of the last buffer that went through the wave kernels (sizes the descriptors)
    std::vector<char> wave_off;       // per mission, for the buffer in hand: the wave-cooperative stage B gave up on it (sx_wave.cpp)
    std::vector<sx::RunList> shard_runs;  // device runs of the last sx_scan_shard* buffer (reuse_runs)
    bool shard_runs_valid = false;
    // ... of which buffer: reuse_runs only counts for the very same one (ADVICE, round 1)
    uint64_t shard_runs_off = 0, shard_runs_len = 0;
    const void* shard_runs_ptr = nullptr;
    uint8_t* h_pin = nullptr;   uint64_t h_pin_cap = 0;
    uint8_t* h_pin2 = nullptr;  uint64_t h_pin2_cap = 0;   // device replay traffic (h_pin may back a live byte view)
    uint8_t* d_scratch = nullptr; uint64_t d_scratch_cap = 0;
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->set_err(std::string(#expr) + ": " + hipGetErrorString(e_));                 \
            return SX_E_HIP;                                                                   \
        }                                                                                      \
    } while (0)

namespace sx {

// Device-resident chunk of which only some byte ranges were downloaded.
class SparseDeviceBytes : public ByteView {
public:
    // (len: the buffer's length if the caller knows it — on-demand fetches then come in blocks of 64 KiB, round 5; 0: exactly what is asked for)
    SparseDeviceBytes(sx_ctx* ctx, const uint8_t* d_base, uint64_t len = 0) : ctx_(ctx), d_base_(d_base), len_(len) {}
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
        // rare: the replay ran further than planned.  Round 5: a block of 64 KiB around what is asked for, kept — a walk through a fill of
        // lead-range bytes (the two-byte family's way back to a token boundary, the exit state of a buffer that ends in one) asked for its
        // bytes four at a time: 272 000 copies of 4 bytes for 300 KB, 3.6 us each.  (Not across a block's end: then exactly what is asked for.)
        std::lock_guard<std::mutex> g(mu_);
        if (len_) {
            const uint64_t b = off / kBlock;
            if (off + n <= std::min<uint64_t>(len_, (b + 1) * kBlock)) {
                auto it = blocks_.find(b);
                if (it == blocks_.end()) {
                    const uint64_t lo = b * kBlock, hi = std::min<uint64_t>(len_, lo + kBlock);
                    std::vector<uint8_t> v((size_t)(hi - lo));
                    if (hipMemcpy(v.data(), d_base_ + lo, (size_t)(hi - lo), hipMemcpyDeviceToHost) != hipSuccess) memset(v.data(), 0, v.size());
                    ctx_->ondemand_fetches++;
                    it = blocks_.emplace(b, std::move(v)).first;
                }
                return it->second.data() + (off - b * kBlock);
            }
        }
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
    uint64_t len_ = 0;
    static constexpr uint64_t kBlock = 65536;
    std::unordered_map<uint64_t, std::vector<uint8_t>> blocks_;   // (node-based: the vectors' storage stays where it is)
    std::vector<Seg> segs_;
    std::deque<std::vector<uint8_t>> extra_, owned_;
    std::mutex mu_;
};

// grow-only buffers of the context (sx_api.cpp)
int ensure_pinned(sx_ctx* ctx, uint64_t bytes);
int ensure_pinned2(sx_ctx* ctx, uint64_t bytes);
// `bytes` (<= kSmallReadBytes, 4-aligned) from device memory to `host_dst` (any host memory), waited for: a one-wavefront kernel into the
// Mission's pinned staging words + a stream sync — not the runtime's blit copy (sx_sort.hip small_copy_kernel).  SX_SMALL_COPY=0: the runtime's
int read_back_sync(sx_ctx* ctx, sx::MissionDev& d, hipStream_t s, void* host_dst, const void* dev_src, size_t bytes);
// the same without the wait, into PINNED host memory (read after the caller's own sync)
int read_back_async(sx_ctx* ctx, hipStream_t s, void* pinned_dst, const void* dev_src, size_t bytes);
int ensure_scratch(sx_ctx* ctx, uint64_t bytes);
int ensure_capacity(sx_ctx* ctx, ScanSlot& s, uint32_t cap);
int ensure_rp(sx_ctx* ctx, MissionDev& d, int slot, uint64_t bytes);
int ensure_cache(sx_ctx* ctx, uint64_t bytes);
unsigned usable_cpus();
unsigned replay_threads(const sx_ctx* ctx);
void begin_call(sx_ctx* ctx);

// ---- stage A (sx_stage_a.cpp)
ScanParams scan_params(const sx_ctx* ctx, int mission, const ScanSlot& s, const uint8_t* d_bytes, uint64_t len,
                       uint32_t parity, uint64_t min_chars);
int stage_a_launch(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                   const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars, int si);
struct ReplayJob;
int ensure_copy_stream(sx_ctx* ctx);   // sx_stage_b.cpp
int merge_drain(sx_ctx* ctx);          // sx_stage_b.cpp: waits for what device_merge left in flight (merge_async)
int stage_a_finish(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                   const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars, int si,
                   std::vector<RunList>* out, bool cut_into_pieces = false, const ReplayJob* wave_job = nullptr);
int device_runs(sx_ctx* ctx, const std::vector<int>& which, const uint8_t* d_bytes, uint64_t len,
                const std::vector<uint32_t>& parity, const std::vector<uint64_t>& min_chars,
                std::vector<RunList>* out);

// ---- stage B (sx_stage_b.cpp)
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




// Missions whose stage B already ran (on the device, while later missions were still being scanned).
struct PreReplayed {
    std::vector<char> done;
    std::vector<MissionFindings> per;
    std::vector<uint64_t> ends;
    explicit PreReplayed(size_t nm) : done(nm, 0), per(nm), ends(nm, 0) {}
};


// wave-cooperative stage B (sx_wave.cpp): for whole buffers of a Mission it covers, when the buffer is string-dense
constexpr int SX_WAVE_FALLBACK = 1001;   // (internal) nothing was produced: use the lane-per-region path
constexpr int SX_NEED_RUNS = 1002;       // (internal) ... which needs the run list that stage A skipped (RunList::skipped)
bool wave_replay_wanted(const sx_ctx* ctx, const ReplayJob& job, size_t k, size_t n_runs, uint64_t heavy_tiles = 0);
int wave_replay_mission(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, MissionFindings* out, uint64_t* end_pos,
                        uint64_t defer_min_bytes, bool own_stream = false);
bool device_replay_wanted(const sx_ctx* ctx, const ReplayJob& job, size_t k, size_t n_runs);
int device_replay_mission(sx_ctx* ctx, size_t k, ByteView& view, const ReplayJob& job, const RunList& runs,
                          MissionFindings* out, uint64_t* end_pos, uint64_t defer_min_bytes);
int replay_all(sx_ctx* ctx, ByteView& bytes, const ReplayJob& job, const std::vector<RunList>& runs,
               Result* into, uint64_t* end_pos, PreReplayed* pre = nullptr);
ReplayJob whole_chunk_job(sx_ctx* ctx, uint64_t len, int file_id, bool is_last);
int download_for_replay(sx_ctx* ctx, const uint8_t* d_bytes, uint64_t len,
                        const std::vector<RunList>* runs_opt, SparseDeviceBytes* view,
                        const ReplayJob& job, const std::vector<char>* skip = nullptr,
                        bool* used_base = nullptr);

// ---- schedule (sx_schedule.cpp) and ingest (sx_ingest.cpp)
struct ResultHolder {
    sx_result* r = new sx_result();
    ~ResultHolder() { delete r; }
    sx_result* release() { sx_result* x = r; r = nullptr; return x; }
};

int set_entry_params(sx_ctx* ctx, bool carried_state_is_entry, const uint8_t* host_bytes, const uint8_t* d_bytes, uint64_t len,
                     uint64_t stream_off, std::vector<uint32_t>* parity);
int scan_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes, uint64_t len, int file_id,
                int is_last, sx_result** out, uint32_t slice_base0 = 0, sx_result* append_to = nullptr);
int shard_common(sx_ctx* ctx, const uint8_t* host_bytes, const uint8_t* d_bytes,
                 const sx_run* const* given_runs, const uint64_t* given_n, uint64_t buf_off, uint64_t buf_len,
                 uint64_t own_lo, uint64_t own_hi, const uint64_t* start_at, uint64_t file_stream_off, int file_id,
                 int reuse_runs, sx_result** out, uint64_t* end_pos);
int stream_core(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                sx_result_fn sink, void* sink_user, sx_result* accumulate, int is_last_at_eof,
                const uint8_t* direct = nullptr, uint64_t direct_len = 0);

}  // namespace sx
