// sx_api.cpp — the C-ABI of include/stringsext_amd.h: context and HIP resources, the lower-level
// stages as entry points, results, utilities.  (sx_scan* live in sx_ingest.cpp.)
#include "sx_ctx.hpp"

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
    static const unsigned flags = [] { const char* e = getenv("SX_POOL_PIN_FLAGS"); return e ? (unsigned)strtoul(e, nullptr, 0) : (unsigned)hipHostMallocNonCoherent; }();
    if (hipHostMalloc(&b.p, cap, flags) != hipSuccess) { b.p = nullptr; return b; }
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
}  // namespace

namespace sx {

double g_tl_t0 = 0;
int g_tl_on = 0;

double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}


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
static bool small_copy_on() { const char* e = getenv("SX_SMALL_COPY"); return !(e && !atoi(e)); }
int read_back_async(sx_ctx* ctx, hipStream_t s, void* pinned_dst, const void* dev_src, size_t bytes) {
    if (small_copy_on() && bytes <= (1u << 20) && ((((uintptr_t)pinned_dst | (uintptr_t)dev_src | bytes) & 3) == 0))
        HIP_TRY(ctx, launch_small_copy(pinned_dst, dev_src, bytes, s));
    else HIP_TRY(ctx, hipMemcpyAsync(pinned_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
    return SX_OK;
}
int read_back_sync(sx_ctx* ctx, MissionDev& d, hipStream_t s, void* host_dst, const void* dev_src, size_t bytes) {
    if (!small_copy_on() || bytes > kSmallReadBytes || ((((uintptr_t)dev_src | bytes) & 3) != 0)) {
        HIP_TRY(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        return SX_OK;
    }
    if (!d.h_small) HIP_TRY(ctx, hipHostMalloc((void**)&d.h_small, kSmallReadBytes, hipHostMallocDefault));
    HIP_TRY(ctx, launch_small_copy(d.h_small, dev_src, bytes, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    memcpy(host_dst, d.h_small, bytes);
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
    std::lock_guard<std::recursive_mutex> g(ctx->grow_mu);
    if (ctx->d_scratch_cap >= bytes) return SX_OK;
    if (ctx->merge_async) { const int rc = merge_drain(ctx); if (rc != SX_OK) return rc; }   // (a queued sort may read its cut table here)
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
// Pass 1's output cache: one for all missions (their stage B runs one after the other).
int ensure_cache(sx_ctx* ctx, uint64_t bytes) {
    if (ctx->d_cache_cap >= bytes) return SX_OK;
    if (ctx->d_cache) HIP_TRY(ctx, hipFree(ctx->d_cache));
    ctx->d_cache = nullptr; ctx->d_cache_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cache, bytes));
    ctx->d_cache_cap = bytes;
    return SX_OK;
}
int ensure_rp(sx_ctx* ctx, MissionDev& d, int slot, uint64_t bytes) {
    if (d.d_rp_cap[slot] >= bytes) return SX_OK;   // (a Mission's buffers are its own thread's)
    std::lock_guard<std::recursive_mutex> g(ctx->grow_mu);
    if (ctx->merge_async) { const int rc = merge_drain(ctx); if (rc != SX_OK) return rc; }   // (a queued sort may still read this Mission's findings)
    if (d.d_rp[slot]) HIP_TRY(ctx, hipFree(d.d_rp[slot]));
    d.d_rp[slot] = nullptr; d.d_rp_cap[slot] = 0;
    bytes += bytes / 4 + 4096;
    HIP_TRY(ctx, hipMalloc(&d.d_rp[slot], bytes));
    d.d_rp_cap[slot] = bytes;
    return SX_OK;
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


void begin_call(sx_ctx* ctx) {
    memset(&ctx->stats, 0, sizeof ctx->stats);
    ctx->err.clear();
    ctx->sharded_call = false;
    ctx->dev_epoch->fetch_add(1);   // (results left on the device by an earlier call — SX_OPT_RESULT_ON_DEVICE — are the context's memory: gone now)
}

// SX_OPT_RESULT_ON_DEVICE: a segment that still lies in HBM comes to the host (the host accessors' first use of it)
static int fetch_to_host(sx_result* r, MissionFindings& s) {
    if (!s.dev_only) return SX_OK;
    if (!s.keep_on_device || !s.dev_epoch_ref || s.dev_epoch_ref->load() != s.dev_epoch) return SX_E_STATE;   // (a later scan has reused the memory)
    const size_t bytes = s.ext_nf * s.rec_size() + s.ext_na;
    PinnedPool::Block blk = r->r.pool ? r->r.pool->take(bytes + 64) : PinnedPool::Block{};
    if (!blk.p) return SX_E_NOMEM;
    if (hipMemcpy(blk.p, s.dev_copy, bytes, hipMemcpyDeviceToHost) != hipSuccess) { r->r.pool->give(blk); return SX_E_HIP; }
    s.ext = blk; s.dev_only = false;
    return SX_OK;
}
static int fetch_all_to_host(const sx_result* r) {
    sx_result* w = const_cast<sx_result*>(r);
    for (auto& s : w->r.segs) { const int rc = fetch_to_host(w, s); if (rc != SX_OK) return rc; }
    return SX_OK;
}


}  // namespace sx

extern "C" {

int sx_abi_version(void) { return SX_ABI_VERSION; }

const uint16_t* sx_decoder_table(uint32_t encoding, uint64_t* n_words) {
    size_t n = 0;
    const uint16_t* t = decoder_table((int)encoding, &n);
    if (n_words) *n_words = n;
    return t;
}

const uint32_t* sx_wave_pair_codes(const sx_mission* mission, uint32_t* out8192) {
    if (!mission || !out8192) return nullptr;
    Mission m;
    std::string err;
    if (Mission::from_c(*mission, false, &m, &err) != SX_OK || m.wave_pairs.size() != 8192) return nullptr;
    memcpy(out8192, m.wave_pairs.data(), 8192 * 4);
    return out8192;
}

const uint32_t* sx_wave_pair_codes2(const sx_mission* mission, uint32_t* out4096) {
    if (!mission || !out4096) return nullptr;
    Mission m;
    std::string err;
    if (Mission::from_c(*mission, false, &m, &err) != SX_OK || m.wave_pairs2.size() != 4096) return nullptr;
    memcpy(out4096, m.wave_pairs2.data(), 4096 * 4);
    return out4096;
}

int sx_wave_swar(const sx_mission* mission, uint32_t* out26) {
    if (!mission || !out26) return SX_E_INVALID;
    Mission m;
    std::string err;
    const int rc = Mission::from_c(*mission, false, &m, &err);
    if (rc != SX_OK) return rc;
    static_assert(sizeof(WvSwar) == 26 * 4, "sx_wave_swar hands the struct out as 26 words");
    memcpy(out26, &m.wave_swar, sizeof(WvSwar));
    return m.wave_ok && m.wave_swar.cls ? 1 : 0;
}

int sx_scan_classifier(const sx_mission* mission, int generic, uint32_t* out20) {
    if (!mission || !out20) return SX_E_INVALID;
    Mission m;
    std::string err;
    const int rc = Mission::from_c(*mission, generic != 0, &m, &err);
    if (rc != SX_OK) return rc;
    const ScanParams& p = m.proto;
    const uint32_t head[7] = { p.a_lo, p.a_hi, p.u_lo, p.u_hi, p.l3_lo, p.l3_hi, p.n_ranges };
    memcpy(out20, head, sizeof head);
    memcpy(out20 + 7, p.rng_c1, 6 * sizeof(uint32_t));
    memcpy(out20 + 13, p.rng_c2, 6 * sizeof(uint32_t));
    out20[19] = 0;
    return (int)m.kind;
}

int sx_wave_classes(const sx_mission* mission, uint8_t* classes) {
    if (!mission || !classes) return SX_E_INVALID;
    Mission m;
    std::string err;
    const int rc = Mission::from_c(*mission, false, &m, &err);
    if (rc != SX_OK) return rc;
    if (!m.wave_ok || m.wave_lead_check || (m.wave_lut.size() != 256 && !(m.wave_family == 2 && m.wave_lut.size() == 512))) return 0;   // (wave_lead_check: -r, covered per buffer only)
    memcpy(classes, m.wave_lut.data(), m.wave_lut.size());   // (UTF-16: 512 bytes)
    return 1 + (int)m.wave_family;
}

int sx_create(sx_ctx** out, const sx_mission* missions, int n_missions, int hip_device, const sx_options* opt) {
    if (!out || !missions || n_missions <= 0 || n_missions > 26) { g_create_error = "bad arguments"; return SX_E_INVALID; }
    sx_ctx* ctx = new sx_ctx();
    if (opt) ctx->opt = *opt;
    if (const char* e = getenv("SX_RESULT_ON_DEVICE")) if (atoi(e)) ctx->opt.flags |= SX_OPT_RESULT_ON_DEVICE;   // (tests / fuzz: the flag through the environment)
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
            if ((e = hipMalloc((void**)&s.d_counters, kCounterWords * sizeof(uint32_t))) != hipSuccess) return fail("hipMalloc", e);
            if ((e = hipMalloc((void**)&s.d_recs, (size_t)cap * sizeof(DevRun))) != hipSuccess) return fail("hipMalloc", e);
            s.capacity = cap;
        }
    }
    for (size_t k = 0; k < ctx->dev.size(); k++) {
        size_t n_words = 0;
        if (const uint16_t* t = decoder_table(ctx->missions[k].c.encoding, &n_words)) {
            if ((e = hipMalloc((void**)&ctx->dev[k].d_table, n_words * 2)) != hipSuccess) return fail("hipMalloc", e);
            if ((e = hipMemcpy(ctx->dev[k].d_table, t, n_words * 2, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
        }
        const std::vector<uint32_t>& pl = ctx->missions[k].pair_lut;
        if (!pl.empty()) {
            if ((e = hipMalloc((void**)&ctx->dev[k].d_pair_lut, pl.size() * 4)) != hipSuccess) return fail("hipMalloc", e);
            if ((e = hipMemcpy(ctx->dev[k].d_pair_lut, pl.data(), pl.size() * 4, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy", e);
        }
    }
    *out = ctx;
    return SX_OK;
}

void sx_destroy(sx_ctx* ctx) {
    if (!ctx) return;
    // results may outlive the context (they share the pinned pool and the epoch word): what they left in HBM is freed below, so the
    // accessors must answer SX_E_STATE from here on
    ctx->dev_epoch->fetch_add(1);
    if (!ctx->host_only && ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        for (auto& d : ctx->dev) {
            if (d.stream) (void)hipStreamSynchronize(d.stream);
            if (d.stream_b) (void)hipStreamSynchronize(d.stream_b);
            for (ScanSlot& s : d.slot) {
                if (s.d_recs) (void)hipFree(s.d_recs);
                if (s.d_cnt) (void)hipFree(s.d_cnt);
                if (s.d_grid) (void)hipFree(s.d_grid);
                if (s.d_packed) (void)hipFree(s.d_packed);
                if (s.d_counters) (void)hipFree(s.d_counters);
                if (s.ev0) (void)hipEventDestroy(s.ev0);
                if (s.ev1) (void)hipEventDestroy(s.ev1);
                if (s.ev_free) (void)hipEventDestroy(s.ev_free);
            }
            if (d.d_table) (void)hipFree(d.d_table);
            if (d.d_pair_lut) (void)hipFree(d.d_pair_lut);
            if (d.d_wave_lut) (void)hipFree(d.d_wave_lut);
            if (d.d_wave_pairs) (void)hipFree(d.d_wave_pairs);
            if (d.d_wave_pairs2) (void)hipFree(d.d_wave_pairs2);
            for (void* q : d.d_rp) if (q) (void)hipFree(q);
            if (d.h_runs) (void)hipHostFree(d.h_runs);
            if (d.ev_runs) (void)hipEventDestroy(d.ev_runs);
            if (d.stream_w) (void)hipStreamDestroy(d.stream_w);
            if (d.h_tot) (void)hipHostFree(d.h_tot);
            if (d.h_small) (void)hipHostFree(d.h_small);
            for (hipEvent_t e : d.wave_ev) if (e) (void)hipEventDestroy(e);
            if (d.stream && d.stream != ctx->scan_stream) (void)hipStreamDestroy(d.stream);
        }
        for (int i = 0; i < 2; i++) {
            if (ctx->ing_pin[i]) (void)hipHostFree(ctx->ing_pin[i]);
            if (ctx->ing_dev[i]) (void)hipFree(ctx->ing_dev[i]);
        }
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
        if (ctx->scan_stream) (void)hipStreamDestroy(ctx->scan_stream);
        if (ctx->d_cache) (void)hipFree(ctx->d_cache);
        if (ctx->post_stream) (void)hipStreamDestroy(ctx->post_stream);
        if (ctx->merge_copy_stream) (void)hipStreamDestroy(ctx->merge_copy_stream);
        if (ctx->d_merge) (void)hipFree(ctx->d_merge);
        for (hipEvent_t e : ctx->merge_ev) if (e) (void)hipEventDestroy(e);
        if (ctx->ev_interleaved) (void)hipEventDestroy(ctx->ev_interleaved);
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

int sx_device_runs(sx_ctx* ctx, int mission_index, const void* device_bytes, uint64_t len, int stream_parity,
                   uint64_t min_chars, sx_run** runs, uint64_t* n_runs) {
    if (!ctx || !runs || !n_runs || mission_index < 0 || (size_t)mission_index >= ctx->missions.size()) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    if ((uintptr_t)device_bytes & 15) { ctx->err = "device_bytes must be 16-byte aligned"; return SX_E_INVALID; }
    if (ctx->missions[(size_t)mission_index].host_sequential()) { ctx->err = "an ISO-2022-JP mission has no stage A (one sequential pass on the host)"; return SX_E_INVALID; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<RunList> out;
    ctx->shard_runs_valid = false;  // the mission's device-side run list is about to be replaced
    int rc = device_runs(ctx, { mission_index }, (const uint8_t*)device_bytes, len, { (uint32_t)(stream_parity & 1) },
                         { min_chars }, &out);
    if (rc != SX_OK) return rc;
    HIP_TRY(ctx, out[0].wait());
    *n_runs = out[0].size();
    *runs = (sx_run*)malloc(sizeof(sx_run) * (out[0].size() ? out[0].size() : 1));
    if (!*runs) return SX_E_NOMEM;
    memcpy(*runs, out[0].data(), sizeof(sx_run) * out[0].size());
    return SX_OK;
}

int sx_device_runs_multi(sx_ctx* ctx, const int* mission_indices, int n, const void* device_bytes, uint64_t len, int stream_parity,
                         const uint64_t* min_chars, sx_run** runs, uint64_t* n_runs) {
    if (!ctx || !mission_indices || n <= 0 || !min_chars || !runs || !n_runs) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    if ((uintptr_t)device_bytes & 15) { ctx->err = "device_bytes must be 16-byte aligned"; return SX_E_INVALID; }
    std::vector<int> which; std::vector<uint32_t> par; std::vector<uint64_t> mc;
    for (int i = 0; i < n; i++) {
        const int k = mission_indices[i];
        if (k < 0 || (size_t)k >= ctx->missions.size() || std::find(which.begin(), which.end(), k) != which.end()) { ctx->err = "bad mission index"; return SX_E_INVALID; }
        if (ctx->missions[(size_t)k].host_sequential()) { ctx->err = "an ISO-2022-JP mission has no stage A (one sequential pass on the host)"; return SX_E_INVALID; }
        which.push_back(k); par.push_back((uint32_t)(stream_parity & 1)); mc.push_back(min_chars[i]);
        runs[i] = nullptr; n_runs[i] = 0;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<RunList> out;
    ctx->shard_runs_valid = false;
    int rc = device_runs(ctx, which, (const uint8_t*)device_bytes, len, par, mc, &out);
    if (rc != SX_OK) return rc;
    for (int i = 0; i < n; i++) {
        if (!out[(size_t)i].own.empty()) out[(size_t)i].use_own();
        HIP_TRY(ctx, out[(size_t)i].wait());
        n_runs[i] = out[(size_t)i].size();
        runs[i] = (sx_run*)malloc(sizeof(sx_run) * (n_runs[i] ? n_runs[i] : 1));
        if (!runs[i]) { for (int j = 0; j < i; j++) { free(runs[j]); runs[j] = nullptr; } return SX_E_NOMEM; }
        memcpy(runs[i], out[(size_t)i].data(), sizeof(sx_run) * n_runs[i]);
    }
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
    std::vector<uint32_t> entry;
    int rc = set_entry_params(ctx, true, bytes, nullptr, len, 0, &entry);
    if (rc != SX_OK) return rc;
    rc = replay_all(ctx, view, whole_chunk_job(ctx, len, input_file_id, is_last_input_buffer != 0), r, &res.r->r, nullptr);
    if (rc == SX_OK) *out = res.release();
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
    // (sx_scan_shard uploads into the context's own staging buffer: the runs of the previous call belong to the same
    //  host bytes only if offset, length and pointer agree — shard_common checks that again on its side)
    if (!(reuse_runs && ctx->shard_runs_valid && ctx->shard_runs_off == buf_off && ctx->shard_runs_len == buf_len && ctx->shard_runs_ptr == (const void*)ctx->d_input)) {
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
    { const int rc = fetch_to_host(const_cast<sx_result*>(r), const_cast<sx_result*>(r)->r.segs[(size_t)i]); if (rc != SX_OK) return rc; }
    const MissionFindings& s = r->r.segs[(size_t)i];
    if (findings) *findings = s.data();
    if (n_findings) *n_findings = s.count();
    if (arena) *arena = (const uint8_t*)s.strings();
    if (arena_len) *arena_len = s.strings_len();
    return SX_OK;
}
int sx_result_segment_packed(const sx_result* r, uint64_t i, const void** findings, uint64_t* n_findings, const uint8_t** arena,
                             uint64_t* arena_len, int* packed, sx_segment_info* info) {
    if (!r || i >= r->r.segs.size()) return SX_E_INVALID;
    { const int rc = fetch_to_host(const_cast<sx_result*>(r), const_cast<sx_result*>(r)->r.segs[(size_t)i]); if (rc != SX_OK) return rc; }
    const MissionFindings& s = r->r.segs[(size_t)i];
    if (findings) *findings = s.packed ? (const void*)s.data16() : (const void*)s.data();
    if (n_findings) *n_findings = s.count();
    if (arena) *arena = (const uint8_t*)s.strings();
    if (arena_len) *arena_len = s.strings_len();
    if (packed) *packed = s.packed ? 1 : 0;
    if (info) {
        memset(info, 0, sizeof *info);
        info->packed = s.packed ? 1 : 0;
        info->input_file_id = -1;
        if (s.packed && s.info) {
            info->input_file_id = s.info->file_id; info->slice_base = s.info->slice_base;
            memcpy(info->position0, s.info->pos0, sizeof info->position0);
        }
    }
    return SX_OK;
}
int sx_result_segment_device(const sx_result* r, uint64_t i, const void** d_records, uint64_t* n_findings, const uint8_t** d_arena,
                             uint64_t* arena_len, int* packed, sx_segment_info* info) {
    if (!r || i >= r->r.segs.size()) return SX_E_INVALID;
    const MissionFindings& s = r->r.segs[(size_t)i];
    if (d_records) *d_records = nullptr;
    if (d_arena) *d_arena = nullptr;
    if (n_findings) *n_findings = s.count();
    if (arena_len) *arena_len = s.strings_len();
    if (packed) *packed = s.packed ? 1 : 0;
    if (info) {
        memset(info, 0, sizeof *info);
        info->packed = s.packed ? 1 : 0;
        info->input_file_id = -1;
        if (s.packed && s.info) {
            info->input_file_id = s.info->file_id; info->slice_base = s.info->slice_base;
            memcpy(info->position0, s.info->pos0, sizeof info->position0);
        }
    }
    if (!s.dev_only) return SX_OK;   // (in host memory: the host accessors)
    if (!s.keep_on_device || !s.dev_epoch_ref || s.dev_epoch_ref->load() != s.dev_epoch) return SX_E_STATE;
    if (d_records) *d_records = s.dev_copy;
    if (d_arena) *d_arena = (const uint8_t*)s.dev_copy + s.ext_nf * s.rec_size();
    return SX_OK;
}
// The contiguous view: joins the segments on first use (a copy; none if there is one segment).
const sx_finding* sx_result_findings(const sx_result* r) {
    if (!r || fetch_all_to_host(r) != SX_OK || !const_cast<sx_result*>(r)->r.flatten(nullptr) || r->r.segs.empty()) return nullptr;
    return r->r.segs[0].data();
}
const uint8_t* sx_result_arena(const sx_result* r, uint64_t* len) {
    if (len) *len = 0;
    if (!r || fetch_all_to_host(r) != SX_OK || !const_cast<sx_result*>(r)->r.flatten(nullptr) || r->r.segs.empty()) return nullptr;
    if (len) *len = r->r.segs[0].strings_len();
    return (const uint8_t*)r->r.segs[0].strings();
}
void sx_result_free(sx_result* r) { delete r; }

int sx_print_findings(const sx_ctx* ctx, const sx_result* r, int n_inputs, int radix, int no_metadata, uint8_t** out,
                      uint64_t* out_len) {
    if (!ctx || !r || !out || !out_len) return SX_E_INVALID;
    if (radix != 0 && radix != 'x' && radix != 'd' && radix != 'o') return SX_E_INVALID;
    { const int rc = fetch_all_to_host(r); if (rc != SX_OK) return rc; }
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
