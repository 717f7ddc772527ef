// sx_sort.hip — order the device run records by start offset on the device (rocPRIM radix
// sort), so that the host only has to walk them once.  Records are 16 bytes: the key is
// `start` (u64), the payload the other 8 bytes.
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "sx_device.hpp"

namespace sx {

__global__ __launch_bounds__(256) void split_records_kernel(const DevRun* recs, uint32_t n, uint64_t* keys, uint64_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const DevRun r = recs[i];
    const bool invalid = r.len == kRecInvalidLen && r.chars_flags == kRecInvalidFlags;
    keys[i] = invalid ? ~0ull : r.start;  // unused slots sort to the end
    vals[i] = ((uint64_t)r.chars_flags << 32) | r.len;
}
__global__ __launch_bounds__(256) void join_records_kernel(const uint64_t* keys, const uint64_t* vals, uint32_t n, DevRun* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    DevRun r;
    r.start = keys[i]; r.len = (uint32_t)vals[i]; r.chars_flags = (uint32_t)(vals[i] >> 32);
    out[i] = r;
}

// scratch must hold sort_scratch_bytes(n) bytes; sorted records are written back to `recs`
size_t sort_scratch_bytes(uint32_t n) {
    size_t tmp = 0;
    uint64_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)n, 0, 64, (hipStream_t)0);
    return (size_t)n * 8 * 4 + tmp + 1024;
}

hipError_t sort_records(DevRun* recs, uint32_t n, uint64_t max_key, void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint64_t* k0 = (uint64_t*)scratch;
    uint64_t* v0 = k0 + n;
    uint64_t* k1 = v0 + n;
    uint64_t* v1 = k1 + n;
    uint8_t* tmp = (uint8_t*)(((uintptr_t)(v1 + n) + 255) & ~(uintptr_t)255);  // rocPRIM wants its storage aligned
    size_t tmp_bytes = scratch_bytes - (size_t)(tmp - (uint8_t*)scratch);
    const unsigned blocks = (n + 255) / 256;
    static const bool dbg = getenv("SX_TIMING2") != nullptr;
    auto stamp = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(stream);
        timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
        fprintf(stderr, "[sx]     %s at %.3f ms\n", what, ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6);
    };
    stamp("sort: begin");
    hipLaunchKernelGGL(split_records_kernel, dim3(blocks), dim3(256), 0, stream, recs, n, k0, v0);
    stamp("sort: split done");
    // only the bits a position can have, and one more: the key of an unused slot (all ones) stays the largest
    unsigned end_bit = 1;
    while (end_bit < 63 && (max_key >> end_bit)) end_bit++;
    end_bit++;
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, end_bit, stream);
    if (e != hipSuccess) return e;
    stamp("sort: radix done");
    hipLaunchKernelGGL(join_records_kernel, dim3(blocks), dim3(256), 0, stream, k1, v1, n, recs);
    return hipGetLastError();
}


// ---- region mode: pack the per-sub-chunk record regions ------------------------------------
// Small blocks, a few hundred bytes of LDS: these kernels must find room next to a scan kernel
// that fills the device (a rocPRIM radix sort does not: its large blocks starve).
constexpr uint32_t kScanPerBlock = 1024;  // 256 threads x 4 counts
__global__ __launch_bounds__(256) void region_block_scan_kernel(const uint32_t* counts, uint64_t n, uint32_t* local_off,
                                                                uint32_t* block_sum) {
    __shared__ uint32_t wsum[4];
    const uint64_t i0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t c[4], t = 0;
    for (int j = 0; j < 4; j++) { c[j] = i0 + j < n ? counts[i0 + j] : 0u; t += c[j]; }
    uint32_t inc = t;
    const uint32_t lane = threadIdx.x & 63;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)lane >= o) inc += v; }
    if (lane == 63) wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
    uint32_t run = before + inc - t;
    for (int j = 0; j < 4; j++) { if (i0 + j < n) local_off[i0 + j] = run; run += c[j]; }
    if (threadIdx.x == 255) block_sum[blockIdx.x] = before + inc;
}
__global__ __launch_bounds__(256) void region_sum_scan_kernel(uint32_t* block_sum, uint32_t n_blocks, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n_blocks ? block_sum[i] : 0u;
        uint32_t inc = v;
        const uint32_t lane = threadIdx.x & 63;
        for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((int)lane >= o) inc += u; }
        if (lane == 63) wsum[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t before = carry_s;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < n_blocks) block_sum[i] = before + inc - v;  // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry_s = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
// A region holds its sub-chunk's records almost in order (tile by tile; inside a tile the lanes
// emit their first stretches, then their second ones, ...), so every record is placed by its
// rank among the region's records (<= region_cap of them: a handful of cached loads).
// Round 5: kPackLanes lanes per REGION, each taking the region's records j, j + kPackLanes, ... (rounds 2-4: a thread per SLOT — with 64
// slots per region and a dozen records in them five lanes of six did nothing, and the launch was 262 144 wavefronts per 64 GiB,
// 1.4 ms next to a scan kernel).
constexpr uint32_t kPackLanes = 16;
__global__ __launch_bounds__(256) void region_pack_kernel(const DevRun* recs, const uint32_t* counts, const uint32_t* local_off,
                                                          const uint32_t* block_off, uint64_t n_regions, uint32_t region_cap,
                                                          DevRun* out) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t w = t / kPackLanes;
    if (w >= n_regions) return;
    const uint32_t n = counts[w];
    const DevRun* reg = recs + w * region_cap;
    const uint64_t base = (uint64_t)block_off[w / kScanPerBlock] + local_off[w];
    for (uint32_t j = (uint32_t)(t % kPackLanes); j < n; j += kPackLanes) {
        const DevRun r = reg[j];
        uint32_t rank = 0;
        for (uint32_t i = 0; i < n; i++) rank += reg[i].start < r.start ? 1u : 0u;
        out[base + rank] = r;
    }
}
// Large regions (string-dense input): instead of packing, the unused slots are marked like the pool's
// unused slots and the whole array goes through the radix sort.
__global__ __launch_bounds__(256) void region_invalidate_kernel(DevRun* recs, const uint32_t* counts, uint64_t n_regions,
                                                                uint32_t region_cap) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t w = s / region_cap;
    if (w >= n_regions) return;
    if ((uint32_t)(s - w * region_cap) >= counts[w]) {
        DevRun r; r.start = 0; r.len = kRecInvalidLen; r.chars_flags = kRecInvalidFlags;
        recs[s] = r;
    }
}
hipError_t invalidate_region_slack(DevRun* recs, const uint32_t* counts, uint64_t n_regions, uint32_t region_cap, hipStream_t stream) {
    const uint64_t slots = n_regions * region_cap;
    if (slots == 0) return hipSuccess;
    hipLaunchKernelGGL(region_invalidate_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream, recs, counts, n_regions,
                       region_cap);
    return hipGetLastError();
}
size_t compact_scratch_bytes(uint64_t n_regions) {
    return n_regions * 4 + ((n_regions + kScanPerBlock - 1) / kScanPerBlock) * 4 + 1024;
}
hipError_t compact_regions(const DevRun* recs, const uint32_t* counts, uint64_t n_regions, uint32_t region_cap, DevRun* out,
                           uint32_t* total, void* scratch, size_t scratch_bytes, hipStream_t stream) {
    if (n_regions == 0) return hipMemsetAsync(total, 0, 4, stream);
    const uint32_t n_blocks = (uint32_t)((n_regions + kScanPerBlock - 1) / kScanPerBlock);
    if (scratch_bytes < compact_scratch_bytes(n_regions)) return hipErrorInvalidValue;
    uint32_t* local_off = (uint32_t*)scratch;
    uint32_t* block_sum = local_off + n_regions;
    hipLaunchKernelGGL(region_block_scan_kernel, dim3(n_blocks), dim3(256), 0, stream, counts, n_regions, local_off, block_sum);
    hipLaunchKernelGGL(region_sum_scan_kernel, dim3(1), dim3(256), 0, stream, block_sum, n_blocks, total);
    const uint64_t threads = n_regions * kPackLanes;
    hipLaunchKernelGGL(region_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, recs, counts, local_off,
                       block_sum, n_regions, region_cap, out);
    return hipGetLastError();
}

// ---- join the sorted records into runs on the device -------------------------------------
// Same rule as merge_sorted_device_runs (sx_replay.cpp): a record continues its predecessor if
// that one reaches its sub-chunk end, this one begins at a sub-chunk start, and they touch.
// Runs with fewer than min_chars characters cannot produce a Finding and are dropped.
struct RecIsHead {
    const DevRun* v;
    __device__ uint32_t operator()(uint32_t i) const {
        const DevRun r = v[i];
        if (r.start == ~0ull) return 0u;
        if (i == 0 || !(r.chars_flags & kRecStartOpen)) return 1u;
        const DevRun p = v[i - 1];
        return ((p.chars_flags & kRecEndOpen) && p.start + p.len == r.start) ? 0u : 1u;
    }
};
__global__ __launch_bounds__(256) void accumulate_runs_kernel(const DevRun* v, uint32_t n, const uint32_t* rid, sx_run* tmp) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const DevRun r = v[i];
    if (r.start == ~0ull) return;
    const uint32_t id = rid[i] - 1u;
    const bool head = i == 0 || rid[i - 1] != rid[i];
    const bool tail = i + 1 >= n || rid[i + 1] != rid[i] || v[i + 1].start == ~0ull;
    const unsigned long long chars = r.chars_flags & kRecCharsMask, end = r.start + r.len;
    if (head && tail) { sx_run o; o.start = r.start; o.end = end; o.chars = chars; tmp[id] = o; return; }  // the usual case
    if (head) tmp[id].start = r.start;
    atomicAdd((unsigned long long*)&tmp[id].chars, chars);
    atomicMax((unsigned long long*)&tmp[id].end, end);
}

// The joined runs that are long enough, compacted: flags -> exclusive scan -> scatter.  (rocprim::select did this in one
// pass but took 1.75 ms for 13 M runs of 24 bytes — rocprofv3, string-dense input; these three kernels take 0.3 ms.)
struct RunLongFlag {
    const sx_run* t; uint64_t min_chars;
    __device__ uint32_t operator()(uint32_t i) const { return t[i].chars >= min_chars ? 1u : 0u; }
};
__global__ __launch_bounds__(256) void scatter_long_runs_kernel(const sx_run* tmp, uint32_t n, uint64_t min_chars, const uint32_t* pos,
                                                                sx_run* out, uint32_t* out_count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const sx_run r = tmp[i];
    const bool keep = r.chars >= min_chars;
    if (keep) out[pos[i]] = r;
    if (i == n - 1) *out_count = pos[i] + (keep ? 1u : 0u);
}
static size_t merge_tmp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    auto heads = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), RecIsHead{ nullptr });
    (void)rocprim::inclusive_scan(nullptr, a, heads, (uint32_t*)nullptr, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), RunLongFlag{ nullptr, 1 });
    (void)rocprim::exclusive_scan(nullptr, b, flags, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t)0);
    return (a > b ? a : b) + 256;
}
size_t merge_scratch_bytes(uint32_t n) {
    return ((size_t)n * 4 + 255) / 256 * 256 + ((size_t)n * sizeof(sx_run) + 511) / 256 * 256 + merge_tmp_bytes(n);
}

// recs: sorted (sort_records); out: room for n runs; out_count: device u32
hipError_t merge_sorted_records(const DevRun* recs, uint32_t n, uint64_t min_chars, void* scratch, size_t scratch_bytes,
                                sx_run* out, uint32_t* out_count, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(out_count, 0, 4, stream);
    uint8_t* base = (uint8_t*)scratch;
    uint32_t* rid = (uint32_t*)base;
    const size_t rid_bytes = ((size_t)n * 4 + 255) / 256 * 256;
    sx_run* tmp = (sx_run*)(base + rid_bytes);
    const size_t run_bytes = ((size_t)n * sizeof(sx_run) + 511) / 256 * 256;  // rocPRIM wants its storage aligned
    void* rp_tmp = base + rid_bytes + run_bytes;
    size_t rp_bytes = scratch_bytes - rid_bytes - run_bytes;
    auto heads = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), RecIsHead{ recs });
    hipError_t e = rocprim::inclusive_scan(rp_tmp, rp_bytes, heads, rid, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(tmp, 0, (size_t)n * sizeof(sx_run), stream);  // unused entries: chars 0 -> dropped below
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(accumulate_runs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, recs, n, rid, tmp);
    const uint64_t minc = min_chars ? min_chars : 1;
    auto flags = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), RunLongFlag{ tmp, minc });
    rp_bytes = scratch_bytes - rid_bytes - run_bytes;
    uint32_t* pos = rid;   // (the run ids have done their work)
    e = rocprim::exclusive_scan(rp_tmp, rp_bytes, flags, pos, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(scatter_long_runs_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tmp, n, minc, pos, out, out_count);
    return hipGetLastError();
}


// ---- interleave the findings of several missions (src/main.rs:118-136: the merger) ---------
// Every mission's findings are ordered by position; across missions the merger orders by position
// and, on ties, by mission.  All missions of a call count bytes from the same origin, so the key is
// the position alone, and a finding's place in the output is its index in its own list plus, for
// every other mission, the number of that mission's findings in front of it: those with a smaller
// position, and for the missions listed before its own also those with the same one (lower_bound /
// upper_bound).  The searches are short: the position range is cut into T tiles of 2^shift bytes
// (about 64 findings each), a table holds where every list enters every tile, and a finding looks
// only at the other lists' stretch of its own tile — neighbours in a list read the same cache lines.
// One pass over the findings (32 bytes read, 32 written each) instead of the eight passes of a
// 64-bit radix sort over (position, index) pairs plus two gathers: 40 -> 3 ms for 46 M findings.
// Up to kMergeLists missions; beyond that the stable radix sort over the concatenation (mission 0's
// findings first, then mission 1's, ...), which yields the same order.
constexpr int kMergeLists = 16;
struct MergeLists {
    const sx_finding* f[kMergeLists];
    uint64_t nf[kMergeLists];
    uint64_t out_base[kMergeLists];   // unused by the placement (ranks are absolute), kept for the arena bases below
    uint32_t arena_adj[kMergeLists];  // added to str_off (mod 2^32)
    int n;
};
struct MergeRange { uint64_t base; uint32_t shift, pad; };

__global__ void merge_range_kernel(MergeLists L, uint32_t T, MergeRange* out) {
    if (threadIdx.x || blockIdx.x) return;
    uint64_t lo = ~0ull, hi = 0;
    for (int m = 0; m < L.n; m++)
        if (L.nf[m]) {
            const uint64_t a = L.f[m][0].position, b = L.f[m][L.nf[m] - 1].position;
            lo = a < lo ? a : lo; hi = b > hi ? b : hi;
        }
    if (lo > hi) { lo = 0; hi = 0; }
    const uint64_t span = hi - lo;
    uint32_t s = 0;
    while (s < 63 && (span >> s) + 1 > (uint64_t)T) s++;
    out->base = lo; out->shift = s; out->pad = 0;
}
__device__ __forceinline__ uint64_t merge_lower(const sx_finding* f, uint64_t lo, uint64_t hi, uint64_t p) {   // first index with position >= p
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (f[mid].position < p) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint64_t merge_upper(const sx_finding* f, uint64_t lo, uint64_t hi, uint64_t p) {   // first index with position > p
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (f[mid].position <= p) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// bounds[m * (T + 1) + t] = where list m enters tile t (t = 0: 0, t = T: its end)
__global__ __launch_bounds__(256) void merge_bounds_kernel(MergeLists L, uint32_t T, const MergeRange* rg, uint32_t* bounds) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t per = (uint64_t)T + 1;
    if (i >= per * (uint64_t)L.n) return;
    const int m = (int)(i / per);
    const uint32_t t = (uint32_t)(i % per);
    uint64_t v;
    if (t == 0) v = 0;
    else if (t == T) v = L.nf[m];
    else v = merge_lower(L.f[m], 0, L.nf[m], rg->base + ((uint64_t)t << rg->shift));
    bounds[i] = (uint32_t)v;
}
__global__ __launch_bounds__(256) void merge_place_kernel(MergeLists L, int m, uint32_t T, const MergeRange* rg, const uint32_t* bounds,
                                                          void* out, int packed) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= L.nf[m]) return;
    sx_finding f = L.f[m][i];
    const uint64_t p = f.position, base = rg->base;
    uint64_t t = p <= base ? 0 : (p - base) >> rg->shift;
    if (t > (uint64_t)T - 1) t = (uint64_t)T - 1;
    uint64_t rank = i;
    for (int o = 0; o < L.n; o++) {
        if (o == m || !L.nf[o]) continue;
        const uint32_t* b = bounds + (uint64_t)o * ((uint64_t)T + 1) + t;
        const uint64_t lo = b[0], hi = b[1];
        rank += o < m ? merge_upper(L.f[o], lo, hi, p) : merge_lower(L.f[o], lo, hi, p);
    }
    f.str_off += L.arena_adj[m];   // (mod 2^32: a part's base may be "negative", see merge_findings_device_part)
    if (packed) {   // include/stringsext_amd.h sx_finding16: what crosses PCIe is half the size
        sx_finding16 p;
        p.position = f.position; p.str_off = f.str_off; p.str_len = (uint16_t)f.str_len;
        p.flags = (uint8_t)((f.precision & 3u) | (f.completes_previous ? 4u : 0u)); p.mission_id = f.mission_id;
        ((sx_finding16*)out)[rank] = p;
    } else ((sx_finding*)out)[rank] = f;
}
static uint32_t merge_tiles(uint64_t n) {
    const uint64_t t = n / 64;
    return (uint32_t)(t < 1 ? 1 : t > (1u << 22) ? (1u << 22) : t);
}

__global__ __launch_bounds__(256) void merge_gather_in_kernel(const sx_finding* src, uint64_t n, uint32_t arena_base, uint64_t out_base,
                                                              sx_finding* all, uint64_t* keys, uint64_t* vals) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    sx_finding f = src[i];
    f.str_off += arena_base;
    all[out_base + i] = f;
    keys[out_base + i] = f.position;
    vals[out_base + i] = out_base + i;
}
__global__ __launch_bounds__(256) void merge_gather_out_kernel(const sx_finding* all, const uint64_t* order, uint64_t n, sx_finding* out) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = all[order[i]];
}
static size_t merge_sort_scratch_bytes(uint64_t n) {
    size_t tmp = 0;
    uint64_t* nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, nul, nul, nul, nul, (size_t)n, 0, 64, (hipStream_t)0);
    return n * sizeof(sx_finding) + 4 * n * 8 + tmp + 2048;
}
bool merge_part_can_pack(uint64_t n, int n_missions) { return n_missions <= kMergeLists && n < 0xFFFFFFFFull; }
size_t merge_findings_scratch_bytes(uint64_t n, int n_missions) {
    if (n_missions > kMergeLists || n >= 0xFFFFFFFFull) return merge_sort_scratch_bytes(n);
    return 512 + ((size_t)merge_tiles(n) + 1) * 4 * (size_t)n_missions + 256;
}
// One part of the interleave: of mission m the findings f[m][0 .. nf[m]) whose strings are a[m][0 .. nb[m]) and whose str_off
// count from off0[m] (the part's first string); out = [sum nf findings][sum nb bytes].
hipError_t merge_findings_device_part(const sx_finding* const* f, const uint8_t* const* a, const uint64_t* nf, const uint64_t* nb,
                                      const uint32_t* off0, int n_missions, void* out, void* scratch, size_t scratch_bytes,
                                      hipStream_t stream, int packed) {
    uint64_t n = 0;
    for (int m = 0; m < n_missions; m++) n += nf[m];
    if (n == 0) return hipSuccess;
    if (scratch_bytes < merge_findings_scratch_bytes(n, n_missions)) return hipErrorInvalidValue;
    uint8_t* base = (uint8_t*)(((uintptr_t)scratch + 255) & ~(uintptr_t)255);
    if (packed && !(n_missions <= kMergeLists && n < 0xFFFFFFFFull)) return hipErrorInvalidValue;   // (merge_part_can_pack)
    sx_finding* out_f = (sx_finding*)out;
    uint8_t* out_a = (uint8_t*)out + n * (packed ? sizeof(sx_finding16) : sizeof(sx_finding));
    if (n_missions <= kMergeLists && n < 0xFFFFFFFFull) {
        MergeLists L{};
        L.n = n_missions;
        uint64_t ab = 0;
        for (int m = 0; m < n_missions; m++) {
            L.f[m] = f[m]; L.nf[m] = nf[m]; L.arena_adj[m] = (uint32_t)ab - off0[m];
            if (nf[m] && nb[m]) {
                hipError_t e = hipMemcpyAsync(out_a + ab, a[m], nb[m], hipMemcpyDeviceToDevice, stream);
                if (e != hipSuccess) return e;
            }
            ab += nb[m];
        }
        const uint32_t T = merge_tiles(n);
        MergeRange* rg = (MergeRange*)base;
        uint32_t* bounds = (uint32_t*)(base + 256);
        hipLaunchKernelGGL(merge_range_kernel, dim3(1), dim3(64), 0, stream, L, T, rg);
        const uint64_t nb_threads = ((uint64_t)T + 1) * (uint64_t)n_missions;
        hipLaunchKernelGGL(merge_bounds_kernel, dim3((unsigned)((nb_threads + 255) / 256)), dim3(256), 0, stream, L, T, rg, bounds);
        for (int m = 0; m < n_missions; m++)
            if (nf[m])
                hipLaunchKernelGGL(merge_place_kernel, dim3((unsigned)((nf[m] + 255) / 256)), dim3(256), 0, stream, L, m, T, rg, bounds, (void*)out_f, packed);
        return hipGetLastError();
    }
    sx_finding* all = (sx_finding*)base;
    uint64_t* k0 = (uint64_t*)(base + ((n * sizeof(sx_finding) + 255) & ~(size_t)255));
    uint64_t* v0 = k0 + n;
    uint64_t* k1 = v0 + n;
    uint64_t* v1 = k1 + n;
    uint8_t* tmp = (uint8_t*)(((uintptr_t)(v1 + n) + 255) & ~(uintptr_t)255);
    size_t tmp_bytes = scratch_bytes - (size_t)(tmp - (uint8_t*)scratch);
    uint64_t fb = 0, ab = 0;
    for (int m = 0; m < n_missions; m++) {
        if (nf[m]) {
            hipLaunchKernelGGL(merge_gather_in_kernel, dim3((unsigned)((nf[m] + 255) / 256)), dim3(256), 0, stream, f[m], nf[m],
                               (uint32_t)ab - off0[m], fb, all, k0, v0);
            hipError_t e = hipMemcpyAsync(out_a + ab, a[m], nb[m], hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
        }
        fb += nf[m]; ab += nb[m];
    }
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0, 64, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_gather_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, all, v1, n, out_f);
    return hipGetLastError();
}

// A plain copy by a few workgroups (dst may be pinned host memory: the bytes then cross PCIe as the kernel's own writes, and
// the copy engine stays free for the small transfers stage B waits on).
typedef unsigned int copy_v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(1024) void copy_bytes_kernel(copy_v4u* __restrict__ dst, const copy_v4u* __restrict__ src, uint64_t n16,
                                                          uint8_t* dst_tail, const uint8_t* src_tail, uint32_t n_tail) {
    // every workgroup copies one contiguous stretch (a sequential stream of writes each)
    const uint64_t per = (n16 + gridDim.x - 1) / gridDim.x;
    const uint64_t begin = per * blockIdx.x, end = begin + per < n16 ? begin + per : n16;
    const uint64_t stride = blockDim.x;
    uint64_t i = begin + threadIdx.x;
    n16 = end;
    for (; i + 3 * stride < n16; i += 4 * stride) {   // four loads in flight per lane, then four posted writes
        const copy_v4u a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const copy_v4u c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        if (NT) {
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
        } else { dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d; }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
// A few words from device memory into pinned host memory by ONE wavefront (round 5).  The runtime's device-to-host copy is a blit
// kernel, and next to a scan kernel that keeps every wave slot filled its workgroups wait for room: the dozen small read-backs of a
// stage B (counters, totals, list lengths) took anything from 0.05 to 10 ms each (profiles/r05b_kernel_timeline_*: `copyBuffer`
// rows of 4.9 and 9.8 ms for 4 and 100 000 bytes).  A block of 64 threads finds a slot within microseconds.
__global__ __launch_bounds__(64) void small_copy_kernel(uint32_t* dst, const uint32_t* src, uint32_t n_words) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, nt = gridDim.x * 64;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const uint32_t n4 = n_words / 4;
        for (uint32_t i = t; i < n4; i += nt) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        for (uint32_t i = n4 * 4 + t; i < n_words; i += nt) dst[i] = src[i];
    } else for (uint32_t i = t; i < n_words; i += nt) dst[i] = src[i];
}
hipError_t launch_small_copy(void* pinned_dst, const void* dev_src, size_t bytes, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    if ((((uintptr_t)pinned_dst | (uintptr_t)dev_src | bytes) & 3) != 0 || bytes > (1u << 20)) return hipErrorInvalidValue;
    const unsigned blocks = bytes <= 16384 ? 1u : (bytes <= 131072 ? 2u : 4u);   // (a wavefront each: room is found at once)
    hipLaunchKernelGGL(small_copy_kernel, dim3(blocks), dim3(64), 0, stream, (uint32_t*)pinned_dst, (const uint32_t*)dev_src, (uint32_t)(bytes / 4));
    return hipGetLastError();
}
hipError_t launch_copy_bytes(void* dst, const void* src, uint64_t bytes, uint32_t workgroups, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) != 0) return hipErrorInvalidValue;
    static const int nt = [] { const char* e = getenv("SX_MERGE_COPY_NT"); return e ? atoi(e) : 1; }();
    static const int threads = [] { const char* e = getenv("SX_MERGE_COPY_THREADS"); return e ? std::max(64, std::min(1024, atoi(e))) : 512; }();   // (round 4, C5 with 16-byte records: 256 -> 474 ms per step, 512 -> 444, 1024 -> 454)
    const uint64_t n16 = bytes / 16;
    if (nt)
        hipLaunchKernelGGL(copy_bytes_kernel<true>, dim3(workgroups ? workgroups : 2), dim3((unsigned)threads), 0, stream, (copy_v4u*)dst, (const copy_v4u*)src, n16,
                           (uint8_t*)dst + n16 * 16, (const uint8_t*)src + n16 * 16, (uint32_t)(bytes - n16 * 16));
    else
        hipLaunchKernelGGL(copy_bytes_kernel<false>, dim3(workgroups ? workgroups : 2), dim3((unsigned)threads), 0, stream, (copy_v4u*)dst, (const copy_v4u*)src, n16,
                           (uint8_t*)dst + n16 * 16, (const uint8_t*)src + n16 * 16, (uint32_t)(bytes - n16 * 16));
    return hipGetLastError();
}

// Where a mission's findings (ordered by position, strings laid out in the same order) are cut at the given positions:
// idx[j] = the first finding at or behind cuts[j], off[j] = where its string begins (= the bytes of all strings before it).
__global__ __launch_bounds__(64) void merge_cuts_kernel(const sx_finding* f, uint64_t n, uint64_t nb, const uint64_t* cuts, uint32_t n_cuts,
                                                        uint64_t* idx, uint64_t* off) {
    const uint32_t j = blockIdx.x * 64u + threadIdx.x;
    if (j >= n_cuts) return;
    const uint64_t c = cuts[j];
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (f[mid].position < c) lo = mid + 1; else hi = mid;
    }
    idx[j] = lo;
    off[j] = lo < n ? f[lo].str_off : nb;
}
hipError_t launch_merge_cuts(const sx_finding* f, uint64_t n, uint64_t nb, const uint64_t* cuts, uint32_t n_cuts, uint64_t* idx,
                             uint64_t* off, hipStream_t stream) {
    if (n_cuts == 0) return hipSuccess;
    hipLaunchKernelGGL(merge_cuts_kernel, dim3((n_cuts + 63) / 64), dim3(64), 0, stream, f, n, nb, cuts, n_cuts, idx, off);
    return hipGetLastError();
}

}  // namespace sx
