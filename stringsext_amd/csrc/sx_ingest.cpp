// sx_ingest.cpp — bytes on their way to the device: sx_scan (host buffer), sx_scan_device,
// sx_scan_stream / sx_scan_file (reader thread, pinned double buffer, H2D next to the scan).
#include "sx_ctx.hpp"

using namespace sx;

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

namespace sx {
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
int stream_core(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
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

}  // namespace sx

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

extern "C" {

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

int sx_scan_stream(sx_ctx* ctx, sx_read_fn read, void* read_user, uint64_t chunk_bytes, int input_file_id,
                   sx_result_fn sink, void* sink_user) {
    if (!ctx || !read || !sink) return SX_E_INVALID;
    begin_call(ctx);
    if (ctx->host_only) { ctx->err = "host-only context: no device scan"; return SX_E_STATE; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return stream_core(ctx, read, read_user, chunk_bytes, input_file_id, sink, sink_user, nullptr, 0);
}


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


}  // extern "C"
