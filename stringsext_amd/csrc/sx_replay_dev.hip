// sx_replay_dev.hip — stage B on the device: the exact replay of FindingCollection::from
// (reference src/finding_collection.rs:84-342) around long runs, one lane per run.
//
// This is the device twin of sx_replay.cpp (RangeReplay) and sx_decoder.cpp: same rules,
// same order of operations, so that positions, precision marks, cuts and strings are
// identical to the host replay (which stays the fallback and the reference for the tests).
//   * lane i owns the region that begins in the window of run i's first byte, with the
//     carried state re-derived from the bytes before that window (decoder state; one
//     accepted char as leftover if a short run hangs over the edge);
//   * it follows runs i, i+1, .. until no cut string is pending and no long leftover is
//     carried (RangeReplay::scan_from's stop rule);
//   * pass 1 counts (findings, string bytes, end position); the host decides which regions
//     stand (a region is void if an earlier one ran over its start) and assigns output
//     offsets; pass 2 replays the standing regions again and writes findings + strings, in
//     order, at those offsets — no atomics, deterministic output order.
// Regions longer than kMaxWindows windows, or anything this path does not cover, are
// reported with a status and replayed by the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sx_device.hpp"

namespace sx {

#define SXD __device__ __forceinline__
}  // namespace sx
#include "sx_replay_core.hpp"
namespace sx {

// Pass 1: one lane per run; a run whose window is certainly inside the region of the run
// before it is marked kRegionChained right away (the earlier region follows it).
__global__ __launch_bounds__(64) void replay_count_kernel(const ReplayParams P, ReplayRegionOut* out) {
    const u64 i = (u64)blockIdx.x * 64 + threadIdx.x;
    if (i >= P.n_runs) return;
    ReplayRegionOut o;
    o.end = 0; o.n_find = 0; o.n_bytes = 0; o.status = kRegionOk;
    const u64 want = win_start(P.runs[i].start, P.W);
    if (want < P.lo || want >= P.hi) o.status = kRegionNotMine;
    else if (i > 0 && want <= win_start(P.runs[i - 1].end - 1, P.W)) o.status = kRegionChained;
    else replay_region<false>(P, i, o, nullptr, nullptr, 0);
    out[i] = o;
}

// Pass 2: the standing regions write their findings and strings at the offsets the host assigned.
__global__ __launch_bounds__(64) void replay_write_kernel(const ReplayParams P, const u64* region_index, const u64* fbase,
                                                          const u64* abase, u64 n_regions, sx_finding* findings, u8* arena) {
    const u64 k = (u64)blockIdx.x * 64 + threadIdx.x;
    if (k >= n_regions) return;
    ReplayRegionOut o;
    replay_region<true>(P, region_index[k], o, findings + fbase[k], arena + abase[k], abase[k]);
}

hipError_t launch_replay_count(const ReplayParams& P, ReplayRegionOut* out, hipStream_t stream) {
    if (P.n_runs == 0) return hipSuccess;
    hipLaunchKernelGGL(replay_count_kernel, dim3((unsigned)((P.n_runs + 63) / 64)), dim3(64), 0, stream, P, out);
    return hipGetLastError();
}
hipError_t launch_replay_write(const ReplayParams& P, const u64* region_index, const u64* fbase, const u64* abase,
                               u64 n_regions, sx_finding* findings, u8* arena, hipStream_t stream) {
    if (n_regions == 0) return hipSuccess;
    hipLaunchKernelGGL(replay_write_kernel, dim3((unsigned)((n_regions + 63) / 64)), dim3(64), 0, stream, P, region_index,
                       fbase, abase, n_regions, findings, arena);
    return hipGetLastError();
}

}  // namespace sx
