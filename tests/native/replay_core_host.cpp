// Test-only harness: compiles the device replay core (sx_replay_core.hpp) as host code.
#include <stdint.h>
#include <string.h>
#define SXD inline
#include "../../stringsext_amd/csrc/sx_replay_core.hpp"

extern "C" int sxd_replay_region_host(const sx::ReplayParams* P, uint64_t i, sx::ReplayRegionOut* o, sx_finding* fout,
                                      uint8_t* aout, uint32_t fcap, uint32_t acap) {
    sx::ReplayRegionOut c;
    sx::replay_region<false>(*P, i, c, nullptr, nullptr, 0);
    *o = c;
    if (c.n_find > fcap || c.n_bytes > acap) return -1;
    sx::ReplayRegionOut w;
    sx::replay_region<true>(*P, i, w, fout, aout, 0);
    return (w.end == c.end && w.n_find == c.n_find && w.n_bytes == c.n_bytes) ? 0 : -2;
}
