// Test-only harness: compiles the device replay core (sx_replay_core.hpp) as host code.
#include <stdint.h>
#include <string.h>
#define SXD inline
#define SXD_NOINLINE inline
#include "../../stringsext_amd/csrc/sx_replay_core.hpp"

extern "C" int sxd_replay_region_host(const sx::ReplayParams* P, uint64_t i, sx::ReplayRegionOut* o, sx_finding* fout,
                                      uint8_t* aout, uint32_t fcap, uint32_t acap) {
    sx::ReplayRegionOut c;
    sx::replay_region_any<0>(*P, i, c, nullptr, nullptr, 0);
    *o = c;
    if (c.n_find > fcap || c.n_bytes > acap) return -1;
    sx::ReplayRegionOut w;
    sx::replay_region_any<1>(*P, i, w, fout, aout, 0);
    if (!(w.end == c.end && w.n_find == c.n_find && w.n_bytes == c.n_bytes)) return -2;
    // pass 1 with the output cache: same counts; if it says "kept", the slot holds exactly pass 2's output
    // two geometries: the smallest slot and a roomy one
    for (uint64_t heads : { (uint64_t)1 << 40, (uint64_t)1 }) {
    const sx::CacheGeom g = sx::cache_geom(1 << 20, heads);
    sx_finding cf[64];
    uint8_t cs[4096];
    memset(cf, 0, sizeof cf); memset(cs, 0, sizeof cs);
    sx::ReplayRegionOut k;
    sx::replay_region_any<2>(*P, i, k, cf, cs, 0, g.cap_f, g.cap_b);
    if (!(k.end == c.end && k.n_find == c.n_find && k.n_bytes == c.n_bytes && k.status == c.status)) return -3;
    if (k.status == sx::kRegionOk) {
        const bool fits = c.n_find <= g.cap_f && c.n_bytes <= g.cap_b;
        if (k.pad && !fits) return -4;
        if (k.pad && (memcmp(cf, fout, c.n_find * sizeof(sx_finding)) != 0 || memcmp(cs, aout, c.n_bytes) != 0)) return -5;
        if (fits && !k.pad) return -6;
    }
    }
    return 0;
}

// Runs cut into pieces at the window starts they cross, exactly as the device does it (split_count / split_piece).
// (tests/test_replay_core.py mirrors ReplayParams with ctypes: the sizes must agree)
extern "C" uint64_t sxd_sizeof_replay_params() { return sizeof(sx::ReplayParams); }
extern "C" uint64_t sxd_split_runs_host(const sx::ReplayParams* P, sx_run* out, uint64_t cap) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < P->n_runs; i++) {
        uint64_t c;
        switch (sx::enc_family(P->encoding)) {
            case 1: c = sx::split_count<1>(*P, i); break;
            case 2: c = sx::split_count<2>(*P, i); break;
            case 3: c = sx::split_count<3>(*P, i); break;
            case 4: c = sx::split_count<4>(*P, i); break;
            case 5: c = sx::split_count<5>(*P, i); break;
            default: c = sx::split_count<0>(*P, i); break;
        }
        for (uint64_t k = 0; k < c; k++) { if (n < cap) out[n] = sx::split_piece(*P, i, k, c); n++; }
    }
    return n;
}

// The product's decoder (sx_codec_core.hpp: the source the device kernels are compiled from) on its own.
extern "C" void* sxd_decoder_new(int enc, const uint16_t* table) {
    sx::DDecoder* d = new sx::DDecoder;
    sx::ddec_reset(*d, enc, table);
    return d;
}
extern "C" int sxd_decoder_step(void* d, const uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, int last, uint32_t* rd, uint32_t* wr) {
    const sx::DStep r = sx::ddecode_any(*(sx::DDecoder*)d, src, n, dst, cap, last != 0);
    *rd = r.read; *wr = r.written;
    return r.result;
}
extern "C" void sxd_decoder_free(void* d) { delete (sx::DDecoder*)d; }
// how many bytes at a buffer start belong to the token pending in the decoder (two-byte family incl. gb18030)
extern "C" uint32_t sxd_entry_skip(const void* d, const uint8_t* next, uint64_t avail) {
    return sx::dbcs_entry_skip<4>(*(const sx::DDecoder*)d, next, avail);
}
