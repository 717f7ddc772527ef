// Test-only harness: the wave-cooperative stage B (stringsext_amd/csrc/sx_wave_core.hpp) compiled as host code and
// driven the way sx_wave_dev.hip drives it — wavefronts that own `nwin` windows, batches of 64 windows with one "lane"
// each, bit masks staged in an array that stands for LDS, entry states exchanged lane to lane until they are consistent,
// warm-up windows, the verification of the assumed entry states, count pass then write pass.  Only the wave intrinsics
// are replaced by loops; every per-lane function is the kernels' own.
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>
#define SXD inline
#define SXD_NOINLINE inline
#include "../../stringsext_amd/csrc/sx_device.hpp"
#include "../../stringsext_amd/csrc/sx_wave_core.hpp"

using namespace sx;

namespace {

struct CountEmit {
    u32 nf = 0, nb = 0;
    void operator()(u32, u32, bool, i32, u32, u32 out_len) { nf++; nb += out_len; }
};
struct WriteEmit {
    const WaveParams* P;
    sx_finding* f;
    u8* a;
    u64 a_off, win_pos;
    bool bad_len = false;   // the transcoded string is not as long as the count pass said
    void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) {
        const u64 soff = win_pos / kWvSlice * kWvSlice;
        sx_finding r;
        memset(&r, 0, sizeof r);
        r.position = P->consumed0 + win_pos + din;
        r.str_off = (u32)(a_off + P->str_off_base);
        r.str_len = out_len;
        if ((prec & 0xFFu) == WV_PROBE) {   // sx_wave_core.hpp WV_PROBE: this call starts at the slice's byte 0
            const u32 lb = (prec >> 8) & 511u, lback = (prec >> 17) & 1023u, hb = wv_probe_hb(prec), pend = wv_probe_pend(prec);
            const u64 avail = P->len - win_pos;
            if (P->family >= 4)   // (a leftover at a second call at byte 0: the bytes in front of the slice were the pending token's, not the leftover's)
                prec = wv_resolve_probe_dbcs((int)P->encoding, P->table, P->data + win_pos, avail < 32 ? (u32)avail : 32u,
                                             P->data + (win_pos - lback), lb ? lback - pend : 0u, lb, hb);
            else prec = wv_resolve_probe(P->data + win_pos, avail < 32 ? (u32)avail : 32u, P->data + (win_pos - lback), lb);
        }
        r.precision = (u8)prec;
        r.completes_previous = completes ? 1 : 0;
        r.mission_id = (u8)P->mission_id;
        r.input_file_id = (int16_t)P->file_id;
        r.slice_index = (u32)(soff / kWvSlice) + P->slice_base;
        *f++ = r;
        const u8* s = P->data + (u64)((long long)win_pos + src_rel);
        if (P->family >= 4) { if (wv_transcode_dbcs((int)P->encoding, P->table, s, src_len, a) != out_len) bad_len = true; }
        else if (P->family == 2) { if (wv_transcode_utf16(P->encoding == kEncUtf16be, s, src_len, a) != out_len) bad_len = true; }
        else if (out_len == src_len) memcpy(a, s, src_len);
        else {
            u32 w = 0;
            for (u32 t = 0; t < src_len; t++) {
                const u8 b = s[t];
                if (b < 0x80) a[w++] = b;
                else w += dput_cp(a + w, P->table ? (u32)P->table[b - 0x80] : 0xF780u + (b - 0x80u));
            }
        }
        a += out_len; a_off += out_len;
    }
};

// both drivers of one window (sx_wave_core.hpp: call by call — the statement — and stretch by stretch — what the kernels run) must
// emit the same findings and leave the same state
struct RecEmit {
    struct E { u32 din, prec; bool comp; i32 src; u32 len, out; };
    std::vector<E> v;
    void operator()(u32 din, u32 prec, bool completes, i32 src_rel, u32 src_len, u32 out_len) { v.push_back(E{ din, prec, completes, src_rel, src_len, out_len }); }
};
int g_driver_mismatch = 0, g_gave_up = 0;
unsigned long long g_rounds_total = 0, g_batches_total = 0, g_redo_lanes = 0, g_repairs = 0;
template <int KIND>
bool drivers_agree(const WvParams& WP, const WvWin& w, u32 in) {
    RecEmit a, b;
    WvState sa = wv_unpack(in), sb = wv_unpack(in);
    wv_window<KIND>(WP, w, sa, a);
    wv_window_calls<KIND>(WP, w, sb, b, false);
    if (getenv("SXW_DEBUG")) {
        bool same = wv_pack(sa) == wv_pack(sb) && a.v.size() == b.v.size();
        for (size_t i = 0; same && i < a.v.size(); i++) same = a.v[i].din == b.v[i].din && a.v[i].prec == b.v[i].prec && a.v[i].comp == b.v[i].comp && a.v[i].src == b.v[i].src && a.v[i].len == b.v[i].len && a.v[i].out == b.v[i].out;
        if (!same) {
            fprintf(stderr, "drivers differ: in %08x out %08x / %08x  n %u pre_empty %u tail_empty %u head_back %u head_pend %u\n", in, wv_pack(sa), wv_pack(sb), w.n, w.pre_empty, w.tail_empty, w.head_back, w.head_pend);
            fprintf(stderr, "  MBA %016llx%016llx\n  D  %016llx%016llx  mb0 %u/%u mbl %u\n", (unsigned long long)w.MBA.hi, (unsigned long long)w.MBA.lo, (unsigned long long)w.D.hi, (unsigned long long)w.D.lo, w.mb0_e, w.mb0_code, w.mbl_code);
            fprintf(stderr, "  E  %016llx%016llx\n  A  %016llx%016llx\n  F  %016llx%016llx\n  CS %016llx%016llx\n  PB %016llx%016llx\n", (unsigned long long)w.E.hi, (unsigned long long)w.E.lo, (unsigned long long)w.A.hi, (unsigned long long)w.A.lo, (unsigned long long)w.F.hi, (unsigned long long)w.F.lo, (unsigned long long)w.CS.hi, (unsigned long long)w.CS.lo, (unsigned long long)w.PB.hi, (unsigned long long)w.PB.lo);
            for (auto& e : a.v) fprintf(stderr, "  new: din %u prec %u comp %d src %d len %u out %u\n", e.din, e.prec, (int)e.comp, e.src, e.len, e.out);
            for (auto& e : b.v) fprintf(stderr, "  old: din %u prec %u comp %d src %d len %u out %u\n", e.din, e.prec, (int)e.comp, e.src, e.len, e.out);
        }
    }
    if (wv_pack(sa) != wv_pack(sb) || a.v.size() != b.v.size()) return false;
    for (size_t i = 0; i < a.v.size(); i++)
        if (a.v[i].din != b.v[i].din || a.v[i].prec != b.v[i].prec || a.v[i].comp != b.v[i].comp || a.v[i].src != b.v[i].src ||
            a.v[i].len != b.v[i].len || a.v[i].out != b.v[i].out) return false;
    return true;
}

// one wavefront, MODE 0 count / 1 write; returns false if an iteration did not settle (cannot happen)
template <int MODE>
bool wave(const WaveParams& P, u64 v, bool skip_idle, u32* rounds_max) {
    const u64 own_start = P.g_lo + v * P.nwin;
    if (own_start >= P.g_hi) return true;
    const u64 own_end = own_start + P.nwin < P.g_hi ? own_start + P.nwin : P.g_hi;
    // (as the kernel: a repair launch of the count pass / the writer after repairs start from what the wavefront in front left)
    bool known = false;
    u32 known_state = 0;
    if (v != 0 && ((MODE == 0 && P.redo) || (MODE == 1 && P.use_entry))) {
        known_state = P.wave_out[v - 1];
        if (MODE == 0 && (P.wave_in[v] == known_state || P.wave_in[v] == 0xFFFFFFFEu)) return true;
        known = true;
    }
    const u64 gw = (v == 0 || known) ? own_start : own_start - kWvWarm;
    u32 carry = v == 0 ? P.inject : (known ? known_state : 0u), assumed_in = carry;
    u32 tot_f = 0, tot_b = 0;
    u64 fbase = 0, abase = 0;
    if (MODE == 1) { fbase = P.wave_fbase[v] - P.f_sub; abase = P.wave_abase[v] - P.a_sub; }
    const WvParams WP{ P.q, P.n_min, P.grep_char >= 0 ? 1u : 0u, P.same };
    std::vector<u32> lds[9];
    u32 dbcs_cov = 0;
    bool dbcs_valid = false;
    for (auto& l : lds) l.assign(kWvMaxTiles * 32 + 8, 0xDEADBEEFu);   // stale bits must not matter
    for (u64 g0 = gw; g0 < own_end; g0 += kWvBatch) {
        u64 ws[64]; u32 wn[64]; bool active[64], owned[64];
        for (u32 l = 0; l < 64; l++) {
            const u64 g = g0 + l;
            active[l] = g < own_end; owned[l] = active[l] && g >= own_start;
            ws[l] = 0; wn[l] = 0;
            if (active[l]) wv_window_at(g, P.W, P.wps, P.len, &ws[l], &wn[l]);
        }
        const u32 n_act = own_end - g0 < kWvBatch ? (u32)(own_end - g0) : kWvBatch;
        const u64 span_lo = ws[0], span_hi = ws[n_act - 1] + wn[n_act - 1];
        const u64 tile0 = wv_tile0(span_lo);
        const u32 n_tiles = (u32)((span_hi - tile0 + kTileBytes - 1) / kTileBytes);
        if (n_tiles > kWvMaxTiles) return false;
        int t_first = 0;
        const u64 next_t0 = wv_tile0(span_hi);
        bool have_next = false;
        u32 cov_next = 0;
        if (P.family >= 4 && (g0 == gw || !dbcs_valid)) {   // as the kernel: back to a tile that holds a byte outside the lead range
            dbcs_cov = 0;
            long long lo = (long long)tile0;
            while (lo > 0) {
                lo -= kTileBytes; t_first--;
                bool any = false;
                for (u32 l = 0; l < 64 && !any; l++) {
                    const long long o = lo + 16ll * l;
                    if (o < 0) { any = true; break; }
                    for (int k = 0; k < 16; k++) any = any || !(P.lut[P.data[o + k]] & WVC_LEAD);
                }
                if (any) break;
            }
        }
        for (int t = t_first; t < (int)n_tiles; t++) {
            u32 outs[64];
            for (u32 l = 0; l < 64; l++) {
                const long long soff = (long long)tile0 + (long long)t * (long long)kTileBytes + 16ll * l;
                const u64 off = soff < 0 ? 0ull : (u64)soff;
                u32 xs[4] = { 0, 0, 0, 0 };
                const u32 avail = soff < 0 || off >= P.len ? 0u : (P.len - off >= 16 ? 16u : (u32)(P.len - off));
                for (u32 k = 0; k < avail; k++) xs[k >> 2] |= (u32)P.data[off + k] << (8 * (k & 3));
                const u32 idx = (u32)(t < 0 ? 0 : t) * 64 + l;
                if (P.family == 0 && P.swar.cls) {   // as the kernel: classes from ranges — and they must be the table's
                    const WvMasks16R r = wv_classify16_single_swar(P.swar, xs[0], xs[1], xs[2], xs[3], avail);
                    const WvMasks16 m = wv_classify16_single(P.lut, xs[0], xs[1], xs[2], xs[3], avail);
                    const u32 keep = avail >= 16 ? 0xFFFFu : ((1u << avail) - 1u);
                    if (m.v != keep || m.a != r.a || (m.o2 & m.a) != (P.swar.hi_len == 2 ? r.a & r.hi : 0u) || (m.o3 & m.a) != (P.swar.hi_len == 3 ? r.a & r.hi : 0u)) return false;
                    ((uint16_t*)lds[0].data())[idx] = (uint16_t)r.a;
                    ((uint16_t*)lds[1].data())[idx] = (uint16_t)r.hi;
                    continue;
                }
                if (P.family == 0) {
                    const WvMasks16 m = wv_classify16_single(P.lut, xs[0], xs[1], xs[2], xs[3], avail);
                    ((uint16_t*)lds[0].data())[idx] = (uint16_t)m.v;
                    ((uint16_t*)lds[1].data())[idx] = (uint16_t)m.a;
                    ((uint16_t*)lds[2].data())[idx] = (uint16_t)m.o2;
                    ((uint16_t*)lds[3].data())[idx] = (uint16_t)m.o3;
                    continue;
                }
                if (P.family == 2) {   // UTF-16: the lanes in order; what a lane hands on must not depend on what it was handed (else: give up, as the kernel)
                    static u32 c_prev;
                    const bool be = P.encoding == kEncUtf16be;
                    const u32 n_units = avail >> 1;
                    const WvU16Lane L = wv_utf16_lane_units(P.lut, be, xs, n_units);
                    u32 wbm = 0;
                    for (u32 j = 0; j <= 8; j++) { const u64 pp = off + 2 * j; if ((pp % kWvSlice) % P.W == 0 || pp == P.len) wbm |= 1u << j; }   // (the buffer's end ends a window too)
                    auto unit_at = [&](u64 pp) -> u32 { return be ? ((u32)P.data[pp] << 8) | P.data[pp + 1] : ((u32)P.data[pp + 1] << 8) | P.data[pp]; };
                    u32 prev_h = 0, prev_acc = 0, next_l = 0;
                    if (off >= 2 && avail) { const WvU16Unit u = wv_utf16_unit(P.lut, unit_at(off - 2)); prev_h = u.kind == 1; prev_acc = u.acc; }
                    if (avail == 16 && off + 18 <= P.len) next_l = wv_utf16_unit(P.lut, unit_at(off + 16)).kind == 2;
                    u32 c0;
                    if (t == 0 && l == 0) {   // the batch's first lane: whether its first unit is read in slow mode is not known — it cannot matter 16 bytes on, unless ...
                        c0 = prev_h;
                        if (prev_h && L.hm == 0xFFu) { g_gave_up = 1; return false; }
                    } else c0 = c_prev;
                    if (wv_utf16_transparent(L.hm, wbm)) { g_gave_up = 1; return false; }
                    c_prev = wv_utf16_chain(L.hm, wbm, 0u) >> 8;
                    const WvMasks16W m = wv_classify16_utf16(L, n_units, wbm, prev_h, prev_acc, c0, next_l);
                    if (m.exotic) { g_gave_up = 1; return false; }
                    const WvU16Packed pk = wv_utf16_pack(m, L.hm);
                    ((uint16_t*)lds[0].data())[idx] = (uint16_t)pk.m0;
                    ((uint16_t*)lds[1].data())[idx] = (uint16_t)pk.m1;
                    ((uint16_t*)lds[2].data())[idx] = (uint16_t)pk.m2;
                    ((uint16_t*)lds[3].data())[idx] = (uint16_t)pk.m3;
                    continue;
                }
                u32 back = 0, ahead = 0, n_ahead = 0;
                if (off >= 4 && avail) memcpy(&back, P.data + off - 4, 4);
                if (avail == 16 && off + 16 < P.len) {
                    n_ahead = P.len - (off + 16) >= 4 ? 4u : (u32)(P.len - (off + 16));
                    for (u32 k = 0; k < n_ahead; k++) ahead |= (u32)P.data[off + 16 + k] << (8 * k);
                }
                u8 b[24];
                const u32 ws6[6] = { back, xs[0], xs[1], xs[2], xs[3], ahead };
                for (int k = 0; k < 24; k++) b[k] = (u8)(ws6[k >> 2] >> (8 * (k & 3)));
                const u32 have_lo = off >= 4 ? 0u : 4u, have_hi = 4u + avail + n_ahead;
                if (P.family == 1 && P.swar.cls) {   // as the kernel: SWAR, five masks — which must be the statement's
                    const WvMasks16U m0 = wv_classify16_utf8(P.lut, b, have_lo, have_hi);
                    u32 wz[6] = { have_lo ? 0u : ws6[0], ws6[1], ws6[2], ws6[3], ws6[4], ws6[5] };
                    const WvMasks16V m = wv_classify16_utf8_swar<6>(P.swar, wz, avail);
                    if (m.e != m0.e || m.a != m0.a || m.f != m0.f || m.ma != m0.ma || m.mb != m0.mb) return false;
                    const u32 vals5[5] = { m.e, m.a, m.f, m.ma, m.mb };
                    for (int k = 0; k < 5; k++) ((uint16_t*)lds[k].data())[idx] = (uint16_t)vals5[k];
                    continue;
                }
                if (P.family == 1) {
                    const WvMasks16U m = wv_classify16_utf8(P.lut, b, have_lo, have_hi);
                    ((uint16_t*)lds[0].data())[idx] = (uint16_t)m.e;
                    ((uint16_t*)lds[1].data())[idx] = (uint16_t)m.a;
                    ((uint16_t*)lds[2].data())[idx] = (uint16_t)m.f;
                    ((uint16_t*)lds[3].data())[idx] = (uint16_t)m.ma;
                    ((uint16_t*)lds[4].data())[idx] = (uint16_t)m.mb;
                    ((uint16_t*)lds[5].data())[idx] = (uint16_t)m.g;
                    continue;
                }
                if (P.family == 5) {   // EUC-JP: the statement byte by byte, the kernels' bit arithmetic next to it, marks beyond the lane handed on
                    static u32 spill[5];   // (of the lane in front: e, a, f, ma, mb bits 16..17)
                    if (t == t_first && l == 0) for (u32& v5 : spill) v5 = 0;
                    u32 cov_in = l == 0 ? dbcs_cov : outs[l - 1];
                    if (soff == 0) cov_in = P.entry_skip;
                    const u32 n_exist = avail + n_ahead;
                    u8 bz[24];
                    for (int k = 0; k < 24; k++) bz[k] = ((u32)k >= have_lo && (u32)k < have_hi) ? b[k] : (u8)0;
                    u32 over = 0;
                    const WvMasks18 m0 = wv_eucjp_walk(P.lut, P.pairs2, P.swar.kana, bz, n_exist, cov_in, &over);
                    const WvEucPre pc = wv_eucjp_classes_swar<6>(P.swar, &ws6[1], n_exist);
                    WvEucOrbit ob = wv_eucjp_orbit_init(pc);
                    while (wv_eucjp_orbit_step(ob)) {}
                    const WvMasks18 m = wv_classify16_eucjp_swar(P.pairs2, P.swar.kana, ws6, pc, ob, cov_in, n_exist);
                    if (soff >= 0 && avail == 16 && wv_eucjp_over(ob, cov_in) != over) return false;
                    if (m.e != m0.e || m.a != m0.a || m.f != m0.f || m.ma != m0.ma || m.mb != m0.mb) return false;
                    outs[l] = soff < 0 ? 0u : (avail == 16 ? over : 0u);
                    if (l == 63) {
                        dbcs_cov = outs[63];
                        if ((long long)tile0 + (long long)(t + 1) * (long long)kTileBytes == (long long)next_t0) { cov_next = dbcs_cov; have_next = true; }
                    }
                    const u32 vals5[5] = { m.e, m.a, m.f, m.ma, m.mb };
                    for (int k = 0; k < 5; k++) {
                        if (t >= 0) ((uint16_t*)lds[k].data())[idx] = (uint16_t)(vals5[k] | spill[k]);
                        spill[k] = vals5[k] >> 16;
                    }
                    continue;
                }
                // the two-byte family: the lanes in order (the kernel composes their in -> out functions along the wavefront)
                const u32 lr = soff < 0 ? 0u : wv_dbcs_lead_mask(P.lut, b, have_hi);
                u32 o0, o1;
                const u32 s0 = wv_dbcs_walk(lr, 0, &o0), s1 = wv_dbcs_walk(lr, 1, &o1);
                u32 cov_in = l == 0 ? dbcs_cov : outs[l - 1];
                if (soff == 0) cov_in = P.entry_skip ? 1u : 0u;
                outs[l] = cov_in ? o1 : o0;
                if (l == 63) {
                    dbcs_cov = outs[63];
                    if ((long long)tile0 + (long long)(t + 1) * (long long)kTileBytes == (long long)next_t0) { cov_next = dbcs_cov; have_next = true; }
                }
                if (t < 0) continue;
                const WvMasks16D m0 = wv_classify16_dbcs(P.lut, P.pairs, b, have_lo, have_hi, lr, cov_in ? s1 : s0, cov_in);
                if (P.swar.cls) {   // as the kernel: SWAR classes, 2 bits per pair, five masks — which must be the statement's
                    const WvDbcsPreS ps = wv_dbcs_classes_swar<6>(P.swar, &ws6[1], avail);
                    const u32 t0 = wv_dbcs_trails(ps.lr, 0u), t1 = wv_dbcs_trails(ps.lr, 1u);
                    if ((t0 >> 16) != o0 || (t1 >> 16) != o1) return false;
                    const WvMasks16E m = wv_classify16_dbcs_swar(P.pairs2, ws6, ps, cov_in ? t1 : t0, cov_in, off > 0, avail + n_ahead);
                    if (m.e != m0.e || m.a != m0.a || m.f != m0.f || m.ma != m0.ma || m.mb != m0.mb) return false;
                    const u32 vals5[5] = { m.e, m.a, m.f, m.ma, m.mb };
                    for (int k = 0; k < 5; k++) ((uint16_t*)lds[k].data())[idx] = (uint16_t)vals5[k];
                    continue;
                }
                // the kernels' path: the same as bit arithmetic — compared with the statement byte by byte above
                const WvDbcsPre pc = wv_dbcs_classes(P.lut, &ws6[1], avail);
                const u32 tr0 = wv_dbcs_trails(pc.lr, 0u), tr1 = wv_dbcs_trails(pc.lr, 1u);
                if ((tr0 >> 16) != o0 || (tr1 >> 16) != o1 || (~tr0 & 0xFFFFu) != s0 || (~tr1 & 0xFFFFu & ~1u) != (s1 & ~1u)) return false;
                const WvMasks16D m = wv_classify16_dbcs_bits(P.pairs, ws6, pc, cov_in ? tr1 : tr0, cov_in, off > 0, avail + n_ahead);
                if (m.e != m0.e || m.a != m0.a || m.f != m0.f || m.g != m0.g || m.ma != m0.ma || m.mb != m0.mb || m.o2 != m0.o2 ||
                    m.o3 != m0.o3 || m.o4 != m0.o4) return false;
                const u32 vals[9] = { m.e, m.a, m.f, m.ma, m.mb, m.g, m.o2, m.o3, m.o4 };
                for (int k = 0; k < 9; k++) ((uint16_t*)lds[k].data())[idx] = (uint16_t)vals[k];
            }
        }
        if (P.family >= 4) { dbcs_valid = have_next; if (have_next) dbcs_cov = cov_next; }
        WvWin w[64];
        for (u32 l = 0; l < 64; l++) {
            const u32 o = active[l] ? (u32)(ws[l] - tile0) : 0u, n = active[l] ? wn[l] : 0u;
            if (P.family == 0 && P.swar.cls) w[l] = wv_win_single_swar(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), P.swar.hi_len, n, P.n_min);
            else if (P.family == 0)
                w[l] = wv_win_single(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[3], o, n), n, P.n_min);
            else if (P.family == 2) {
                const bool hb = active[l] && ws[l] >= 2 && o >= 2 && wv_extract(lds[2], o - 2, 1).lo;
                w[l] = wv_win_utf16(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[3], o, n), hb,
                                    ws[l] % kWvSlice == 0, n, P.n_min);
                if (active[l] && n) {   // the statement: the decoder's state machine over the window, byte by byte
                    WvMasks16W sm;
                    u32 acc_in = 0, acc_out = 0;
                    if (hb) { const u64 pp = ws[l] - 2; const bool be = P.encoding == kEncUtf16be;
                              acc_in = wv_utf16_unit(P.lut, be ? ((u32)P.data[pp] << 8) | P.data[pp + 1] : ((u32)P.data[pp + 1] << 8) | P.data[pp]).acc; }
                    // (128 bytes: four words per mask)
                    WvMask E{ 0, 0 }, A{ 0, 0 }, F{ 0, 0 }, MA{ 0, 0 }, MB{ 0, 0 }, O2{ 0, 0 }, O3{ 0, 0 }, O4{ 0, 0 };
                    bool hs = hb;
                    bool exo = false;
                    // the statement works on <= 32 bytes at a time only through its masks' width: run it over the window in one piece with 128-bit marks
                    {
                        const u8* wp = P.data + ws[l];
                        bool h = hs; u32 hacc = acc_in; i32 hstart = -2;
                        auto put = [&](i32 fs, u32 end, u32 acc, u32 len) {
                            if (fs >= 0) F = wm_or(F, wm_bit((u32)fs));
                            E = wm_or(E, wm_bit(end)); if (acc) A = wm_or(A, wm_bit(end));
                            if (len >= 2) O2 = wm_or(O2, wm_bit(end));
                            if (len >= 3) O3 = wm_or(O3, wm_bit(end));
                            if (len >= 4) O4 = wm_or(O4, wm_bit(end));
                        };
                        const bool be = P.encoding == kEncUtf16be;
                        for (u32 i = 0; i + 2 <= n; i += 2) {
                            const u32 u = be ? ((u32)wp[i] << 8) | wp[i + 1] : ((u32)wp[i + 1] << 8) | wp[i];
                            const WvU16Unit x = wv_utf16_unit(P.lut, u);
                            const bool last_unit = i + 4 > n;
                            if (h) {
                                if (x.kind == 2) { put(hstart, i + 1, hacc, 4); h = false; }
                                else if (x.kind == 1) { MA = wm_or(MA, wm_bit(i + 1)); hstart = (i32)i; hacc = x.acc; }
                                else { MB = wm_or(MB, wm_bit(i)); put((i32)i, i + 1, x.acc, x.len); if (last_unit) exo = true; h = false; }
                                continue;
                            }
                            if (x.kind == 0) put((i32)i, i + 1, x.acc, x.len);
                            else if (x.kind == 2) MA = wm_or(MA, wm_bit(i + 1));
                            else if (last_unit) { h = true; hstart = (i32)i; hacc = x.acc; }
                            else {
                                const u32 v2 = be ? ((u32)wp[i + 2] << 8) | wp[i + 3] : ((u32)wp[i + 3] << 8) | wp[i + 2];
                                if (wv_utf16_unit(P.lut, v2).kind == 2) { put((i32)i, i + 3, x.acc, 4); i += 2; }
                                else MA = wm_or(MA, wm_bit(i + 1));
                            }
                        }
                        (void)sm; (void)acc_out;
                    }
                    if (exo) { if (getenv("SXW_DEBUG")) fprintf(stderr, "exo not flagged: window at %llu\n", (unsigned long long)ws[l]); return false; }   // (the lanes must have given up)
                    const WvMask MAw = wm_and(wv_extract(lds[2], o, n), WvMask{ 0xAAAAAAAAAAAAAAAAull, 0xAAAAAAAAAAAAAAAAull });
                    auto eq = [](WvMask a, WvMask b) { return a.lo == b.lo && a.hi == b.hi; };
                    // (a window's last unit, a high surrogate with a low one behind it: the lanes mark the character's first byte there, its
                    // end lies in the next window — a mark nothing in this window looks at)
                    if (n >= 2 && wm_test(wv_extract(lds[2], o, n), n - 2)) F = wm_or(F, wm_and(w[l].F, wm_bit(n - 2)));
                    if (!eq(w[l].E, E) || !eq(w[l].A, A) || !eq(w[l].F, F) || !eq(MAw, MA) || !eq(w[l].PB, MB) || !eq(w[l].O2, O2) || !eq(w[l].O3, O3) ||
                        !eq(w[l].O4, O4)) {
                        if (getenv("SXW_DEBUG")) {
                            fprintf(stderr, "utf16 masks differ: window at %llu n %u hb %d\n", (unsigned long long)ws[l], n, (int)hb);
                            const char* nm[8] = { "E", "A", "F", "MA", "MB", "O2", "O3", "O4" };
                            const WvMask gotm[8] = { w[l].E, w[l].A, w[l].F, MAw, w[l].PB, w[l].O2, w[l].O3, w[l].O4 }, wantm[8] = { E, A, F, MA, MB, O2, O3, O4 };
                            for (int k = 0; k < 8; k++) if (!eq(gotm[k], wantm[k])) fprintf(stderr, "  %s got %016llx%016llx want %016llx%016llx\n", nm[k], (unsigned long long)gotm[k].hi, (unsigned long long)gotm[k].lo, (unsigned long long)wantm[k].hi, (unsigned long long)wantm[k].lo);
                        }
                        return false;
                    }
                }
            }
            else if (P.family == 5) {
                auto bit = [&](int k, u32 at) -> u32 { return (u32)wv_extract(lds[k], at, 1).lo; };
                const bool has1 = ws[l] >= 1 && o >= 1, has2 = ws[l] >= 2 && o >= 2;
                const bool done1 = !has1 || (bit(0, o - 1) | bit(3, o - 1)) != 0, done2 = !has2 || (bit(0, o - 2) | bit(3, o - 2)) != 0;
                w[l] = wv_win_eucjp_swar(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[3], o, n),
                                         wv_extract(lds[4], o, n), P.swar.hi_len, done1, done2, has1 && bit(2, o - 1), has2 && bit(2, o - 2), has1, has2,
                                         ws[l] % kWvSlice == 0, n, P.n_min);
            } else if (P.family == 4) {
                const u32 eb = o >= 1 ? (u32)wv_extract(lds[0], o - 1, 1).lo : 1u, mab = o >= 1 ? (u32)wv_extract(lds[3], o - 1, 1).lo : 0u;
                const u32 fb1 = o >= 1 ? (u32)wv_extract(lds[2], o - 1, 1).lo : 0u;
                if (P.swar.cls) w[l] = wv_win_dbcs_swar(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[3], o, n),
                                                        wv_extract(lds[4], o, n), P.swar.hi_len, (eb | mab) != 0, fb1 != 0, ws[l] > 0, ws[l] % kWvSlice == 0, n, P.n_min);
                else w[l] = wv_win_dbcs(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[5], o, n),
                                   wv_extract(lds[3], o, n), wv_extract(lds[4], o, n), wv_extract(lds[6], o, n), wv_extract(lds[7], o, n),
                                   wv_extract(lds[8], o, n), (eb | mab) != 0, fb1 != 0, ws[l] > 0, ws[l] % kWvSlice == 0, n, P.n_min);
            } else {
                const u32 fb = o >= 3 ? (u32)wv_extract(lds[2], o - 3, 3).lo : 0u;
                if (P.swar.cls) {
                    const WvMask A_ = wv_extract(lds[1], o, n), F_ = wv_extract(lds[2], o, n);
                    w[l] = wv_win_utf8(wv_extract(lds[0], o, n), A_, F_, wv_utf8_good_from(A_, F_), wv_extract(lds[3], o, n), wv_extract(lds[4], o, n), fb,
                                       ws[l] % kWvSlice == 0, n, P.n_min);
                } else
                w[l] = wv_win_utf8(wv_extract(lds[0], o, n), wv_extract(lds[1], o, n), wv_extract(lds[2], o, n), wv_extract(lds[5], o, n),
                                   wv_extract(lds[3], o, n), wv_extract(lds[4], o, n), fb, ws[l] % kWvSlice == 0, n, P.n_min);
            }
        }
        for (u32 l = 0; l < 64 && WP.grep; l++) {   // -g: which characters are the grep char (every lane from its own window's bytes, as the kernel)
            if (!active[l]) { w[l].GC = wm_zero(); continue; }
            const bool be = P.encoding == kEncUtf16be;
            if (P.family == 0) wv_set_grep<0>(w[l], WP, P.data + ws[l], (u32)P.grep_char, be);
            else if (P.family == 1) wv_set_grep<1>(w[l], WP, P.data + ws[l], (u32)P.grep_char, be);
            else if (P.family == 2) wv_set_grep<3>(w[l], WP, P.data + ws[l], (u32)P.grep_char, be);
            else wv_set_grep<2>(w[l], WP, P.data + ws[l], (u32)P.grep_char, be);
        }
        if (!WP.grep) for (u32 l = 0; l < 64; l++) w[l].GC = wm_zero();
        for (u32 l = 0; l < 64; l++) {   // -r: the accepted multi-byte characters and where their lead byte changes (as the kernel)
            if (!active[l] || !WP.same || !wn[l]) { w[l].MBA = wm_zero(); w[l].D = wm_zero(); w[l].mb0_e = 128; w[l].mb0_code = 0; w[l].mbl_code = 0; continue; }
            const bool be = P.encoding == kEncUtf16be;
            if (P.family == 0) wv_set_same<0>(w[l], P.data + ws[l], P.ubf, be, WvLeadOfTable{ P.table });
            else if (P.family == 1) wv_set_same<1>(w[l], P.data + ws[l], P.ubf, be, WvLeadOfTable{ P.table });
            else wv_set_same<3>(w[l], P.data + ws[l], P.ubf, be, WvLeadOfTable{ P.table });
        }
        u32 in[64], out[64], nf[64], nb[64];
        std::vector<u32> stage(kWvStageSame * 192u, 0xDEADBEEFu);   // (the kernel's count pass stages a window's first findings as descriptors in LDS)
        bool todo[64], injected[64];
        // (as the kernel: the exchange starts from every window's guess of what it hands on, sx_wave_core.hpp wv_exit_guess)
        for (u32 l = 0; l < 64; l++)
            out[l] = !active[l] ? 0u : P.family == 0 ? wv_exit_guess<0>(WP, w[l]) : P.family == 1 ? wv_exit_guess<1>(WP, w[l]) : P.family == 2 ? wv_exit_guess<3>(WP, w[l]) : wv_exit_guess<2>(WP, w[l]);
        for (u32 l = 0; l < 64; l++) {
            injected[l] = g0 + l == P.g_lo;
            in[l] = l == 0 ? carry : out[l - 1];
            if (injected[l]) in[l] = P.inject;
            nf[l] = nb[l] = 0; todo[l] = true;
        }
        u32 rounds = 0;
        for (;;) {
            if (++rounds > 70) return false;
            for (u32 l = 0; l < 64; l++) {
                if (todo[l] && active[l]) {
                    WvState st = wv_unpack(in[l]);
                    WvStageEmit<u32*> ce;
                    ce.cap = P.same ? kWvStageSame : kWvStage;
                    ce.stage = stage.data(); ce.lane = l;
                    ce.widx = (u32)(g0 + l - own_start);
                    if (P.family == 0) wv_window<0>(WP, w[l], st, ce, skip_idle);
                    else if (P.family == 1) wv_window<1>(WP, w[l], st, ce, skip_idle);
                    else if (P.family == 2) wv_window<3>(WP, w[l], st, ce, skip_idle);
                    else wv_window<2>(WP, w[l], st, ce, skip_idle);
                    out[l] = wv_pack(st); nf[l] = ce.nf; nb[l] = ce.nb;
                } else if (!active[l]) out[l] = in[l];
            }
            bool any = false;
            u32 pin[64];
            for (u32 l = 0; l < 64; l++) pin[l] = l == 0 ? carry : out[l - 1];
            for (u32 l = 0; l < 64; l++) {
                if (injected[l]) pin[l] = P.inject;
                todo[l] = active[l] && pin[l] != in[l];
                if (MODE == 0 && todo[l]) g_redo_lanes++;
                in[l] = pin[l];
                any = any || todo[l];
            }
            if (!any) break;
        }
        if (rounds > *rounds_max) *rounds_max = rounds;
        if (MODE == 0) { g_rounds_total += rounds; g_batches_total++; }
        if (MODE == 0)
            for (u32 l = 0; l < 64; l++)
                if (active[l] && !(P.family == 0 ? drivers_agree<0>(WP, w[l], in[l]) : P.family == 1 ? drivers_agree<1>(WP, w[l], in[l]) : P.family == 2 ? drivers_agree<3>(WP, w[l], in[l]) : drivers_agree<2>(WP, w[l], in[l]))) {
                    g_driver_mismatch++;
                    return false;
                }
        if (g0 == gw && v != 0 && !known) assumed_in = in[kWvWarm];
        carry = out[63];
        const u32 last_out = out[n_act - 1];
        u32 bf = 0, bb = 0;
        for (u32 l = 0; l < 64; l++) {
            if (!owned[l]) { nf[l] = 0; nb[l] = 0; }
            if (MODE == 0 && P.desc && nf[l]) {   // as the kernel: the descriptors of the lane-per-finding writer
                const u32 at = tot_f + bf, ab = tot_b + bb;
                WvDesc* slot = (WvDesc*)P.desc + v * (u64)P.desc_cap + at;
                const u32 room = at < P.desc_cap ? P.desc_cap - at : 0u;
                if (nf[l] <= (P.same ? kWvStageSame : kWvStage)) {
                    const u32 k = nf[l] < room ? nf[l] : room;
                    for (u32 j = 0; j < k; j++) {
                        const u32* sp = stage.data() + j * 192u + l;
                        slot[j] = WvDesc{ sp[0] + ab, sp[64], sp[128] };
                    }
                } else {
                    WvDescEmit de{ slot, room, ab, (u32)(g0 + l - own_start) };
                    WvState st = wv_unpack(in[l]);
                    if (P.family == 0) wv_window<0>(WP, w[l], st, de, skip_idle);
                    else if (P.family == 1) wv_window<1>(WP, w[l], st, de, skip_idle);
                    else if (P.family == 2) wv_window<3>(WP, w[l], st, de, skip_idle);
                    else wv_window<2>(WP, w[l], st, de, skip_idle);
                    if (de.a_local != ab + nb[l]) return false;
                }
            }
            if (MODE == 1 && (nf[l] | nb[l])) {
                const u64 fo = fbase + tot_f + bf, ao = abase + tot_b + bb;
                WriteEmit we{ &P, P.findings + fo, P.arena + ao, ao, ws[l] };
                WvState st = wv_unpack(in[l]);
                if (P.family == 0) wv_window<0>(WP, w[l], st, we, skip_idle);
                else if (P.family == 1) wv_window<1>(WP, w[l], st, we, skip_idle);
                else if (P.family == 2) wv_window<3>(WP, w[l], st, we, skip_idle);
                else wv_window<2>(WP, w[l], st, we, skip_idle);
                if ((u64)(we.f - (P.findings + fo)) != nf[l] || we.a_off - ao != nb[l] || we.bad_len) return false;   // both passes must agree
            }
            bf += nf[l]; bb += nb[l];
        }
        if (bf >= (1u << 14) || bb >= (1u << 18)) return false;   // the kernels pack a batch's totals into 32 bits
        tot_f += bf; tot_b += bb;
        if (g0 + kWvBatch >= own_end && MODE == 0)
            P.wave_out[v] = last_out | (P.family >= 4 && own_end == P.g_hi ? w[n_act - 1].tail_pend << 27 : 0u);
    }
    if (MODE == 0) { P.wave_nf[v] = tot_f; P.wave_nb[v] = tot_b; P.wave_in[v] = assumed_in; }
    return true;
}

}  // namespace

// Replays windows [g_lo, all) of the buffer from the exact state `inject` at window g_lo.  Returns 0, or -1 (a pass did not
// settle / the passes disagree), -2 (output capacity).  *bad_waves = wavefronts whose assumed entry state was wrong.
extern "C" int sxw_emulate(const uint8_t* data, uint64_t len, uint64_t consumed0, uint32_t slice_base, uint32_t W, uint32_t q, uint32_t n_min,
                           uint64_t g_lo, uint32_t inject, uint32_t nwin, const uint8_t* lut, const uint16_t* table, int mission_id,
                           int file_id, sx_finding* fout, uint64_t fcap, uint8_t* aout, uint64_t acap, uint64_t* nf, uint64_t* nb,
                           uint32_t* final_state, uint64_t* bad_waves, int skip_idle, uint32_t* rounds_max, uint32_t family,
                           const uint32_t* pairs, uint32_t encoding, uint32_t entry_skip, const uint32_t* swar25, const uint32_t* pairs2, int grep_char, int same, uint64_t ubf) {
    WaveParams P;
    memset(&P, 0, sizeof P);
    P.data = data; P.len = len; P.consumed0 = consumed0; P.slice_base = slice_base; P.W = W; P.wps = wv_wps(W); P.q = q; P.n_min = n_min;
    P.v0 = 0; P.v1 = ~0ull;
    P.g_lo = g_lo; P.g_hi = wv_window_count(len, W); P.nwin = nwin; P.inject = inject; P.mission_id = mission_id; P.file_id = file_id;
    P.lut = lut; P.table = table; P.family = family; P.pairs = pairs; P.encoding = encoding; P.entry_skip = entry_skip;
    if (swar25) memcpy(&P.swar, swar25, sizeof P.swar);
    P.pairs2 = pairs2;
    P.grep_char = grep_char;
    P.same = same && family <= 2 ? 1u : 0u; P.ubf = ubf;
    if (family == 4 && !pairs2) P.swar.cls = 0;
    if (family == 5 && (!pairs2 || !P.swar.cls)) return -8;
    *nf = *nb = 0; *bad_waves = 0; *final_state = inject; *rounds_max = 0;
    g_gave_up = 0;
    if (P.g_hi <= P.g_lo) return 0;
    const u64 n_waves = (P.g_hi - P.g_lo + nwin - 1) / nwin;
    std::vector<u32> wnf(n_waves), wnb(n_waves), win(n_waves), wout(n_waves);
    std::vector<u64> fb(n_waves), ab(n_waves);
    P.wave_nf = wnf.data(); P.wave_nb = wnb.data(); P.wave_in = win.data(); P.wave_out = wout.data();
    // descriptors: two per window, or few enough that some wavefronts overflow (then only the window-parallel writer's output is checked)
    P.desc_cap = (nwin & 3u) == 1u ? nwin / 8 + 1 : (P.same ? 16 : 2) * nwin + 64;   // (-r: as sx_wave.cpp)
    std::vector<u32> desc((size_t)(n_waves * P.desc_cap * 3 + 3), 0xDEADBEEFu);
    P.desc = nwin <= kWvDescMaxWin ? desc.data() : nullptr;   // (as sx_wave.cpp: larger wavefronts do without)
    for (u64 v = 0; v < n_waves; v++) if (!wave<0>(P, v, skip_idle != 0, rounds_max)) return g_gave_up ? -9 : g_driver_mismatch ? -7 : -1;
    // repairs (sx_wave.cpp): wavefronts whose assumption was wrong run again from what their predecessor left, launch after launch.  A
    // launch's wavefronts run in any order; here from the last to the first, so that every one reads what the launch BEFORE left in front of
    // it — the order in which a chain of wrong wavefronts takes the most launches.
    auto count_bad = [&]() { u64 b = 0; for (u64 v = 1; v < n_waves; v++) if (win[v] != wout[v - 1]) b++; return b; };
    bool repaired = false;
    if (!getenv("SXW_NO_REPAIR")) {
        for (int round = 0; round < 400 && count_bad(); round++) {
            repaired = true;
            g_repairs++;
            P.redo = 1;
            bool any = false;
            for (u64 v = n_waves - 1; v >= 1; v--) {
                if (win[v] == wout[v - 1] || win[v] == 0xFFFFFFFEu) continue;
                any = true;
                if (!wave<0>(P, v, skip_idle != 0, rounds_max)) return g_gave_up ? -9 : g_driver_mismatch ? -7 : -1;
            }
            if (!any) break;
        }
    }
    P.redo = 0;
    u64 f = 0, a = 0;
    for (u64 v = 0; v < n_waves; v++) {
        fb[v] = f; ab[v] = a; f += wnf[v]; a += wnb[v];
        if (v > 0 && win[v] != wout[v - 1]) (*bad_waves)++;
    }
    P.use_entry = repaired && family < 2 ? 1u : 0u;
    const bool skip_window_writer = repaired && family >= 2;   // (the product goes back to the other path for that combination unless the descriptors hold everything)
    *nf = f; *nb = a; *final_state = wout[n_waves - 1];
    if (f > fcap || a > acap) return -2;
    P.wave_fbase = fb.data(); P.wave_abase = ab.data(); P.findings = fout; P.arena = aout;
    u32 dummy = 0;
    bool overflow = false;
    for (u64 v = 0; v < n_waves; v++) overflow = overflow || wnf[v] > P.desc_cap;
    if (skip_window_writer && (overflow || !P.desc)) return -10;   // (the product: SX_WAVE_FALLBACK)
    if (!skip_window_writer) for (u64 v = 0; v < n_waves; v++) if (!wave<1>(P, v, skip_idle != 0, &dummy)) return -1;
    // the lane-per-finding writer (sx_wave_dev.hip wave_emit_kernel) from the count pass' descriptors: the same records and strings
    if (!overflow && P.desc) {
        std::vector<sx_finding> f2((size_t)f + 1);
        std::vector<u8> a2((size_t)a + 8, 0);
        for (u64 v = 0; v < n_waves; v++) {
            const u64 own_start = P.g_lo + v * P.nwin;
            const WvDesc* d = (const WvDesc*)P.desc + v * (u64)P.desc_cap;
            for (u32 i = 0; i < wnf[v]; i++) {
                const WvDesc x = d[i];
                u64 ws; u32 wn;
                wv_window_at(own_start + wv_desc_widx(x), P.W, P.wps, P.len, &ws, &wn);
                const u64 ao = ab[v] + wv_desc_a_local(x);
                WriteEmit we{ &P, f2.data() + fb[v] + i, a2.data() + ao, ao, ws };
                we(wv_desc_din(x), wv_desc_prec(x), wv_desc_completes(x), wv_desc_src_rel(x), wv_desc_src_len(x), wv_desc_out_len(x));
                if (we.bad_len) return -3;
            }
        }
        if (skip_window_writer) { if (f) memcpy(fout, f2.data(), (size_t)f * sizeof(sx_finding)); if (a) memcpy(aout, a2.data(), (size_t)a); }
        if (f && memcmp(f2.data(), fout, (size_t)f * sizeof(sx_finding)) != 0) return -4;
        if (a && memcmp(a2.data(), aout, (size_t)a) != 0) return -5;
    }
    return 0;
}

extern "C" uint32_t sxw_pack_state(uint32_t lc, uint32_t lb, uint32_t lback, uint32_t cut) { return wv_pack(WvState{ lc, lb, lback, cut, 0 }); }
extern "C" uint32_t sxw_pack_state_g(uint32_t lc, uint32_t lb, uint32_t lback, uint32_t cut, uint32_t lg) { return wv_pack(WvState{ lc, lb, lback, cut, lg }); }

extern "C" uint32_t sxw_pack_state_m(uint32_t lc, uint32_t lb, uint32_t lback, uint32_t cut, uint32_t lm) { return wv_pack(WvState{ lc, lb, lback, cut, 0, lm }); }
extern "C" uint32_t sxw_lead_code(uint64_t ubf, uint32_t lead) { return wv_lead_code(ubf, lead); }

extern "C" void sxw_round_stats(unsigned long long* out) { out[0] = g_rounds_total; out[1] = g_batches_total; out[2] = g_redo_lanes; out[3] = g_repairs; g_rounds_total = g_batches_total = g_redo_lanes = g_repairs = 0; }
