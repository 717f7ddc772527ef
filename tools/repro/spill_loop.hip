// Stand-alone attempt at the round-5 wrong value (profiles/r06_spill_note.md): a 64-bit value computed before a loop whose body is divergent and
// whose tail is a DPP wave shift at loop level, forced into a spill slot by a register cap, used behind the loop.
//   hipcc --offload-arch=gfx950 -O3 -o spill_loop tools/repro/spill_loop.hip && ./spill_loop      (prints the lanes whose value came back wrong)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32; typedef uint64_t u64;
__device__ __forceinline__ u32 from_prev(u32 v, u32 edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xF, 0xF, false); }
// a body with enough live values to need every register the cap leaves
__device__ __forceinline__ u32 heavy(u32 in, const u32* __restrict__ tab, u32 lane) {
    u32 a[24];
#pragma unroll
    for (int k = 0; k < 24; k++) a[k] = tab[(in + 7u * k + lane) & 1023u];
    u32 s = in;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 24; k++) { s = s * 1664525u + a[k]; a[k] ^= s >> (k & 15); }
#pragma unroll
    for (int k = 0; k < 24; k++) s += a[k];
    return s & 7u;
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) void k(const u32* tab, const u32* seed, u64* out_ws, u32* out_rounds, u32* out_in) {
    const u32 lane = threadIdx.x;
    const u64 ws = (u64)seed[lane] * 0x100000001ull + lane;     // lives across the loop
    const bool active = lane < 61;
    u32 carry = seed[64], out = seed[lane] & 7u, in = from_prev(out, carry), rounds = 0;
    bool todo = true;
    for (;;) {
        if (todo && active) { out = heavy(in, tab, lane); rounds++; }
        else if (!active) out = in;
        const u32 pin = from_prev(out, carry);
        todo = active && pin != in;
        in = pin;
        if (!__ballot(todo)) break;
    }
    out_ws[lane] = ws + in;      // (in: so that the loop cannot be dropped)
    out_rounds[lane] = rounds;
    out_in[lane] = in;
}
int main() {
    u32 h_tab[1024], h_seed[65];
    for (int i = 0; i < 1024; i++) h_tab[i] = (u32)i * 2654435761u;
    u32 *d_tab, *d_seed, *d_r, *d_in; u64* d_ws;
    hipMalloc(&d_tab, sizeof h_tab); hipMalloc(&d_seed, sizeof h_seed); hipMalloc(&d_ws, 64 * 8); hipMalloc(&d_r, 64 * 4); hipMalloc(&d_in, 64 * 4);
    hipMemcpy(d_tab, h_tab, sizeof h_tab, hipMemcpyHostToDevice);
    int bad = 0;
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 65; i++) h_seed[i] = (u32)(trial * 977 + i * 131071) * 2246822519u;
        hipMemcpy(d_seed, h_seed, sizeof h_seed, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_tab, d_seed, d_ws, d_r, d_in);
        u64 ws[64]; u32 r[64], in[64];
        hipMemcpy(ws, d_ws, sizeof ws, hipMemcpyDeviceToHost); hipMemcpy(r, d_r, sizeof r, hipMemcpyDeviceToHost); hipMemcpy(in, d_in, sizeof in, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; l++) {
            const u64 want = (u64)h_seed[l] * 0x100000001ull + l + in[l];
            if (ws[l] != want) { if (bad < 10) printf("trial %d lane %d rounds %u: %llx != %llx\n", trial, l, r[l], (unsigned long long)ws[l], (unsigned long long)want); bad++; }
        }
    }
    printf("spill_loop: %d wrong values in 2000 launches\n", bad);
    return bad != 0;
}
