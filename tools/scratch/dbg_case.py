import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tools")
import fuzz_case, stringsext_amd as sx
c = fuzz_case.make(951892060)
fuzz_case.set_switches({"SX_REGION_CAP": "0"})
data = c["files"][0]
sc = sx.Scanner(c["missions"], device=0)
chunk = 16384
for k in range(12):
    res = sc.scan(data[k * chunk:(k + 1) * chunk], file_id=1)
    if k == 11:
        st = sc.stats()
        print("wave windows", st.wave_windows)
        for packed, v, n, arena, info in res.packed_segments():
            print("segment packed", packed, "n", n, "arena", len(arena))
            rows = []
            for i in range(n):
                f = v[i]
                if f.mission_id == 1 and f.position < 0x2c464 + 200:
                    rows.append((hex(f.position), f.str_off, f.str_len, arena[f.str_off:f.str_off + f.str_len]))
            for r in rows[:8]: print(r)
            # overlaps
            iv = sorted((v[i].str_off, v[i].str_off + v[i].str_len, v[i].mission_id, hex(v[i].position)) for i in range(n))
            for a, b in zip(iv, iv[1:]):
                if a[1] > b[0]: print("OVERLAP", a, b)
            gaps = [(a, b) for a, b in zip(iv, iv[1:]) if a[1] != b[0]]
            print("gaps/overlaps", len(gaps), gaps[:5])
    res.free()
