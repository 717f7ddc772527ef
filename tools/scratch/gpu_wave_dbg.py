"""debug: the ingest test's data, Mission by Mission, chunked through sx_scan (host buffers) and sx_scan_file"""
import os, random, sys, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import refconfig as rc, stringsext_amd as sx, sxo_binding as sxo
from product_harness import run_cli_product
from test_host_logic import synth
rng = random.Random(31)
data = synth(rng, (5 << 20) + 4096 * 3 + 77, 1 / 500)
path = os.path.join(tempfile.mkdtemp(), "image.bin"); open(path, "wb").write(data)
for encs in (["ascii"], ["utf-8"], ["utf-8", "ascii"], ["utf-8", "utf-16le", "ascii"]):
    ms = rc.missions(encodings=encs, chars_min="6")
    want = sxo.run_cli(ms, [data], radix="x")
    for how in ("scan", "file"):
        for chunk in (1 << 20, 64 << 10):
            if how == "scan":
                got = run_cli_product(ms, [data], radix="x", device=0, chunk_bytes=chunk)
            else:
                sc = sx.Scanner(ms, device=0)
                parts = sc.scan_file(path, chunk_bytes=chunk, file_id=1)
                got = sx.OUTPUT_BOM + b"".join(r.printed(n_inputs=1, radix="x") for r in parts) + b"\n"
                sc.close()
            ok = got == want
            print(encs, how, chunk, "OK" if ok else "MISMATCH", flush=True)
            if not ok:
                gl, wl = got.split(b"\n"), want.split(b"\n")
                for i, (a, b) in enumerate(zip(gl, wl)):
                    if a != b:
                        print("   line", i, "got", a[:100], "want", b[:100]); print("   before:", gl[i - 1][:100]); break
                print("   lines", len(gl), len(wl))
