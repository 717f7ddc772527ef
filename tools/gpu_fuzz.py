"""Long-running randomized parity check on the GPU (or, with SX_FUZZ_HOST=1, of the host replay stage alone,
fed with the oracle's runs — no GPU needed): random missions (encodings, -n, -q, filters,
-g, -r), random inputs (planted strings, adversarial soup, dense strings, text), random chunking,
stage-B placement and alternative code paths (the environment switches of DESIGN.md §9), each compared
byte for byte with the oracle's CLI output.  A failing case is replayed with tools/gpu_repro.py CASE_SEED.
SX_FUZZ_TRACE=FILE: the case in hand is written there before it runs (a hang or a fault then names its case).
usage: tools/gpu_fuzz.py SECONDS [SEED]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_case
import stringsext_amd as sx, sxo_binding as sxo
from product_harness import run_cli_product

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = random.Random(seed)
host_only = bool(os.environ.get("SX_FUZZ_HOST"))
t0 = time.time(); n = 0
while time.time() - t0 < budget:
    n += 1
    case_seed = rng.randrange(1 << 31)
    c = fuzz_case.make(case_seed)
    want = sxo.run_cli(c["missions"], c["files"], radix="x", flush_at_eof=c["flush"])
    fuzz_case.set_switches({} if host_only else c["switches"])
    if os.environ.get("SX_FUZZ_TRACE"):
        with open(os.environ["SX_FUZZ_TRACE"], "w") as tf:
            tf.write(f"seed {seed} case {n} case_seed {case_seed}: {fuzz_case.describe(c)}\n")
    try:
        if host_only:
            got = run_cli_product(c["missions"], c["files"], radix="x", chunk_bytes=c["chunk"], device=None, flush_at_eof=c["flush"])
        else:
            got = run_cli_product(c["missions"], c["files"], radix="x", chunk_bytes=c["chunk"], device=0, subchunk_bytes=c["sub"],
                                  flush_at_eof=c["flush"], device_replay=c["replay"], generic_kernels=c["generic"])
    except sx.SxError as e:
        got = ("error: %s" % e).encode()
    if got != want:
        print(f"MISMATCH fuzz seed {seed} case {n} case_seed {case_seed}: {fuzz_case.describe(c)}")
        gl, wl = got.split(b"\n"), want.split(b"\n")
        for i, (a, b) in enumerate(zip(gl, wl)):
            if a != b:
                print("  first differing line", i, a[:120], b[:120]); break
        print("  lines", len(gl), len(wl))
        sys.exit(1)
print(f"fuzz seed {seed}: {n} cases in {time.time() - t0:.0f} s, all equal to the oracle")
