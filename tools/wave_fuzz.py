"""CPU-only randomized check of the wave-cooperative stage B (no GPU): sx_wave_core.hpp compiled for the host and driven as
sx_wave_dev.hip drives it (tests/native/wave_core_host.cpp) against the oracle, on the cases of gpu_fuzz.py whose Missions
the wave path covers, with random wavefront sizes.  usage: tools/wave_fuzz.py SECONDS [SEED]
SAME=1: every Mission with -r and without -g (round 5: -r in the wave kernels), half the inputs text that changes script every few characters;
SAME=2: with -r and, most of them, with -g too"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_case
import test_wave_core as twc
from test_sharded_gloo import oracle_findings

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = random.Random(seed)
L = twc.load_wave()
t0 = time.time(); n = checked = gave_up = 0; max_rounds = 0
while time.time() - t0 < budget:
    n += 1
    case_seed = rng.randrange(1 << 31)
    c = fuzz_case.make(case_seed)
    data = c["files"][0][:300_000]
    if not data:
        continue
    for m in c["missions"]:
        if os.environ.get("SAME"):
            m = dict(m, require_same_unicode_block=True, grep_char=None if os.environ["SAME"] == "1" else rng.choice([None, 32, 101, 58, m.get("grep_char")]))
            if rng.random() < 0.5:
                txt = twc.same_text(rng, rng.choice([300, 3000, 30_000]), runs=rng.choice([(1, 1, 2, 3, 4, 5, 8, 13, 40), (1, 2, 3), (1, 7, 30, 64, 65, 130)]))
                codec = {1: "utf-8", 2: "utf-16-le", 3: "utf-16-be"}.get(m["encoding"])
                data = txt.encode(codec) if codec else txt.encode("utf-8") if rng.random() < 0.3 else bytes(rng.choice(b"abc \xc1\xd2\xe3\xf4\xa5\xb6\n\x00") for _ in range(len(txt)))
        if twc.wave_classes(m) is None:
            continue
        want = oracle_findings([dict(m, mission_id=0)], data)
        nwin = rng.choice([508, 508, 60, 5, 17, 64, 1000])
        if m["encoding"] in (2, 3) and len(data) % 2:   # (UTF-16: the wave path takes buffers of whole units)
            data = data[:-1]
            want = oracle_findings([dict(m, mission_id=0)], data)
        skip = rng.choice([0, 1, 1])
        if os.environ.get("SX_FUZZ_TRACE"):   # (the case in hand, before it runs: a crash of the harness names it)
            with open(os.environ["SX_FUZZ_TRACE"], "w") as tf:
                tf.write(f"wave seed {seed} case {n} case_seed {case_seed} mission {m} nwin={nwin} skip={skip} len={len(data)}\n")
        try:
            got, info = twc.emulate(L, m, data, nwin=nwin, skip_idle=skip, may_give_up=True)
        except AssertionError as e:
            print(f"HARNESS ERROR {e} wave seed {seed} case_seed {case_seed} mission {m} nwin={nwin} skip={skip} len={len(data)}: {fuzz_case.describe(c)}")
            sys.exit(1)
        if got is None:   # UTF-16: the wavefronts gave the buffer back (the product then takes the lane-per-region path)
            gave_up += 1
            continue
        max_rounds = max(max_rounds, info["rounds"])
        if info["bad"] and m.get("grep_char") is not None:   # -g: a chain of wrong wavefronts longer than the repair rounds (the product: the other path)
            gave_up += 1
            continue
        if got != want or info["bad"]:
            print(f"MISMATCH wave seed {seed} case_seed {case_seed} mission {m} nwin={nwin} info={info}: {fuzz_case.describe(c)}")
            print("  first diff", next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
            sys.exit(1)
        checked += 1
print(f"wave fuzz seed {seed}: {n} cases, {checked} mission replays (most rounds to settle a batch: {max_rounds}; UTF-16 buffers given back: {gave_up}), all equal to the oracle")
