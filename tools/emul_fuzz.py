"""CPU-only randomized check of the device stage-B pipeline (no GPU): the replay core compiled for the host
(tests/native), driven the way the device drives it — head runs, regions from derived state, sequential
stitch (tests/test_replay_core.py emulate_device_stage_b) — against the oracle, on the cases of gpu_fuzz.py.
usage: tools/emul_fuzz.py SECONDS [SEED]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_case
import sxo_binding as sxo
import test_replay_core as trc
from test_sharded_gloo import oracle_findings

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = random.Random(seed)
core = trc.load_core()
t0 = time.time(); n = checked = 0
while time.time() - t0 < budget:
    n += 1
    case_seed = rng.randrange(1 << 31)
    c = fuzz_case.make(case_seed)
    data = c["files"][0][:200_000]
    if not data:
        continue
    for m in c["missions"]:
        if m["output_line_char_nb_max"] > 255 or m["chars_min_nb"] == 0 or m["encoding"] == 71:
            continue   # never replayed on the device (sx_stage_b.cpp device_replay_wanted)
        long_run = max(1, min(m["chars_min_nb"], m["output_line_char_nb_max"]))
        runs = sxo.runs(m, data, stream_parity=0, min_chars=long_run)
        for skip in (1, 0):
            slabs = rng.choice([1, 1, 2, 3, 7, 20])   # the list replayed in slabs (sx_stage_b.cpp device_replay_mission)
            got = trc.emulate_device_stage_b(core, m, data, runs, skip=skip, slabs=slabs)
            if got is None:
                break
            want = [(p, pr, s, c_, si) for p, pr, s, c_, _, si in oracle_findings([dict(m, mission_id=0)], data)]
            if got != want:
                print(f"MISMATCH emul seed {seed} case_seed {case_seed} mission {m} skip={skip} slabs={slabs}: {fuzz_case.describe(c)}")
                print("  first diff", next(((a, b) for a, b in zip(got, want) if a != b), (len(got), len(want))))
                sys.exit(1)
            checked += 1
print(f"emul fuzz seed {seed}: {n} cases, {checked} mission replays, all equal to the oracle")
