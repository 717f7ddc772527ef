"""What the compiler made of the kernels (no GPU needed): the per-kernel resource remarks hipcc prints with
-Rpass-analysis=kernel-resource-usage, kept by the Makefile next to the objects (stringsext_amd/csrc/build/*.remarks).

VERDICT round 5, weak #3: a kernel with the wave path's exchange loop (sx_wave_dev.hip: DPP at loop level around a divergent body)
returned a wrong value from a spilled register.  The cause is unknown (profiles/r06_spill_note.md), so the rule is: no kernel
that holds that loop has vector-register spill code, and the headline kernels do not lose registers or occupancy unnoticed."""
import os
import re
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stringsext_amd", "csrc")


def remarks(name):
    path = os.path.join(CSRC, "build", name + ".remarks")
    src = os.path.join(CSRC, name + ".hip")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        obj = os.path.join(CSRC, "build", name + ".o")
        if os.path.exists(obj):
            os.remove(obj)   # (an object from before the Makefile kept the remarks: compile it again)
        subprocess.check_call(["make", "-C", CSRC, "build/%s.o" % name], env=dict(os.environ, HIPCC=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")))
    rows, cur = {}, None
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); rows[cur] = {}; continue
        m = re.search(r"remark: +([A-Za-z ]+?(?: \[[^\]]*\])?): (\S+) \[-Rpass", line)
        if cur and m:
            rows[cur][m.group(1).strip()] = m.group(2)
    names = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.split("\n")
    out = {}
    for mangled, name in zip(rows, names):
        out[re.sub(r"\(.*", "", name).replace("void ", "").replace("sx::", "")] = {k: int(v) if v.isdigit() else v for k, v in rows[mangled].items()}
    return out


# the two-byte families' count kernels with SWAR classes: 65 / 63 registers spilled at four wavefronts per SIMD, none at two — at two BASELINE
# config 5 runs 387 -> 465 ms per step (profiles/r06i_*).  They stay as they are until the classification is a kernel of its own
# ("Next"); round 5's and this round's GPU fuzz ran clean on them.  Nothing else may join this list.
SPILL_ALLOWED = {"wave_replay_kernel<0, 4, 4, 1, 0>", "wave_replay_kernel<0, 5, 4, 1, 0>"}


def test_no_kernel_with_the_exchange_loop_spills_vector_registers():
    rows = remarks("sx_wave_dev")
    wave = {k: v for k, v in rows.items() if k.startswith("wave_replay_kernel<")}
    assert len(wave) >= 40, sorted(wave)
    bad = {k: v["VGPRs Spill"] for k, v in wave.items() if v["VGPRs Spill"] > 0 and k not in SPILL_ALLOWED}
    assert not bad, bad
    for k in SPILL_ALLOWED:
        assert k in wave, k     # (the list must not outlive the kernels it names)


def test_headline_kernels_keep_their_registers():
    fused = remarks("sx_fused")
    head = [k for k in fused if k.startswith("scan_kernel_fused<Utf8Range2, Utf16RangeT<0, 0>, Utf16RangeT<1, 0>, 1>")]
    assert len(head) == 1, sorted(fused)
    for k, v in fused.items():
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0, (k, v)       # (the late parameters must not turn into a stack copy of the argument)
    h = fused[head[0]]
    assert h["Occupancy [waves/SIMD]"] >= 6 and h["VGPRs"] <= 80, h
    scan = remarks("sx_kernels")
    for k in ("scan_kernel<Utf8Range2, false>", "scan_kernel<Utf16RangeT<0, 0>, false>", "scan_kernel<Utf16RangeT<1, 0>, false>"):
        v = scan[k]
        assert v["VGPRs Spill"] == 0 and v["ScratchSize [bytes/lane]"] == 0 and v["Occupancy [waves/SIMD]"] >= 7, (k, v)
