"""Builds the test-only harness libraries (the device replay core, the wave core and the range classifiers compiled as host code) — no pytest, no torch, no
package import: __graft_entry__.build() and the test modules both call this.  A library is rebuilt when its source or a header is newer."""
import os
import subprocess

NATIVE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(NATIVE))


def _build(so_name, src_name, headers):
    so, src = os.path.join(NATIVE, so_name), os.path.join(NATIVE, src_name)
    hdrs = [os.path.join(ROOT, "stringsext_amd", "csrc", h) for h in headers]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        tmp = f"{so}.{os.getpid()}.tmp"   # (several pytest-xdist workers may get here at once)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                               "-Wno-unknown-pragmas", "-o", tmp, src])
        os.replace(tmp, so)
    return so


def build_replay_core():
    return _build("libreplay_core_host.so", "replay_core_host.cpp", ("sx_replay_core.hpp", "sx_codec_core.hpp", "sx_device.hpp"))


def build_wave_core():
    return _build("libwave_core_host.so", "wave_core_host.cpp", ("sx_wave_core.hpp", "sx_codec_core.hpp", "sx_device.hpp"))


def build_classify():
    return _build("libclassify_host.so", "classify_host.cpp", ("sx_classify_ranges.hpp", "sx_device.hpp"))


if __name__ == "__main__":
    print(build_replay_core())
    print(build_wave_core())
    print(build_classify())
