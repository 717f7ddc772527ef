"""A bounded slice of the randomized tools (tools/emul_fuzz.py, tools/shard_fuzz.py, tools/gpu_fuzz.py in host mode)
with fixed seeds, so that the CPU suite keeps exercising them.  No GPU needed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,env", [("emul_fuzz.py", {}), ("shard_fuzz.py", {}), ("gpu_fuzz.py", {"SX_FUZZ_HOST": "1"})])
def test_randomized_tools_run_clean_for_a_few_seconds(tool, env):
    e = dict(os.environ, **env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "8", "20260928"], env=e, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "all equal to the oracle" in out.stdout
