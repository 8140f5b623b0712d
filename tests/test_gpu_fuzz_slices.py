"""Fixed-seed slices of the fuzzers (tools/fuzz_gpu.py, tools/fuzz_exact_math.py, tools/fuzz_gpu_mutated.py, tools/fuzz_gpu_mutated_db.py) under `pytest -m gpu`, so that what the driver
runs at the end of a round includes them: random clip shapes x random settings x consumers through the C ABI against the oracle, and the
registration-time analysis behind the short exact square root / reciprocal under adversarial rotation ranges. A few seconds each;
the tools run the same loops for as long as one likes. Needs a GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tool, seconds, seed):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    completed = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(seconds), str(seed)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert completed.returncode == 0, completed.stdout[-2000:] + completed.stderr[-2000:]
    return completed.stdout


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_and_settings_against_the_oracle(seed):
    out = _run("fuzz_gpu.py", 8, seed)
    assert "fuzz ok" in out and "rejected 0" in out, out[-500:]


@pytest.mark.parametrize("seed", [21])
def test_short_exact_math_analysis_under_adversarial_ranges(seed):
    out = _run("fuzz_exact_math.py", 8, seed)
    assert "exact math fuzz ok" in out, out[-500:]


@pytest.mark.parametrize("seed", [31])
def test_mutated_clips_that_are_accepted_decode_to_the_oracle(seed):
    out = _run("fuzz_gpu_mutated.py", seed, 8)          # (this tool takes seed, seconds)
    assert "gpu mutated fuzz ok" in out and " 0 clips differ" in out, out[-500:]


@pytest.mark.parametrize("seed", [41])
def test_mutated_databases_that_are_accepted_stream_and_decode_to_the_oracle(seed):
    """mutated compressed_database headers / bulk data: refused, or bound, streamed in and out in random chunk counts and decoded bit
    identically to the restated database_context (validate_database.cpp's walk on hostile input)"""
    out = _run("fuzz_gpu_mutated_db.py", seed, 10)
    assert "gpu mutated database fuzz ok" in out and " 0 differences" in out, out[-500:]
