"""A slice of every sweep script in each GPU run.  The sweeps (tests/*_sweep.py) compare the library with the oracle over
seeded random chunks - valid streams of every encoding, edited ones, the kernel variants, several handles at once, the
entry point that allocates nothing; run by hand they go through thousands of chunks.  Here each runs for a few seconds in
a process of its own (they are scripts), with seeds that move from run to run only when this file changes."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))

SWEEPS = [
    ("fused_sweep.py", ["52000", "60"]),
    ("fused_sweep.py", ["53000", "40", "0"]),
    ("delta_sweep.py", ["54000", "40", "1"]),
    ("delta_sweep.py", ["55000", "30", "2"]),
    ("bytes_sweep.py", ["56000", "50"]),
    ("levels_sweep.py", ["57000", "60"]),
    ("valid_sweep.py", ["58000", "300"]),
    ("batch_sweep.py", ["59000", "40"]),
    ("fuzz_sweep.py", ["60000", "2", "60"]),
    ("noalloc_sweep.py", ["61000", "200"]),
    ("concurrent_sweep.py", ["62000", "100"]),
]


@pytest.mark.parametrize("script,args", SWEEPS, ids=["%s-%s" % (s[:-3], "-".join(a)) for s, a in SWEEPS])
def test_sweep_slice(script, args):
    r = subprocess.run([sys.executable, os.path.join(HERE, script)] + args, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
