"""The host-side thrift walks (page headers, footer) under AddressSanitizer and UBSan on mutated pyarrow files.
The two sources are plain C++ (no HIP), so the sanitizers run here on the CPU; tests/host_fuzz.py is the driver."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_page_header_and_footer_walk_survive_mutated_files(tmp_path):
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    lib = str(tmp_path / "libpqhost_asan.so")
    src = os.path.join(ROOT, "parquet-go_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-fno-sanitize-recover=undefined", "-shared", "-o", lib,
                           os.path.join(src, "page_scan.cpp"), os.path.join(src, "footer.cpp")])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", PQD_HOST_ASAN_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(HERE, "host_fuzz.py"), "25", "77"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "host fuzz ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
