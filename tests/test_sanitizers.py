"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code: the product's host logic and both oracles
(the reference offers the same switch for its own targets, cmake/asan.cmake).  GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_logic_and_oracles_are_clean_under_asan_ubsan():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "host_asan")
        flags = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-mfma", "-mavx2",
                 "-ffp-contract=off"]
        srcs_c = [os.path.join(ROOT, "oracle", "phmm_oracle.c"), os.path.join(ROOT, "oracle", "sw_oracle.c")]
        objs = []
        for s in srcs_c:
            o = os.path.join(d, os.path.basename(s) + ".o")
            subprocess.run(["gcc", "-std=gnu11"] + flags + ["-c", s, "-o", o], check=True)
            objs.append(o)
        subprocess.run(["g++", "-std=c++17"] + flags + [os.path.join(ROOT, "tests", "sanitize", "host_asan.cpp"),
                                                         os.path.join(ROOT, "rovaca_amd", "host", "pairhmm_host_logic.cpp")] + objs +
                       ["-o", exe, "-lm", "-lpthread"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "sanitizer run clean" in r.stdout
