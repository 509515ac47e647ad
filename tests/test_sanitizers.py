"""AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code: the product's host logic, the CPU backend of the C ABI and both oracles
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
                                                         os.path.join(ROOT, "rovaca_amd", "host", "pairhmm_host_logic.cpp"),
                                                         os.path.join(ROOT, "rovaca_amd", "csrc", "pairhmm_cpu.cpp"),
                                                         os.path.join(ROOT, "rovaca_amd", "csrc", "pairhmm_tables.cpp")] + objs +
                       ["-o", exe, "-lm", "-lpthread"], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "sanitizer run clean" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_batchers_are_race_free_under_tsan_and_clean_under_asan():
    # the two cross-region batchers (adapter code as shipped) over stand-in device libraries: ThreadSanitizer watches the
    # hand-offs between 12 workers and the dispatcher threads; every worker checks that it got its own region's numbers
    host = os.path.join(ROOT, "rovaca_amd", "host")
    srcs = [os.path.join(ROOT, "tests", "sanitize", "batcher_tsan.cpp"), os.path.join(host, "pairhmm_gpu_engine.cpp"),
            os.path.join(host, "pairhmm_host_logic.cpp"), os.path.join(host, "sw_gpu_engine.cpp")]
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "batcher_tsan")
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-fno-omit-frame-pointer", *srcs, "-o", exe, "-lpthread"]
        subprocess.run(cmd, check=True)
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 mismatches" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr
        # ... and heavily oversubscribed: 48 workers on whatever cores there are -- the stage reservation is a lock-free compare-and-swap
        # on a 128-bit cursor word (pairhmm_gpu_engine.cpp), and a worker preempted half way through one must hold nobody up
        r = subprocess.run([exe], capture_output=True, text=True,
                           env=dict(os.environ, BATCHER_THREADS="48", TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 mismatches" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr
        # the same harness (both seams, synchronous and in two halves, tickets dropped unfinished) under AddressSanitizer + UBSan:
        # lifetimes of the tickets' buffers against the dispatchers that write into them (leak check off: the dispatcher threads
        # and their queues are never torn down, by design)
        exe = os.path.join(d, "batcher_asan")
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", *srcs,
               "-o", exe, "-lpthread"]
        subprocess.run(cmd, check=True)
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "0 mismatches" in r.stdout
