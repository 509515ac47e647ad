"""The adapter's N-device path (SURVEY.md 8e through the seam: init_pairhmm_ptr_gpu(-1) makes one cross-region batcher per device and
deals the regions out round-robin, pairhmm_gpu_engine.cpp) executed with N > 1 on a one-GPU box: the device-list hook
ROVACA_AMD_PAIRHMM_DEVICES=0,0,0 -- dead without PHMM_ENABLE_TEST_HOOKS and PHMM_MULTI_ALLOW_SHARED_DEVICE -- puts every batcher on
device 0.  The batchers are process-wide and never destroyed, so each configuration runs in a child interpreter (this module sorts
first among the GPU tests: the pytest process has not touched the GPU when the children start)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _child(devices, n_workers=30, per_worker=12, hooks=True):
    env = dict(os.environ)
    for k in ("ROVACA_AMD_PAIRHMM_DEVICES", "PHMM_MULTI_ALLOW_SHARED_DEVICE", "PHMM_ENABLE_TEST_HOOKS"):
        env.pop(k, None)
    if devices:
        env["ROVACA_AMD_PAIRHMM_DEVICES"] = devices
    if hooks:
        env["PHMM_ENABLE_TEST_HOOKS"] = "1"
        env["PHMM_MULTI_ALLOW_SHARED_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "adapter_devices_child.py"), str(n_workers), str(per_worker)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_n_batchers_on_one_card_give_the_single_batcher_matrices(devices):
    n = len(devices.split(","))
    d = _child(devices)
    assert d["n_mismatches"] == 0, d["mismatches"]  # every region bit-equal to the oracle's call, whichever batcher took it
    st = d["batchers"]
    assert len(st) == n and all(dev == 0 for dev, _, _ in st)
    per = [r for _, r, _ in st]
    assert sum(per) == d["regions"] and max(per) - min(per) <= 2  # dealt out round-robin
    assert all(s >= 1 for _, _, s in st)


def test_device_list_is_dead_without_the_hooks():
    d = _child("0,0,0", n_workers=4, per_worker=3, hooks=False)
    assert d["n_mismatches"] == 0 and len(d["batchers"]) >= 1
    # one batcher per VISIBLE device, not three on device 0
    assert sorted(dev for dev, _, _ in d["batchers"]) == list(range(len(d["batchers"])))
