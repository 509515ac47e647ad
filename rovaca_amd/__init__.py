"""MI355X-native PairHMM likelihood engine behind Rovaca's seam: C ABI bindings (capi, swapi, hostapi) and the synthetic generator."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_id():
    """Content hash of the device code of the sweep kernels: profiles/hbm_traffic.json records it when the PMC counters are
    collected (tools/summarize_profiles.py), and bench.py / tests refuse counters taken from other kernel sources."""
    h = hashlib.sha256()
    for name in ("pairhmm_kernels.hip", "pairhmm_device.h"):
        h.update(open(os.path.join(_HERE, "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def sw_kernel_source_id():
    """The same for the Smith-Waterman kernel (profiles/sw_counters.json)."""
    h = hashlib.sha256()
    for name in ("sw_kernels.hip", "sw_device.h"):
        h.update(open(os.path.join(_HERE, "csrc", name), "rb").read())
    return h.hexdigest()[:16]
