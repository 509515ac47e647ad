"""MI355X-native PairHMM likelihood engine behind Rovaca's seam: C ABI bindings (capi, swapi, hostapi) and the synthetic generator."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _hipflags():
    """The HIPFLAGS of csrc/Makefile as make sees them (`:=`, so $(ARCH) is expanded where the line stands)."""
    arch, flags = None, None
    for line in open(os.path.join(_HERE, "csrc", "Makefile")):
        if line.startswith("ARCH") and "?=" in line:
            arch = line.split("?=", 1)[1].strip()
        if line.startswith("HIPFLAGS :="):
            flags = line.split(":=", 1)[1].strip()
    if arch is None or flags is None:
        raise RuntimeError("csrc/Makefile has no ARCH / HIPFLAGS line")
    return flags.replace("$(ARCH)", os.environ.get("ARCH", arch))


def _source_id(names):
    h = hashlib.sha256()
    for name in names:
        h.update(open(os.path.join(_HERE, "csrc", name), "rb").read())
    h.update((_hipflags() + "\n").encode())
    return h.hexdigest()[:16]


def kernel_source_id():
    """Content hash of the device code of the sweep kernels AND of the flags they are compiled with -- the same hash csrc/Makefile links
    into the library as phmm_kernel_build_id().  profiles/hbm_traffic.json records it when the PMC counters are collected
    (tools/summarize_profiles.py); bench.py and the tests refuse counters taken from other sources, other flags or another build."""
    return _source_id(("pairhmm_kernels.hip", "pairhmm_systolic.h", "pairhmm_systolic_sweep.inc", "pairhmm_systolic_ugo_sweep.inc",
                       "pairhmm_kernels_ragged.hip", "pairhmm_kernels_tab.hip", "pairhmm_kernels_ragged_tab.hip", "pairhmm_systolic_tab.h",
                       "pairhmm_systolic_tab_round.inc",
                       "pairhmm_device.h", "pairhmm_tab_sched.mk"))


def sw_kernel_source_id():
    """The same for the Smith-Waterman kernel (profiles/sw_counters.json, swhip_kernel_build_id())."""
    return _source_id(("sw_kernels.hip", "sw_device.h"))


def loaded_kernel_build_id():
    """The id linked into the PairHMM library this process uses (a stale or variant library differs from kernel_source_id())."""
    from . import capi
    return capi.lib().phmm_kernel_build_id().decode()


def loaded_sw_kernel_build_id():
    from . import swapi
    return swapi.lib().swhip_kernel_build_id().decode()
