"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/pairhmm_hip.h declares, builds the
reference's lookup tables exactly, and refuses to run without a device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

from rovaca_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "pairhmm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phmm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(capi.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/pairhmm_hip.h but not exported"
    assert set(names) == set(capi.EXPORTS)
    assert capi.lib().phmm_abi_version() == 5


def test_product_tables_equal_reference_tables(golden):
    t = capi.tables()
    for k in ("ph2pr_f32", "mm_f32", "ph2pr_f64", "mm_f64"):
        assert np.array_equal(t[k], golden[k]), k


def test_struct_layout_matches_header():
    # phmm_batch: 2 x u32 + u64 + 12 pointers; phmm_stats: 2 x u64 + 2 x f64 + 2 x u32
    assert ctypes.sizeof(capi.PhmmBatch) == 16 + 13 * 8
    assert ctypes.sizeof(capi.PhmmStats) == 40


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.PhmmError, match="no HIP device|no CPU fallback|does not exist"):
        capi.Engine(0)


# ---- Smith-Waterman library (include/smithwaterman_hip.h) ----
def test_sw_library_exports_every_declared_symbol_and_refuses_without_device():
    from rovaca_amd import swapi

    text = open(os.path.join(ROOT, "include", "smithwaterman_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(swhip_[a-z0-9_]+)\s*\(", text)))
    assert names == ["swhip_abi_version", "swhip_align", "swhip_create", "swhip_destroy", "swhip_device_count", "swhip_kernel_build_id", "swhip_last_error"]
    L = swapi.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/smithwaterman_hip.h but not exported"
    assert L.swhip_abi_version() == 1
    # swhip_batch: 2 x u32 + u64 + 8 pointers; swhip_alignment: 20 bytes; swhip_stats: u64 + 2 x f64 + u32 (+ pad)
    assert ctypes.sizeof(swapi.SwBatch) == 16 + 8 * 8
    assert swapi.ALIGNMENT_DTYPE.itemsize == 20
    assert ctypes.sizeof(swapi.SwStats) == 32
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback|swhip_create failed"):
            swapi.Aligner(0)


def test_sw_host_adapter_exports_the_seam_candidates():
    L = ctypes.CDLL(os.path.join(ROOT, "rovaca_amd", "lib", "librovaca_sw_host.so"))
    for n in ("sw_hip_run", "sw_hip_install", "rovaca_amd_sw_run_through_handle"):
        assert hasattr(L, n)


def test_sw_handle_mirror_has_the_reference_layout():
    # rovaca_amd/host/sw_seam.h restates lib_sw_avx; where the reference's header is available (oracle/_ref) the offsets of
    # every member the adapter touches must agree with it
    import oracle
    ref = oracle.sw_ref()
    if ref is None:
        pytest.skip("oracle/_ref/libsw_ref.so not available")
    L = ctypes.CDLL(os.path.join(ROOT, "rovaca_amd", "lib", "librovaca_sw_host.so"))
    a, b = (ctypes.c_int32 * 32)(), (ctypes.c_int32 * 32)()
    na, nb = L.rovaca_amd_sw_layout(a, 32), ref.ref_sw_layout(b, 32)
    assert na == nb == 16 and list(a[:na]) == list(b[:nb])


def test_sw_adapter_compiles_against_the_reference_header():
    # the in-tree build (-DROVACA_AMD_IN_ROVACA_TREE) uses the reference's own smithwaterman_common.h instead of the mirror
    import shutil
    import subprocess
    import tempfile
    inc = "/root/reference/src/haplotypecaller/common/base/include"
    if not os.path.exists(os.path.join(inc, "smithwaterman_common.h")) or shutil.which("g++") is None:
        pytest.skip("reference tree not available")
    with tempfile.TemporaryDirectory() as d:
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-Wall", "-Wextra", "-Werror", "-DROVACA_AMD_IN_ROVACA_TREE", "-I" + inc,
               "-I/root/reference/third_lib/htslib/include", "-c", os.path.join(ROOT, "rovaca_amd", "host", "sw_gpu_engine.cpp"),
               "-o", os.path.join(d, "sw_intree.o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_pairhmm_seam_mirror_against_the_reference_headers():
    # rovaca_amd/host/rovaca_seam.h restates the few Rovaca types the seam touches.  genotype/forward.h needs boost (absent
    # here), but common/enum.h and genotype/genotype_struct.h do not: build the check TU and the adapter itself with the
    # reference's own two headers swapped in, and compare the mangled seam signature of both builds.
    import shutil
    import subprocess
    import tempfile
    ref = "/root/reference/src/haplotypecaller"
    host = os.path.join(ROOT, "rovaca_amd", "host")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    have_ref = os.path.exists(os.path.join(ref, "common", "enum.h"))
    with tempfile.TemporaryDirectory() as d:
        names, symbols = [], []
        for tag, extra in (("mirror", []), ("reference", ["-DROVACA_AMD_WITH_REFERENCE_ENUM_AND_STRUCT", "-I" + ref])):
            if tag == "reference" and not have_ref:
                continue
            exe = os.path.join(d, "seam_" + tag)
            r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + host, *extra,
                                os.path.join(ROOT, "tests", "seam_check.cpp"), "-o", exe], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            names.append(subprocess.run([exe], capture_output=True, text=True, check=True).stdout.strip())
            obj = os.path.join(d, "engine_" + tag + ".o")
            r = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-Wall", "-Wextra", "-Werror", "-I" + host,
                                "-I" + os.path.join(ROOT, "include"), *extra, "-c", os.path.join(host, "pairhmm_gpu_engine.cpp"),
                                "-o", obj], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            nm = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True, check=True).stdout
            defined = [l.split()[-1] for l in nm.splitlines() if " T " in l]
            symbols.append(sorted(x for x in defined if "16call_gpu_pairhmmE" in x))  # the synchronous seam itself
            # its two halves ("return futures") come out of both builds too
            assert any("22call_gpu_pairhmm_beginE" in x for x in defined) and any("23call_gpu_pairhmm_finishE" in x for x in defined)
        # PcrIndelModel is mangled as a GLOBAL enum (13PcrIndelModel, not N6rovaca13PcrIndelModelE)
        assert all("13PcrIndelModel" in n and "6rovaca13PcrIndelModel" not in n for n in names), names
        assert all(len(s) == 1 for s in symbols), symbols
        assert symbols[0][0] == ("_ZN6rovaca16call_gpu_pairhmmERKSt6vectorIPNS_9HaplotypeENSt3pmr21polymorphic_allocatorIS2_EEERS0_IPNS_10ReadRecordE"
                                 "NS4_ISA_EEEi13PcrIndelModelPNS3_15memory_resourceE")
        if have_ref:
            assert names[0] == names[1] and symbols[0] == symbols[1]
        # the shipped stand-alone library exports exactly that symbol
        nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "rovaca_amd", "lib", "librovaca_pairhmm_host.so")],
                            capture_output=True, text=True, check=True).stdout
        assert symbols[0][0] in nm


def test_public_headers_are_plain_c_and_link_from_a_c_program():
    # the drop-in boundary is a C ABI (extern "C", plain pointers and sizes): both headers must compile as C99 with -pedantic,
    # and a C program must link against the libraries and reach the version / error entry points (no compute: no device here)
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = os.path.join(ROOT, "rovaca_amd", "lib")
    src = r'''
#include <stdio.h>
#include "pairhmm_hip.h"
#include "smithwaterman_hip.h"
int main(void)
{
    phmm_batch b;
    phmm_stats st;
    swhip_batch sb;
    (void)b; (void)st; (void)sb;
    printf("%d %d %d\n", phmm_abi_version(), swhip_abi_version(), (int)sizeof(phmm_batch));
    return phmm_last_error() == 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "use.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "use")
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), c, "-o", exe,
                            "-L" + lib, "-lrovaca_pairhmm_hip", "-lrovaca_sw_hip", "-Wl,-rpath," + lib], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout.split() == ["5", "1", str(ctypes.sizeof(capi.PhmmBatch))]


def test_sw_kernels_keep_their_column_state_out_of_scratch(tmp_path):
    """A Smith-Waterman kernel whose per-column arrays (H and F of the row above, the bases) end up in scratch still computes the
    right results, eight times more slowly: round 5 saw it when a wrapper lambda kept the step from being inlined (the 16-column
    kernels went from 3.6 to 30 ms on 250-base reads).  Device assembly of every instantiation: at most the few dwords of SGPR
    spills the widest kernels have, and no spilled VGPRs."""
    import re
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rovaca_amd", "csrc")
    flags = subprocess.run(["make", "-s", "-C", csrc, "print-hipflags"], check=True, capture_output=True, text=True).stdout.split()
    asm = tmp_path / "sw.s"
    subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", "-o", str(asm), os.path.join(csrc, "sw_kernels.hip")], check=True,
                   capture_output=True)
    text = asm.read_text()
    seen = 0
    for m in re.finditer(r"\.name:\s+(\S*sw_kernel\S*)\n(.*?)\.vgpr_spill_count:\s+(\d+)", text, re.S):
        seen += 1
        private = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", m.group(2)).group(1))
        assert private <= 64 and int(m.group(3)) == 0, (m.group(1), private, m.group(3))
    assert seen >= 19, seen
