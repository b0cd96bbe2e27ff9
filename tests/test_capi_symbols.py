"""The C-ABI shared library loads on a machine without a GPU and exports every symbol
include/nanorepeat_amd.h declares.  No compute calls."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "nanorepeat_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nra_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(capi):
    names = _declared()
    assert len(names) >= 15
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nanorepeat_amd.h but not exported"
    assert sorted(capi.EXPORTS) == names


def test_abi_version_and_defaults(capi):
    lib = capi.load()
    assert lib.nra_abi_version() == 4
    assert b"gfx950" in lib.nra_version()
    sc = capi.default_scoring()
    assert (sc.match, sc.mismatch, sc.gap_open1, sc.gap_ext1, sc.gap_open2, sc.gap_ext2,
            sc.sc_ambi, sc.min_dp_score) == (2, 4, 4, 2, 24, 1, 1, 80)


def test_structs_match_header_layout(capi):
    assert ctypes.sizeof(capi.Scoring) == 32
    assert ctypes.sizeof(capi.Region) == 3 * ctypes.sizeof(ctypes.c_void_p) + 12 + 4   # padded to 8
    assert ctypes.sizeof(capi.JointRegion) == 5 * ctypes.sizeof(ctypes.c_void_p) + 20 + 4
    assert ctypes.sizeof(capi.Stats) == 5 * 8 + 3 * 8 + 8 + 8 + 4 * 8 + 8


def test_no_cpu_fallback_in_product(capi):
    """The product package never imports, loads or links the oracle."""
    import nanorepeat_amd
    pkg = os.path.dirname(nanorepeat_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                for bad in ("import oracle", "from oracle", "libnr_oracle", "nr_oracle.h", "nr_decomp", "nro_", "nrd_"):
                    assert bad not in src, (f, bad)


def test_fails_loudly_without_a_device_and_validates_arguments_first(capi):
    """No CPU path: on a box without a HIP device every compute entry point returns NRA_E_DEVICE
    (argument errors are reported before the device is touched).  Skipped where a GPU is present --
    the GPU suite covers the error paths there."""
    import numpy as np
    if capi.load().nra_device_count() > 0:
        import pytest
        pytest.skip("a HIP device is present")
    good = ([("ACGTACGTAC", "CAG", "TTGACCA")], ["ACGTACCAGCAGTTGA"], [0], [3])
    for call in (lambda: capi.round3_1d(*good),
                 lambda: capi.joint_2d(("ACGTAC", "CAG", "AT", "CCG", "GGTTAA"), ["ACGTACCAGCAGATCCGGGTT"], [0], [2], [1]),
                 lambda: capi.align_pairs(["ACGTACGT", "ACGTTCGT"], [0], [1]),
                 lambda: capi.align_pairs_cigar(["ACGTACGT", "ACGTTCGT"], [0], [1])):
        try:
            call()
        except capi.NraError as e:
            assert e.code == -2 and "no HIP device" in str(e)
        else:
            raise AssertionError("a compute call succeeded without a HIP device")
    for bad in (lambda: capi.round3_1d([("ACGT", "", "TT")], ["ACGT"], [0], [3]),                 # empty unit
                lambda: capi.round3_1d(good[0], good[1], [0], [3], sc=capi.default_scoring(match=0))):
        try:
            bad()
        except capi.NraError as e:
            assert e.code == -1, e
        else:
            raise AssertionError("an invalid argument was accepted")
