"""The in-LDS FFT core of the frequency-domain convolutions (applecider_amd/csrc/ac_fft_core.h) on the CPU: the
harness runs the very pass / untangle functions the HIP kernels call, work item by work item, for N = 8 ... 2048
against a direct fp64 DFT (transform in bit-reversed order, conjugate-partner positions, the two-real-sequences-as-one
split, and the inverse round trip).  No GPU involved."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compiler():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def test_fft_core_against_direct_dft(tmp_path):
    cxx = _compiler()
    if cxx is None:
        pytest.skip("no clang++ (the header uses ext_vector_type)")
    exe = str(tmp_path / "fft_core_harness")
    subprocess.run([cxx, "-O2", "-std=c++17", "-x", "c++", os.path.join(ROOT, "tests", "fft_core_harness.cpp"), "-o", exe],
                   check=True, capture_output=True)
    out = subprocess.run([exe], check=False, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("logn")]
    assert len(lines) == 9 and all("partner ok" in l for l in lines), out.stdout
    sizes = [l for l in out.stdout.splitlines() if l.startswith("N ")]
    assert len(sizes) == 21 and all(l.endswith("ok") for l in sizes), out.stdout        # 5 sizes 9 * 2^m, 7 of 3 * 2^m, 9 of 2^m
    assert float(out.stdout.split("worst")[1].split()[0]) < 1e-6 and out.stdout.strip().endswith("fails 0")


def test_fft_plans_of_the_default_stages():
    """Host logic: transform plans and the cost rule for SpectraNet's default stages (default_config.toml:104-114)."""
    from applecider_amd import hipops as H
    H.set_math("bf16x3")
    try:
        assert H.fft_plan(1024, 251) == (7, 2, 1, 1024)        # one 1152-point sequence per sample (needs 1149)
        assert H.fft_plan(256, 61) == (5, 2, 1, 256) and H.fft_plan(16, 13) == (3, 1, 1, 16)       # 288, 24 points
        H._FFT_RADIX9 = False                                  # lengths 2^m and 3 * 2^m only
        try:
            assert H.fft_plan(1024, 251) == (9, 1, 1, 1024) and H.fft_plan(1024, 31) == (7, 1, 3, 354)
        finally:
            H._FFT_RADIX9 = True
        assert H.fft_plan(4096, 1021) is None                  # stage 1 stays on the Toeplitz window kernels
        logm, r3, blocks, step = H.fft_plan(2048, 31)          # overlap-save where one sequence would not fit
        assert blocks > 1 and step == H._fft_size((logm, r3))[2] - 31 + 1
        H._FFT_RADIX3 = False                                  # power-of-two lengths only
        try:
            assert H.fft_plan(1024, 251) == (11, 0, 1, 1024) and H.fft_plan(1024, 31) == (8, 0, 5, 226)
        finally:
            H._FFT_RADIX3 = True
        chosen = {(L, k): H.fftconv_covered(512, L, ci, co, k) for L, ci, co, ks in
                  ((1024, 64, 128, (3, 31, 251)), (256, 128, 256, (3, 15, 61)), (64, 256, 512, (3, 11, 31)),
                   (16, 512, 1024, (3, 7, 13))) for k in ks}
        assert [kk for kk, v in chosen.items() if v] == [(1024, 31), (1024, 251), (256, 15), (256, 61), (64, 11), (64, 31),
                                                          (16, 7), (16, 13)]
        assert not H.fftconv_covered(2, 1024, 64, 128, 251)        # two samples: the taps' spectrum dominates
        assert not H.fftconv_covered(512, 1024, 1, 64, 1021)       # Cin = 1
        H.set_math("bf16")
        assert not H.fftconv_covered(512, 1024, 64, 128, 251)      # the unqualified fast mode keeps its bf16 kernels
    finally:
        H.set_math("f32")
