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
    assert float(out.stdout.split("worst")[1]) < 1e-6


def test_fft_plans_of_the_default_stages():
    """Host logic: transform plans and the cost rule for SpectraNet's default stages (default_config.toml:104-114)."""
    from applecider_amd import hipops as H
    H.set_math("bf16x3")
    try:
        assert H.fft_plan(1024, 251) == (11, 1, 1024)      # one 2048-point sequence per sample
        assert H.fft_plan(1024, 31) == (8, 5, 226)         # overlap-save: 5 windows of 256
        assert H.fft_plan(256, 61) == (9, 1, 256) and H.fft_plan(16, 13) == (5, 1, 16)
        assert H.fft_plan(4096, 1021) is None              # stage 1 stays on the Toeplitz window kernels
        for blocks_plan in (H.fft_plan(1024, 31), H.fft_plan(256, 15), H.fft_plan(64, 11)):
            logn, blocks, step = blocks_plan
            assert blocks > 1 and step == (1 << logn) - {5: 31, 6: 15, 3: 11}[blocks] + 1
        chosen = {(L, k): H.fftconv_covered(512, L, ci, co, k) for L, ci, co, ks in
                  ((1024, 64, 128, (3, 31, 251)), (256, 128, 256, (3, 15, 61)), (64, 256, 512, (3, 11, 31)),
                   (16, 512, 1024, (3, 7, 13))) for k in ks}
        assert [kk for kk, v in chosen.items() if v] == [(1024, 31), (1024, 251), (256, 15), (256, 61), (64, 11), (64, 31), (16, 13)]
        assert not H.fftconv_covered(2, 1024, 64, 128, 251)        # two samples: the taps' spectrum dominates
        assert not H.fftconv_covered(512, 1024, 1, 64, 1021)       # Cin = 1
        H.set_math("bf16")
        assert not H.fftconv_covered(512, 1024, 64, 128, 251)      # the unqualified fast mode keeps its bf16 kernels
    finally:
        H.set_math("f32")
