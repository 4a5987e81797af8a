"""The K loop of gemm_x3_kernel keeps a whole tile's loads in flight (round 4: for two rounds hipcc had drained every load
at the top of each trip - a `break` between the two halves of the trip and a conditional scalar load between the loads;
DESIGN section 4).  Nothing in the source shows that, so the built library is disassembled.  CPU only."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "applecider_amd", "csrc", "libapplecider_hip.so")
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_x3_loops as chk  # noqa: E402


@pytest.fixture(scope="module")
def report():
    if not os.path.exists(LIB):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    return chk.scan(LIB)


def test_every_instantiation_was_found(report):
    # <A_KC, B_KC, BATCH, B_PL>: NT / NN / TN, their batched forms, NT / NN with plane-fed weights
    assert len(report) == 8, sorted(report)
    for name, trips in report.items():
        assert len(trips) >= 4, (name, len(trips))          # four loop copies (fast / general per operand)


def test_interior_tile_trip_keeps_a_tile_of_loads_in_flight(report):
    for name, trips in report.items():
        lo, hi, waits = min(trips, key=lambda t: t[1] - t[0])
        counts = chk.vm(waits)
        assert counts and min(counts) >= 8, (name, waits)   # 8 loads = one K tile of A and B
        assert max(counts) >= 12, (name, waits)             # and the oldest tile is consumed load by load, not all at once


def test_no_trip_of_the_unbatched_kernels_drains_the_loads(report):
    for name, trips in report.items():
        if "ELb1ELb0EEE" in name or "ELb1ELb1EEE" in name:
            continue    # batched form: its edge-tile copies still drain (tools/check_x3_loops.py reports them)
        for lo, hi, waits in trips:
            assert 0 not in chk.vm(waits), (name, waits)


def test_the_checker_recognises_a_draining_loop():
    """Known answer for the parser: a synthetic disassembly with one 48-MFMA loop that waits for vmcnt(0)."""
    head = "0000000000001000 <_ZN12_GLOBAL__N_114gemm_x3_kernelILb1ELb1ELb0ELb0EEEvNS_10GemmParamsE>:\n"
    body, addr = [], 0x1000
    def ins(text, extra=""):
        nonlocal addr
        body.append(f"\t{text:<58} // {addr:012X}: 00000000{extra}")
        addr += 4
    ins("s_nop 0")
    ins("s_waitcnt vmcnt(0)")
    for _ in range(48):
        ins("v_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]")
    ins("s_cbranch_scc1 65000", " <_ZN12_GLOBAL__N_114gemm_x3_kernelILb1ELb1ELb0ELb0EEEvNS_10GemmParamsE+0x4>")
    fns = chk.functions(head + "\n".join(body) + "\n")
    (name, (start, insts)), = fns.items()
    (lo, hi, waits), = chk.trips(start, insts)
    assert chk.vm(waits) == [0] and hi - lo == 49
