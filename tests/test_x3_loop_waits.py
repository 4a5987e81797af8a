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


def test_depthwise_backward_with_inline_asm_loads_never_spills():
    """dwconv_pipe_bwd_kernel<.., RES = true> loads the shortcut's gradient from inline assembly and waits for it by
    hand: the compiler does not know those registers are in flight, so it must never move them -
    a spill to scratch between the load and the wait would save a stale value.  No scratch instruction may appear in
    these kernels, and the hand-written wait must sit in front of the stores."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    ks = {n: t for n, t in chk.kernels(LIB, "dwconv_pipe_bwd_kernel").items() if "ELb1EEE" in n}
    assert len(ks) == 2, sorted(ks)                      # 15 x 15 and 7 x 7
    for name, insts in ks.items():
        assert not [t for t in insts if t.startswith(("scratch_", "buffer_store", "buffer_load"))], name
        text = [t.split("//")[0].strip() for t in insts]
        first_store = next(i for i, t in enumerate(text) if t.startswith("global_store_dword"))
        before = text[:first_store]
        last_wait = max(i for i, t in enumerate(before) if t.startswith("s_waitcnt") and "vmcnt(0)" in t)
        loads_after_wait = [t for t in before[last_wait:] if t.startswith("global_load_dword ")]
        assert not loads_after_wait, (name, loads_after_wait)
