"""What hipcc made of the loops of a kernel file: for every innermost loop that loads from global memory, the order of
loads (ld), LDS-DMA (DMA), stores (st), barriers (|) and `s_waitcnt vmcnt(N)` in the generated code, with the count of
matrix instructions.  A software pipeline that works shows COUNTED waits (vmcnt(8), vmcnt(15) ...); `vmcnt(0)` inside a
loop means every load in flight is drained there - round 4 found four such stalls this way that no
profile had pointed at.

usage: python tools/asm_loops.py [--outer] applecider_amd/csrc/ac_dwconv.hip [kernel-name-substring]
       (--outer: also the loops that contain other loops, e.g. a persistent item loop around DMA loops)
(compiles the file to assembly with the library's flags: seconds for most files, ~10 min for ac_gemm.hip)"""
import itertools
import os
import re
import subprocess
import sys
import tempfile

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-Wno-pass-failed", "-fno-slp-vectorize",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S", "--cuda-device-only"]


def loops_of(asm, only=None, outer=False):
    lines = asm.split("\n")
    names = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:", l)]
    for start, name in names:
        if only and only not in name:
            continue
        try:
            end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        except StopIteration:
            continue
        fn = lines[start:end]
        labels = {l.split(":")[0]: i for i, l in enumerate(fn) if l.startswith(".LBB")}
        loops = []
        for i, l in enumerate(fn):
            m = re.search(r"s_cbranch_\w+ (\.LBB\w+)|s_branch (\.LBB\w+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < i:
                    loops.append((labels[t], i))
        out = []
        for lo, hi in loops:
            if not outer and any(l2 >= lo and h2 <= hi and (l2, h2) != (lo, hi) for l2, h2 in loops):
                continue
            body = fn[lo:hi + 1]
            if not any("global_load" in x or "buffer_load" in x for x in body):
                continue
            seq = []
            for x in body:
                if "vmcnt" in x:
                    seq.append(re.search(r"vmcnt\(\d+\)", x).group(0))
                elif "global_load_lds" in x or ("buffer_load" in x and "lds" in x):
                    seq.append("DMA")
                elif "global_load" in x or "buffer_load" in x:
                    seq.append("ld")
                elif "global_store" in x or "buffer_store" in x:
                    seq.append("st")
                elif "s_barrier" in x:
                    seq.append("|")
            comp = [f"{k}x{n}" if n > 1 else k for k, n in ((k, len(list(g))) for k, g in itertools.groupby(seq))]
            valu = sum(1 for x in body if x.startswith("\tv_") and "mfma" not in x)
            out.append(f"   loop of {hi - lo} instructions, {sum('v_mfma' in x for x in body)} mfma, {valu} valu: " + " ".join(comp))
        if out:
            yield name, out


def main(argv):
    outer = "--outer" in argv
    argv = [a for a in argv if a != "--outer"]
    src, only = argv[0], (argv[1] if len(argv) > 1 else None)
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        subprocess.run([HIPCC] + FLAGS + ["-I", os.path.dirname(os.path.abspath(src)), src, "-o", asm], check=True,
                       capture_output=True)
        for name, out in loops_of(open(asm).read(), only, outer):
            print("==", name)
            print("\n".join(out))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
