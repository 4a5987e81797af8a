"""Disassemble the built library and check the K loops of gemm_x3_kernel: no `s_waitcnt vmcnt(0)` inside a two-tile trip.

Why (profiles/r04_x3_ktile_before_after.txt): the loop prefetches two K tiles ahead; for two rounds hipcc
drained every load at the top of each trip because the loop had an exit between its two halves (a second back edge on
which the first half's loads are in flight) and a conditional scalar load between the loads - the prefetch distance was 1
in effect and nobody saw it in the source.  This check makes the finding impossible to lose: a loop is a backward branch;
the trips are the loops with exactly 48 matrix instructions (2 tiles x 24) and no inner loop.

usage: check_x3_loops.py lib.so      exit 0 = clean, 1 = a trip drains the loads, 2 = could not disassemble / no loops found"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
HEAD = re.compile(r"^([0-9a-f]+) <(\S*gemm_x3_kernel\S*)>:")
ANYHEAD = re.compile(r"^[0-9a-f]+ <\S+>:")
ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
TARGET = re.compile(r"<\S+\+0x([0-9a-fA-F]+)>\s*$")


def functions(dis, head=None):
    """-> {name: (start address, [(address, text)])} of the gemm_x3_kernel instantiations (or of the kernels `head` matches)"""
    head = head or HEAD
    out, cur = {}, None
    for ln in dis.splitlines():
        m = head.match(ln)
        if m:
            cur = out.setdefault(m.group(2), (int(m.group(1), 16), []))
            continue
        if ANYHEAD.match(ln):
            cur = None
            continue
        if cur is not None and "\t" in ln:
            a = ADDR.search(ln)
            if a:
                cur[1].append((int(a.group(1), 16), ln.strip()))
    return out


def trips(start, insts):
    """innermost loops with exactly 48 MFMAs -> [(first index, last index, [waits on vmcnt])]"""
    index = {a: i for i, (a, _) in enumerate(insts)}
    loops = []
    for i, (a, t) in enumerate(insts):
        if not t.startswith(("s_cbranch", "s_branch")):
            continue
        m = TARGET.search(t)
        if not m:
            continue
        j = index.get(start + int(m.group(1), 16))
        if j is not None and j <= i:
            loops.append((j, i))
    res = []
    for lo, hi in loops:
        body = [t for _, t in insts[lo:hi + 1]]
        if sum("v_mfma" in t for t in body) != 48:
            continue
        if any(lo <= l2 and h2 <= hi and (l2, h2) != (lo, hi) and sum("v_mfma" in t for _, t in insts[l2:h2 + 1]) >= 24
               for l2, h2 in loops):
            continue   # not innermost
        res.append((lo, hi, [t.split("//")[0].strip() for t in body if "vmcnt" in t]))
    return res


def kernels(lib, substr):
    """-> {name: [instruction text]} of every kernel of the library whose mangled name contains `substr`"""
    head = re.compile(r"^([0-9a-f]+) <(\S*" + re.escape(substr) + r"\S*)>:")
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = {}
        for o in sorted(glob.glob(local + ".*gfx950*")):
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", o], check=True, capture_output=True, text=True).stdout
            if substr in dis:
                for name, (_, insts) in functions(dis, head).items():
                    out[name] = [t for _, t in insts]
        return out


def scan(lib):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        report = {}
        for o in sorted(glob.glob(local + ".*gfx950*")):
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", o], check=True, capture_output=True, text=True).stdout
            if "gemm_x3_kernel" not in dis:
                continue
            for name, (start, insts) in functions(dis).items():
                report[name] = trips(start, insts)
        return report


def vm(w):
    return [int(re.search(r"vmcnt\((\d+)\)", x).group(1)) for x in w if "vmcnt(" in x]


def main(argv):
    try:
        report = scan(argv[0])
    except (OSError, subprocess.CalledProcessError) as e:
        print(f"{argv[0]}: cannot disassemble: {e}")
        return 2
    n = sum(len(v) for v in report.values())
    if len(report) < 6 or n < 3 * len(report):
        print(f"{argv[0]}: {len(report)} gemm_x3_kernel instantiations, {n} two-tile trips - nothing to check?")
        return 2
    rc, notes = 0, []
    for name, tr in sorted(report.items()):
        batched = "ELb1ELb0EEE" in name or "ELb1ELb1EEE" in name      # <A_KC, B_KC, BATCH = true, B_PL>
        # the copy for interior tiles is the shortest trip: at EVERY wait a whole tile's loads (8) must stay in flight
        lo, hi, w = min(tr, key=lambda t: t[1] - t[0])
        if not vm(w) or min(vm(w)) < 8:
            print(f"{name}: the interior-tile trip waits below a tile's loads in flight: {w}")
            rc = 1
        for lo, hi, w in tr:
            if 0 in vm(w):
                if batched:
                    notes.append(name)      # edge-tile copies of the batched form (the step's products are whole tiles)
                else:
                    print(f"{name}: a trip drains every load: {w}")
                    rc = 1
    if rc == 0:
        extra = f"; edge-tile copies of {len(set(notes))} batched instantiation(s) still drain" if notes else ""
        print(f"{argv[0]}: {len(report)} gemm_x3_kernel instantiations, {n} two-tile trips: a tile's loads stay in flight at "
              f"every wait of the interior-tile trips, no vmcnt(0) in the unbatched ones{extra}")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
