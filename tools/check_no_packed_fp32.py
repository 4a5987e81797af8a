"""Disassemble the gfx950 code objects of the built libraries and fail on packed-fp32 VALU instructions.

Why (profiles/r03_fft_coresidency_root_cause.txt): `v_pk_add_f32` / `v_pk_mul_f32` / `v_pk_fma_f32`
returned wrong high halves in transform workgroups that shared a CU with certain other kernels; the library is
therefore built with `-Xclang -target-feature -Xclang -packed-fp32-ops` (csrc/Makefile NOPK).  This check is what makes
that fix impossible to lose: the Makefile runs it after linking, tests/test_no_packed_fp32.py runs it on CPU.

usage: check_no_packed_fp32.py lib.so [lib2.so ...]     exit 0 = clean, 1 = found, 2 = could not disassemble"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
PACKED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


def scan(lib):
    """-> (number of gfx950 code objects, number of instructions disassembled, {mnemonic: count})"""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        objs = sorted(glob.glob(local + ".*gfx950*"))
        insts, found = 0, {}
        for o in objs:
            out = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", o], check=True, capture_output=True, text=True).stdout
            for ln in out.splitlines():
                if "\t" not in ln:
                    continue
                insts += 1
                m = PACKED.search(ln)
                if m:
                    found[m.group(0)] = found.get(m.group(0), 0) + 1
        return len(objs), insts, found


def main(argv):
    rc = 0
    for lib in argv:
        try:
            n, insts, found = scan(lib)
        except (OSError, subprocess.CalledProcessError) as e:
            print(f"{lib}: cannot disassemble: {e}")
            return 2
        if n == 0 or insts < 1000:
            print(f"{lib}: {n} gfx950 code objects, {insts} instructions - nothing to check?")
            return 2
        if found:
            print(f"{lib}: packed-fp32 instructions present (NOPK lost?): {found}")
            rc = 1
        else:
            print(f"{lib}: {n} gfx950 code objects, {insts} instructions, no v_pk_{{add,mul,fma}}_f32")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
