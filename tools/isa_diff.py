#!/usr/bin/env python3
"""Per-kernel comparison of the gfx950 code of two object files / shared libraries:

    python tools/isa_diff.py old.o new.o

Extracts the device code objects (llvm-objdump --offloading), disassembles them and compares the instruction streams
kernel by kernel (addresses and branch targets stripped).  Used to show that a refactoring left the serial kernels'
code untouched (round 6: the device building blocks moved into csrc/tree_device.h and gained FOREST template flags)."""
import hashlib
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump")


def kernels(path):
    tmp = tempfile.mkdtemp()
    try:
        local = os.path.join(tmp, "x.o")
        shutil.copy(path, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        out = {}
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", "--demangle", os.path.join(tmp, f)], check=True, capture_output=True,
                                 text=True).stdout
            name = None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    name = m.group(1)
                    out[name] = []
                    continue
                if name is None or not line.strip():
                    continue
                ins = line.split("//")[0].strip()
                ins = re.sub(r"\b(s_cbranch_\w+|s_branch)\s+\S+", r"\1 L", ins)
                out[name].append(ins)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    same = changed = 0
    for k in sorted(set(a) | set(b)):
        if k not in a:
            print("NEW      %6d instr  %s" % (len(b[k]), k))
        elif k not in b:
            print("GONE     %6d instr  %s" % (len(a[k]), k))
        elif hashlib.md5("\n".join(a[k]).encode()).digest() != hashlib.md5("\n".join(b[k]).encode()).digest():
            changed += 1
            print("CHANGED  %6d -> %6d instr  %s" % (len(a[k]), len(b[k]), k))
        else:
            same += 1
    print("%d kernels identical, %d changed" % (same, changed))


if __name__ == "__main__":
    main()
