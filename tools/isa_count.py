#!/usr/bin/env python3
"""Instruction accounting of a kernel's inner loops, read from the code object inside libhgmm_hip.so.

    python tools/isa_count.py [kernel-name-substring ...]

The gfx950 code objects are pulled out of the shared library (llvm-objdump --offloading on a temporary
copy), disassembled, and every loop (a backward branch inside the kernel) is summarised: VALU
instructions by class -- packed fp32 (v_pk_*), transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt), DPP,
fp64, other -- plus SALU / waitcnt / nop counts.  bench.py uses `loop_profile()` to turn the measured
launch time of the VALU-bound fused EM kernel into an issue-slot fraction without hard-coded counts.
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gpu-accelerated-point-cloud-registration-using-hierarchical-gmm_amd", "libhgmm_hip.so")
OBJDUMP = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "llvm-objdump")
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")

_cache = {}


def disassemble(lib=LIB):
    """-> {kernel symbol: [(address, mnemonic, full text)]} for every gfx950 code object in `lib`."""
    key = (lib, os.path.getmtime(lib))
    if key in _cache:
        return _cache[key]
    tmp = tempfile.mkdtemp(prefix="hgmm_isa_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        funcs = {}
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True,
                                 text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^([0-9a-f]+) <(\S+)>:$", line)
                if m:
                    cur = funcs.setdefault(m.group(2), [])
                    continue
                m = re.match(r"^\s+(\S+)(.*?)//\s*([0-9A-F]+):", line)
                if m and cur is not None:
                    cur.append((int(m.group(3), 16), m.group(1), line.strip()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    _cache[key] = funcs
    return funcs


def classify(mn, text):
    if mn.startswith("v_"):
        if mn.startswith(("v_mfma", "v_smfma")):
            return "mfma"
        if mn.startswith("v_pk_"):
            return "valu_pk"
        if mn.startswith(TRANS):
            return "valu_trans"
        if "_dpp" in mn or "quad_perm" in text or "row_" in text:
            return "valu_dpp"
        if "_f64" in mn:
            return "valu_f64"
        if mn.startswith("v_mfma") or mn.startswith("v_smfma"):
            return "mfma"
        if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        return "valu_other"
    if mn.startswith("s_nop"):
        return "nop"
    if mn.startswith("s_waitcnt"):
        return "waitcnt"
    if mn.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if mn.startswith("s_"):
        return "salu"
    if mn.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if mn.startswith("ds_"):
        return "lds"
    return "other"


def loops(instrs):
    """Innermost-first list of loops: each a dict of class counts + 'n' + 'span'."""
    base = instrs[0][0]
    index = {a: i for i, (a, _, _) in enumerate(instrs)}
    out = []
    for i, (addr, mn, text) in enumerate(instrs):
        if not (mn.startswith("s_cbranch") or mn == "s_branch"):
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", text)
        if not m:
            continue
        tgt = base + int(m.group(1), 16)
        if tgt <= addr and tgt in index:
            body = instrs[index[tgt]:i + 1]
            c = collections.Counter(classify(mn_, t_) for _, mn_, t_ in body)
            c["n"] = len(body)
            c["valu"] = sum(v for k, v in c.items() if k.startswith("valu_"))
            c["span"] = (index[tgt], i)
            out.append(dict(c))
    return out


def find_kernel(substr, lib=LIB):
    funcs = disassemble(lib)
    hits = [k for k in funcs if substr in k and funcs[k]]
    if not hits:
        raise KeyError("no kernel matching %r in %s" % (substr, lib))
    return sorted(hits, key=len)[0]


def loop_profile(substr, want=None, lib=LIB):
    """Class counts of ONE loop of the kernel matching `substr`.  `want(counts) -> bool` selects the loop
    (default: the loop with the most VALU instructions that contains no other loop)."""
    name = find_kernel(substr, lib)
    ls = loops(disassemble(lib)[name])
    inner = [l for l in ls if not any(o is not l and l["span"][0] <= o["span"][0] and o["span"][1] <= l["span"][1]
                                      for o in ls)]
    cands = [l for l in inner if want(l)] if want else inner
    if not cands:
        raise KeyError("no loop of %s satisfies the selector" % name)
    best = max(cands, key=lambda l: l["valu"])
    best = dict(best)
    best["kernel"] = name
    return best


def main():
    pats = sys.argv[1:] or ["flat_fused_pk_kernelILi13", "flat_estep_rows_pk_kernelILi3ELi1ELi4"]
    for p in pats:
        name = find_kernel(p)
        print(name)
        for l in loops(disassemble()[name]):
            print("   loop @instr %5d..%5d: %4d instr | VALU %4d (pk %d, trans %d, dpp %d, f64 %d, lane %d, other %d) "
                  "mfma %d salu %d smem %d vmem %d lds %d nop %d waitcnt %d"
                  % (l["span"][0], l["span"][1], l["n"], l["valu"], l.get("valu_pk", 0), l.get("valu_trans", 0),
                     l.get("valu_dpp", 0), l.get("valu_f64", 0), l.get("valu_lane", 0), l.get("valu_other", 0),
                     l.get("mfma", 0), l.get("salu", 0), l.get("smem", 0), l.get("vmem", 0), l.get("lds", 0),
                     l.get("nop", 0), l.get("waitcnt", 0)))


if __name__ == "__main__":
    main()
