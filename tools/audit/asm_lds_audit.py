"""Audit of the inline-asm LDS reads (score_sweep.h, compiled as part of coarse_match.hip; namespace sweep): hipcc does not know that the destination
registers of an asm `ds_read` are not valid until the matching `s_waitcnt lgkmcnt`, so a compiler-generated copy /
spill / use of them in between would read stale data (cdna_hip_programming.md §5.7 item 1).  This script compiles the
file to ISA and checks, for every asm block of ds_reads, that no instruction names a destination register before the
next lgkmcnt wait in program order, and that the kernels have no scratch.

    python tools/audit/asm_lds_audit.py [extra hipcc flags]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_regs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs(tok)
    return out


def audit(asm_path):
    lines = open(asm_path).read().split("\n")
    bad, blocks, kernel = [], 0, None
    i = 0
    while i < len(lines):
        l = lines[i].strip()
        if l.endswith(":") and l.startswith("_Z"):
            kernel = l[:-1]
        if l.startswith(";;#ASMSTART"):
            j, dst = i + 1, set()
            while not lines[j].strip().startswith(";;#ASMEND"):
                t = lines[j].strip().split()
                if t and t[0].startswith("ds_read"):
                    dst |= regs(t[1].rstrip(","))
                j += 1
            if dst:
                blocks += 1
                k = j + 1
                while k < len(lines):
                    t = lines[k].strip()
                    if t.startswith("s_waitcnt") and "lgkmcnt" in t:
                        break
                    if t.startswith(".Lfunc_end"):
                        bad.append((kernel, i, "no lgkmcnt wait after asm ds_read"))
                        break
                    if t and not t.startswith(";") and not t.startswith(".") and not t.endswith(":"):
                        ins = t.split(";")[0]
                        if not ins.startswith("ds_read") and all_regs(ins) & dst:
                            bad.append((kernel, k, ins))
                    k += 1
            i = j
        i += 1
    scratch = [l for l in lines if re.search(r"\.private_segment_fixed_size:\s+[1-9]", l) or re.search(r"\.vgpr_spill_count:\s+[1-9]", l)]
    return blocks, bad, scratch


def main():
    src = os.path.join(ROOT, "loftr_amd", "csrc", "coarse_match.hip")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "cm.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
               "-S", "--cuda-device-only", "-o", out, src, *sys.argv[1:]]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        blocks, bad, scratch = audit(out)
    print(f"{blocks} asm ds_read blocks; {len(bad)} early uses; {len(scratch)} kernels with scratch / spills")
    for b in bad[:20]:
        print("  EARLY USE", b)
    for s_ in scratch:
        print("  SCRATCH", s_.strip())
    return 1 if bad or scratch else 0


if __name__ == "__main__":
    sys.exit(main())
