"""Where a kernel's scratch accesses (register spills, stack objects) sit in the source: compiles the HIP library with line
tables (-gline-tables-only: same code as the shipped build), cuts one kernel out of the assembly and prints every scratch_*
instruction with the source line it belongs to.  Cross-compiled for gfx950: no GPU needed.
usage: python tools/spill_map.py [mangled-name-substring, default spikeILi4ELb1E] [--asm=FILE: keep the kernel's assembly] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "metrotrpl_amd", "csrc", "mtrpl_capi.hip")


def main():
    key = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "spikeILi4ELb1E"
    keep = [a[6:] for a in sys.argv[1:] if a.startswith("--asm=")]
    extra = [a for a in sys.argv[1:] if a.startswith("-") and not a.startswith("--asm=")]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-gline-tables-only",
                        "-save-temps", "-o", os.path.join(d, "k.so"), SRC] + extra, cwd=d, capture_output=True, check=True)
        asm = open(os.path.join(d, "mtrpl_capi-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    files = {}
    for l in asm:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
    start = next(i for i, l in enumerate(asm) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
    end = next(i for i in range(start, len(asm)) if asm[i].lstrip().startswith(".amdhsa_kernel"))
    if keep:
        open(keep[0], "w").write("\n".join(asm[start:end]))
    loc, n = None, 0
    for i in range(start, end):
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", asm[i])
        if m:
            loc = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        if "scratch_" in asm[i]:
            n += 1
            print(f"{i - start:6d}  {loc[0]:20s} {loc[1]:5d}  {asm[i].strip()[:100]}")
    print(f"{n} scratch instructions in {end - start} lines of {asm[start][:60]}")


if __name__ == "__main__":
    main()
