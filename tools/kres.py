"""Per-kernel register / LDS / occupancy table of the HIP library as hipcc reports it
(-Rpass-analysis=kernel-resource-usage), cross-compiled for gfx950: no GPU needed.
usage: python tools/kres.py [filter-substring] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(REPO, "metrotrpl_amd", "csrc", "mtrpl_capi.hip")


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else ""
    extra = [a for a in sys.argv[1:] if a.startswith("-")]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/kres.so", SRC] + extra
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(\w[\w ]*?):\s+(\S+)", line) or re.search(r":\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]):\s+(\S+)", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        k, v = m.group(1), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'sSpl':>5s} {'vSpl':>5s} {'scr':>5s} {'LDS':>7s} {'occ':>4s}")
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name)
        if flt and flt not in name:
            continue
        g = lambda k: r.get(k, "?")
        print(f"{name[:70]:70s} {g('VGPRs'):>5s} {g('AGPRs'):>5s} {g('TotalSGPRs'):>5s} {g('SGPRs Spill'):>5s} "
              f"{g('VGPRs Spill'):>5s} {g('ScratchSize [bytes/lane]'):>5s} {g('LDS Size [bytes/block]'):>7s} {g('Occupancy [waves/SIMD]'):>4s}")


if __name__ == "__main__":
    main()
