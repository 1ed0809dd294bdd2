#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from `make -C dvi_ekf_amd/csrc resource-usage` output
(stdin or a file): one line per kernel, demangled."""
import re
import subprocess
import sys

def main():
    src = open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin
    rows, cur = [], None
    for l in src:
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", l)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split()[0]] = v
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("void eskf::", "")
        print(f"{n:64s} VGPR {r.get('VGPRs'):>4s} SGPR {r.get('TotalSGPRs'):>4s} scratch {r.get('ScratchSize'):>5s} occ {r.get('Occupancy')} lds {r.get('LDS')}")

if __name__ == "__main__":
    main()
