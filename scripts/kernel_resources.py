#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip TU (hipcc -Rpass-analysis=kernel-resource-usage).

    python scripts/kernel_resources.py psfm_solver.hip [extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "particle-sfm_amd"))
import build as psfm_build


def main():
    src = os.path.join(psfm_build.CSRC, sys.argv[1])
    cmd = [psfm_build.HIPCC] + psfm_build.FLAGS + sys.argv[2:] + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s+\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s+\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem)
        print("%-52s VGPR %-4s AGPR %-3s spillV %-3s spillS %-3s scratch %-5s LDS %-6s occ %s" % (
            dem[:52], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"),
            r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
