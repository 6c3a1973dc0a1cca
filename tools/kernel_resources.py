"""Registers, spills, LDS and occupancy of every kernel of one csrc/*.hip unit, from hipcc's
-Rpass-analysis=kernel-resource-usage (no GPU needed):  python tools/kernel_resources.py roi_align_records.hip [filter]"""
import os
import re
import subprocess
import sys

from detectron_pytorch_amd import build as hip_build


def main():
    unit = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    src = os.path.join(hip_build.CSRC, unit)
    cmd = [hip_build.hipcc()] + hip_build.flags() + ["-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    rows, cur = [], None
    for line in out.splitlines():
        m = re.search(r"remark: +([A-Za-z][A-Za-z /\[\]]*?): (.+?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2).strip()
        if k == "Function Name":
            name = subprocess.run(["c++filt", v], stdout=subprocess.PIPE, text=True).stdout.strip()
            name = name.replace("mi::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
            cur = {"name": re.sub(r"\(.*", "", name)}
            rows.append(cur)
        elif cur is not None:
            cur[k] = v
    for r in rows:
        if pat in r["name"]:
            print("%-60s VGPR %3s AGPR %3s SGPR %3s spill(s/v) %s/%s scratch %s occ %s LDS %s" % (
                r["name"], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))


if __name__ == "__main__":
    main()
