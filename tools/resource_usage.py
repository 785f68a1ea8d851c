#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of a kernel source file (hipcc
-Rpass-analysis=kernel-resource-usage), e.g.  python tools/resource_usage.py encode_encrypt [-D...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "encode_encrypt"
csrc = os.path.join(ROOT, "seal-embedded_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950",
       "-I" + csrc, "-I" + os.path.join(ROOT, "include"), "-c", os.path.join(csrc, "kernels", name + ".hip"),
       "-o", "/tmp/ru_" + name + ".o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("%-70s %5s %5s %8s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    nm = re.sub(r"\(.*", "", r["name"]).replace("void seamd::", "")
    print("%-70s %5s %5s %8s %4s %7s" % (nm[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                        r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                                        r.get("LDS Size [bytes/block]", "?")))
