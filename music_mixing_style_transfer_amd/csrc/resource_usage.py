"""Print VGPR/AGPR/LDS/occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage."""
import os, re, subprocess, sys
out = subprocess.run(["make", "-s", "resource-usage"], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (.+?): (.+?) \[-Rpass", line) or re.search(r"remark: (.+?): (.+)$", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    print(f'{name[:70]:70s} vgpr {r.get("VGPRs","?"):>4} agpr {r.get("AGPRs","?"):>4} spill {r.get("VGPRs Spill","?"):>3} '
          f'scratch {r.get("ScratchSize [bytes/lane]","?"):>4} occ {r.get("Occupancy [waves/SIMD]","?"):>2} lds {r.get("LDS Size [bytes/block]","?"):>6}')
