"""Per-kernel register / scratch / occupancy table of the precise TU (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(root, "mitsuba-im_amd", "csrc"), "resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None; rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("mi_precise::", "").replace("void ", ""); rows[cur] = {}
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "TotalSGPRs"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur: rows[cur][key.split(" ")[0]] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for k, v in rows.items():
    if flt in k: print(f"{k:55s} VGPR {v.get('VGPRs', 0):4d} SGPR {v.get('TotalSGPRs', 0):4d} scratch {v.get('ScratchSize', 0):4d} occ {v.get('Occupancy', 0)} lds {v.get('LDS', 0)}")
