"""Dev tool (CPU): registers / scratch / occupancy per kernel from build/resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/resource_table.py [pattern]"""
import re
import subprocess
import sys

pat = sys.argv[1] if len(sys.argv) > 1 else "minimize"
txt = open("build/resource_usage.txt").read()
cur, d = None, {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        d[cur] = {}
        continue
    for key, short in (("VGPRs:", "v"), ("AGPRs:", "a"), ("ScratchSize [bytes/lane]:", "scratch"), ("Occupancy [waves/SIMD]:", "occ"), ("LDS Size [bytes/block]:", "lds")):
        m = re.search(re.escape(key) + r"\s*(\d+)", line)
        if m and cur and short not in d[cur]:
            d[cur][short] = int(m.group(1))
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
for mangled, name in zip(d, names):
    if pat in name:
        print(f"{name.split('(')[0][-64:]:64s} {d[mangled]}")
