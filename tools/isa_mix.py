"""Dev tool: static instruction mix of one kernel in build/slam_hip-hip-amdgcn-amd-amdhsa-gfx950.s, split at basic-block labels.
usage: tools/isa_mix.py <mangled-name-prefix> [--blocks]"""
import collections, re, sys
t = open("build/slam_hip-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
pref = sys.argv[1]
start = next(i for i, l in enumerate(t) if l.startswith(pref) and ":" in l)
end = next(i for i in range(start, len(t)) if t[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in t[start + 1 : end + 1]:
    s = l.strip()
    if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
        continue
    if re.match(r"^\.LBB\d+_\d+:", s):
        blocks.append((name, cur)); name, cur = s.split(":")[0], []
        continue
    cur.append(s.split()[0])
blocks.append((name, cur))
allins = [i for _, b in blocks for i in b]
def mix(ins):
    c = collections.Counter()
    for k in ins:
        if not k.startswith("v_"):
            c["~" + k.split("_")[0] + "_" + (k.split("_")[1] if "_" in k else "")] += 1
        else:
            c[re.sub(r"_(e32|e64|dpp|sdwa)$", "", k)] += 1
    return c
print("instructions", len(allins), "valu", sum(1 for i in allins if i.startswith("v_")))
if "--blocks" in sys.argv:
    for n, b in blocks:
        if len(b) >= 40:
            v = [i for i in b if i.startswith("v_")]
            f64 = sum(1 for i in v if "f64" in i)
            print(f"{n:12s} total {len(b):5d} valu {len(v):5d} f64 {f64:5d} mov {sum(1 for i in v if i.startswith('v_mov') or i.startswith('v_accvgpr')):4d} cnd {sum(1 for i in v if 'cndmask' in i):4d}")
else:
    for k, v in mix(allins).most_common(70):
        print(f"{k:34s}{v}")
