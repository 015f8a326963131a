#!/usr/bin/env python3
"""Dev tool (CPU): per-basic-block instruction mix of the optimizer kernel in build/kc/one-*.s (tools/kcompile.sh)."""
import re, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else "build/kc/one-hip-amdgcn-amd-amdhsa-gfx950.s"
kern = sys.argv[2] if len(sys.argv) > 2 else "minimize_kernel"
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l))
end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
cats = [("fma64", r"v_fma_f64|v_fmac_f64"), ("mul64", r"v_mul_f64"), ("add64", r"v_add_f64"), ("oth64", r"v_(max|min|rcp|rsq|cvt_f64|cvt_f32_f64|cvt_i32_f64|rndne|trunc|floor|fract|ldexp|cmp_\w+_f64|cmpx?_\w+_f64|div)\w*f64|v_cvt_f64"),
        ("pk32", r"v_pk_"), ("dpp", r"_dpp"), ("cnd", r"v_cndmask"), ("rdlane", r"v_readlane|v_readfirstlane"), ("wrlane", r"v_writelane"),
        ("mov", r"v_mov_b32|v_accvgpr"), ("valu_other", r"^\s+v_"), ("ds", r"^\s+ds_"), ("smem", r"s_load"), ("vmem", r"global_|scratch_|buffer_|flat_"),
        ("wait", r"s_waitcnt"), ("salu", r"^\s+s_")]
blocks, cur, name = [], collections.Counter(), "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blocks.append((name, cur)); cur = collections.Counter(); name = m.group(1); continue
    if not re.match(r"^\s+[a-z]", l) or l.strip().startswith((".", ";")):
        continue
    for c, pat in cats:
        if re.search(pat, l):
            cur[c] += 1
            break
    cur["n"] += 1
    if re.search(r"s_cbranch|s_branch", l):
        cur["br"] = l.split()[-1]
blocks.append((name, cur))
hdr = ["n"] + [c for c, _ in cats]
print("%-12s" % "block" + "".join("%7s" % h for h in hdr) + "  branch")
tot = collections.Counter()
for nm, c in blocks:
    if c["n"] >= int(sys.argv[3]) if len(sys.argv) > 3 else c["n"] >= 30:
        print("%-12s" % nm + "".join("%7d" % c[h] for h in hdr) + "  " + str(c.get("br", "")))
    for h in hdr:
        tot[h] += c[h]
print("%-12s" % "TOTAL" + "".join("%7d" % tot[h] for h in hdr))
