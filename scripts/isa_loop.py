"""Compact view of one kernel of a hipcc `-save-temps` assembly file: registers, scratch, and the region around its MFMAs as one
line per scheduling-relevant event (runs of VALU instructions are counted, SALU dropped).
    python scripts/isa_loop.py file.s <mangled-name-substring> [--full]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
full = "--full" in sys.argv
s = open(path).read()
names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", s, re.M) if key in m.group(1)]
if not names:
    sys.exit(f"no kernel matching {key}")
name = names[0]
i = s.index(name + ":")
body = s[i:s.index(".end_amdhsa_kernel", i)]
for k in (".amdhsa_next_free_vgpr", ".amdhsa_accum_offset", ".amdhsa_private_segment_fixed_size", ".amdhsa_next_free_sgpr"):
    m = re.search(re.escape(k) + r"\s+(\S+)", body)
    print(k, m.group(1) if m else None)
lines = body.split("\n")
mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
print("instructions ~", sum(1 for l in lines if l.startswith("\t") and not l.startswith("\t.")), " mfma", len(mf), " scratch ops",
      sum("scratch_" in l for l in lines), " first/last mfma line", mf[0], mf[-1])
lo, hi = max(0, mf[0] - 80), min(len(lines), mf[-1] + 60)
valu = 0
mrun = 0
out = []


def flush():
    global valu, mrun
    if mrun:
        out.append(f"  {mrun} x mfma")
        mrun = 0
    if valu:
        out.append(f"  [{valu} valu]")
        valu = 0


for k in range(lo, hi):
    l = lines[k].split(";")[0].strip()
    if not l or (l.startswith(".") and not l.startswith(".LBB")):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"):
        if valu:
            flush()
        if full:
            out.append(f"{k}: {l[:80]}")
        else:
            mrun += 1
        continue
    if op.startswith("v_"):
        if mrun:
            flush()
        valu += 1
        continue
    if op.startswith("s_") and not any(op.startswith(p) for p in ("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "s_endpgm")):
        continue
    flush()
    if full or not op.startswith("s_waitcnt") or "vmcnt" in l:
        out.append(f"{k}: {l[:80]}")
flush()
# collapse repeated ds_read / global_load lines
res = []
for o in out:
    t = o.split(": ", 1)[-1].split()[0] if ": " in o else o
    if res and ": " in o and res[-1][0] == t and t in ("ds_read_b128", "global_load_dwordx4", "ds_write_b128"):
        res[-1][1] += 1
    else:
        res.append([t if ": " in o else None, 1, o])
for t, n, o in res:
    print(o if n == 1 else f"{o}   (x{n})")
