"""Instruction histogram of one kernel in a hipcc -S dump.
usage: isa_histogram.py <asm file> <mangled-name substring> [top]"""
import collections, re, sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].rstrip().endswith(":"))
c = collections.Counter()
for l in txt[start + 1:]:
    s = l.strip()
    if s.startswith(".Lfunc_end"):
        break
    if not s or s[0] in ".;/" or s.split(";")[0].rstrip().endswith(":"):
        continue
    c[s.split()[0]] += 1
tot = sum(c.values())
grp = collections.Counter()
for k, v in c.items():
    g = ("lds_read" if k.startswith("ds_read") else "lds_write" if k.startswith("ds_write") else "lds_other" if k.startswith("ds_")
         else "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "waitcnt" if k.startswith("s_waitcnt")
         else "salu" if k.startswith("s_") else "valu_f64" if k.endswith("_f64") else "valu_other")
    grp[g] += v
print("total", tot, dict(grp))
for k, v in c.most_common(top):
    print(f"{k:28s}{v}")
