"""histogram of instruction mnemonics of one kernel in a hipcc -S listing (CPU-side proxy for VALU-bound kernels).
usage: isa_hist.py file.s <substring of the mangled kernel name> [--loops]"""
import re, sys, collections
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for n, l in enumerate(lines):
    if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0] and not l.startswith("."):
        start = n
        break
assert start is not None, "kernel not found"
end = next(n for n in range(start, len(lines)) if lines[n].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
cls = collections.Counter(); ops = collections.Counter()
def klass(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): return "valu_trans"
    if m.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")): return "valu_quarter"
    if m.startswith("v_"): return "valu"
    if m.startswith("ds_"): return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith("s_waitcnt"): return "waitcnt"
    if m.startswith("s_barrier"): return "barrier"
    if m.startswith("s_"): return "salu"
    return "other"
# basic blocks
blocks, cur, name = [], [], "entry"
for l in body[1:]:
    s = l.strip()
    if not s or s.startswith((";", "//")): continue
    if re.match(r"^\.LBB\d+_\d+:", s):
        blocks.append((name, cur)); name, cur = s[:-1], []; continue
    if s.startswith("."): continue
    m = s.split()[0]
    cur.append(m)
blocks.append((name, cur))
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(klass(m) for m in ins)
    tot.update(c)
    if "--loops" in sys.argv and len(ins) > 40:
        print("%-12s n=%5d  " % (name, len(ins)) + "  ".join("%s=%d" % kv for kv in sorted(c.items())))
        if "--ops" in sys.argv:
            oc = collections.Counter(ins)
            print("     " + " ".join("%s:%d" % kv for kv in oc.most_common(40)))
print("TOTAL", sum(tot.values()), dict(tot))
