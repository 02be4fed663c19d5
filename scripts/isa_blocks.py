#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -save-temps .s file.
usage: isa_blocks.py file.s <kernel-symbol-substring> [--dump BLOCKLABEL]"""
import collections, re, sys
path, sym = sys.argv[1], sys.argv[2]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().split(":")[0].endswith("E") and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = m.group(1); blocks[cur] = []
        continue
    blocks[cur].append(s.split(";")[0].strip())
def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_exp"): return "exp"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_"): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    return "salu"
for name, ins in blocks.items():
    c = collections.Counter(kind(i.split()[0]) for i in ins)
    if dump == name:
        print("\n".join(ins)); continue
    if dump is None and len(ins) > 20:
        valu = collections.Counter(i.split()[0] for i in ins if kind(i.split()[0]) == "valu")
        print(f"{name:12s} n={len(ins):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        print("             valu: " + " ".join(f"{k}={v}" for k, v in valu.most_common()))
