#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  bash scripts/profile.sh <tag> [bench args, e.g. --config cfg3]
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv, pmc_*.csv summaries, hbm_traffic_entry.json}.  Counters are
# collected in their own passes (never together with sys/hip tracing), as the pool requires.
# hbm_traffic_entry.json = the record to merge into profiles/hbm_traffic.json under the config's name: HBM bytes per
# launch (2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE), the kernel's name and the
# sha256 of the kernel sources it was measured on - bench.py reports `traffic` only when both still match.
set -u
TAG=${1:-run}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --no-cpu-baseline --no-secondary $*"

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH --steps 20 --warmup 5 > $OUT/kt.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/kt/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/kernel_stats.csv", "w") as w:
        w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            name = r["Name"]
            name = name if len(name) < 90 else name[:87] + "..."
            w.write(",".join([f'"{name}"', r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
for f in glob.glob(out + "/kt/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "fa_fwd" in r["Kernel_Name"]]
    if rows:
        r = rows[-1]
        open(out + "/dispatch_info.txt", "w").write(
            f'kernel={r["Kernel_Name"]}\nLDS={r["LDS_Block_Size"]} scratch={r["Scratch_Size"]} VGPR={r["VGPR_Count"]} AGPR={r["Accum_VGPR_Count"]} SGPR={r["SGPR_Count"]} wg={r["Workgroup_Size_X"]} grid={r["Grid_Size_X"]}\n')
PY
cat $OUT/kernel_stats.csv | cut -c1-200

pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $BENCH --steps 4 --warmup 2 > $OUT/pmc_$name.log 2>&1
  python3 - "$OUT" "$name" <<'PY'
import csv, glob, sys, collections
out, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/pmc_{name}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_{name}.csv", "w") as w:
    w.write("counter,dispatches,mean_per_dispatch\n")
    for k, v in sorted(agg.items()):
        w.write(f"{k},{len(v)},{sum(v)/len(v):.1f}\n")
print(open(f"{out}/pmc_{name}.csv").read())
PY
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
pmc sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum
pmc grbm GRBM_GUI_ACTIVE
python3 - "$OUT" "$TAG" "$(python3 $R/bench.py --print-csrc-fingerprint)" <<'PY'
import csv, json, sys
out, tag, sha = sys.argv[1:4]
def val(name, counter):
    for r in csv.DictReader(open(f"{out}/pmc_{name}.csv")):
        if r["counter"] == counter:
            return float(r["mean_per_dispatch"])
    return None
fetch, write = val("fetch", "FETCH_SIZE"), val("write", "WRITE_SIZE")   # KiB per dispatch
import re
kern = ""
try:
    m = re.search(r"fa_fwd_[a-z0-9]+_kernel", open(f"{out}/dispatch_info.txt").read())
    kern = m.group(0) if m else ""
except Exception:
    pass
rec = {"kernel": kern, "csrc_sha256_16": sha, "source": f"profiles/{tag}",
       "fetch_size_kib": fetch, "write_size_kib": write,
       "hbm_bytes_per_launch": round((2 * fetch + write) * 1024) if fetch is not None and write is not None else None}
json.dump(rec, open(f"{out}/hbm_traffic_entry.json", "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/kt $OUT/pmc_*/
ls $OUT
