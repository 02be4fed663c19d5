#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:
#     bash scripts/profile.sh <tag> [bench args, e.g. --config cfg3 | --config cfg2 --backward | --config cfg4_attn]
# Produces gpurun_out/prof_<tag>/{kernel_stats.csv, kernel_stats_timed.csv, pmc_*.csv summaries, hbm_traffic_entry.json}.
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats summary of the whole command (first call, clock ramp and warm-up
#                           launches included: its mean is an upper bound of the timed region's)
#   kernel_stats_timed.csv  the same trace restricted to the TIMED region: bench.py runs with --no-probes, so the last
#                           --steps dispatches of each attention kernel are exactly the launches between its two events;
#                           mean / median / min / max over them - this is the figure that reproduces the bench line
# Counters are collected in their own passes (never together with sys/hip tracing), as the pool requires; they are read
# from the timed-region dispatches too.
# hbm_traffic_entry.json = the record to merge into profiles/hbm_traffic.json under the config's name: HBM bytes per
# launch (2 x FETCH_SIZE, the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE; summed over the kernels of one step
# for the backward), the kernel's name and the sha256 of the kernel sources it was measured on - bench.py reports
# `traffic` only when both still match.
set -u
TAG=${1:-run}; shift || true
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${PROF_STEPS:-20}
FILTER="fa_fwd"
case " $* " in *" --backward "*) FILTER="fa_bwd";; esac
BENCH="python3 $R/bench.py --no-cpu-baseline --no-secondary --no-probes $*"

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $BENCH --steps $STEPS --warmup 5 > $OUT/kt.log 2>&1
grep '^{' $OUT/kt.log | tail -1 > $OUT/bench_line_under_rocprof.json
python3 - "$OUT" "$FILTER" "$STEPS" <<'PY'
import csv, glob, statistics, sys
out, flt, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
for f in glob.glob(out + "/kt/*/*_kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(out + "/kernel_stats.csv", "w") as w:
        w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
        for r in rows:
            name = r["Name"]
            name = name if len(name) < 90 else name[:87] + "..."
            w.write(",".join([f'"{name}"', r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]]) + "\n")
for f in glob.glob(out + "/kt/*/*_kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if flt in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append(r)
    with open(out + "/kernel_stats_timed.csv", "w") as w:
        w.write("Name,Calls,AverageNs,MedianNs,MinNs,MaxNs,note\n")
        for name, rs in by.items():
            last = rs[-steps:]
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
            short = name if len(name) < 90 else name[:87] + "..."
            w.write(f'"{short}",{len(d)},{sum(d) / len(d):.0f},{statistics.median(d):.0f},{min(d)},{max(d)},last {len(d)} of {len(rs)} dispatches = the timed region\n')
        if len(by) > 1:     # several kernels per step (backward): the step's total
            tot = [sum(int(rs[-steps:][i]["End_Timestamp"]) - int(rs[-steps:][i]["Start_Timestamp"]) for rs in by.values() if len(rs) >= steps) for i in range(steps)]
            w.write(f'"(sum over the kernels of one step)",{len(tot)},{sum(tot) / len(tot):.0f},{statistics.median(tot):.0f},{min(tot)},{max(tot)},\n')
    if rows:
        with open(out + "/dispatch_info.txt", "w") as w:
            for name, rs in by.items():
                r = rs[-1]
                w.write(f'kernel={r["Kernel_Name"]}\nLDS={r["LDS_Block_Size"]} scratch={r["Scratch_Size"]} VGPR={r["VGPR_Count"]} AGPR={r["Accum_VGPR_Count"]} SGPR={r["SGPR_Count"]} wg={r["Workgroup_Size_X"]} grid={r["Grid_Size_X"]}\n')
PY
cat $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/kernel_stats_timed.csv | cut -c1-250

pmc() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- $BENCH --steps 4 --warmup 2 --ramp-ms 200 > $OUT/pmc_$name.log 2>&1
  python3 - "$OUT" "$name" "$FILTER" <<'PY'
import csv, glob, re, sys, collections
out, name, flt = sys.argv[1], sys.argv[2], sys.argv[3]
def short(n):
    m = re.search(r"fa_(fwd|bwd)_[a-z0-9_]*kernel", n)
    return m.group(0) if m else n
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/pmc_{name}/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_{name}.csv", "w") as w:
    w.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for kern, cs in sorted(agg.items()):
        for k, v in sorted(cs.items()):
            v = v[-4:]          # the timed steps (the ramp's dispatches run at a lower clock)
            w.write(f"{kern},{k},{len(v)},{sum(v)/len(v):.1f}\n")
print(open(f"{out}/pmc_{name}.csv").read())
PY
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
pmc sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc l2 TCC_HIT_sum TCC_MISS_sum
pmc grbm GRBM_GUI_ACTIVE
python3 - "$OUT" "$TAG" "$R" <<'PY'
import csv, json, subprocess, sys
out, tag, root = sys.argv[1:4]
def val(name, counter):     # summed over the kernels of one step
    tot, kerns = 0.0, []
    for r in csv.DictReader(open(f"{out}/pmc_{name}.csv")):
        if r["counter"] == counter:
            tot += float(r["mean_per_dispatch"]); kerns.append(r["kernel"])
    return (tot, kerns) if kerns else (None, [])
(fetch, kerns), (write, _) = val("fetch", "FETCH_SIZE"), val("write", "WRITE_SIZE")   # KiB per dispatch
# the fingerprint of the sources these kernels are built from (bench.py csrc_fingerprint)
sha = subprocess.run([sys.executable, f"{root}/bench.py", "--print-csrc-fingerprint", " + ".join(kerns)], capture_output=True, text=True).stdout.strip()
rec = {"kernel": " + ".join(kerns), "csrc_sha256_16": sha, "source": f"profiles/{tag}",
       "fetch_size_kib": fetch, "write_size_kib": write,
       "hbm_bytes_per_launch": round((2 * fetch + write) * 1024) if fetch is not None and write is not None else None}
json.dump(rec, open(f"{out}/hbm_traffic_entry.json", "w"), indent=1)
print(json.dumps(rec))
PY
rm -rf $OUT/kt $OUT/pmc_*/
ls $OUT
