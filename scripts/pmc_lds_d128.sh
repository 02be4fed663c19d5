# Developer aid: LDS counters of the forward kernel at head_dim 128 (bench.py --config cfg3) - run on the GPU box from the repo root.
# Result this round: SQ_LDS_BANK_CONFLICT 0.2 % of SQ_LDS_IDX_ACTIVE, LDS index unit ~26 % busy: the LDS is not what bounds D=128.
R=$PWD
OUT=$R/gpurun_out/pmc128
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/raw -- python3 $R/bench.py --no-cpu-baseline --config cfg3 --steps 3 --warmup 1 --ramp-ms 10 > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/raw/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fa_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(f"{k},{len(v)},{sum(v)/len(v):.1f}")
PY
rm -rf $OUT/raw
