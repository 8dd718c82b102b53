#!/bin/bash
# Runs on the GPU box: SQ / LDS / TCP counters of tools/run_one.py for a list of GNNA_TUNE variants (what a kernel
# variant spends its wave cycles on).  Usage: tools/pmc_sq.sh <tag> "<config> <D> <steps>" "VARIANT1" ...
TAG=$1; shift
WL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcsq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for V in "$@"; do
  i=$((i+1))
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM" \
             "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_sum"; do
    name=$(echo $set | tr ' ' '+' | cut -c1-30)
    GNNA_TUNE="$V" rocprofv3 --pmc $set --kernel-include-regex "stream_kernel|sweep_kernel" -T -d $OUT/v$i/$name -o pmc -f csv -- python $R/tools/run_one.py $WL > $OUT/v${i}_$name.log 2>&1
  done
  echo "$V" > $OUT/v$i/VARIANT
done
cd $R
python - "$OUT" <<'PY' > $OUT/SUMMARY.md
import csv, glob, os, sys, json
out = sys.argv[1]
rows_out = {}
names = []
for vd in sorted((d for d in glob.glob(os.path.join(out, "v[0-9]*")) if os.path.isdir(d)), key=lambda p: int(os.path.basename(p)[1:])):
    var = open(os.path.join(vd, "VARIANT")).read().strip() or "(default)"
    names.append(var)
    for f in glob.glob(os.path.join(vd, "**", "*counter_collection.csv"), recursive=True):
        by = {}
        for r in csv.DictReader(open(f)):
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            v = v[1:] if len(v) > 1 else v
            rows_out.setdefault(k, {})[var] = sum(v) / len(v)
    for lf in glob.glob(vd + "_*.log"):
        for line in open(lf):
            if line.startswith("{"):
                rows_out.setdefault("kernel_ms (profiled)", {})[var] = json.loads(line)["kernel_ms"]
print("| counter | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
for k in sorted(rows_out):
    print("| " + k + " | " + " | ".join(f"{rows_out[k].get(n, float('nan')):.6g}" for n in names) + " |")
PY
cat $OUT/SUMMARY.md
