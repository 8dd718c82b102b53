#!/bin/bash
# Runs on the GPU box: PMC passes (fabric bytes, L2 hit rate, atomics) of tools/run_one.py for a list of GNNA_TUNE
# variants.  Usage: tools/pmc_variants.sh <tag> "<config> <D>" "VARIANT1" "VARIANT2" ...   (a variant is a GNNA_TUNE string)
# -> gpurun_out/pmcv_<tag>/SUMMARY.md (one table row per variant and counter, per-dispatch means of the main kernel)
TAG=$1; shift
WL=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmcv_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for V in "$@"; do
  i=$((i+1))
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum"; do
    name=$(echo $set | tr ' ' '+' | cut -c1-30)
    GNNA_TUNE="$V" rocprofv3 --pmc $set --kernel-include-regex "stream_kernel|sweep_kernel" -T -d $OUT/v$i/$name -o pmc -f csv -- python $R/tools/run_one.py $WL > $OUT/v${i}_$name.log 2>&1
  done
  echo "$V" > $OUT/v$i/VARIANT
done
cd $R
python - "$OUT" <<'PY' > $OUT/SUMMARY.md
import csv, glob, os, sys, json
out = sys.argv[1]
print("| variant (GNNA_TUNE) | kernel ms | FETCH GB (x2 calibrated) | WRITE GB | L2 hit | L2 req M | EA RD M | EA WR M | EA ATOMIC M |")
print("|---|---|---|---|---|---|---|---|---|")
for vd in sorted((d for d in glob.glob(os.path.join(out, "v[0-9]*")) if os.path.isdir(d)),
                 key=lambda p: int(os.path.basename(p)[1:])):
    var = open(os.path.join(vd, "VARIANT")).read().strip()
    vals = {}
    for f in glob.glob(os.path.join(vd, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        by = {}
        for r in rows:
            by.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in by.items():
            v = v[1:] if len(v) > 1 else v          # drop the warm-up dispatch
            vals[k] = sum(v) / len(v)
    ms = "?"
    for lf in glob.glob(vd + "_*.log"):
        for line in open(lf):
            if line.startswith("{"):
                ms = json.loads(line)["kernel_ms"]
    g = lambda k: vals.get(k, float("nan"))
    hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if "TCC_HIT_sum" in vals else float("nan")
    print(f"| {var or '(default)'} | {ms} | {2 * g('FETCH_SIZE') * 1024 / 1e9:.3f} | {g('WRITE_SIZE') * 1024 / 1e9:.3f} | {hit:.3f} | "
          f"{g('TCC_REQ_sum') / 1e6:.1f} | {g('TCC_EA0_RDREQ_sum') / 1e6:.1f} | {g('TCC_EA0_WRREQ_sum') / 1e6:.1f} | {g('TCC_EA0_ATOMIC_sum') / 1e6:.1f} |")
PY
cat $OUT/SUMMARY.md
