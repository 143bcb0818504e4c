#!/bin/bash
# Hardware counters of the Gray short-row kernel on the banded C5 instances (diagnostic): two rocprofv3 --pmc passes
# over tools/c5_probe.py.  usage (gpurun, repo root): tools/pmc_gray.sh  -> gpurun_out/pmc_gray.json
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp C5_ONLY_BANDED=1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pmcg_$i
  timeout 150 rocprofv3 --kernel-include-regex "k_gray_rows_short" --pmc $set --output-format csv -d /tmp/pmcg_$i -o p -- python3 tools/c5_probe.py > "$OUT/pmc_gray_$i.log" 2>&1
done
python3 - "$OUT/pmc_gray.json" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("/tmp/pmcg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gray_rows_short" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
out = {c: v / cnt[c] for c, v in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
