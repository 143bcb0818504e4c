#!/bin/bash
# Hardware counters of the Gray kernels (diagnostic): two rocprofv3 --pmc passes over a probe script.
# usage (gpurun, repo root): tools/pmc_gray.sh [kernel-regex] [probe.py]  -> gpurun_out/pmc_gray.json
#   defaults: the short-row kernel on the banded C5 instances (tools/c5_probe.py);
#   tools/pmc_gray.sh "k_gray_rows" tools/gray_once.py = the power-law path on the RMAT bench matrix
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp C5_ONLY_BANDED=1
RE=${1:-k_gray_rows_short}; PROBE=${2:-tools/c5_probe.py}
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS_ATOMIC" \
           "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmcg_$i
  timeout 150 rocprofv3 --kernel-include-regex "$RE" --pmc $set --output-format csv -d /tmp/pmcg_$i -o p -- python3 $PROBE > "$OUT/pmc_gray_$i.log" 2>&1
done
python3 - "$OUT/pmc_gray.json" "$RE" <<'PY'
import csv, glob, json, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not re.search(sys.argv[2], k): continue
        k = re.search(r"k_gray_\w+", k).group(0)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
