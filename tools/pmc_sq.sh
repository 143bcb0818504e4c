#!/bin/bash
# SQ counters of the permute row / tile kernels (diagnostic). usage: tools/pmc_sq.sh <tag> [--rcm] -> gpurun_out/pmc_sq_<tag>.txt
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1)); rm -rf /tmp/pmcr_$i
  timeout 200 rocprofv3 --kernel-include-regex "k_permute_tile|k_permute_block" --pmc $set --output-format csv -d /tmp/pmcr_$i -o p -- python3 tools/permute_only.py "$@" > "$OUT/pmc_sq_$i.log" 2>&1
done
python3 - > "$OUT/pmc_sq_$TAG.txt" <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcr_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0]
        k += " g" + r.get("Grid_Size", r.get("Grid_Size_X", "?")) + " w" + r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    d = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    print(k)
    print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
cat "$OUT/pmc_sq_$TAG.txt"
