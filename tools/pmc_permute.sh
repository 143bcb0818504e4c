#!/bin/bash
# Hardware counters of the permute kernels (diagnostic): several rocprofv3 --pmc passes over tools/permute_only.py,
# summed per kernel.  usage (gpurun, repo root): tools/pmc_permute.sh [--rcm]  -> gpurun_out/pmc_permute.json
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1)); rm -rf /tmp/pmcp_$i
  # (a TA_* counter set aborted rocprofv3 and hung its finalisation for the whole time limit: every pass is bounded)
  timeout 150 rocprofv3 --kernel-include-regex "k_permute|k_rows_quad|k_short|k_long|k_rowwise|k_rec_" --pmc $set --output-format csv -d /tmp/pmcp_$i -o p -- python3 tools/permute_only.py "$@" > "$OUT/pmc_permute_$i.log" 2>&1
done
python3 - "$OUT/pmc_permute.json" <<'PY'
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_permute" not in k and "k_rows_quad" not in k and "k_short" not in k and "k_long" not in k and "k_rowwise" not in k and "k_rec_" not in k: continue
        k = k.replace("void (anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {k: {c: v / cnt[k][c] for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, d in sorted(out.items()):
    print(k); print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
