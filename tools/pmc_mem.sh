#!/bin/bash
# memory-pipeline counters (TA / TCP / UTCL1 / TCC / TD) of the permute kernels (diagnostic). usage: tools/pmc_mem.sh <tag> [--rcm]
set -u
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0
i=0
for set in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
           "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_GATE_EN1_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "GRBM_GUI_ACTIVE TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1)); rm -rf /tmp/pmcm_$i
  timeout 200 rocprofv3 --kernel-include-regex "k_rows|k_permute_tile|k_permute_block|k_tile" --pmc $set --output-format csv -d /tmp/pmcm_$i -o p -- python3 tools/permute_only.py "$@" > "$OUT/pmc_mem_$i.log" 2>&1
done
python3 - > "$OUT/pmc_mem_$TAG.txt" <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcm_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0]
        k += " g" + r.get("Grid_Size", "?") + " w" + r.get("Workgroup_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    d = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    print(k)
    print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
cat "$OUT/pmc_mem_$TAG.txt"
