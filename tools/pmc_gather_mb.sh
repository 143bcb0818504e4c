#!/bin/bash
# L2-side counters of the gather replay microbenchmark (diagnostic) -> gpurun_out/pmc_gather_mb.txt
export TMPDIR=/tmp
i=0
for set in "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum"; do
  i=$((i+1)); rm -rf /tmp/pmcg_$i
  timeout 200 rocprofv3 --kernel-include-regex "k_replay" --pmc $set --output-format csv -d /tmp/pmcg_$i -o p -- python3 tools/gather_ceiling2.py > gpurun_out/pmc_gather_mb_$i.log 2>&1
done
python3 - > gpurun_out/pmc_gather_mb.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcg_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0] + " g" + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    d = {c: acc[k][c] / cnt[k][c] for c in acc[k]}
    print(k); print("   ", {c: round(v) for c, v in sorted(d.items())})
PY
cat gpurun_out/pmc_gather_mb.txt
