#!/bin/bash
# Hardware counters of the kernels matching a regex, summed per kernel over several rocprofv3 --pmc passes.
# usage (gpurun, repo root): tools/pmc_kernel.sh <kernel regex> <python script> [args]  -> gpurun_out/pmc_kernel.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
RE=$1; shift
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_ATOMIC_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmck_$i
  timeout 200 rocprofv3 --kernel-include-regex "$RE" --pmc $set --output-format csv -d /tmp/pmck_$i -o p -- python3 "$@" > "$OUT/pmc_kernel_$i.log" 2>&1
done
python3 - "$OUT/pmc_kernel.txt" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmck_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
with open(sys.argv[1], "w") as o:
    for k, d in sorted(acc.items()):
        o.write(k + "  (launches %d)\n" % max(cnt[k].values()))
        for c, v in sorted(d.items()):
            o.write("    %-36s total %16.0f   per launch %14.0f\n" % (c, v, v / cnt[k][c]))
print(open(sys.argv[1]).read())
PY
