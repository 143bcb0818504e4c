#!/bin/bash
# Dynamic instruction counts of k_permute_tile by phase: SBX_DEBUG_TILE_STOP=k leaves the kernel behind phase k
# (1 row map, 2 loads + gathers, 3 bucket words, 4 counts + scans, 5 placement + ranking, 0 everything), one rocprofv3
# --pmc pass per counter set and stop.  usage (gpurun, repo root): tools/tile_phase_pmc.sh [--rcm] -> gpurun_out/tile_phase_pmc.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0 SBX_PROBE_LIB=${SBX_PROBE_LIB:-tuning}  # SBX_DEBUG_TILE_STOP: tuning build only
: > "$OUT/tile_phase_pmc.txt"
for stop in 1 2 3 4 5 0; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/tpp_${stop}_$i
    SBX_DEBUG_TILE_STOP=$stop timeout 150 rocprofv3 --kernel-include-regex "k_permute_tile<" --pmc $set --output-format csv -d /tmp/tpp_${stop}_$i -o p -- python3 tools/permute_only.py "$@" > /dev/null 2>&1
  done
  python3 - $stop >> "$OUT/tile_phase_pmc.txt" <<'PY'
import csv, glob, sys, collections
stop = sys.argv[1]
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(f"/tmp/tpp_{stop}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_permute_tile<" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print("stop", stop, {c: round(v / cnt[c] / 1e6, 2) for c, v in sorted(acc.items())})
PY
done
cat "$OUT/tile_phase_pmc.txt"
