#!/bin/bash
# VALU / SALU / memory instruction counts and busy cycles per kernel of the bench step (diagnostic: which kernels are
# bound by instruction issue).  usage (gpurun, repo root): tools/pmc_bench.sh  -> gpurun_out/pmc_bench.json
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp SBX_PERMUTE_OVERLAP=0 SBX_RCM_OVERLAP=0
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmcb_$i
  timeout 200 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcb_$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sharded > "$OUT/pmc_bench_$i.log" 2>&1
done
python3 - "$OUT/pmc_bench.json" <<'PY'
import csv, glob, json, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("/tmp/pmcb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "at::" in k or "rocprim" in k or "rocclr" in k: continue
        k = re.sub(r"\(.*", "", k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", ""))
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
out = {}
for k, d in acc.items():
    o = {c: v / cnt[k][c] for c, v in d.items()}
    o["launches"] = max(cnt[k].values())
    # VALU busy: a wave64 VALU instruction occupies its SIMD for 4 cycles; 1024 SIMDs
    if o.get("GRBM_GUI_ACTIVE") and o.get("SQ_INSTS_VALU"):
        o["valu_busy_frac"] = round(o["SQ_INSTS_VALU"] * 4 / (o["GRBM_GUI_ACTIVE"] * 1024), 3)
    out[k] = o
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
for k, o in sorted(out.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0) * kv[1]["launches"])[:22]:
    print(f"{k[:60]:60s} n={o['launches']:4d} cyc={o.get('GRBM_GUI_ACTIVE',0):10.0f} valu={o.get('SQ_INSTS_VALU',0):12.0f} salu={o.get('SQ_INSTS_SALU',0):11.0f} vmem={o.get('SQ_INSTS_VMEM_RD',0)+o.get('SQ_INSTS_VMEM_WR',0):10.0f} lds={o.get('SQ_INSTS_LDS',0):10.0f} valu_busy={o.get('valu_busy_frac')}")
PY
