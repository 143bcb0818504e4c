#!/bin/bash
# Round-4 evidence for profiles/ (run through gpurun from the repo root): the bench command's kernel stats and traffic
# counters (tools/collect_profiles.sh), the permute kernels' instruction / wait / LDS / TCP counters for both orders
# (tools/pmc_permute.sh), serialised kernel traces, the row-class ablation and the memory-side replays.
set -u
mkdir -p gpurun_out
tools/collect_profiles.sh r4 > gpurun_out/r4_collect.log 2>&1
tools/pmc_permute.sh --rcm > gpurun_out/r4_pmc_permute_rcm.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r4_pmc_permute_rcm.json
tools/pmc_permute.sh > gpurun_out/r4_pmc_permute_random.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r4_pmc_permute_random.json
KT_N=40 tools/kt_permute.sh r4_rcm --rcm > /dev/null
KT_N=40 tools/kt_permute.sh r4_random > /dev/null
SBX_PERMUTE_C512_QUAD=1 tools/kt_ablate.sh > gpurun_out/r4_rows_ablation_random.txt 2>&1
SBX_PERMUTE_C512_QUAD=1 tools/kt_ablate.sh --rcm > gpurun_out/r4_rows_ablation_rcm.txt 2>&1
python tools/gather_ceiling2.py > gpurun_out/r4_gather_ceiling.log 2>&1
python tools/replay_policy.py > gpurun_out/r4_replay_policy.log 2>&1
./tools/rowperm_bench > gpurun_out/r4_rowperm_bench.log 2>&1
./tools/buffer_oob_test > gpurun_out/r4_buffer_oob.log 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rcm_kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/rcm_kt -o kt -- python3 $GRAFT_REPO_ROOT/tools/rcm_trace.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python3 tools/rcm_timeline.py /tmp/rcm_kt --all > gpurun_out/r4_rcm_timeline.txt 2>&1
python tools/c4_probe.py > gpurun_out/r4_c4_probe.json 2> gpurun_out/r4_c4_probe.err
python tools/ops_table.py --gpu-only > gpurun_out/r4_ops_table.txt 2>&1
tail -3 gpurun_out/r4_rcm_timeline.txt; tail -c 400 gpurun_out/r4_c4_probe.json; tail -c 300 gpurun_out/r4_bench_line.json
