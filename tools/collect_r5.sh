#!/bin/bash
# Round-5 evidence for profiles/ (run through gpurun from the repo root): the bench command's kernel stats and traffic
# counters (tools/collect_profiles.sh), the permute kernels' counters for both orders (tools/pmc_permute.sh), serialised
# kernel traces, the row-class ablation with its sort-only leg, the RCM and sort timelines, the ops table, the C4 probe.
set -u
mkdir -p gpurun_out
tools/collect_profiles.sh r5 > gpurun_out/r5_collect.log 2>&1
tools/pmc_permute.sh --rcm > gpurun_out/r5_pmc_permute_rcm.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r5_pmc_permute_rcm.json
tools/pmc_permute.sh > gpurun_out/r5_pmc_permute_random.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r5_pmc_permute_random.json
KT_N=40 tools/kt_permute.sh r5_rcm --rcm > /dev/null; cp gpurun_out/kt_r5_rcm.txt gpurun_out/r5_permute_kernels_rcm.txt
KT_N=40 tools/kt_permute.sh r5_random > /dev/null; cp gpurun_out/kt_r5_random.txt gpurun_out/r5_permute_kernels_random.txt
tools/kt_ablate.sh > gpurun_out/r5_rows_ablation_random.txt 2>&1
tools/kt_ablate.sh --rcm > gpurun_out/r5_rows_ablation_rcm.txt 2>&1
tools/permute_span.sh r5 --rcm > /dev/null; cp gpurun_out/permute_span_r5.txt gpurun_out/r5_permute_span_rcm.txt
tools/rcm_kt.sh r5 > /dev/null; cp gpurun_out/rcm_timeline_r5.txt gpurun_out/r5_rcm_timeline.txt
export TMPDIR=/tmp; rm -rf /tmp/c2b_kt
COO_PROBE_ONLY=c2b rocprofv3 --kernel-trace --output-format csv -d /tmp/c2b_kt -o kt -- python3 tools/coo_sort_probe.py > /dev/null 2>&1
python3 tools/timeline.py /tmp/c2b_kt k_coo_is_sorted --all > gpurun_out/r5_sort_timeline_c2b.txt 2>&1
python tools/c4_probe.py > gpurun_out/r5_c4_probe.json 2> gpurun_out/r5_c4_probe.err
python tools/ops_table.py --gpu-only > gpurun_out/r5_ops_table.txt 2>&1
python tools/int64_probe.py > gpurun_out/r5_int64_probe.log 2>&1
python tools/gray_e2e_probe.py > gpurun_out/r5_gray_e2e.json 2> gpurun_out/r5_gray_e2e.err
tail -1 gpurun_out/r5_rcm_timeline.txt; tail -c 400 gpurun_out/r5_c4_probe.json; tail -c 300 gpurun_out/r5_bench_line.json
