#!/bin/bash
# Round-6 evidence for profiles/ (run through gpurun from the repo root): the bench command's kernel stats and traffic
# counters (tools/collect_profiles.sh), the permute kernels' counters for both orders (tools/pmc_permute.sh: the new tile
# kernel's instruction counts and LDS bank conflicts), serialised kernel traces, the RCM timeline, the ops table, the C4
# probe, the 64-bit probe, GrayReorder end to end (pinned staging) with the kernel traces and the stall probe that found
# what the 14 - 25 ms in front of its key stage were.
set -u
# (every step under its own timeout: a profiler run behind a faulting process once hung until gpurun's limit)
mkdir -p gpurun_out
timeout 900 tools/collect_profiles.sh r6 > gpurun_out/r6_collect.log 2>&1
timeout 900 tools/pmc_permute.sh --rcm > gpurun_out/r6_pmc_permute_rcm.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r6_pmc_permute_rcm.json
timeout 900 tools/pmc_permute.sh > gpurun_out/r6_pmc_permute_random.txt 2>&1; cp gpurun_out/pmc_permute.json gpurun_out/r6_pmc_permute_random.json
KT_N=40 timeout 900 tools/kt_permute.sh r6_rcm --rcm > /dev/null; cp gpurun_out/kt_r6_rcm.txt gpurun_out/r6_permute_kernels_rcm.txt
KT_N=40 timeout 900 tools/kt_permute.sh r6_random > /dev/null; cp gpurun_out/kt_r6_random.txt gpurun_out/r6_permute_kernels_random.txt
timeout 900 tools/permute_span.sh r6 --rcm > /dev/null; cp gpurun_out/permute_span_r6.txt gpurun_out/r6_permute_span_rcm.txt
timeout 900 tools/rcm_kt.sh r6 > /dev/null; cp gpurun_out/rcm_timeline_r6.txt gpurun_out/r6_rcm_timeline.txt
timeout 900 python tools/c4_probe.py > gpurun_out/r6_c4_probe.json 2> gpurun_out/r6_c4_probe.err
timeout 900 python tools/ops_table.py --gpu-only > gpurun_out/r6_ops_table.txt 2>&1
timeout 900 python tools/int64_probe.py > gpurun_out/r6_int64_probe.log 2>&1
timeout 900 python tools/gray_e2e_probe.py > gpurun_out/r6_gray_e2e.json 2> gpurun_out/r6_gray_e2e.err
timeout 900 tools/gray_kt.sh > gpurun_out/r6_gray_kt.log 2>&1
timeout 900 tools/gray_kt2.sh > gpurun_out/r6_gray_kt2_pinned.txt 2>&1
SBX_HOST_PINNED_STAGING=0 timeout 900 tools/gray_kt2.sh > gpurun_out/r6_gray_kt2_pageable.txt 2>&1
timeout 900 tools/gray_stall_probe.sh > gpurun_out/r6_gray_stall_probe.txt 2>&1
timeout 900 python3 tools/idle_latency_probe.py > gpurun_out/r6_idle_latency_probe.txt 2>&1
tail -1 gpurun_out/r6_rcm_timeline.txt; tail -c 300 gpurun_out/r6_c4_probe.json; tail -c 300 gpurun_out/r6_bench_line.json
