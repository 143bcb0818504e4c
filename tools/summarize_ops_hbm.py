#!/usr/bin/env python3
"""Per-kernel achieved HBM bandwidth of the streaming kernels: average duration from the rocprofv3 kernel-trace
stats, bytes per launch from separate FETCH_SIZE / WRITE_SIZE counter passes (units of 1 KiB; FETCH_SIZE doubled on
gfx950 as MI355X_MICROARCH.md prescribes — exact for wide streaming reads, an upper bound otherwise).
usage: summarize_ops_hbm.py <kernel_stats.csv> <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, sys
# (first match wins: longer names in front of their prefixes)
KERNELS = ["k_coo_to_csr", "k_csr_to_coo", "k_permute_copy", "k_permute_tile_radix", "k_permute_tile",
           "k_permute_block_rows", "k_rows_quad", "k_permute_rows_radix", "k_long_seg_gather", "k_long_seg_partition", "k_gray_tile",
           "k_gray_rows_short", "k_onesweep_pass", "k_bandwidth_csr", "k_profile_csr", "k_tile_spans",
           "k_degrees", "k_rowwise_prep", "k_classify_scan"]

def key_of(name):
    for k in KERNELS:
        if k in name:
            return k
    return None

stats = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = key_of(r["Name"])
    if k:
        s = stats.setdefault(k, [0, 0])
        s[0] += int(r["Calls"]); s[1] += int(r["TotalDurationNs"])

def load(d, counter):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = key_of(r["Kernel_Name"])
                if k:
                    tot[k] += float(r["Counter_Value"]); cnt[k] += 1
    return tot, cnt

ft, fc = load(sys.argv[2], "FETCH_SIZE")
wt, wc = load(sys.argv[3], "WRITE_SIZE")
out = {}
for k, (calls, ns) in sorted(stats.items()):
    avg_us = ns / calls / 1e3
    fetch = 2 * ft.get(k, 0.0) * 1024 / max(fc.get(k, 1), 1)
    write = wt.get(k, 0.0) * 1024 / max(wc.get(k, 1), 1)
    out[k] = {"launches": calls, "avg_us": round(avg_us, 2), "hbm_read_mb_per_launch": round(fetch / 1e6, 2),
              "hbm_write_mb_per_launch": round(write / 1e6, 2),
              "achieved_hbm_gbs": round((fetch + write) / (avg_us * 1e-6) / 1e9, 1),
              "frac_of_8tbs": round((fetch + write) / (avg_us * 1e-6) / 1e9 / 8000, 4)}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
