#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel-group HBM traffic.

MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB-units of 1024 B as reported by
rocprofv3; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced
streaming read (16 B/lane), so it is doubled for the streaming kernels; other access widths
are uncalibrated (both raw and doubled figures are kept; `hbm_bytes_per_launch` uses the
doubled read side, i.e. an upper bound for the scattered kernels).
usage: summarize_pmc.py <fetch_dir> <write_dir> <out.json>
"""
import csv, glob, json, sys, collections

GROUPS = [("k_bfs_bottom_up", "bfs_bottom_up"), ("k_ubfs_bottom_up", "bfs_bottom_up"), ("k_ubfs_collect", "level_order"),
          ("k_fresh_words", "level_order"), ("k_keys_from_fresh", "level_order"), ("k_visited_from_ppos", "level_order"),
          ("k_ubfs_", "rcm_misc"), ("k_classify_scan", "permute_prep"),
          ("k_bfs_expand_heavy", "bfs_heavy"), ("k_bfs_expand", "bfs_expand"),
          ("k_bfs_small_levels", "bfs_small_levels"), ("k_permute_tile", "permute_tile"), ("k_permute_copy", "permute_tile"),
          ("k_permute_block_rows", "permute_block"), ("k_rows_quad", "permute_block"), ("k_permute_rows_radix", "permute_block"), ("k_long_", "permute_long"),
          ("k_rowwise_prep", "permute_prep"), ("k_rec_classify", "permute_prep"), ("k_tile_first", "permute_prep"),
          ("k_onesweep_pass", "radix_scatter"), ("k_onesweep_hist", "radix_hist"), ("k_onesweep_bins", "radix_hist"),
          ("k_scan_", "scan"), ("k_cc_", "cc"), ("k_classify", "cc"), ("k_level_", "level_order"),
          ("k_coo_to_csr", "coo_to_csr"), ("k_csr_to_coo", "csr_to_coo"), ("k_gray", "gray"), ("k_degree", "degree")]

def group_of(name):
    for pat, g in GROUPS:
        if pat in name:
            return g
    return None

def load(d, counter):
    tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            g = group_of(r["Kernel_Name"])
            if g is None:
                continue
            tot[g] += float(r["Counter_Value"]); cnt[g] += 1
    return tot, cnt

ft, fc = load(sys.argv[1], "FETCH_SIZE")
wt, wc = load(sys.argv[2], "WRITE_SIZE")
out = {}
for g in sorted(set(ft) | set(wt)):
    launches = max(fc.get(g, 0), wc.get(g, 0)) or 1
    fetch = ft.get(g, 0.0) * 1024 / max(fc.get(g, 1), 1)
    write = wt.get(g, 0.0) * 1024 / max(wc.get(g, 1), 1)
    out[g] = {"launches_profiled": launches, "fetch_bytes_raw_per_launch": fetch, "write_bytes_per_launch": write,
              "hbm_bytes_per_launch": 2 * fetch + write,
              "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM (exact for 16 B/lane streaming reads; upper bound otherwise)"}
# which collection this is: date, and a digest of the kernel sources the profiled library was built from (the GPU box has
# no .git; bench.py quotes this beside roofline.traffic)
import datetime, hashlib, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hsh = hashlib.sha256()
csrc = os.path.join(root, "sparsebase_amd", "csrc")
for f in sorted(os.listdir(csrc)):
    hsh.update(open(os.path.join(csrc, f), "rb").read())
out["_meta"] = {"collected": datetime.date.today().isoformat(), "build": "csrc-sha256:" + hsh.hexdigest()[:12]}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
