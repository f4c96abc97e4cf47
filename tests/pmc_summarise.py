#!/usr/bin/env python3
"""Aggregate tests/collect_pmc.sh output into one JSON: per dominant kernel, counters averaged over its FULL-GRID launches
(the wavefront's fill / drain launches have smaller grids).  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in
KB, and on gfx950 FETCH_SIZE counts wide streaming reads at half their bytes (x2 correction).
python tests/pmc_summarise.py gpurun_out/pmc profiles/<name>.json [T of the micro-benchmark]"""
import collections
import csv
import glob
import json
import re
import sys

root, out = sys.argv[1], sys.argv[2]
KEEP = ("lstm_step_fwd", "lstm_step_bwd", "lstm_bwd_epi", "gemm_tn_bf16", "lstm_persist_fwd", "lstm_persist_bwd")
T_MICRO = int(sys.argv[3]) if len(sys.argv) > 3 else 16          # time steps of the micro-benchmark: a persistent launch covers T + 3 diagonals
acc = collections.defaultdict(lambda: collections.defaultdict(list))     # kernel -> counter -> [(grid, value)]
for f in glob.glob(f"{root}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if not any(k in name for k in KEEP):
            continue
        short = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
        acc[short][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
res = {}
for k, cs in acc.items():
    d = {}
    for c, vals in cs.items():
        g = max(v[0] for v in vals)
        full = [v[1] for v in vals if v[0] == g]
        d[c] = round(sum(full) / len(full), 1)
        d["full_launches"] = len(full)
    if "FETCH_SIZE" in d:
        d["hbm_read_MB_corrected"] = round(2 * d["FETCH_SIZE"] / 1024, 1)
    if "WRITE_SIZE" in d:
        d["hbm_write_MB"] = round(d["WRITE_SIZE"] / 1024, 1)
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 3)
    if "SQ_WAVE_CYCLES" in d:
        d["wait_any_frac"] = round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 3)
        d["active_inst_frac"] = round(d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"], 3)
    if "lstm_persist" in k:
        d["diagonals_per_launch"] = T_MICRO + 3
    res[k] = d
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
