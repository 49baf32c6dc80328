#!/usr/bin/env python3
"""HBM bytes per launch of the x3 kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected
separately, as MI355X_MICROARCH.md prescribes):  pmc_traffic.py fetch_counter_collection.csv write_counter_collection.csv out.json"""
import collections
import csv
import json
import re
import sys


def norm(name):
    """k_gemm_pb<BM, BN, WM, WN, BK, PB, EPI, NOISE, AB[, RP]>: the labels below were written before the RP argument existed."""
    m = re.match(r"(k_gemm_pb<)([^>]*)(>)", name)
    if not m:
        return name
    args = [a.strip() for a in m.group(2).split(",")]
    return m.group(1) + ", ".join(args[:9]) + m.group(3) if len(args) == 10 and args[9] == "false" else name


def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void kurbm::", "").replace("kurbm::", "")
        name = norm(name)
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        acc[(name, wgs)].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
label = {}
# (the last template argument: round 1 / early round 2 "wave-specialised", since then "A operand is a byte plane")
for ws in ("", ", true", ", false"):
    label[("k_gemm_pb<128, 128, 2, 4, 64, 3, 0, 1%s>" % ws, 256)] = "x3_half_step_vh_sample"
    label[("k_gemm_pb<128, 128, 2, 4, 64, 3, 0, 1%s>" % ws, 224)] = "x3_half_step_hv_sample"
    label[("k_gemm_pb<128, 128, 2, 4, 64, 3, 0, 0%s>" % ws, 256)] = "x3_half_step_vh_prob"
    label[("k_gemm_pb<128, 128, 2, 4, 64, 3, 1, 0%s>" % ws, 224)] = "x3_stats_gemm"
# half steps on 256 x 64 tiles (tall_ok)
label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true>", 256)] = "x3_half_step_vh_sample"
label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true>", 224)] = "x3_half_step_hv_sample"
label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 0, true>", 256)] = "x3_half_step_vh_prob"
label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 1, 0, false>", 256)] = "x3_stats_gemm"
label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 1, 0, true>", 256)] = "x3_stats_gemm"      # (... with v_neg^T as a byte plane)     # (256 x 64 statistics tiles)
for nz, wgs, name in ((1, 256, "x3_half_step_vh_sample"), (1, 224, "x3_half_step_hv_sample"), (0, 256, "x3_half_step_vh_prob")):
    label[("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, %d, false>" % nz, wgs)] = name      # (real-valued A operand: bf16 planes)
out = {}
for key in sorted(set(fetch) | set(write)):
    name = label.get(key)
    if name is None and "reduce_apply" in key[0]:
        name = "x3_reduce_apply"
    if name is None and "f32_to_bf16" in key[0]:
        name = "x3_f32_to_bf16_wgs%d" % key[1]
    if name is None:
        continue
    f, w = fetch.get(key, 0.0), write.get(key, 0.0)
    out[name] = {"rocprof_kernel": key[0], "workgroups": key[1], "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                 "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                 "note": "FETCH_SIZE doubled (gfx950 counts 64 B per 128-B request on wide coalesced reads, "
                         "MI355X_MICROARCH.md HBM section); WRITE_SIZE as read"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
