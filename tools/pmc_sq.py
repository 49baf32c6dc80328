#!/usr/bin/env python3
"""Per-launch averages of SQ / TCC counters of the x3 kernels from rocprofv3 --pmc passes (one CSV per pass):
    pmc_sq.py out.json pass1_counter_collection.csv [pass2 ...]"""
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

LABEL = {
    ("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true>", 256): "x3_half_step_vh_sample",
    ("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 1, true>", 224): "x3_half_step_hv_sample",
    ("k_gemm_pb<256, 64, 4, 2, 64, 3, 0, 0, true>", 256): "x3_half_step_vh_prob",
    ("k_gemm_pb<128, 128, 2, 4, 64, 3, 1, 0, true>", 224): "x3_stats_gemm",      # (rounds 1-2a: last argument = wave-specialised)
    ("k_gemm_pb<128, 128, 2, 4, 64, 3, 1, 0, false>", 224): "x3_stats_gemm",     # (since: last argument = A is a byte plane)
    ("k_gemm_pb<256, 64, 4, 2, 64, 3, 1, 0, false>", 256): "x3_stats_gemm",      # (256 x 64 statistics tiles)
    ("k_gemm_pb<256, 64, 4, 2, 64, 3, 1, 0, true>", 256): "x3_stats_gemm",       # (... with v_neg^T as a byte plane)
}
out = collections.defaultdict(dict)
for path in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void kurbm::", "").replace("kurbm::", "")
        name = norm(name)
        wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        lab = LABEL.get((name, wgs))
        if lab is None and "k_reduce_apply_split" in name:
            lab = "x3_reduce_apply"            # (the slab reduce + weight-piece mirror launch: no MFMA, HBM / latency bound)
        if lab:
            acc[(lab, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (lab, c), v in acc.items():
        out[lab][c] = sum(v) / len(v)
for lab, d in out.items():
    if d.get("TCC_REQ_sum"):
        d["l2_hit_rate"] = d.get("TCC_HIT_sum", 0.0) / d["TCC_REQ_sum"]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("SQ_WAVE_CYCLES"):
        d["wait_any_share"] = d.get("SQ_WAIT_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
        d["wait_inst_any_share"] = d.get("SQ_WAIT_INST_ANY", 0.0) / d["SQ_WAVE_CYCLES"]
    if d.get("SQ_BUSY_CU_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        # MFMA-pipe busy cycles summed over the SIMDs (= MFMA instructions x 16 for v_mfma_f32_16x16x32_bf16) over the
        # CUs' busy cycles x 4 SIMDs
        d["mfma_busy_share"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * d["SQ_BUSY_CU_CYCLES"])
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(v, 4) for c, v in d.items() if c.endswith("share") or c.endswith("rate")} for k, d in out.items()}))
