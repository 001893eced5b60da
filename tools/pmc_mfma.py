"""Summarise rocprofv3 --pmc passes over `tools/gen_only.py` into MFMA utilisation and wave-stall shares per generator
kernel and stage (the stage is told by the grid size).  Pass 1: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F16; pass 2: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM (each with --kernel-trace; kernels run serialised under --pmc).
  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)   (MI355X_MICROARCH.md: the busy
  counter counts cycles, 32 per 32x32x16 f16 MFMA, summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
Usage: python tools/pmc_mfma.py MFMA_counter_collection.csv WAIT_counter_collection.csv [forwards_in_run]"""
import collections
import csv
import json
import re
import sys


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, dur = collections.Counter(), collections.defaultdict(float)
    first = None
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sat::", "")
        key = (n, int(r["Grid_Size"]))
        per[key][r["Counter_Name"]] += float(r["Counter_Value"])
        first = first or r["Counter_Name"]
        if r["Counter_Name"] == first:
            cnt[key] += 1
            dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return per, cnt, dur


# (clock_GHz = GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches under 0.3 ms: reported as null there)
def main():
    a, cnt, dur = load(sys.argv[1])
    b, _, _ = load(sys.argv[2])
    fw = int(sys.argv[3]) if len(sys.argv) > 3 else 13
    rows, tot_busy, tot_cyc, tot_ms = [], 0.0, 0.0, 0.0
    for k in sorted(a, key=lambda k: -dur[k]):
        v, w = a[k], b.get(k, {})
        if not any(s in k[0] for s in ("conv1d_", "resblock_pair", "convpost", "act_split", "gemm_f16x3", "attention", "layernorm", "w2v2_", "mrf16", "pairw", "pair32s", "ups2")):
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        wc = max(w.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append({"kernel": k[0], "grid": k[1], "launches_per_forward": round(cnt[k] / fw, 1),
                     "avg_us": round(dur[k] / cnt[k] / 1e3, 1), "ms_per_forward": round(dur[k] / fw / 1e6, 3),
                     "clock_GHz": (round(cyc / dur[k], 2) if dur[k] / cnt[k] >= 3e5 else None), "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 3),
                     "wait_any": round(w.get("SQ_WAIT_ANY", 0) / wc, 2), "wait_inst_any": round(w.get("SQ_WAIT_INST_ANY", 0) / wc, 2),
                     "wait_inst_lds": round(w.get("SQ_WAIT_INST_LDS", 0) / wc, 2),
                     "lds_bank_conflict_share": round(w.get("SQ_LDS_BANK_CONFLICT", 0) / max(w.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3)})
        tot_busy += v["SQ_VALU_MFMA_BUSY_CYCLES"]
        tot_cyc += cyc
        tot_ms += dur[k] / fw / 1e6
    print(json.dumps({"forwards_in_run": fw, "generator_ms_per_forward_serialised": round(tot_ms, 3),
                      "mfma_util_all_generator_kernels": round(tot_busy / 1024 / tot_cyc, 3), "kernels": rows}, indent=1))


if __name__ == "__main__":
    main()
