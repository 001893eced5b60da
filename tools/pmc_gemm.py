"""1x1 convs on split planes (wav2vec2 Linear shapes, 32 utterances x 249 frames) through the 128 x 128 GEMM kernel
(k1_gemm = 1) and the LDS-DMA ring GEMM (k1_gemm = 2), for rocprofv3 --kernel-trace --pmc passes:
  pass 1  SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
  pass 2  SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
`python tools/pmc_gemm.py summarise <pass1.csv> <pass2.csv>` prints MFMA utilisation, clock and wave-stall shares per
(kernel, grid)."""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    sys.path.insert(0, ROOT)
    import torch
    from satools_amd import _lib, ops, packing
    B, T = 32, 249
    for cin, cout in ((1024, 4096), (4096, 1024), (1024, 1024)):
        x = torch.randn(B, cin, T, device="cuda")
        w = torch.randn(cout, cin, 1, device="cuda") / cin ** 0.5
        wp = packing.pack_conv_weight_f16x3(w)
        xs = ops.act_split(x, 1.0)
        for opt in (1, 2):
            _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "set_option")
            for _ in range(6):
                ops.conv1d(x, wp, cout, 1, mode=1, x_split=xs)
            torch.cuda.synchronize()


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


def summarise(p1, p2):
    a, cnt, dur = load(p1)
    b, _, _ = load(p2)
    rows = []
    for k in sorted(a, key=lambda k: (k[0], k[1])):
        if not ("k1_kernel" in k[0] or "ring" in k[0]):
            continue
        v, w = a[k], b.get(k, {})
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        wc = max(w.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        rows.append({"kernel": k[0], "grid": k[1], "launches": cnt[k], "avg_us": round(dur[k] / cnt[k] / 1e3, 1),
                     "clock_GHz": (round(cyc / dur[k], 2) if dur[k] / cnt[k] >= 3e5 else None), "mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc, 3),
                     "wait_any": round(w.get("SQ_WAIT_ANY", 0) / wc, 2), "wait_inst_any": round(w.get("SQ_WAIT_INST_ANY", 0) / wc, 2),
                     "wait_inst_lds": round(w.get("SQ_WAIT_INST_LDS", 0) / wc, 2),
                     "lds_active_share": round(w.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3),
                     "lds_bank_conflict_share": round(w.get("SQ_LDS_BANK_CONFLICT", 0) / max(w.get("SQ_LDS_IDX_ACTIVE", 0), 1), 3)})
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarise":
        summarise(sys.argv[2], sys.argv[3])
    else:
        run()
