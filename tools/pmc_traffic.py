"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `tools/gen_only.py` (generator forwards only)
into HBM traffic per generator forward (batch of 32).  Units and corrections per MI355X_MICROARCH.md §HBM:
counters are KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads, so
the read side is reported raw and doubled (upper bound).  Usage: python tools/pmc_traffic.py FETCH.csv WRITE.csv"""
import csv
import json
import re
import sys


def load(path, counter):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void ", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)
        d = per.setdefault(name, [0, 0.0])
        d[0] += 1
        d[1] += float(r["Counter_Value"])
    return per


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    w2v2 = len(sys.argv) > 3 and sys.argv[3] == "w2v2"
    if w2v2:
        return main_w2v2(fetch, write, int(sys.argv[4]) if len(sys.argv) > 4 else 5)
    # every kernel of the run that is not torch's (`at::`: the load-time fold / pack of the weights) or the runtime's: tools/gen_only.py
    # launches nothing but generator forwards.  (Rounds 1-5 kept a list of name fragments here, which `pairw_kernel` and
    # `planes_f8_sidecar_kernel` matched none of: the round-4 / round-5 totals lack their 1.5-2.5 GB.)
    gen = lambda n: not (n.startswith("at::") or "rocclr" in n or n.startswith("void at::"))
    # generator forwards in that run = launches of the output stage
    n_post = max(1, sum(v[0] for k, v in fetch.items() if "convpost_kernel" in k))
    out = {"generator_forwards_in_run": n_post, "kernels": {}}
    tot_f = tot_w = 0.0
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0, 0.0]), write.get(name, [0, 0.0])
        if not gen(name):
            continue
        out["kernels"][name] = {"launches": f[0], "fetch_GB": round(f[1] * 1024 / 1e9, 3), "write_GB": round(w[1] * 1024 / 1e9, 3)}
        tot_f += f[1] * 1024
        tot_w += w[1] * 1024
    out["per_forward"] = {"fetch_GB_raw": round(tot_f / n_post / 1e9, 3), "fetch_GB_doubled": round(2 * tot_f / n_post / 1e9, 3),
                          "write_GB": round(tot_w / n_post / 1e9, 3),
                          "traffic_GB_raw": round((tot_f + tot_w) / n_post / 1e9, 3),
                          "traffic_GB_fetch_doubled": round((2 * tot_f + tot_w) / n_post / 1e9, 3),
                          "algorithmic_GB_per_layer_model": 23.405}
    print(json.dumps(out, indent=1))


def main_w2v2(fetch, write, n_fwd):
    """every launch of `tools/w2v2_only.py` (get_bn of a batch of 32 = the wav2vec2 extractor + TDNNF tail + VQ), per forward"""
    out = {"forwards_in_run": n_fwd, "kernels": {}}
    tot_f = tot_w = 0.0
    for name in sorted(set(fetch) | set(write)):
        if name.startswith("at::") or "rocclr" in name:
            continue                                   # load-time packing / fills of the process, not the path
        f, w = fetch.get(name, [0, 0.0]), write.get(name, [0, 0.0])
        out["kernels"][name] = {"launches": f[0], "fetch_GB": round(f[1] * 1024 / 1e9, 3), "write_GB": round(w[1] * 1024 / 1e9, 3)}
        tot_f += f[1] * 1024
        tot_w += w[1] * 1024
    out["per_forward"] = {"fetch_GB_raw": round(tot_f / n_fwd / 1e9, 3), "fetch_GB_doubled": round(2 * tot_f / n_fwd / 1e9, 3),
                          "write_GB": round(tot_w / n_fwd / 1e9, 3),
                          "traffic_GB_fetch_doubled": round((2 * tot_f + tot_w) / n_fwd / 1e9, 3),
                          "algorithmic_GB_per_layer_model": round((625e6 * 32 + 1.26e9) / 1e9, 3)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
