# HBM traffic of the fbank-tag bottleneck extractor (tools/getbn_only.py: 23 get_bn calls): two separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE),
# once as configured and once with SATOOLS_AMD_TDNNF_PLANES_ONLY=0:  bash tools/prof_getbn_traffic.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r06}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
for v in 1 0; do
  export SATOOLS_AMD_TDNNF_PLANES_ONLY=$v
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gf$v -- python3 $R/tools/getbn_only.py > $O/gf$v.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/gw$v -- python3 $R/tools/getbn_only.py > $O/gw$v.log 2>&1
done
cd $R
python3 - "$O" > $R/gpurun_out/${TAG}_getbn_traffic.json <<'PY'
import csv, glob, json, re, sys
O = sys.argv[1]
out = {"calls_in_run": 23, "unit": "MB per get_bn (batch 32 x 5 s); FETCH_SIZE raw (KiB) and doubled (MI355X_MICROARCH.md: wide streaming reads are tallied at half), WRITE_SIZE raw"}
for v in ("1", "0"):
    tot = {}
    for kind, d in (("FETCH_SIZE", "gf"), ("WRITE_SIZE", "gw")):
        f = glob.glob(f"{O}/{d}{v}/**/*counter_collection.csv", recursive=True)[0]
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != kind:
                continue
            name = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"]))
            if name.startswith("at::") or "rocclr" in name:
                continue
            per[name] = per.get(name, 0.0) + float(r["Counter_Value"])
        tot[kind] = per
    names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda n: -(tot["FETCH_SIZE"].get(n, 0) * 2 + tot["WRITE_SIZE"].get(n, 0)))
    rows = [{"kernel": n[:70], "fetch_MB_x2": round(tot["FETCH_SIZE"].get(n, 0) * 2 * 1024 / 1e6 / 23, 1), "write_MB": round(tot["WRITE_SIZE"].get(n, 0) * 1024 / 1e6 / 23, 1)} for n in names[:8]]
    out["planes_only=" + v] = {"fetch_MB_x2": round(sum(tot["FETCH_SIZE"].values()) * 2 * 1024 / 1e6 / 23, 1),
                               "write_MB": round(sum(tot["WRITE_SIZE"].values()) * 1024 / 1e6 / 23, 1), "kernels": rows}
print(json.dumps(out, indent=1))
PY
rm -rf $O/gf1 $O/gw1 $O/gf0 $O/gw0
head -c 1500 $R/gpurun_out/${TAG}_getbn_traffic.json
