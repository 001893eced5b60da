# HBM traffic of the generator: two separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) over tools/gen_only.py
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02k}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/gen_only.py > $O/f.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/gen_only.py > $O/w.log 2>&1
cd $R
F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W > $R/gpurun_out/${TAG}_generator_traffic.json
rm -rf "$O/f" "$O/w"
tail -12 $R/gpurun_out/${TAG}_generator_traffic.json
