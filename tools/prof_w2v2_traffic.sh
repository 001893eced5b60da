# HBM traffic of the wav2vec2 bottleneck extractor: two separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) over tools/w2v2_only.py
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r03}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/wf -- python3 $R/tools/w2v2_only.py > $O/wf.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ww -- python3 $R/tools/w2v2_only.py > $O/ww.log 2>&1
cd $R
F=$(find $O/wf -name "*counter_collection.csv" | head -1); W=$(find $O/ww -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $F $W w2v2 7 > $R/gpurun_out/${TAG}_w2v2_traffic.json
rm -rf "$O/wf" "$O/ww"
tail -12 $R/gpurun_out/${TAG}_w2v2_traffic.json
