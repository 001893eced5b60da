# rocprofv3 kernel stats of the fbank-tag bottleneck extractor alone: bash tools/prof_getbn.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r06x}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd $R
python3 tools/getbn_only.py | tail -1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/getbn_only.py > $O/getbn.log 2>&1
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $R/gpurun_out/${TAG}_getbn_kernel_stats.csv
rm -rf "$O/stats"
tail -1 $O/getbn.log
