# rocprofv3 kernel stats of 13 generator forwards (tools/gen_only.py) -> gpurun_out/<tag>_generator_only_kernel_stats.csv
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02h}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gen_only.py > $O/gen_stats.log 2>&1
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $R/gpurun_out/${TAG}_generator_only_kernel_stats.csv
rm -rf "$O/stats"
tail -1 $O/gen_stats.log
