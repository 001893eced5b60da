# launch-by-launch view of one wav2vec2-tag get_bn (tools/trace_forward.py over a rocprofv3 kernel trace of tools/w2v2_only.py): bash tools/prof_trace_w2v2.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r04u}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd $R
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/w2v2_only.py > $O/run.log 2>&1
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_forward.py $T "layernorm_ch_kernel<16, 32, true>" > $R/gpurun_out/${TAG}_w2v2_forward_launches.txt
rm -rf "$O/trace"
cat $R/gpurun_out/${TAG}_w2v2_forward_launches.txt
