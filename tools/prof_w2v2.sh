# rocprofv3 kernel stats + the two PMC passes tools/pmc_mfma.py summarises over tools/w2v2_only.py: bash tools/prof_w2v2.sh <tag>
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r02g}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/w2v2_only.py > $O/stats.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 $R/tools/w2v2_only.py > $O/pmc1.log 2>&1
timeout 500 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc2 -- python3 $R/tools/w2v2_only.py > $O/pmc2.log 2>&1
cd $R
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp $S $R/gpurun_out/${TAG}_w2v2_extractor_kernel_stats.csv
P1=$(find $O/pmc1 -name "*counter_collection.csv" | head -1); P2=$(find $O/pmc2 -name "*counter_collection.csv" | head -1)
python3 tools/pmc_mfma.py $P1 $P2 7 > $R/gpurun_out/${TAG}_w2v2_extractor_mfma_util.json 2> $O/pmc_mfma.err
rm -rf "$O/stats" "$O/pmc1" "$O/pmc2"
tail -2 $O/stats.log; head -c 600 $R/gpurun_out/${TAG}_w2v2_extractor_mfma_util.json
