# every rocprofv3 summary a round commits under profiles/: bash tools/prof_round.sh <tag>   (writes gpurun_out/<tag>_*)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r04}
bash $R/tools/prof_gen_pmc.sh $TAG > /dev/null 2>&1        # <tag>_generator_only_kernel_stats.csv, <tag>_generator_mfma_util.json
bash $R/tools/prof_gen_traffic.sh $TAG > /dev/null 2>&1    # <tag>_generator_traffic.json
bash $R/tools/prof_w2v2.sh $TAG > /dev/null 2>&1           # <tag>_w2v2_extractor_kernel_stats.csv, <tag>_w2v2_extractor_mfma_util.json
bash $R/tools/prof_w2v2_traffic.sh $TAG > /dev/null 2>&1   # <tag>_w2v2_traffic.json
bash $R/tools/prof_bench.sh $TAG > /dev/null 2>&1          # <tag>_bench_headline_kernel_stats.csv
ls -la $R/gpurun_out/${TAG}_* 
