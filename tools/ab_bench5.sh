# the headline bench (four jobs in flight) with the ring conv's launches capped at fewer blocks than CUs: do the small kernels of the
# other job streams gain more from the CUs left to them than the ring launches lose?   bash tools/ab_bench5.sh
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() {
  local label=$1; shift
  env "$@" python3 bench.py --headline-only --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', d['value'], d['ms_per_step'], 'generator alone', d['roofline']['timing_ms']['median'])"
}
for rep in 1 2; do
  run "all CUs   " SATOOLS_AMD_CONV_OPTIONS=convring_blocks=0
  run "248 blocks" SATOOLS_AMD_CONV_OPTIONS=convring_blocks=248
  run "240 blocks" SATOOLS_AMD_CONV_OPTIONS=convring_blocks=240
  run "224 blocks" SATOOLS_AMD_CONV_OPTIONS=convring_blocks=224
  run "192 blocks" SATOOLS_AMD_CONV_OPTIONS=convring_blocks=192
done
