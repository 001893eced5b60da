# headline-only bench lines with the VQ near-tie guard off / on, plain convert() ("sync") and deferred status, interleaved on one box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
line() { python -c "import sys,json; o=json.loads(sys.stdin.read()); print('$1', o['value'], o['ms_per_step'], o['repeats']['ms_per_step_windows'])"; }
for i in 1 2; do
  SATOOLS_AMD_VQ_TIE_SIGMAS=0 python bench.py --headline-only --no-cpu-baseline --f0-status sync 2>/dev/null | line "sync     guard off        :"
  SATOOLS_AMD_VQ_TIE_SIGMAS=4 python bench.py --headline-only --no-cpu-baseline --f0-status sync 2>/dev/null | line "sync     guard 4 sigma    :"
  SATOOLS_AMD_VQ_TIE_SIGMAS=3 python bench.py --headline-only --no-cpu-baseline --f0-status sync 2>/dev/null | line "sync     guard 3 sigma    :"
  SATOOLS_AMD_VQ_TIE_SIGMAS=4 SATOOLS_AMD_VQ_TIE_STREAM_PRIORITY=0 python bench.py --headline-only --no-cpu-baseline --f0-status sync 2>/dev/null | line "sync     guard 4 s, prio 0:"
  SATOOLS_AMD_VQ_TIE_SIGMAS=0 python bench.py --headline-only --no-cpu-baseline --f0-status deferred 2>/dev/null | line "deferred guard off        :"
  SATOOLS_AMD_VQ_TIE_SIGMAS=4 python bench.py --headline-only --no-cpu-baseline --f0-status deferred 2>/dev/null | line "deferred guard 4 sigma    :"
done
