# rocprofv3 PMC passes over tools/bench_convring_var.py (VARS=0): MFMA busy share, wave stall shares and instruction-class activity of
# conv1d_f16x3_ring16_kernel next to the register-staged tile -> gpurun_out/<tag>_convring_pmc.txt
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r04c}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
export VARS=0
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/pmc1 -- python3 $R/tools/bench_convring_var.py > $O/pmc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc2 -- python3 $R/tools/bench_convring_var.py > $O/pmc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/pmc3 -- python3 $R/tools/bench_convring_var.py > $O/pmc3.log 2>&1 || true
cd $R
python3 - "$O" > $R/gpurun_out/${TAG}_convring_pmc.txt <<'PY'
import collections, csv, glob, re, sys
O = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(float)
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True):
        first = None
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["Kernel_Name"])).replace("sat::", "")
            if "conv1d_f16x3" not in n: continue
            key = (n[:60], int(r["Grid_Size"]), int(r.get("LDS_Block_Size", 0) or 0))
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            first = first or r["Counter_Name"]
            if d == "pmc1" and r["Counter_Name"] == first:
                cnt[key] += 1; dur[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k in sorted(per, key=lambda k: -dur[k]):
    v = per[k]
    if not cnt[k]: continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    wc = max(v.get("SQ_WAVE_CYCLES", 0) / 2, 1)   # (collected in two passes)
    print(f"{k[0]:60s} grid {k[1]:7d} n {cnt[k]:4d} avg {dur[k] / cnt[k] / 1e3:7.1f} us clock {cyc / dur[k]:4.2f} GHz mfma_busy {v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc:5.3f}"
          f" mops/launch {v['SQ_INSTS_VALU_MFMA_MOPS_F16'] / cnt[k]:.3e}"
          f" | wait_any {v['SQ_WAIT_ANY'] / wc:4.2f} wait_inst {v['SQ_WAIT_INST_ANY'] / wc:4.2f} wait_inst_lds {v['SQ_WAIT_INST_LDS'] / wc:4.2f}"
          f" act_lds {v['SQ_ACTIVE_INST_LDS'] / wc:4.2f} act_vmem {v['SQ_ACTIVE_INST_VMEM'] / wc:4.2f} act_sca {v['SQ_ACTIVE_INST_SCA'] / wc:4.2f} act_valu {v['SQ_ACTIVE_INST_VALU'] / wc:4.2f}"
          f" | insts/launch valu {v['SQ_INSTS_VALU'] / cnt[k]:.3e} salu {v['SQ_INSTS_SALU'] / cnt[k]:.3e} lds {v['SQ_INSTS_LDS'] / cnt[k]:.3e} vmem_rd {v['SQ_INSTS_VMEM_RD'] / cnt[k]:.3e} coexec {v['SQ_VALU_MFMA_COEXEC_CYCLES'] / 1024 / cyc:5.3f}")
PY
rm -rf "$O/pmc1" "$O/pmc2" "$O/pmc3"
cat $R/gpurun_out/${TAG}_convring_pmc.txt
