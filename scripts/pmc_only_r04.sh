# PMC passes only (a few minutes): MFMA busy and HBM traffic of the headline launch on the current build, both forms.
#   bash scripts/pmc_only_r04.sh <commit>
set -e
export TMPDIR=/tmp
C=${1:-unknown}
O=gpurun_out
mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-shapes --no-secondary"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/r04_pmc_fetch -- $B > /dev/null 2>> $O/r04_prof.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/r04_pmc_write -- $B > /dev/null 2>> $O/r04_prof.err
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r04_pmc_mfma -- $B > /dev/null 2>> $O/r04_prof.err
python scripts/pmc_summary.py traffic $(find $O/r04_pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/r04_pmc_write -name "*counter_collection.csv" | head -1) $C $O/r04_roofline_traffic.json > /dev/null
python scripts/pmc_summary.py mfma $(find $O/r04_pmc_mfma -name "*counter_collection.csv" | head -1) $C $O/r04_mfma_utilisation.json > /dev/null
rm -rf $O/r04_pmc_fetch $O/r04_pmc_write $O/r04_pmc_mfma
export GLOWK_CO_OFF=1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/r04_pmc_mfma_u -- $B > /dev/null 2>> $O/r04_prof.err
unset GLOWK_CO_OFF
python scripts/pmc_summary.py mfma $(find $O/r04_pmc_mfma_u -name "*counter_collection.csv" | head -1) $C $O/r04_mfma_utilisation_one_per_cu.json > /dev/null
rm -rf $O/r04_pmc_mfma_u
python - <<'PY'
import json
for f in ("r04_mfma_utilisation", "r04_mfma_utilisation_one_per_cu"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, {k: (v.get("kernel"), round(v.get("mfma_utilisation", 0), 4), round(v.get("kernel_cycles", 0))) for k, v in d["level0"].items()})
d = json.load(open("gpurun_out/r04_roofline_traffic.json"))
print("traffic", d["hbm_bytes_per_level0_launch"])
PY
echo "pmc done"
