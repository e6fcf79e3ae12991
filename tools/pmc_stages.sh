# PMC evidence for one stage: bash tools/pmc_stages.sh {flow|hift|decode} [reps]  ->  gpurun_out/r6_pmc_<stage>.json
# (program directly after `--`; counters in passes of their own; no trace domains beside --pmc)
STAGE=$1; REPS=${2:-3}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in trace mfma fetch write; do rm -rf $R/gpurun_out/pmc_${STAGE}_$d; done
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${STAGE}_trace -- python3 $R/tools/prof_stage_run.py $STAGE $REPS > $R/gpurun_out/pmc_${STAGE}.log 2>&1
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_${STAGE}_mfma -- python3 $R/tools/prof_stage_run.py $STAGE $REPS >> $R/gpurun_out/pmc_${STAGE}.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${STAGE}_fetch -- python3 $R/tools/prof_stage_run.py $STAGE $REPS >> $R/gpurun_out/pmc_${STAGE}.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${STAGE}_write -- python3 $R/tools/prof_stage_run.py $STAGE $REPS >> $R/gpurun_out/pmc_${STAGE}.log 2>&1
cd $R
python tools/pmc_stage.py $STAGE $REPS gpurun_out/r6_pmc_${STAGE}.json gpurun_out/pmc_${STAGE}_trace gpurun_out/pmc_${STAGE}_mfma gpurun_out/pmc_${STAGE}_fetch gpurun_out/pmc_${STAGE}_write
find gpurun_out/pmc_${STAGE}_trace gpurun_out/pmc_${STAGE}_mfma gpurun_out/pmc_${STAGE}_fetch gpurun_out/pmc_${STAGE}_write -name '*.csv' -size +4M -delete
