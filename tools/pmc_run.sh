# HBM traffic of the decode step from PMC counters: two separate rocprofv3 --pmc passes (no trace domains with --pmc)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/tools/bench_llm.py 1 40 255 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -- python3 $GRAFT_REPO_ROOT/tools/bench_llm.py 1 40 255 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_decode.py gpurun_out/pmc_fetch FETCH_SIZE
python tools/pmc_decode.py gpurun_out/pmc_write WRITE_SIZE
