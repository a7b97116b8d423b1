set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2p; mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 tools/pmc_traffic.py c4 64 > $O/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 tools/pmc_traffic.py c4 64 > $O/pmc_write.log 2>&1 || exit 4
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_sq --output-format csv -- python3 tools/pmc_traffic.py c4 64 > $O/pmc_sq.log 2>&1 || exit 5
python3 tools/pmc_to_json.py $O/pmc_fetch $O/pmc_write $O/pmc_sq c4 64 $O/r2_pmc_traffic.json $O/r2_pmc_mfma_busy.json || exit 6
