cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/dev/time_train.py nearhover 1048576 > gpurun_out/r03_time_train2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_train256 -- python3 tools/dev/time_coop.py > gpurun_out/r03_prof_train256.log 2>&1
tail -12 gpurun_out/r03_time_train2.log
find gpurun_out/prof_r03_train256 -name "*kernel_stats.csv" | head -2
