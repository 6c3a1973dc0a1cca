cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
ROUND=r06d bash tools/gpu_profiles.sh poolcrop > gpurun_out/r06d_poolcrop.log 2>&1
cat gpurun_out/prof_r06d/pool_crop_bwd_kernel_durations.csv
grep -E "FETCH_SIZE|WRITE_SIZE|INSTS_VALU|INSTS_SALU|SQ_WAVES |THREAD_CYCLES" gpurun_out/prof_r06d/pmc_roi_pool_bwd.txt | grep bwd_tiles
cut -c1-400 gpurun_out/prof_r06d/pool_crop_time.json
