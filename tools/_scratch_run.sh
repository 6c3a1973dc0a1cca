cd /tmp && export TMPDIR=/tmp
rm -rf /root/repo/gpurun_out/prof_inf
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_inf -o inf --output-format csv -- python /root/repo/bench.py --child-inference-graph box > /root/repo/gpurun_out/r6m_inf.log 2>&1
tail -2 /root/repo/gpurun_out/r6m_inf.log | cut -c1-600
find /root/repo/gpurun_out/prof_inf -name "*kernel_stats.csv" | head
