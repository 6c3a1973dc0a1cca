#!/bin/bash
# forward tuning iteration: parity tests (roi_align), timing sweep, PMC group 1+2 for ring=192
TAG=${1:-i}; R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "roi_align" > $R/gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest exit $?" >> $R/gpurun_out/${TAG}_pytest.log
: > $R/gpurun_out/${TAG}_sweep.log
for ring in 192 256; do
  MI_ROI_ALIGN_RING=$ring timeout 300 python bench.py --only-roofline >> $R/gpurun_out/${TAG}_sweep.log 2>&1
done
cd /tmp; export TMPDIR=/tmp; export MI_ROI_ALIGN_RING=192
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc$i -o p -- python $R/tools/run_one_kernel.py roi_align_fwd 5 > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
done
tail -3 $R/gpurun_out/${TAG}_pytest.log; cat $R/gpurun_out/${TAG}_sweep.log
