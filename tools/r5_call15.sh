#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
timeout -k 5 800 python -m pytest tests/test_gpu_regs.py tests/test_gpu_lanes.py tests/test_gpu_parity.py -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp; export TMPDIR=/tmp
RMI_CFG_TRACE=1 RMI_CFG_BW=0 timeout -k 5 200 rocprofv3 --kernel-trace --stats -T -d $OLDPWD/$O/kt -o kt -f csv -- python $OLDPWD/tools/cfg_run.py M - 50 > $OLDPWD/$O/kt.log 2>&1
cd $OLDPWD
python tools/summarize_prof.py $O/kt | head -12
{
TAG=new python tools/cfg_run.py M
TAG=new python tools/cfg_run.py C3
TAG=new python tools/cfg_run.py C4s
} > $O/times.log 2>&1
grep -v "^  File\|^Traceback\|amdgpu.ids\|^    " $O/times.log
