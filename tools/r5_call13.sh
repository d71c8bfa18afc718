#!/bin/bash
O=gpurun_out/r5m; mkdir -p $O
{
for ch in 2 4 16; do python tools/stream_dbg2.py uniform_u32 radix 1024 150000 $ch 2>&1 | tail -2; done
for ch in 2 16; do python tools/stream_dbg2.py dups_u32 radix 1024 150000 $ch 2>&1 | tail -2; done
python tools/stream_dbg2.py uniform_u64 linear 1024 150000 8 2>&1 | tail -2
python tools/stream_dbg2.py dups_u64 linear 4096 1500000 16 2>&1 | tail -2
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt | cut -c1-220
