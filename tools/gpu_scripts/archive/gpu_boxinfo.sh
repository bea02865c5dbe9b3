#!/bin/bash
# What kind of box is this?  (the pool's boxes differ by +-12 % on one binary: clocks / CU count / partition mode next to one UNet latency)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out; mkdir -p $O
{
/opt/rocm/bin/rocminfo 2>/dev/null | grep -i "Marketing Name\|Compute Unit\|Max Clock Freq\|Wavefront Size\|Cacheline\|L2:\|L3:\|Chip ID\|ASIC Revision" | sort | uniq -c | head -20
rocm-smi --showclocks --showperflevel --showpower --showmemuse --showcomputepartition --showmemorypartition --showtemp 2>/dev/null | grep -v "^=\|^$" | head -30
timeout 300 python tools/unet_latency.py "this box" 20 3 2>/dev/null | grep round
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|fclk\|mclk\|Power" | head
} > $O/boxinfo.txt 2>&1
cat $O/boxinfo.txt
