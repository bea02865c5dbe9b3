#!/bin/bash
# Round 6, pass L: key-split attention with the second key group rotated (SDMI_ATTN_ROT=1): bit-identity, kernel time + SQ counters, UNet A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6l}
O=$PWD/gpurun_out; mkdir -p $O/${P}_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -s -k "attention_key_split" > $O/${P}_attn.log 2>&1; el "attention tests exit $? : $(tail -1 $O/${P}_attn.log)"
grep -a "Error\|assert" $O/${P}_attn.log | head -5
for rot in 0 1; do
  SDMI_ATTN_ROT=$rot timeout 300 rocprofv3 --kernel-trace --stats -d $O/${P}_pmc/kt$rot -o kt -- python tools/attn_one.py > $O/${P}_pmc/kt$rot.log 2>&1
  SDMI_ATTN_ROT=$rot timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/${P}_pmc/sq$rot -o pmc -- python tools/attn_one.py > $O/${P}_pmc/sq$rot.log 2>&1
done; el "pmc exit $?"
python - "$P" <<'PY' | tee gpurun_out/r6l_attn_pmc.txt
import sqlite3, glob, sys
P = sys.argv[1]
O = f'gpurun_out/{P}_pmc'
for rot in (0, 1):
    dur = {}
    for f in glob.glob(f'{O}/kt{rot}/**/*_results.db', recursive=True):
        for name, calls, total, avg, pct in sqlite3.connect(f).execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if 'attn' in name: dur[name] = (calls, avg * 1e3)
    ctr = {}
    for f in glob.glob(f'{O}/sq{rot}/**/*_results.db', recursive=True):
        for k, cn, v, n in sqlite3.connect(f).execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            if 'attn' in k: ctr[cn] = v / max(n, 1)
    for name, (calls, avg) in dur.items():
        wc = max(ctr.get('SQ_WAVE_CYCLES', 1), 1)
        print(f'SDMI_ATTN_ROT={rot}  {name[60:110]:50s} n={calls} avg {avg/1e3:7.2f} us | wave cycles: wait {ctr.get("SQ_WAIT_ANY",0)/wc:.3f} stall {ctr.get("SQ_WAIT_INST_ANY",0)/wc:.3f} '
              f'active {ctr.get("SQ_ACTIVE_INST_ANY",0)/wc:.3f} | MFMA busy {ctr.get("SQ_VALU_MFMA_BUSY_CYCLES",0) / (1024 * avg * 1e-9 * 2.4e9):.3f} of (1024 SIMDs x duration x 2.4 GHz)')
PY
timeout 900 python tools/unet_ab.py SDMI_ATTN_ROT=0 SDMI_ATTN_ROT=1 --rounds 5 > $O/${P}_ab.log 2>&1; el "ab exit $?"; tail -2 $O/${P}_ab.log
rm -rf $O/${P}_pmc
el done
