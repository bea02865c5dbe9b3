#!/bin/bash
# Round 6, pass M: attention with one barrier per two key tiles (SDMI_ATTN_TPB=2): bit-identity, kernel time + SQ counters, UNet A/B at 64x64 and 96x96.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6m}
O=$PWD/gpurun_out; mkdir -p $O/${P}_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -s -k "attention_key_split or test_attention" > $O/${P}_attn.log 2>&1; el "attention tests exit $? : $(tail -1 $O/${P}_attn.log)"
grep -a "Error\|assert " $O/${P}_attn.log | head -5
for tpb in 1 2; do
  SDMI_ATTN_TPB=$tpb timeout 300 rocprofv3 --kernel-trace --stats -d $O/${P}_pmc/kt$tpb -o kt -- python tools/attn_one.py > $O/${P}_pmc/kt$tpb.log 2>&1
  SDMI_ATTN_TPB=$tpb timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/${P}_pmc/sq$tpb -o pmc -- python tools/attn_one.py > $O/${P}_pmc/sq$tpb.log 2>&1
done; el "pmc exit $?"
python - "$P" <<'PY' | tee gpurun_out/r6m_attn_pmc.txt
import sqlite3, glob, sys
P = sys.argv[1]
O = f'gpurun_out/{P}_pmc'
for tpb in (1, 2):
    dur = {}
    for f in glob.glob(f'{O}/kt{tpb}/**/*_results.db', recursive=True):
        for name, calls, total, avg, pct in sqlite3.connect(f).execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if 'attn' in name: dur[name] = (calls, avg * 1e3)
    ctr = {}
    for f in glob.glob(f'{O}/sq{tpb}/**/*_results.db', recursive=True):
        for k, cn, v, n in sqlite3.connect(f).execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
            if 'attn' in k: ctr[cn] = v / max(n, 1)
    for name, (calls, avg) in dur.items():
        wc = max(ctr.get('SQ_WAVE_CYCLES', 1), 1)
        print(f'SDMI_ATTN_TPB={tpb}  {name[60:118]:58s} n={calls} avg {avg/1e3:7.2f} us | wave cycles: wait {ctr.get("SQ_WAIT_ANY",0)/wc:.3f} stall {ctr.get("SQ_WAIT_INST_ANY",0)/wc:.3f} '
              f'active {ctr.get("SQ_ACTIVE_INST_ANY",0)/wc:.3f} | MFMA busy {ctr.get("SQ_VALU_MFMA_BUSY_CYCLES",0) / (1024 * avg * 1e-9 * 2.4e9):.3f} of (1024 SIMDs x duration x 2.4 GHz)')
PY
timeout 900 python tools/unet_ab.py SDMI_ATTN_TPB=1 SDMI_ATTN_TPB=2 --rounds 5 > $O/${P}_ab64.log 2>&1; el "ab64 exit $?"; tail -2 $O/${P}_ab64.log
timeout 900 python tools/unet_ab.py SDMI_ATTN_TPB=1 SDMI_ATTN_TPB=2 --rounds 3 --latent 96 --iters 10 > $O/${P}_ab96.log 2>&1; el "ab96 exit $?"; tail -2 $O/${P}_ab96.log
rm -rf $O/${P}_pmc
el done
