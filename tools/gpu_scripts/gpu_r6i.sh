#!/bin/bash
# Round 6, pass I: attention d = 40 with the key range split over two 8-wave groups of one 16-wave workgroup (attn_dma_kernel KVS = 2): parity,
# stand-alone kernel time and SQ counters with / without it, UNet call A/B at 64x64 and 96x96.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-r6i}
O=$PWD/gpurun_out; mkdir -p $O/${P}_pmc
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -p no:cacheprovider -s -k "attention_key_split or test_attention" > $O/${P}_attn.log 2>&1; el "attention tests exit $? : $(tail -1 $O/${P}_attn.log)"
grep -a "key split\|one group" $O/${P}_attn.log | cut -c1-160
for kvs in 0 1; do
  SDMI_ATTN_KVS=$kvs timeout 300 rocprofv3 --kernel-trace --stats -d $O/${P}_pmc/kt$kvs -o kt -- python tools/attn_one.py > $O/${P}_pmc/kt$kvs.log 2>&1
  SDMI_ATTN_KVS=$kvs timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -d $O/${P}_pmc/sq$kvs -o pmc -- python tools/attn_one.py > $O/${P}_pmc/sq$kvs.log 2>&1
  SDMI_ATTN_KVS=$kvs timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $O/${P}_pmc/sqb$kvs -o pmc -- python tools/attn_one.py > $O/${P}_pmc/sqb$kvs.log 2>&1
done; el "pmc exit $?"
python - "$P" <<'PY' | tee gpurun_out/r6i_attn_pmc.txt
import sqlite3, glob, sys
P = sys.argv[1]
O = f'gpurun_out/{P}_pmc'
for kvs in (0, 1):
    dur = {}
    for f in glob.glob(f'{O}/kt{kvs}/**/*_results.db', recursive=True):
        for name, calls, total, avg, pct in sqlite3.connect(f).execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if 'attn' in name: dur[name] = (calls, avg * 1e3)
    ctr = {}
    for sub in ('sq', 'sqb'):
        for f in glob.glob(f'{O}/{sub}{kvs}/**/*_results.db', recursive=True):
            for k, cn, v, n in sqlite3.connect(f).execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
                if 'attn' in k: ctr[cn] = v / max(n, 1)
    for name, (calls, avg) in dur.items():
        wc = max(ctr.get('SQ_WAVE_CYCLES', 1), 1)
        mf24 = ctr.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * avg * 1e-9 * 2.4e9)
        print(f'SDMI_ATTN_KVS={kvs}  {name[:70]:70s} n={calls} avg {avg/1e3:7.2f} us | wave cycles: wait {ctr.get("SQ_WAIT_ANY",0)/wc:.3f} stall {ctr.get("SQ_WAIT_INST_ANY",0)/wc:.3f} '
              f'active {ctr.get("SQ_ACTIVE_INST_ANY",0)/wc:.3f} (VALU {ctr.get("SQ_ACTIVE_INST_VALU",0)/wc:.3f} LDS {ctr.get("SQ_ACTIVE_INST_LDS",0)/wc:.3f}) '
              f'wait-on-LDS {ctr.get("SQ_WAIT_INST_LDS",0)/wc:.3f} bank conflicts {ctr.get("SQ_LDS_BANK_CONFLICT",0):.0f} | MFMA busy {mf24:.3f} of (1024 SIMDs x duration x 2.4 GHz), '
              f'{ctr.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/max(ctr.get("SQ_BUSY_CYCLES",1),1):.3f} x SQ_BUSY_CYCLES')
PY
timeout 900 python tools/unet_ab.py SDMI_ATTN_KVS=0 SDMI_ATTN_KVS=1 --rounds 5 > $O/${P}_ab64.log 2>&1; el "ab 64 exit $?"; tail -2 $O/${P}_ab64.log
timeout 900 python tools/unet_ab.py SDMI_ATTN_KVS=0 SDMI_ATTN_KVS=1 --rounds 3 --latent 96 --iters 10 > $O/${P}_ab96.log 2>&1; el "ab 96 exit $?"; tail -2 $O/${P}_ab96.log
SDMI_ATTN_KVS=1 timeout 900 python -m pytest tests/test_unet_gpu.py -q -m gpu -p no:cacheprovider -s -k "golden or headroom" > $O/${P}_unet.log 2>&1; el "unet goldens (KVS=1) exit $? : $(tail -1 $O/${P}_unet.log)"
grep -a "^.\?\[unet sdv1_64\|^.\?\[unet sdv1_96\|^.\?\[unet sdv1_w._64\|^.\?\[unet sdv1_real_64\|headroom" $O/${P}_unet.log | cut -c1-200
rm -rf $O/${P}_pmc
el done
