#!/bin/bash
# Round-4 GPU pass J: re-tune of ALL workloads on the final kernels, starting from pass I's table (unet64 + unet96 re-tuned there): A/B of the
# committed / pass-I / all-workload tables, the whole GPU suite + smoke + the three bench workloads with the all-workload table.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
P=${1:-j}
O=$PWD/gpurun_out; mkdir -p $O
L=$PWD/stable-diffusion_amd
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cp ${TUNE_I:-$L/tune_gfx950.txt} $O/${P}_tune_i.txt
cp ${TUNE_I:-$L/tune_gfx950.txt} $O/${P}_tune.txt
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 1200 python tools/tune.py --rounds 72 --reps 4 --out $O/${P}_tune.txt --dump $O/${P}_tune_dump.txt > $O/${P}_tune.log 2>&1; el "tune all workloads exit $? : $(tail -1 $O/${P}_tune.log)"
diff <(grep -v "^#" $O/${P}_tune_i.txt | cut -d" " -f1-10 | sort) <(grep -v "^#" $O/${P}_tune.txt | cut -d" " -f1-10 | sort) | grep -c "^>" | xargs echo "rows changed against pass I's table:"
for r in 1 2 3; do
  timeout 300 python tools/unet_latency.py "committed table" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_TUNE_FILE=$O/${P}_tune_i.txt timeout 300 python tools/unet_latency.py "pass-I table (unet64 + unet96)" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
  SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 300 python tools/unet_latency.py "all-workload table" 20 2 2>/dev/null | grep round >> $O/${P}_ab.txt
done
el "A/B 64x64"; cat $O/${P}_ab.txt
for w in txt2img512 txt2img768 img2img512; do
  for t in committed i all; do
    case $t in committed) TF=$L/tune_gfx950.txt;; i) TF=$O/${P}_tune_i.txt;; all) TF=$O/${P}_tune.txt;; esac
    SDMI_TUNE_FILE=$TF timeout 600 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-roofline > $O/${P}_bench_${w}_$t.log 2>&1
    el "bench $w table=$t: $(tail -1 $O/${P}_bench_${w}_$t.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],4), 'unet', round(d.get('unet_ms_per_call',0),3), 'vae', round(d.get('vae_decode_ms',0),3))")"
  done
done
SDMI_TUNE_FILE=$O/${P}_tune.txt timeout 1200 python -m pytest tests -x -q -m gpu -p no:cacheprovider -s > $O/${P}_pytest.log 2>&1; el "pytest -m gpu -x (all-workload table) exit $? : $(tail -1 $O/${P}_pytest.log)"
grep -h "\[unet \|headroom\|^FAILED" $O/${P}_pytest.log | sed 's/^\.*//' | cut -c1-150 | head -24
grep -h "max-abs" $O/${P}_pytest.log | grep -i "vae\|decode\|encode\|clip\|pipeline" | sed 's/^\.*//' | cut -c1-160 | head -30
el done
