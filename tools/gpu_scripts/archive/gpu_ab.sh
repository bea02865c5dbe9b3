#!/bin/bash
# A/B on ONE box: bench.py under several env settings (box-to-box variation is +-4 %, so only same-box numbers compare).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
i=0
for envs in "${@}"; do
  i=$((i+1))
  env $envs timeout 600 python bench.py --steps ${BENCH_STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab_$i.log 2>&1
  python - "$envs" gpurun_out/ab_$i.log <<'PY'
import json, sys
l = [x for x in open(sys.argv[2]) if x.startswith('{')]
if not l:
    print('==', sys.argv[1], 'FAILED'); sys.exit(0)
d = json.loads(l[-1])
print('==', sys.argv[1], 'images/s %.4f  unet ms %.3f  vae ms %.3f' % (d['value'], d['unet_ms_per_call'], d.get('vae_decode_ms', 0)))
if 'roofline' in d:
    print('   ' + '  '.join('%s:%d/%.2f' % (r['name'].replace('igemm_', 'g').replace('splitk_', 'sk_'), r['launches'], r['ms']) for r in d['roofline']['per_class'][:11]))
PY
done
