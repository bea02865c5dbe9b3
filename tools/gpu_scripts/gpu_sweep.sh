#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for w in 256 384 512 768; do
  echo "== SDMI_SPLIT_WANT=$w"; SDMI_SPLIT_WANT=$w python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_ms_per_call'])"
done
echo "== precise 1x1 off"; SDMI_PRECISE_1X1=0 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_ms_per_call'])"
