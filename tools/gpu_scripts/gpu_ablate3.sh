#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/ablate3; mkdir -p $O
for ab in 0 1 2 3 4 5 6 7; do
  SDMI_CONV3GN_ABLATE=$ab rocprofv3 --kernel-trace --stats -d $O/a$ab -o t -- python tools/prof_conv3gn.py > $O/a$ab.log 2>&1
done
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob('gpurun_out/ablate3/*/t_results.db')):
    con=sqlite3.connect(f)
    for name,calls,tot,avg,pct in con.execute("select * from top_kernels"):
        if 'conv3gn' in name: print(f.split('/')[-2], f'{avg:.1f} us  (bits: 1 no weight loads, 2 no MFMA, 4 no input staging)')
PY
