#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/ablate; mkdir -p $O
for ab in 0 1 2 3; do
  for shape in l0conv l2conv big; do
    SDMI_IGEMM_ABLATE=$ab rocprofv3 --kernel-trace --stats -d $O/a${ab}_$shape -o t -- python tools/prof_igemm.py --shape $shape --tiles 0,1,2,3,5 --iters 10 > $O/a${ab}_$shape.log 2>&1
  done
done
python - <<'PY'
import sqlite3, glob
for f in sorted(glob.glob('gpurun_out/ablate/*/t_results.db')):
    con=sqlite3.connect(f)
    print(f.split('/')[-2])
    for name,calls,tot,avg,pct in con.execute("select * from top_kernels"):
        if 'igemm' in name: print('   ', name.split('igemm_kernel<')[1].split('>')[0], f'{avg:.1f} us')
PY
