import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
for f in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us', '/sys/fs/cgroup/cpu/cpu.cfs_period_us'):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, 'n/a')
print('loadavg', open('/proc/loadavg').read().strip())
import torch
from oracle import unet_ref
from oracle.plan import TINY, SMALL40
from oracle.weights import make_inputs, make_state_dict
sd = make_state_dict(SMALL40, 0)
x, t, ctx = make_inputs(SMALL40, 2, 32, 32, seed=1)
for th in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(th)
    unet_ref.unet_forward(sd, SMALL40, x, t, ctx)
    t0 = time.perf_counter(); unet_ref.unet_forward(sd, SMALL40, x, t, ctx); dt = time.perf_counter() - t0
    print(f'threads {th:4d}: {dt:.3f} s', flush=True)
