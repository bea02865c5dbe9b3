"""Time the attention kernel on one shape under the kernel variants (SDMI_ATTN_V1, waves per workgroup) and, with
--ablate, the timing-only ablations of attn.hip (SDMI_ATTN_ABL): one subprocess per setting, HIP events around 20 launches."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
import kernels as K
d, heads, B, nq, nkv = %d, %d, %d, %d, %d
g = torch.Generator().manual_seed(1)
BH = B * heads
q = torch.randn((BH, nq, d), generator=g).half().cuda(); k = torch.randn((BH, nkv, d), generator=g).half().cuda()
vt = torch.randn((BH, d, (nkv + 7) // 8 * 8), generator=g).half().cuda()
for _ in range(5): K.attention(q, k, vt, heads, nkv, d ** -0.5)
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.attention(q, k, vt, heads, nkv, d ** -0.5)
e1.record(); torch.cuda.synchronize()
print('%%.1f' %% (e0.elapsed_time(e1) * 1000 / 20))
'''


def run(env, shape):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', CHILD % ((ROOT, ROOT) + shape)], env=e, capture_output=True, text=True)
    return r.stdout.strip() or r.stderr.strip()[-300:]


if __name__ == '__main__':
    shapes = [(40, 8, 2, 4096, 4096), (80, 8, 2, 1024, 1024), (40, 8, 2, 4096, 77)]
    settings = [('dma ring (default)', {}), ('register staged (v1)', {'SDMI_ATTN_V1': '1'}),
                ('dma ring, 4 waves / workgroup', {'SDMI_ATTN_NW_GT1K': '4', 'SDMI_ATTN_NW_LE1K': '4'}),
                ('dma ring, 8 waves / workgroup', {'SDMI_ATTN_NW_GT1K': '8', 'SDMI_ATTN_NW_LE1K': '8'})]
    if '--ablate' in sys.argv:
        settings += [(f'dma ring, ablation {a} ({n})', {'SDMI_ATTN_ABL': str(a)})
                     for a, n in ((1, 'no wait+barrier'), (2, 'no DMA'), (3, 'no exp'), (4, 'no PV MFMA'), (5, 'no QK MFMA'), (6, 'LDS tile 0 only'))]
    for sh in shapes:
        print('shape d=%d heads=%d B=%d nq=%d nkv=%d' % sh)
        for name, env in settings:
            if 'ablation' in name and sh[0] != 40:
                continue
            print('  %-44s %s us' % (name, run(env, sh)))
