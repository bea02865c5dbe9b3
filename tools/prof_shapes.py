"""In-situ per-shape timing of every igemm launch of one UNet call (HIP events, SDMI_PROF_SHAPES=1)."""
import json, os, sys
os.environ['SDMI_PROF_SHAPES'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
ld, unet, vae = bench.build_gpu_model(torch.device('cuda'))
tab = bench.profile_unet(unet, torch.device('cuda'))
tab.sort(key=lambda r: -r['ms'])
tot = sum(r['ms'] for r in tab)
print(f'total {tot:.3f} ms over {sum(r["launches"] for r in tab)} launches')
for r in tab:
    tf = r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['flops'] else 0
    print(f'{r["name"]:58s} n={r["launches"]:3d} total {r["ms"]*1e3:8.1f} us  avg {r["ms"]*1e3/r["launches"]:7.1f} us  {tf:6.1f} TF/s')
