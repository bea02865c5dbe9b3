"""Localise the > 2 GB decode discrepancy (tests/test_vae_gpu.py::test_vae_decode_outputs_beyond_2_gb): per sample and per image-row band,
B = 4 at 768 x 768 against single-image decodes, under the environment it is started with."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_vae_gpu import _model, make_vae_inputs, SD_VAE
m = _model('sd', 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lat = make_vae_inputs(SD_VAE, B, 96, 96, seed=21).cuda()
allb = m.decode(lat)
torch.cuda.synchronize()
for i in range(B):
    one = m.decode(lat[i:i + 1])
    d = (one - allb[i:i + 1]).abs()[0]          # [3][768][768]
    rows = d.amax(dim=(0, 2))
    bad = (rows > 2e-3).nonzero().flatten()
    print(f'sample {i}: max {d.max().item():.3e}  bad rows {bad.numel()}' + (f' first {bad[0].item()} last {bad[-1].item()}' if bad.numel() else ''), flush=True)
    if bad.numel():
        cols = d.amax(dim=(0, 1)); bc = (cols > 2e-3).nonzero().flatten()
        print(f'   bad cols {bc.numel()} first {bc[0].item()} last {bc[-1].item()}; per-channel max {[f"{x:.2e}" for x in d.amax(dim=(1, 2)).tolist()]}', flush=True)
