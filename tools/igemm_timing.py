"""Per-workgroup phase timing of every generic GEMM launch of one UNet call.  Needs the instrumented library:
    SDMI_CXXFLAGS=-DSDMI_IGEMM_TIMING SDMI_LIB_OUT=libsdmi_timing.so python stable-diffusion_amd/build.py
    SDMI_LIB_PATH=stable-diffusion_amd/libsdmi_timing.so python tools/igemm_timing.py out.txt
(SDMI_IGEMM_TIMING: s_memtime stamps at kernel
entry, k-loop entry, k-loop exit, kernel exit of every workgroup; the library synchronises after each launch -- debug only)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'igemm_timing.txt')
if os.path.exists(out):
    os.remove(out)
os.environ['SDMI_IGEMM_TIMING'] = out
import torch  # noqa: E402
import bench  # noqa: E402
dev = torch.device('cuda:0')
ld, unet, vae = bench.build_gpu_model(dev)
g = torch.Generator().manual_seed(3)
x = torch.randn(2, 4, 64, 64, generator=g).to(dev)
ctx = (0.1 * torch.randn(2, 77, 768, generator=g)).to(dev)
t = torch.tensor([481, 481], device=dev)
unet(x, t, context=ctx)
torch.cuda.synchronize()
open(out, 'w').close()          # keep the second (warm) call only
unet(x, t, context=ctx)
torch.cuda.synchronize()
lines = open(out).read().splitlines()
print(len(lines), 'launches')
seen = set()
for l in lines:
    key = l.split('|')[0]
    if key not in seen:
        seen.add(key)
        print(l)
