import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import kernels as K
g = torch.Generator().manual_seed(0)
B, H, Cin, N, ks = 2, 64, 320, 320, 3
x = torch.randn(B * H * H, Cin, generator=g).half().cuda()
w = (torch.randn(N, ks * ks * Cin, generator=g) / math.sqrt(ks * ks * Cin)).half().cuda()
out = torch.empty(B * H * H, N, device='cuda')
print('tensors ok', flush=True)
for sk in (1, 0):
    for tile in (-1, 2, 3):
        print('launch splitk', sk, 'tile', tile, flush=True)
        K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f32=out, splitk=sk, tile=tile)
        torch.cuda.synchronize()
        print('   ok', float(out.abs().max()), flush=True)
flush = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
flush.fill_(1.0); torch.cuda.synchronize(); print('flush ok', flush=True)
K.igemm(x, w, N, B, H, H, H, H, ks, 1, 0, out_f32=out, splitk=0)
torch.cuda.synchronize(); print('after flush ok', flush=True)
