import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import kernels as K
from stable_diffusion_amd import _lib
B, H, W, C, N = 2, 64, 64, 320, 320
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, C, generator=g).cuda()
gamma = torch.ones(C).cuda(); beta = torch.zeros(C).cuda()
w = (torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C)).cuda()
wp = K.pack_conv_weight(w)
out = torch.empty(B * H * W, N, device='cuda')
n = _lib.load().sdmi_k_groupnorm_ws_floats(B, H * W)
gws = torch.empty((n,), dtype=torch.float32, device='cuda')
for i in range(12):
    _lib.check(_lib.load().sdmi_k_conv3gn(x.data_ptr(), None, C, 0, B, H, W, gamma.data_ptr(), beta.data_ptr(), 1e-5, wp.data_ptr(), N,
                                          None, None, 0, None, 0, out.data_ptr(), N, 1, None, 0, gws.data_ptr(), n, K._s()))
torch.cuda.synchronize(); print('done')
