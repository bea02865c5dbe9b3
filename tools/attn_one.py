"""Run the attention kernel on one shape a few times (target of rocprofv3 --pmc passes, tools/attn_pmc.sh)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
import kernels as K  # noqa: E402
d, heads, B, nq, nkv = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (40, 8, 2, 4096, 4096))]
g = torch.Generator().manual_seed(1)
BH = B * heads
q = torch.randn((BH, nq, d), generator=g).half().cuda(); k = torch.randn((BH, nkv, d), generator=g).half().cuda()
vt = torch.randn((BH, d, (nkv + 7) // 8 * 8), generator=g).half().cuda()
for _ in range(6):
    K.attention(q, k, vt, heads, nkv, d ** -0.5)
torch.cuda.synchronize()
