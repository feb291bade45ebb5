"""Debug: the s_memtime stamps a trace build of linear_fxd_kernel (-DXFH_LFXD_TRACE=1, XFH_LIB_PATH) leaves in the refine workspace: per wave of workgroup 0, chunk 8:
cycles from the chunk's top to [operands pinned, group 0 issued + second reads, group 1's operands pinned, group 1 issued, behind the barrier]."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
x = torch.randn(100000, 128, device='cuda')
for _ in range(3): xf.net._fine_matcher(x)
torch.cuda.synchronize()
t = xf.net._ws['refine']
off = (-t.data_ptr()) % 256
w = t[off:off + ((t.numel() - off) // 4) * 4].view(torch.int32).cpu()
idx = (w == 0x7ace7ace).nonzero().flatten().tolist()
for i in idx[:16]:
    print('wave', (i % 64) // 8, [int(w[i + j]) for j in range(1, 6)])
print(len(idx), 'stamp records')
