import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clip_decontamination_amd import ops
from oracle import ctd as OC
x = torch.from_numpy(OC.make_clustered_tokens(16, 1369, 768, seed=2, spread=0.5, n_centers=9)).cuda()
cls = torch.randn(16, 768).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out, lab = ops.ctd_debias(x, cls, normalize_cls=True)
    torch.cuda.synchronize(); print("ctd 16 tiles of 37x37x768: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
