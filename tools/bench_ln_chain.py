"""out-proj -> ln_2 -> fc and proj -> ln_1 -> QKV chains at the ViT-L/14 launch shape: folded LayerNorm vs the separate pass (sg_op_ln_chain)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import ops

M = int(os.environ.get("GEMM_TILES", "128")) * 1370
dev = "cuda:0"
for name, K1, D, N2, act in [("out-proj->ln2->fc", 1024, 1024, 4096, 1), ("proj->ln1->qkv", 4096, 1024, 3072, 0)]:
    A = torch.randn(M, K1, device=dev); W1 = torch.randn(D, K1, device=dev) * K1 ** -0.5; b1 = torch.randn(D, device=dev) * 0.1
    x = torch.randn(M, D, device=dev); g = 1 + 0.1 * torch.randn(D, device=dev); be = 0.1 * torch.randn(D, device=dev)
    W2 = torch.randn(N2, D, device=dev) * D ** -0.5; b2 = torch.randn(N2, device=dev) * 0.1
    for fold in (True, False):
        ops.ln_chain(A, W1, b1, x, g, be, W2, b2, act, "bf16", fold)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.ln_chain(A, W1, b1, x, g, be, W2, b2, act, "bf16", fold)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:20s} fold={fold}: {e0.elapsed_time(e1) / 5:.3f} ms per chain (includes operand packing / x clone / unpack, identical in both arms)", flush=True)
    del A, W1, x, W2
