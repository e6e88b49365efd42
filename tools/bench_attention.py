"""The fused attention op at the ViT-L/14 tile shape (B tiles x 16 heads x 1370 tokens, head_dim 64): time per launch and MFMA rate.
ATTN_B tiles (default 64); ATTN_VARIANT (vanilla, SegEarth, ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import _lib, ops

lib = _lib.load()
B = int(os.environ.get("ATTN_B", "64"))
N, D, H = 1370, 1024, 16
variant = os.environ.get("ATTN_VARIANT", "vanilla")
prec = os.environ.get("ATTN_PREC", "bf16")
qkv = torch.randn(B, N, 3 * D, device="cuda:0")
ops.attention(qkv, H, variant, precision=prec)
torch.cuda.synchronize()
lib.sg_profile_enable(4096)
for _ in range(int(os.environ.get("ATTN_REPS", "5"))):
    ops.attention(qkv, H, variant, precision=prec)
torch.cuda.synchronize()
import ctypes as C
ms, fl, n, dr = C.c_double(), C.c_double(), C.c_long(), C.c_long()
lib.sg_profile_read(1, C.byref(ms), C.byref(fl), C.byref(n), C.byref(dr))
print(f"{variant} {prec} B={B}: {ms.value / max(n.value, 1) * 1e3:.1f} us per launch, {fl.value / ms.value / 1e9:.1f} TFLOP/s over {n.value} launches", flush=True)
