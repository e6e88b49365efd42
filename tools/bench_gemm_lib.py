"""Calibration only (not product code): what the vendor GEMM library reaches through torch (hipBLASLt / rocBLAS) on the same ViT-L/14
linear shapes and random bf16 operands, next to this repo's persistent kernel.  Plain GEMM + bias, no fused activation / residual."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import _lib
lib = _lib.load()
dev = "cuda:0"
M = int(os.environ.get("GEMM_TILES", "128")) * 1370
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for name, m, n, k in [("qkv", M, 3072, 1024), ("out", M, 1024, 1024), ("fc", M, 4096, 1024), ("proj", M, 1024, 4096)]:
    A = torch.randn(m, k, device=dev).bfloat16()
    W = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    bias16 = bias.bfloat16()
    Cc = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    def mine():
        lib.sg_gemm_bf16_raw(P(A), P(W), P(bias), None, P(Cc), m, n, k, 0, 1, stream)
    def vendor():
        torch.nn.functional.linear(A, W, bias16)
    for label, fn in (("this repo", mine), ("torch/hipBLASLt", vendor)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        med = sorted(ts)[2]
        print(f"{name:5s} {label:16s}: {med * 1e3:8.1f} us -> {2.0 * m * n * k / med / 1e9:7.1f} TFLOP/s", flush=True)
