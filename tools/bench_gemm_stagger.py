"""Start-up stagger of the persistent bf16 GEMM (sg_set_gemm_tuning(0, cycles)) on the ViT-L/14 shapes at the bench launch size."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import _lib

lib = _lib.load()
dev = "cuda:0"
M = int(os.environ.get("GEMM_TILES", "128")) * 1370
SHAPES = [("qkv", M, 3072, 1024, 0, 1), ("out", M, 1024, 1024, 0, 0), ("fc", M, 4096, 1024, 1, 1), ("proj", M, 1024, 4096, 0, 0)]
STAG = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,20000,40000,80000".split(","))]
ROUNDS, ITERS = 5, 6
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for name, m, n, k, act, cbf in SHAPES:
    A = torch.randn(m, k, device=dev).bfloat16()
    W = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    Cc = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if cbf else torch.float32)
    R = None if cbf else torch.randn(m, n, device=dev)
    res = {}
    for rnd in range(ROUNDS):
        for st in STAG:
            lib.sg_set_gemm_tuning(0, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                lib.sg_gemm_bf16_raw(P(A), P(W), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(st, []).append(e0.elapsed_time(e1) / ITERS)
    for st in STAG:
        ts = sorted(res[st])
        med = ts[len(ts) // 2]
        print(f"{name:5s} stagger {st:7d}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  -> {2.0 * m * n * k / med / 1e9:7.1f} TFLOP/s", flush=True)
lib.sg_set_gemm_tuning(0, -1)
