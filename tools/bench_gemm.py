"""A/B of the bf16 GEMM tile variants on the ViT-L/14 shapes (interleaved rounds in ONE process, random data)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import _lib

lib = _lib.load()
dev = "cuda:0"
M = int(os.environ.get("GEMM_TILES", "32")) * 1370
SHAPES = [("qkv", M, 3072, 1024, 0, 1), ("out", M, 1024, 1024, 0, 0), ("fc", M, 4096, 1024, 1, 1), ("proj", M, 1024, 4096, 0, 0)]
CONFIGS = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5,6".split(","))]
# tile order of the persistent kernel (config 30): GEMM_ORDER = auto | raster | <N-group size>
_order = os.environ.get("GEMM_ORDER", "auto")
lib.sg_set_gemm_config(1000 if _order == "auto" else (1001 if _order == "raster" else 1001 + int(_order)))
if os.environ.get("GEMM_GRID_CAP"):
    lib.sg_set_gemm_config(2000 + int(os.environ["GEMM_GRID_CAP"]))   # experiment: fewer persistent workgroups than CUs
ROUNDS, ITERS = 5, 10
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
res = {}
for name, m, n, k, act, cbf in SHAPES:
    A = torch.randn(m, k, device=dev).bfloat16()
    W = (torch.randn(n, k, device=dev) * k ** -0.5).bfloat16()
    bias = torch.randn(n, device=dev)
    Cc = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if cbf else torch.float32)
    R = None if cbf else torch.randn(m, n, device=dev)
    ref = None
    for cfg in CONFIGS:
        lib.sg_set_gemm_config(cfg)
        rc = lib.sg_gemm_bf16_raw(P(A), P(W), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)
        assert rc == 0, lib.sg_last_error()
        torch.cuda.synchronize()
        out = Cc.float().clone()
        if ref is None:
            x = A.float() @ W.float().T + bias
            if act == 1:
                x = x * torch.sigmoid(1.702 * x)
            if R is not None:
                x = x + R
            ref = x
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        assert err < 2e-2 or (cfg >= 10 and cfg != 30) or os.environ.get("GEMM_NOCHECK"), (name, cfg, err)   # GEMM_NOCHECK: ablated builds (tools/ablate_persist.sh)
    for rnd in range(ROUNDS):
        for cfg in CONFIGS:
            lib.sg_set_gemm_config(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                lib.sg_gemm_bf16_raw(P(A), P(W), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault((name, cfg), []).append(e0.elapsed_time(e1) / ITERS)
    for cfg in CONFIGS:
        ts = sorted(res[(name, cfg)])
        med = ts[len(ts) // 2]
        print(f"{name:5s} cfg {cfg}: median {med * 1e3:8.1f} us  min {ts[0] * 1e3:8.1f} us  -> {2.0 * m * n * k / med / 1e9:7.1f} TFLOP/s (median)", flush=True)
lib.sg_set_gemm_config(-1)
