"""fp8 GEMMs (v_mfma_f32_16x16x128_f8f6f4: persistent ping-pong kernel `fp8`, two-stage ring kernel `fp8r`) on the ViT-L/14 linear shapes, random operands; bf16 persistent kernel beside it."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clip_decontamination_amd import _lib, ops

lib = _lib.load()
dev = "cuda:0"
M = int(os.environ.get("GEMM_TILES", "128")) * 1370
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
for name, m, n, k, act, cbf in [("qkv", M, 3072, 1024, 0, 1), ("fc", M, 4096, 1024, 1, 1), ("proj", M, 1024, 4096, 0, 0)]:
    A = torch.randn(m, k, device=dev)
    W = torch.randn(n, k, device=dev) * k ** -0.5
    a8, sa = ops.quantize_rows_fp8(A)
    w8, sw = ops.quantize_rows_fp8(W)
    A16, W16 = A.bfloat16(), W.bfloat16()
    del A, W
    bias = torch.randn(n, device=dev)
    Cc = torch.empty(m, n, device=dev, dtype=torch.bfloat16 if cbf else torch.float32)
    R = None if cbf else torch.randn(m, n, device=dev)
    def fp8_ring():
        lib.sg_set_gemm_config(31)                       # the non-persistent two-stage ring kernel
        rc = lib.sg_gemm_fp8_raw(P(a8), P(sa), P(w8), P(sw), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)
        lib.sg_set_gemm_config(-1)
        return rc
    # the two fp8 kernels must agree on the same operands (same products, f32 accumulation order differs only by K-step grouping)
    lib.sg_set_gemm_config(32)
    assert lib.sg_gemm_fp8_raw(P(a8), P(sa), P(w8), P(sw), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream) == 0, lib.sg_last_error()
    lib.sg_set_gemm_config(-1)
    ref_out = Cc.float().clone()
    assert fp8_ring() == 0, lib.sg_last_error()
    torch.cuda.synchronize()
    d = (Cc.float() - ref_out).abs().max().item() / ref_out.abs().max().item()
    print(f"{name:5s} persistent vs ring fp8 kernel: max rel diff {d:.2e}", flush=True)
    assert d < 2e-2, d
    def fp8_persist():
        lib.sg_set_gemm_config(32)                       # force the persistent kernel on every shape
        rc = lib.sg_gemm_fp8_raw(P(a8), P(sa), P(w8), P(sw), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)
        lib.sg_set_gemm_config(-1)
        return rc
    # the MX forms: fc writing e4m3 + block scales, proj reading them
    c8 = torch.empty(m, n, device=dev, dtype=torch.uint8) if name == "fc" else None
    cs = torch.zeros(n // 128, m, 4, device=dev, dtype=torch.uint8) if name == "fc" else None
    amx = torch.full((k // 128, m, 4), 127, device=dev, dtype=torch.uint8) if name == "proj" else None
    mx = []
    if name == "fc":
        mx = [("mxout", lambda: lib.sg_gemm_fp8_mx_raw(P(a8), P(sa), None, P(w8), P(sw), P(bias), None, None, P(c8), P(cs), m, n, k, act, 1, stream))]
    if name == "proj":
        mx = [("mxin", lambda: lib.sg_gemm_fp8_mx_raw(P(a8), None, P(amx), P(w8), P(sw), P(bias), P(R), P(Cc), None, None, m, n, k, act, cbf, stream))]
    for label, fn in (*mx, ("fp8p", fp8_persist), ("fp8", lambda: lib.sg_gemm_fp8_raw(P(a8), P(sa), P(w8), P(sw), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream)),
                      ("fp8r", fp8_ring),
                      ("bf16", lambda: lib.sg_gemm_bf16_raw(P(A16), P(W16), P(bias), P(R), P(Cc), m, n, k, act, cbf, stream))):
        assert fn() == 0, lib.sg_last_error()
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
        print(f"{name:5s} {label:5s}: {med * 1e3:8.1f} us -> {2.0 * m * n * k / med / 1e9:7.1f} TFLOP/s", flush=True)
