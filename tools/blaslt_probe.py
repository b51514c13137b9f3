"""What the vendor GEMM (hipBLASLt via torch) reaches on the hot-path shapes: a yardstick for gemm.hip, not product code."""
import torch, time, sys
dev = "cuda:0"
shapes = [(3968, 9216, 3072), (3968, 3072, 3072), (3968, 12288, 3072), (3968, 3072, 12288), (3968, 21504, 3072), (3968, 3072, 15360), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev).to(torch.bfloat16)
    for _ in range(3):
        y = torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        y = torch.nn.functional.linear(a, w, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"hipBLASLt linear M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
