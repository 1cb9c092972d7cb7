import torch, time
torch.backends.cuda.matmul.allow_fp16_reduced_precision_reduction = False
dev = "cuda"
def bench(M, N, K, reps=50):
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(5): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): torch.matmul(a, w.t())
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / reps
    print(f"M={M} N={N} K={K}: {us:.1f} us  {2*M*N*K/us/1e6:.0f} TFLOP/s")
for M in (1500, 12000):
    for N, K in ((3840, 1280), (1280, 1280), (5120, 1280), (1280, 5120), (2560, 1280)):
        bench(M, N, K)
