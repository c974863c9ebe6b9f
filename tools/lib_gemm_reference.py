"""Reference point only (never used by the product path): what torch.matmul (hipBLASLt / rocBLAS) reaches on the DiT GEMM
shapes, to compare with tools/kernel_bench.py gemm."""
import torch, time
dev = torch.device("cuda:0")
def bench(M, N, K):
    a = (torch.randn(M, K, device=dev) ).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(2): torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): torch.matmul(a, w.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5e3
    print(f"torch.matmul (hipBLASLt/rocBLAS) M={M} N={N} K={K}: {t*1e3:.3f} ms {2*M*N*K/t/1e12:.1f} TFLOP/s", flush=True)
for shp in ((194400, 7680, 2560), (194400, 2560, 2560), (194400, 13824, 2560), (194400, 2560, 6912), (8192, 8192, 8192)):
    bench(*shp)
