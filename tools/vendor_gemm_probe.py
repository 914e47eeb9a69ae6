"""Development probe (not part of the product): what the vendor BLAS behind torch.bmm reaches on the
headline GEMM shapes, as a yardstick for k_mfma_f32 on the same box."""
import torch

torch.backends.cuda.matmul.allow_tf32 = False
dev = "cuda:0"


def run(R, M, N, K, iters=30):
    a = torch.randn(R, M, K, device=dev) / 16
    b = torch.randn(R, K, N, device=dev) / 16
    c = torch.empty(R, M, N, device=dev)
    for _ in range(5):
        torch.bmm(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.bmm(a, b, out=c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"bmm R={R} M={M} N={N} K={K}: {ms * 1e3:.1f} us  {2.0 * R * M * N * K / ms / 1e9:.1f} TFLOP/s", flush=True)


for R in (192, 256, 1024):
    run(R, 256, 1024, 256)
    run(R, 256, 256, 1024)
run(1, 4096, 4096, 4096)
run(1, 8192, 8192, 8192, iters=10)
