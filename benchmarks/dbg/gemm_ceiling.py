"""What a vendor GEMM reaches on this box (the practical fp16 MFMA ceiling under the chip's power limit), beside the filter's
0.40-0.43 of the nominal 2 516.6 TFLOP/s: torch.matmul (hipBLASLt) at 8192^3 and at the filter's own shape (K = 128, f32 out)."""
import torch
dev = torch.device("cuda", 0)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for dt in (torch.float16, torch.bfloat16):
    A = torch.randn(8192, 8192, device=dev, dtype=dt); B = torch.randn(8192, 8192, device=dev, dtype=dt)
    s = t(lambda: A @ B)
    print(f"{dt}: 8192^3 GEMM {s * 1e3:.3f} ms = {2 * 8192 ** 3 / s / 1e12:.0f} TFLOP/s = {2 * 8192 ** 3 / s / 2516.6e12:.2f} of nominal")
    A = torch.randn(16, 8192, 128, device=dev, dtype=dt); B = torch.randn(16, 128, 8192, device=dev, dtype=dt)
    s = t(lambda: torch.bmm(A, B))
    print(f"{dt}: 16 x (8192 x 128 x 8192) bmm, 16-bit out {s * 1e3:.3f} ms = {16 * 2 * 8192 * 8192 * 128 / s / 1e12:.0f} TFLOP/s (writes 2.1 GB of scores: HBM-bound)")
