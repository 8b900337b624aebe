"""Context for the GEMM roofline fraction: what the vendor BLAS (rocBLAS / hipBLASLt behind torch.mm, fp32 in / fp32
out, no TF32) reaches on the dominant shape -- plain GEMM, no prologue / epilogue fusion."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda:0')
for M, K, N in [(65536, 576, 576), (65536, 576, 576), (16384, 288, 288), (16384, 288, 576), (4096, 288, 288), (65536, 576, 48), (65536, 48, 576)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(K, N, device=dev); c = torch.empty(M, N, device=dev)
    torch.mm(a, b, out=c); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(10): torch.mm(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print('torch.mm fp32 %6d x %3d x %3d: %7.1f us  %6.1f TFLOP/s' % (M, K, N, best * 1e3, 2.0 * M * K * N / best / 1e9))
