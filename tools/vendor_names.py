"""Which vendor-library kernels (macro-tile, wave layout, split) serve the train-step GEMM shapes: run under
rocprofv3 --kernel-trace --stats and read the kernel names.  Measurement aid only."""
import torch
dev = torch.device("cuda:0")
for (M, N, K) in [(12000, 512, 2048), (12000, 512, 1536), (12000, 2048, 512), (8192, 8192, 8192), (2592, 512, 2048)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = torch.randn(N, K, device=dev).bfloat16()
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(10):
        torch.matmul(A, B.t(), out=C)
    torch.cuda.synchronize()
