import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visdial_amd import ops
from microbench import timeit

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
for (M, N, K) in [(20000, 2048, 512), (20000, 512, 2048), (20000, 2048, 2048), (8192, 8192, 512), (8192, 8192, 4096)]:
    A, B = rnd(M, K), rnd(K, N) * 0.05
    C = torch.empty(M, N, device=dev)
    ms = timeit(lambda: ops.gemm_nn(A, B, C), iters=10)
    W = rnd(N, K) * 0.05
    ms2 = timeit(lambda: ops.gemm_nt(A, W, C), iters=10)
    print("M=%d N=%d K=%d: nn %.3f ms %.1f TF | nt %.3f ms %.1f TF" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, ms2, 2.0 * M * N * K / ms2 / 1e9))
