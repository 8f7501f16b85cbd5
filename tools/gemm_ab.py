"""alpha0 GEMM stage (and greedy stage) per call at the n > 64 shapes, from the library's own HIP events (bench._profiled_encode).
usage: python tools/gemm_ab.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lyssandra_amd import engine

for n, K, k, N in [(256, 4096, 20, 1 << 17), (128, 1024, 10, 1 << 19), (128, 8192, 10, 1 << 16)]:
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(5)
    Xs = torch.randn((N, n), device=dev, generator=g)
    D = torch.randn((n, K), device=dev, generator=g)
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set(D / D.norm(dim=0, keepdim=True))
    dd.gram()
    out = None
    gemm_ms, omp_ms, wall = bench._profiled_encode(lambda: engine.bomp_encode(Xs, dd, k), 5)
    fl = 2.0 * n * K * N
    print("n=%d K=%d k=%d N=%d: gemm %.3f ms (%.1f TFLOP/s fp32-equivalent), greedy %.3f ms, call %.3f ms = %.2f M patches/s"
          % (n, K, k, N, gemm_ms, fl / gemm_ms / 1e9, omp_ms, wall, N / wall / 1e3))
    engine.release_workspaces()
