"""Encode throughput at the other configs' shapes (parity-test shapes, not the bench line)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine

dev = torch.device("cuda", 0)
for name, n, K, k, N in [("C1 n=64 K=256 k=5", 64, 256, 5, 1 << 20), ("M  n=64 K=1024 k=10", 64, 1024, 10, 1 << 20),
                         ("n=128 K=1024 k=10", 128, 1024, 10, 1 << 19), ("n=256 K=512 k=20", 256, 512, 20, 1 << 18),
                         ("C3 n=256 K=4096 k=20", 256, 4096, 20, 1 << 14), ("n=64 K=2048 k=10", 64, 2048, 10, 1 << 16)]:
    g = torch.Generator(device=dev).manual_seed(1)
    Dt = torch.randn((n, K), device=dev, generator=g)
    Dt = Dt / Dt.norm(dim=0, keepdim=True)
    Xs = torch.randn((N, n), device=dev, generator=g)
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set(Dt)
    dd.gram()
    out = engine.bomp_encode(Xs, dd, k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        engine.bomp_encode(Xs, dd, k, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    F = 2 * n * K + K * k * (k + 1) + k ** 3
    print("%-24s N=%8d  %.3f ms  %.2f M patches/s  %.1f TFLOP/s (%.1f%% fp32 peak)"
          % (name, N, dt * 1e3, N / dt / 1e6, F * N / dt / 1e12, F * N / dt / 157.3e10))
    engine.release_workspaces()
