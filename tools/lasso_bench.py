"""l1 coder timing (sparse_coding.py:487-509 path): Gram-based coordinate descent, structured synthetic signals."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import engine

dev = torch.device("cuda", 0)
for name, n, K, N, lam in [("metric shape", 64, 1024, 262144, 0.15), ("config-4 shape", 128, 8192, 32768, 0.15)]:
    g = torch.Generator(device=dev).manual_seed(11)
    Dm = torch.randn((K, n), device=dev, generator=g)
    Dm = Dm / Dm.norm(dim=1, keepdim=True)
    sel = torch.randint(0, K, (N, 6), device=dev, generator=g)
    w = torch.randn((N, 6), device=dev, generator=g)
    Xs = (Dm[sel] * w[:, :, None]).sum(1) + 0.05 * torch.randn((N, n), device=dev, generator=g)
    Xs = (Xs / Xs.norm(dim=1, keepdim=True)).contiguous()
    dd = engine.DeviceDictionary(n, K, dev)
    dd.set(Dm.t().contiguous())
    out = engine.lasso_encode(Xs, dd, lam, return_steps=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        idx, coef, nnz, steps = engine.lasso_encode(Xs, dd, lam, out=out[:3], return_steps=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    st = steps.float()
    print("%s n=%d K=%d N=%d lambda=%.2f: %.2f ms -> %.2f M signals/s | nnz mean %.1f max %d | CD steps mean %.1f max %d"
          % (name, n, K, N, lam, ms, N / ms / 1e3, nnz.float().mean().item(), int(nnz.max().item()),
             st.mean().item(), int(st.max().item())))
