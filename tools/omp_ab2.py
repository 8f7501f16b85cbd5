"""Within-process interleaved A/B of the second-generation greedy kernel variants (lys_debug_bomp_variant >= 100)
against the product launch (lys_bomp_from_alpha0), with a parity check of every variant against the product output.
usage: python tools/omp_ab2.py [N] [variant ...]"""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lyssandra_amd import _lib, engine

lib = _lib.load()
n, K, k = 64, 1024, 10
N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
variants = [int(v) for v in sys.argv[2:]] or [100, 101, 106]
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
Dt = torch.randn((n, K), device=dev, generator=g)
Dt = Dt / Dt.norm(dim=0, keepdim=True)
Xs = torch.randn((N, n), device=dev, generator=g)
dd = engine.DeviceDictionary(n, K, dev)
dd.set(Dt)
G = dd.gram()
a0 = torch.empty((N, 1024), dtype=torch.float32, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
_lib.check(lib.lys_alpha0(P(Xs), Xs.stride(0), P(dd.D), n, K, N, P(a0), st))


def outs():
    return (torch.full((N, k), -7, dtype=torch.int32, device=dev), torch.zeros((N, k), dtype=torch.float32, device=dev),
            torch.zeros((N,), dtype=torch.int32, device=dev))


def run(var, o):
    if var < 0:
        _lib.check(lib.lys_bomp_from_alpha0(P(a0), P(G), K, k, N, P(o[0]), P(o[1]), P(o[2]), st))
    else:
        _lib.check(lib.lys_debug_bomp_variant(P(a0), P(G), N, k, P(o[0]), P(o[1]), P(o[2]), var, 0, st))


# reference for the parity check: the FIRST-generation kernel (debug variant 0), an independent implementation -- comparing
# against the product launch would compare the new kernel with itself
ref = outs()
run(0, ref)
torch.cuda.synchronize()
names = {-1: "product", 0: "gen-1 kernel"}
for v in [-1] + variants:
    o = outs()
    try:
        run(v, o)
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print("variant %d failed: %s" % (v, e))
        continue
    same_idx = (o[0] == ref[0]).all(dim=1)
    same_nnz = (o[2] == ref[2])
    scale = ref[1].abs().amax(dim=1).clamp_min(1e-30)
    rel = ((o[1] - ref[1]).abs().amax(dim=1) / scale)
    rel_ok = rel[same_idx]
    print("variant %d vs gen-1: idx rows equal %d / %d, nnz equal %d, max rel coef diff on equal rows %.3e, nan %d"
          % (v, int(same_idx.sum()), N, int(same_nnz.sum()), float(rel_ok.max()) if rel_ok.numel() else -1.0,
             int(torch.isnan(o[1]).sum())))
    names.setdefault(v, "variant %d" % v)

order = [0, -1] + [v for v in variants if v in names]
times = {v: [] for v in order}
o = outs()
for rnd in range(int(os.environ.get("AB_ROUNDS", "24"))):
    for v in order:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(v, o)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 4:
            times[v].append(e0.elapsed_time(e1))
for v in order:
    t = sorted(times[v])
    med = t[len(t) // 2]
    print("%-14s median %.4f ms  min %.4f ms  -> %.1f M sig/s  (%.3f of fp32 peak at 113640 FLOP/patch)"
          % (names[v], med, t[0], N / med / 1e3, 113640.0 * N / (med * 1e-3) / 157.3e12))
