"""what bounds staging from host memory: raw pinned->HBM copies vs threads copying into pinned memory (GPU box probe)"""
import time, torch, numpy as np
from concurrent.futures import ThreadPoolExecutor
dev = torch.device("cuda:0")
for mb in (16, 64, 256):
    h = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    d = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    n = max(4, 2048 // mb)
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    print(f"h2d pinned {mb} MB chunks: {n * mb / 1024 / (time.perf_counter() - t):.1f} GB/s", flush=True)
src = np.random.default_rng(1).integers(0, 255, 2 << 30, dtype=np.uint8)
pin = torch.empty(2 << 30, dtype=torch.uint8).pin_memory().numpy()
plain = np.empty(2 << 30, np.uint8); plain[:] = 0
for nt in (8, 16, 32, 64):
    for name, dst in (("pinned", pin), ("plain", plain)):
        step = (2 << 30) // nt
        def cp(k): dst[k * step:(k + 1) * step] = src[k * step:(k + 1) * step]
        with ThreadPoolExecutor(nt) as ex:
            list(ex.map(cp, range(nt)))
            t = time.perf_counter()
            for _ in range(3): list(ex.map(cp, range(nt)))
            print(f"memcpy {nt} threads -> {name}: {6 / (time.perf_counter() - t):.1f} GB/s", flush=True)
# compare-only (what an alias check costs): two reads, no write
for nt in (16, 64):
    step = (2 << 30) // nt
    def cmp(k): return bool(np.array_equal(src[k * step:(k + 1) * step], pin[k * step:(k + 1) * step]))
    with ThreadPoolExecutor(nt) as ex:
        t = time.perf_counter()
        for _ in range(3): list(ex.map(cmp, range(nt)))
        print(f"compare {nt} threads: {6 / (time.perf_counter() - t):.1f} GB/s per stream", flush=True)
