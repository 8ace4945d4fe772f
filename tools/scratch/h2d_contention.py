"""H2D copies from pinned memory beside a compute kernel that fills the chip: which engine does the runtime use, what is left of 52 GB/s"""
import os, time, torch
dev = torch.device("cuda:0")
h = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.float64)
cs, ks = torch.cuda.Stream(), torch.cuda.Stream()
def copies(n):
    with torch.cuda.stream(cs):
        for _ in range(n): d.copy_(h, non_blocking=True)
copies(4); torch.cuda.synchronize()
t = time.perf_counter(); copies(64); torch.cuda.synchronize(); alone = 4.0 / (time.perf_counter() - t)
with torch.cuda.stream(ks):
    for _ in range(3): b = a @ a
torch.cuda.synchronize()
t = time.perf_counter()
with torch.cuda.stream(ks):
    for _ in range(6): b = a @ a
ks.synchronize(); mm = (time.perf_counter() - t) / 6
with torch.cuda.stream(ks):
    for _ in range(12): b = a @ a
t = time.perf_counter(); copies(64); cs.synchronize(); beside = 4.0 / (time.perf_counter() - t)
torch.cuda.synchronize()
print({k: os.environ.get(k) for k in ("GPU_BLIT_ENGINE_TYPE", "GPU_FORCE_BLIT_COPY_SIZE", "HSA_ENABLE_SDMA")}, f"alone {alone:.1f} GB/s, beside fp64 matmuls ({mm*1e3:.0f} ms each) {beside:.1f} GB/s")
