#!/usr/bin/env python3
"""Host <-> device copy rates of the box (page-locked buffers, one stream each way): the ceiling of the PCIe-inclusive ABI rate."""
import time
import torch

n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in (("h2d", lambda: d.copy_(h, non_blocking=True)), ("d2h", lambda: h.copy_(d, non_blocking=True))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {8 * n / (time.perf_counter() - t0) / 1e9:.1f} GB/s")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    with torch.cuda.stream(s1):
        d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"both ways at once: {8 * n / dt / 1e9:.1f} GB/s each")
