#!/usr/bin/env python3
"""r6: what a 33 MB image costs across PCIe on this box: pageable and pinned, each way, and both ways at once."""
import time
import numpy as np
import torch

N = 3840 * 2160 * 4
dev = torch.empty(N, dtype=torch.uint8, device="cuda")
dev2 = torch.empty(N, dtype=torch.uint8, device="cuda")
page = torch.from_numpy(np.random.default_rng(1).integers(0, 256, N, dtype=np.uint8))
pin = page.clone().pin_memory()
pin2 = torch.empty(N, dtype=torch.uint8).pin_memory()
page2 = torch.empty(N, dtype=torch.uint8)


def t(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, f in [("H2D pageable", lambda: dev.copy_(page)), ("H2D pinned", lambda: dev.copy_(pin, non_blocking=True)),
                ("D2H pageable", lambda: page2.copy_(dev)), ("D2H pinned", lambda: pin2.copy_(dev, non_blocking=True))]:
    dt = t(f)
    print(f"{name:14s} {dt * 1e3:7.3f} ms  {N / dt / 1e9:6.1f} GB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(s1):
        dev.copy_(pin, non_blocking=True)
    with torch.cuda.stream(s2):
        pin2.copy_(dev2, non_blocking=True)


dt = t(both)
print(f"both ways, pinned, two streams: {dt * 1e3:7.3f} ms per pair  {2 * N / dt / 1e9:6.1f} GB/s total")
t0 = time.perf_counter()
for _ in range(10):
    pin2.copy_(page)
dt = (time.perf_counter() - t0) / 10
print(f"host memcpy pageable -> pinned (one thread): {dt * 1e3:7.3f} ms  {N / dt / 1e9:6.1f} GB/s")
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
buf = np.random.default_rng(2).integers(0, 256, N, dtype=np.uint8)
t0 = time.perf_counter()
for _ in range(5):
    rc = hip.hipHostRegister(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(N), 0)
    rc2 = hip.hipHostUnregister(ctypes.c_void_p(buf.ctypes.data))
dt = (time.perf_counter() - t0) / 5
print(f"hipHostRegister + Unregister of 33 MB: {dt * 1e3:7.3f} ms (rc {rc}, {rc2})")
