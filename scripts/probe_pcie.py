import torch, time
for mb in (1, 8, 64, 256):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for _ in range(2): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    for _ in range(5): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 5
    print("%4d MiB  D2H %.2f GB/s (%.3f ms)   H2D %.2f GB/s" % (mb, n / dt / 1e9, dt * 1e3, n / dt2 / 1e9), flush=True)
