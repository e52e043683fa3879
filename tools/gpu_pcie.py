"""D2H / H2D rate of pinned copies on this box, by the NUMA node of the allocating thread (diagnostic)."""
import os, sys, time
import torch
n = 512 << 20
g = torch.empty(n, dtype=torch.uint8, device="cuda")
for label, cpus in (("default", None), ("cpus 0-63", range(0, 64)), ("cpus 64-127", range(64, 128)), ("cpus 128-191", range(128, 192))):
    if cpus is not None:
        try: os.sched_setaffinity(0, cpus)
        except Exception as e: print(label, "affinity failed", e); continue
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    h.zero_()
    for name, fn in (("D2H", lambda: h.copy_(g, non_blocking=True)), ("H2D", lambda: g.copy_(h, non_blocking=True))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4): fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-14s %s %.1f GB/s" % (label, name, 4 * n / dt / 1e9))
    del h
os.system("rocm-smi --showtopo 2>/dev/null | head -30; cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4")
