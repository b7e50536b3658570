#!/usr/bin/env python3
"""Box exploration (development): host cpus actually usable, torch CPU threading behaviour."""
import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except Exception as e:
        print(f, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
a = torch.randn(3, 1024, 8192); b = torch.randn(3, 8192, 64)
for n in (8, 16, 32, 64, 128, 256):
    torch.set_num_threads(n)
    t0 = time.time()
    for _ in range(3):
        (a.softmax(-1) @ b).sum().item()
    print("threads", n, "softmax+bmm x3: %.3f s" % (time.time() - t0))
