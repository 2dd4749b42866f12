#!/usr/bin/env python3
"""The achievable ceiling for a launch as short as one frame: device-to-device copy of one frame's samples
(read P + write P bytes) and of 1 GiB, HIP events around each (torch's elementwise copy kernel)."""
import torch
torch.cuda.set_device(0)
for name, n in (("1440p 8-bit frame (5.5 MB)", 2560 * 1440 * 3 // 2), ("2160p 8-bit frame (12.4 MB)", 3840 * 2160 * 3 // 2),
                ("eight 2160p frames (99.5 MB)", 8 * 3840 * 2160 * 3 // 2), ("1 GiB", 1 << 30)):
    a = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    for _ in range(5):
        b.copy_(a)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = sorted(ts)[len(ts) // 2]
    print(f"{name}: {us:8.1f} us  {2 * n / us / 1e3:7.1f} GB/s (read + write)")
