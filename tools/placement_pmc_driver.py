#!/usr/bin/env python3
"""Driver for tools/placement_pmc.sh: the probe kernel (z = x + y over 70 M floats) on a FAST and on a SLOW placement of its first
read stream inside one 9.5 GiB allocation (profiles/r04_placement_scan.txt part 3: the first 8 GiB read fast, the tail slow):
y, z fixed in the fast part; 12 launches with x at offset 0, then 12 with x at 8704 MiB.  Under rocprofv3 --pmc the dispatches
of gd3d::probe_add_kernel come out in this order: the first 12 are FAST, the last 12 SLOW.  Also prints event timings."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mmdet3d_gaussian_amd as amd
from mmdet3d_gaussian_amd.gd_loss import DispatchTimer
lib = amd.load_library()
N, MB = 70_000_000, 1 << 20
arena = torch.empty(9728 * MB, dtype=torch.uint8, device='cuda').zero_()
torch.cuda.synchronize()
base = arena.data_ptr()
stream = torch.cuda.current_stream().cuda_stream
tm = DispatchTimer()
for name, off in (('fast', 0), ('slow', 8704)):
    ts = []
    for _ in range(12):
        assert lib.gd3d_probe_stream(base + off * MB, base + 4096 * MB, base + 4400 * MB, N, stream, tm.start, tm.stop) == 0
        torch.cuda.synchronize()
        ts.append(tm.elapsed_ms() * 1e3)
    print(name, 'x at', off, 'MiB: median', round(statistics.median(ts), 1), 'us', flush=True)
