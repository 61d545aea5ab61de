#!/usr/bin/env python
"""Drives tools/ubench/handoff.hip: the period of a phase of 256 one-per-CU workgroups when the phases are (a) separate launches on one
stream, (b) one launch whose workgroups wait on the previous phase's completion counters (lower block ids only).
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libhandoff.so tools/ubench/handoff.hip ; python tools/ubench/handoff.py"""
import ctypes, os
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libhandoff.so"))
P = ctypes.c_void_p
lib.handoff_run.argtypes = [P, P, P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
PH = 40
done = torch.zeros(PH * 8, dtype=torch.int32, device="cuda")
data = torch.zeros(PH * 256 * 512, device="cuda")
stamps = torch.zeros(PH * 256 * 3, dtype=torch.int64, device="cuda")
fault = torch.zeros(1, dtype=torch.int32, device="cuda")
for lds in (140 * 1024, 64 * 1024):
    for work_us in (12, 35):
        for one in (0, 1):
            for rep in range(2):
                rc = lib.handoff_run(done.data_ptr(), data.data_ptr(), stamps.data_ptr(), fault.data_ptr(), PH, int(work_us * 2000), one, lds)
            assert rc == 0 and int(fault.item()) == 0, (rc, fault)
            t = stamps.cpu().numpy().reshape(PH, 256, 3).astype(np.float64) * 0.01   # us
            start, ready, end = t[:, :, 0], t[:, :, 1], t[:, :, 2]
            period = np.diff(end.max(1))[5:].mean()              # last end of a phase -> last end of the next
            work = (end - ready)[5:].mean()
            wait = (ready - start)[5:].mean()
            gap = (ready.min(1)[1:] - end.max(1)[:-1])[5:].mean()  # last producer end -> first consumer ready
            gap_med = (np.median(ready, 1)[1:] - np.median(end, 1)[:-1])[5:].mean()
            print("LDS %3d KB  work %2d us  %-12s: phase period %6.2f us  (work %5.2f, in-WG wait %5.2f)  overhead %5.2f us per phase;  last end -> first ready %5.2f, median end -> median ready %5.2f"
                  % (lds // 1024, work_us, "one launch" if one else "P launches", period, work, wait, period - work, gap, gap_med))
