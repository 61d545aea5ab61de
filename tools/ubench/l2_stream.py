#!/usr/bin/env python
"""Drives tools/ubench/l2_stream.hip: B/clk per CU when every CU streams the same weights from L2 (see the .hip header).
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libl2stream.so tools/ubench/l2_stream.hip ; python tools/ubench/l2_stream.py"""
import ctypes, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libl2stream.so"))
P = ctypes.c_void_p
lib.l2stream_run.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.POINTER(ctypes.c_float)]
out = torch.zeros(16, device="cuda")
cyc = torch.zeros(2048, dtype=torch.int64, device="cuda")
print("mode 0 same addresses/order, 1 rotated start per workgroup, 2 private copy per workgroup")
for threads, BLOCKS in ((256, 256), (512, 256), (256, 512), (256, 1024)):   # waves per CU: 4, 8, 8 (2 workgroups), 16 (4 workgroups)
    nw = threads // 64
    for total_kb in (144, 576):             # one 64x64x3x3 conv layer = 144 KB; four layers
        kb_per_wave = total_kb // nw
        w = torch.randn(BLOCKS * total_kb * 256 + 4096, device="cuda")
        for mode in (0, 2):
            for depth in (4, 12):
                reps = 20
                ms = ctypes.c_float(0)
                for _ in range(2):
                    rc = lib.l2stream_run(w.data_ptr(), kb_per_wave, total_kb, mode, reps, depth, BLOCKS, threads, out.data_ptr(), cyc.data_ptr(), ctypes.byref(ms))
                torch.cuda.synchronize()
                c = cyc[:BLOCKS].cpu().numpy().astype(np.float64)
                bytes_per_cu = reps * total_kb * 1024
                print("waves/WG %d x %4d WGs  buffer %4d KB  mode %d  depth %2d : %6.1f B/clk per WG (median), wall %.1f us/pass, %5.2f TB/s aggregate = %5.1f B/clk/CU at 2.1 GHz"
                      % (nw, BLOCKS, total_kb, mode, depth, bytes_per_cu / np.median(c), ms.value * 1e3 / reps, BLOCKS * bytes_per_cu / (ms.value * 1e-3) / 1e12,
                         BLOCKS * bytes_per_cu / (ms.value * 1e-3) / 256 / 2.1e9))
