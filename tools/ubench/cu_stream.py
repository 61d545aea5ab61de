#!/usr/bin/env python
"""Drives tools/ubench/cu_stream.hip (see its header): bytes per clock one CU gets from L2 / L1 / LDS, 4 and 8 waves.
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libcustream.so tools/ubench/cu_stream.hip ; python tools/ubench/cu_stream.py"""
import ctypes, os
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libcustream.so"))
P = ctypes.c_void_p
lib.custream_run.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.POINTER(ctypes.c_float)]
out = torch.zeros(16, device="cuda")
cyc = torch.zeros(2048, dtype=torch.int64, device="cuda")
names = {0: "L2 -> VGPR (global_load_dwordx4)", 1: "L2 -> VGPR, non-temporal", 2: "L1 -> VGPR (4 KB window)", 3: "L2 -> LDS (global_load_lds)", 4: "LDS -> VGPR (ds_read_b128)", 5: "VGPR -> LDS (ds_write_b128)", 6: "VGPR -> LDS (ds_write_b64)", 7: "VGPR -> LDS (ds_write_b32)"}
BLOCKS = 256
for threads in (256, 512):
    nw = threads // 64
    kb_per_wave = 256 // nw          # 256 KB per workgroup and pass = one Winograd layer
    w = torch.randn(256 * 1024 // 4 + 4096, device="cuda")
    for mode in (0, 1, 2, 4, 5, 6, 7):   # (mode 3, global_load_lds, needs a builtin this compiler does not have: the kernel body is empty)
        reps = 20
        ms = ctypes.c_float(0)
        for _ in range(2):
            rc = lib.custream_run(w.data_ptr(), kb_per_wave, mode, reps, BLOCKS, threads, out.data_ptr(), cyc.data_ptr(), ctypes.byref(ms))
        torch.cuda.synchronize()
        c = cyc[:BLOCKS].cpu().numpy().astype(np.float64)
        b = reps * 256 * 1024
        # the cycle count is wave 0's: with 8 waves it leaves early, so the wall-clock figure is the one to read there
        print("waves %d  %-34s rc %d : %6.1f B/clk per CU by wave 0's clock (median over workgroups), %.1f us per 256 KB pass by wall clock" % (nw, names[mode], rc, b / np.median(c), ms.value * 1e3 / reps))
