#!/usr/bin/env python
"""Drives tools/ubench/mfma_stream.hip (see its header): cycles per step of the chain's products loop with the weight fragments arriving
in registers, in LDS by DMA, or not at all.
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libmfmastream.so tools/ubench/mfma_stream.hip ; python tools/ubench/mfma_stream.py"""
import ctypes, os
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libmfmastream.so"))
P = ctypes.c_void_p
lib.mfmastream_run.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, ctypes.POINTER(ctypes.c_float)]
out = torch.zeros(16, device="cuda")
cyc = torch.zeros(2048, dtype=torch.int64, device="cuda")
names = {0: "fragments -> registers (global_load_dwordx4)", 1: "fragments -> LDS by DMA (global_load_lds_dwordx4) + ds_read", 2: "no weight loads"}
BLOCKS, REPS = 256, 10
w = torch.randn(2 * 1024 * 1024, device="cuda")
for threads in (256, 512):
    nw = threads // 64
    steps = 256 // nw * 4          # 1 MB per workgroup and pass in 1 KB fragments: four Winograd layers' worth
    for mode in (2, 0, 1):
        ms = ctypes.c_float(0)
        for _ in range(2):
            rc = lib.mfmastream_run(w.data_ptr(), steps, mode, REPS, BLOCKS, threads, out.data_ptr(), cyc.data_ptr(), ctypes.byref(ms))
        torch.cuda.synchronize()
        c = np.median(cyc[:BLOCKS].cpu().numpy().astype(np.float64))
        per_step = c / (REPS * steps)
        print("waves %d  %-60s rc %d : %6.1f cycles per step of a wave (96 = matrix rate at one wave per SIMD), %5.1f B/clk per CU"
              % (nw, names[mode], rc, per_step, (0 if mode == 2 else nw * 1024 / per_step)))
