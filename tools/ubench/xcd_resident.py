#!/usr/bin/env python
"""Drives tools/ubench/xcd_resident.hip: the period of one simulation = phase A ("chain", one root per workgroup) + phase B
("LSTM", a 16-root tile) when (1) every phase is its own launch -- today's structure --, (0) all 256 workgroups stay resident for the whole
search and the 16 workgroups of a group, all on one XCD, hand over through that XCD's L2 (plain stores + flag, sc1 loads: no write-back, no
invalidate), (2) the same resident kernel with agent-scope release / acquire fences (the form that does not depend on placement).
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o tools/ubench/libxcdresident.so tools/ubench/xcd_resident.hip ; python tools/ubench/xcd_resident.py [out.json]"""
import ctypes, json, os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libxcdresident.so"))
P = ctypes.c_void_p
lib.xcd_resident_run.argtypes = [P, P, P, P, P, P, P, P] + [ctypes.c_int] * 8 + [ctypes.POINTER(ctypes.c_float)]
S = 50
VA, VB = 272, 512
ctl = torch.zeros(16 + 2 + 512, dtype=torch.int32, device="cuda")
flagA = torch.zeros(32 * S, dtype=torch.int32, device="cuda")
flagB = torch.zeros(32 * S, dtype=torch.int32, device="cuda")
payA = torch.zeros(2 * 256 * VA * 4, device="cuda")
payB = torch.zeros(2 * 256 * VB * 4, device="cuda")
STREAM_A = 1310720 // 16            # 1.31 MB of transformed chain weights, the same for every workgroup (16-byte vectors)
STREAM_B = 557056 // 16             # 557 KB LSTM weight slice per workgroup
wA = torch.rand(STREAM_A * 4, device="cuda")
wB = torch.rand(256 * STREAM_B * 4, device="cuda")
stamps = torch.zeros(S * 256 * 4, dtype=torch.int64, device="cuda")
results = []


def run(mode, work_a_us, work_b_us, coop=0, lds=140 * 1024, label=""):
    ms = ctypes.c_float(0)
    for rep in range(3):
        stamps.zero_()
        rc = lib.xcd_resident_run(ctl.data_ptr(), flagA.data_ptr(), flagB.data_ptr(), payA.data_ptr(), payB.data_ptr(), wA.data_ptr(), wB.data_ptr(),
                                  stamps.data_ptr(), S, int(work_a_us * 2100), int(work_b_us * 2100), STREAM_A, STREAM_B, mode, lds, coop, ctypes.byref(ms))
    c = ctl.cpu().numpy()
    fault, stale, xcc = int(c[16]), int(c[17]), c[:8].tolist()
    t = stamps.cpu().numpy().reshape(S, 256, 4).astype(np.float64) * 0.01   # us
    a_ready, a_end, b_ready, b_end = t[:, :, 0], t[:, :, 1], t[:, :, 2], t[:, :, 3]
    lo = 5
    period = (b_end[S - 1].max() - a_ready[lo].min()) / (S - lo)
    work_a, work_b = (a_end - a_ready)[lo:].mean(), (b_end - b_ready)[lo:].mean()
    # hand-off latencies: last producer of a GROUP done -> each consumer of that group ready (resident), or chip-wide (launches)
    gap_ab = (b_ready[lo:] - a_end[lo:].max(1, keepdims=True)).mean() if mode == 1 else None
    rec = dict(mode={0: "resident, XCD-local (plain stores + flag, sc1 loads)", 1: "two launches per simulation", 2: "resident, agent-scope release / acquire"}[mode],
               cooperative=bool(coop), work_a_us=work_a_us, work_b_us=work_b_us, rc=rc, fault=fault, stale_words=stale, workgroups_per_xcc=xcc,
               period_us=period, phase_a_body_us=work_a, phase_b_body_us=work_b, overhead_us=period - work_a - work_b, host_ms_per_search=float(ms.value))
    results.append(rec)
    print("%-58s %s work %4.1f + %4.1f us: period %6.2f us/simulation (bodies %5.2f + %5.2f, overhead %5.2f)  host %6.2f ms/search  rc %d fault %d stale %d  xcc %s"
          % (rec["mode"], "coop" if coop else "    ", work_a_us, work_b_us, period, work_a, work_b, period - work_a - work_b, ms.value, rc, fault, stale, xcc if mode != 1 else ""))
    return rec


for wa, wb in ((30, 10), (0, 0)):
    run(0, wa, wb)            # also fills the groups the launch form reuses
    run(1, wa, wb)
    run(2, wa, wb)
    run(0, wa, wb, coop=1)
if len(sys.argv) > 1:
    json.dump(dict(_how="tools/ubench/xcd_resident.py on one MI355X: 256 workgroups x 512 threads, 140 KB LDS (one per CU), 50 simulations, phase A streams 1.31 MB of "
                        "shared weights, phase B a 557 KB slice per workgroup; hand-off payloads 4.25 KB (A -> the 16 peers) and 8 KB (B -> 512 B to each peer); "
                        "period = (last phase-B end of the last simulation - first phase-A start of simulation 5) / 45; every consumed word checked",
                   runs=results), open(sys.argv[1], "w"), indent=1)
