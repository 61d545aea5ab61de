"""CPU: the N>1 path -- contiguous env sharding and the trajectory-row all-gather -- with gloo, world size 2."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from lightzero_amd import shard
dist.init_process_group("gloo")
rank, world = shard.rank_world()
lo, hi = shard.shard_range(10, rank, world)
A = 6
out = {i: dict(action=i %% A, searched_value=0.5 * i, predicted_value=-0.25 * i,
               visit_count_distributions=[i, 1, 2][: 2 + (i %% 2)]) for i in range(lo, hi)}
rows = torch.from_numpy(shard.pack_rows(out, A))
allrows = shard.all_gather_rows(rows)
assert allrows.shape == (10, 4 + A), allrows.shape
exp_ids = list(range(10))
assert allrows[:, 1].tolist() == [0.5 * i for i in exp_ids]
assert allrows[:, 0].tolist() == [float(i %% A) for i in exp_ids]
assert allrows[:, 3].tolist() == [float(2 + (i %% 2)) for i in exp_ids]
if rank == 0:
    print("OK")
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_covers_everything():
    from lightzero_amd import shard
    for n in (1, 7, 256, 513):
        for w in (1, 2, 3, 8):
            spans = [shard.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_all_gather_rows_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK" in r.stdout
