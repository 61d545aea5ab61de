"""CPU: the N>1 path -- contiguous env sharding, the env-step-row all-gather (uneven blocks, asynchronous form) and the weight
refresh broadcast -- with gloo, world size 2."""
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
# ---- uneven env split (11 envs over 2 ranks: 6 + 5): the blocks are padded for the collective and stripped again
N, A, F = 11, 6, 12
lo, hi = shard.shard_range(N, rank, world)
def outputs(i):
    return dict(action=i %% A, searched_value=0.5 * i, predicted_value=np.array([-0.25 * i], np.float32),
                visit_count_distribution_entropy=0.125 * i, visit_count_distributions=[i, 1, 2][: 2 + (i %% 2)])
def mask(i):
    m = np.zeros(A, np.float32); m[: 2 + (i %% 2)] = 1
    return m
out = {i: outputs(i) for i in range(lo, hi)}
frames = np.stack([np.full(F, i, np.float32) for i in range(lo, hi)])
rows = torch.from_numpy(shard.pack_rows(out, [mask(i) for i in range(lo, hi)], [1 + i %% 2 for i in range(lo, hi)], A, frames=frames,
                                        timestep=list(range(lo, hi))))
assert rows.shape == (hi - lo, shard.row_width(A, F))
allrows = shard.all_gather_rows(rows)
assert allrows.shape == (N, shard.row_width(A, F)), allrows.shape
cols = shard.unpack_rows(allrows.numpy(), A, frame_shape=(F,))
ids = np.arange(N)
assert cols["action"].tolist() == (ids %% A).tolist()
assert cols["root_value"].tolist() == (0.5 * ids).tolist() and cols["predicted_value"].tolist() == (-0.25 * ids).tolist()
assert cols["n_legal"].tolist() == (2 + ids %% 2).tolist() and cols["to_play"].tolist() == (1 + ids %% 2).tolist()
assert cols["timestep"].tolist() == ids.tolist()
assert (cols["frame"] == ids[:, None]).all()
assert np.allclose(cols["child_visits"].sum(1), 1.0) and (cols["action_mask"].sum(1) == cols["n_legal"]).all()
# the asynchronous form (collective of step i overlapped with the work of step i + 1)
work, finish = shard.all_gather_rows(rows, async_op=True)
assert torch.equal(finish(), allrows)
# block sizes known to the caller: no size exchange
known = [b - a for a, b in (shard.shard_range(N, r, world) for r in range(world))]
assert torch.equal(shard.all_gather_rows(rows, counts=known), allrows)
# equal blocks, pre-allocated output
eq = rows[:5].contiguous()
buf, w = shard.all_gather_rows_equal(eq, async_op=True)
w.wait()
assert buf.shape[0] == 10 and torch.equal(buf[:5] if rank == 0 else buf[5:], eq)
# ---- weight refresh: rank 0 holds the new state_dict, rank 1 a stale one of the same layout
rng = np.random.default_rng(5)
new = {"a.weight": rng.standard_normal((3, 4)).astype(np.float32), "a.bias": rng.standard_normal(3).astype(np.float32),
       "bn.running_var": rng.random(3).astype(np.float32), "bn.num_batches_tracked": np.array(7)}
mine = new if rank == 0 else {k: np.zeros_like(v) for k, v in new.items()}
got = shard.broadcast_state_dict(mine, src=0)
assert sorted(got) == ["a.bias", "a.weight", "bn.running_var"]
assert all(np.array_equal(got[k], new[k]) and got[k].dtype == np.float32 for k in got)
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


def test_plan_gpus_keeps_a_full_share_per_gpu():
    """BASELINE configs[3] (512 envs, 8 GPUs available) -> 2 GPUs x 256 roots; configs[4] (256 envs, 4 GPUs) -> 1 GPU; bigger batches
    spread over the whole node, uneven totals by shard_range"""
    from lightzero_amd import shard
    assert shard.plan_gpus(512, 8) == (2, [256, 256])
    assert shard.plan_gpus(256, 4) == (1, [256])
    assert shard.plan_gpus(2048, 8) == (8, [256] * 8)
    assert shard.plan_gpus(4096, 8) == (8, [512] * 8)
    # the launch-bound vector-observation families: BASELINE configs[4] (256 envs "on 4 GPUs") belongs on one
    assert shard.plan_gpus(256, 4, family="mlp") == (1, [256])
    assert shard.plan_gpus(1024, 4, family="mlp") == (1, [1024])
    assert shard.plan_gpus(4096, 8, family="mlp") == (4, [1024] * 4)
    assert shard.plan_gpus(700, 8) == (3, [234, 233, 233])
    assert shard.plan_gpus(10, 8) == (1, [10])
