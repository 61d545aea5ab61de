"""Multi-GPU self-play: one process per GPU, env batches sharded by contiguous blocks, no collective
inside a search.  The only exchange step is pooling finished trajectory rows so that every rank (or
the learner rank) can push them into its replay buffer -- one RCCL all-gather per collect interval
(new relative to the reference, whose DDP ranks keep private buffers).

Row schema (float32, one row per env-step): [action, searched_value, predicted_value, n_legal,
visit_count[0..A-1]] -- the fields MuZeroCollector stores per step (muzero_collector.py:557-568,
game_segment.py:241-263).
"""
import os

import numpy as np


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_envs, rank, world):
    """Contiguous env block of this rank; sizes differ by at most one."""
    base, rem = divmod(n_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_rows(output, action_space_size):
    """policy output dict (env_id -> dict) -> float32 [n_env, 4 + A] rows, ordered by env id."""
    ids = sorted(output)
    rows = np.zeros((len(ids), 4 + action_space_size), np.float32)
    for k, i in enumerate(ids):
        o = output[i]
        d = o["visit_count_distributions"]
        rows[k, 0] = o["action"]
        rows[k, 1] = o["searched_value"]
        rows[k, 2] = o["predicted_value"]
        rows[k, 3] = len(d)
        rows[k, 4:4 + len(d)] = d
    return rows


def all_gather_rows(rows_t):
    """rows_t: torch tensor [n, W] on this rank's device (cuda -> RCCL over xGMI, cpu -> gloo).
    Returns [world * n, W].  Equal n on every rank (pad the last block if the split is uneven)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return rows_t
    out = torch.empty((dist.get_world_size() * rows_t.shape[0],) + tuple(rows_t.shape[1:]), dtype=rows_t.dtype,
                      device=rows_t.device)
    dist.all_gather_into_tensor(out, rows_t.contiguous())
    return out
