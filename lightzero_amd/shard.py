"""Multi-GPU self-play: one process per GPU, env batches sharded by contiguous blocks, no collective inside a search.
Two exchange steps exist, both here:

* ``all_gather_rows``: pooling the finished env-step rows of every rank so that every rank (or the learner rank) can push them
  into its replay buffer -- one RCCL all-gather over xGMI per collect step (new relative to the reference, whose DDP ranks keep
  private buffers).  The row is the field set of ``GameSegment.append`` / ``store_search_stats``
  (lzero/mcts/buffer/game_segment.py:158-182, 241-263; arrays of ``game_segment_to_array`` :265-338): action, reward, searched
  root value, predicted value, to_play, timestep, visit entropy, child visits, action mask and the newest observation frame --
  for BASELINE configs[1] 8 + 2*6 + 96*96 float32 = 36.9 KB per env-step.  On the device the rows are written by one kernel
  (``lz_roots_collect_rows``, include/lz_mi355.h); ``pack_rows`` is the host twin for non-engine paths and tests.
* ``broadcast_state_dict``: the weight refresh after a learner update (checkpoint ``model`` state_dict,
  lzero/policy/muzero.py:1043-1047): one flat fp32 broadcast, then ``model.load_state_dict`` re-ingests in place.
"""
import os

import numpy as np

HEADER = 8  # words before the per-action blocks
F_ACTION, F_REWARD, F_ROOT_VALUE, F_PRED_VALUE, F_TO_PLAY, F_TIMESTEP, F_ENTROPY, F_N_LEGAL = range(HEADER)


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_envs, rank, world):
    """Contiguous env block of this rank; sizes differ by at most one."""
    base, rem = divmod(n_envs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


ROOTS_PER_GPU_FULL = {"conv": 256, "mlp": 1024}


def plan_gpus(total_envs, max_gpus, roots_per_gpu_full=None, family="conv"):
    """How many of the ``max_gpus`` devices a batch of ``total_envs`` roots should be sharded over.  The conv models' chain launch is ONE
    workgroup per root, so below one root per CU (256 on MI355X) a GPU's matrix pipes idle in proportion: measured on BASELINE
    configs[3] (Go 9x9, 200 simulations), 64 / 128 / 256 roots per GPU take 19.1 / 19.5 / 20.4 ms per step -- the same time for a
    quarter of the work -- and the chain runs at 0.22 / 0.86 of the fp32-matrix peak at 64 / 256 roots (profiles/r03_cfg3*.json).
    So: the fewest GPUs that still give every GPU at most ``roots_per_gpu_full`` roots ... unless that leaves GPUs that would each get a
    full share anyway.  -> (n_gpus, [roots per rank]); 512 envs on an 8-GPU node -> 2 GPUs x 256 (12.5k env-steps/s per GPU instead of
    8 x 3.4k), 2048 envs -> 8 x 256.
    ``family="mlp"`` (the vector-observation models: MuZeroModelMLP, EfficientZeroModelMLP, SampledEfficientZeroModelMLP -- BASELINE
    configs[0] / [4]): their recurrent inference is ten launches of 5-10 us each whatever the batch (launch-bound: 4.26 ms per
    50-simulation step for 64 roots, 4.75 ms for 256, profiles/r04_cfg4*.json), so a GPU is "full" only at ~1024 roots and configs[4]'s
    256 envs belong on ONE GPU, not on the 4 BASELINE.json names."""
    if roots_per_gpu_full is None:
        roots_per_gpu_full = ROOTS_PER_GPU_FULL[family]
    total_envs, max_gpus = int(total_envs), max(1, int(max_gpus))
    n = min(max_gpus, max(1, -(-total_envs // int(roots_per_gpu_full))))
    return n, [hi - lo for lo, hi in (shard_range(total_envs, q, n) for q in range(n))]


def row_width(action_space_size, frame_floats, extra_words=0):
    """``extra_words``: the block the sampled / Gumbel families add between the action mask and the frame (root_sampled_actions
    [K * D] / improved_policy_probs [A], game_segment.py:254-258); ``action_space_size`` = K for Sampled-EfficientZero rows"""
    return HEADER + 2 * action_space_size + extra_words + frame_floats


def pack_rows(output, action_mask, to_play, action_space_size, frames=None, timestep=None, extra_key=None):
    """Host twin of lz_roots_collect_rows(_ex): policy output dict (env_id -> dict, efficientzero.py:636-643) + the collector's
    per-env action mask / to_play (/ timestep) + the newest observation frame of every env -> float32 [n_env, W] rows ordered by
    env id.  ``frames``: [n_env, ...] or None.  ``extra_key``: 'root_sampled_actions' (Sampled EfficientZero: the row's visit block
    and mask are over the K sampled actions -- pass a [n, K] mask of ones --, word 0 is the selected POSITION) or
    'improved_policy_probs' (Gumbel MuZero): that entry of the output dict fills the extra block (game_segment.py:254-258)."""
    ids = sorted(output)
    A = action_space_size
    F = 0 if frames is None else int(np.prod(np.asarray(frames).shape[1:]))
    E = 0 if extra_key is None else int(np.asarray(output[ids[0]][extra_key]).size)
    rows = np.zeros((len(ids), row_width(A, F, E)), np.float32)
    for k, i in enumerate(ids):
        o = output[i]
        d = np.asarray(o["visit_count_distributions"], np.float32)
        s = np.float32(d.sum()) if d.sum() != 0 else np.float32(1e-6)  # game_segment.py:244-246
        if extra_key == "root_sampled_actions":   # the position of the chosen action among the root's sampled actions
            acts = np.asarray(o[extra_key], np.float32).reshape(len(d), -1)
            rows[k, F_ACTION] = int(np.nonzero((acts == np.asarray(o["action"], np.float32).reshape(1, -1)).all(1))[0][0])
        else:
            rows[k, F_ACTION] = o["action"]
        rows[k, F_ROOT_VALUE] = np.asarray(o["searched_value"]).reshape(-1)[0]
        rows[k, F_PRED_VALUE] = np.asarray(o["predicted_value"]).reshape(-1)[0]
        rows[k, F_TO_PLAY] = to_play[k] if np.ndim(to_play) else to_play
        rows[k, F_TIMESTEP] = -1 if timestep is None else timestep[k]
        rows[k, F_ENTROPY] = o.get("visit_count_distribution_entropy", 0.0)
        rows[k, F_N_LEGAL] = len(d)
        rows[k, HEADER:HEADER + len(d)] = d / s
        rows[k, HEADER + A:HEADER + 2 * A] = np.asarray(action_mask[k], np.float32)
        if E:
            rows[k, HEADER + 2 * A:HEADER + 2 * A + E] = np.asarray(o[extra_key], np.float32).reshape(-1)
        if F:
            rows[k, HEADER + 2 * A + E:] = np.asarray(frames[k], np.float32).reshape(-1)
    return rows


def unpack_rows(rows, action_space_size, frame_shape=None, extra_words=0):
    """[n, W] rows (numpy) -> dict of column arrays; child visits stay in legal-list order, padded with zeros"""
    rows = np.asarray(rows)
    A, E = action_space_size, int(extra_words)
    out = dict(action=rows[:, F_ACTION].astype(np.int64), reward=rows[:, F_REWARD].copy(), root_value=rows[:, F_ROOT_VALUE].copy(),
               predicted_value=rows[:, F_PRED_VALUE].copy(), to_play=rows[:, F_TO_PLAY].astype(np.int64),
               timestep=rows[:, F_TIMESTEP].astype(np.int64), entropy=rows[:, F_ENTROPY].copy(),
               n_legal=rows[:, F_N_LEGAL].astype(np.int64), child_visits=rows[:, HEADER:HEADER + A].copy(),
               action_mask=rows[:, HEADER + A:HEADER + 2 * A].copy())
    if E:
        out["extra"] = rows[:, HEADER + 2 * A:HEADER + 2 * A + E].copy()
    if rows.shape[1] > HEADER + 2 * A + E:
        fr = rows[:, HEADER + 2 * A + E:]
        out["frame"] = fr.reshape((rows.shape[0],) + tuple(frame_shape)) if frame_shape is not None else fr.copy()
    return out


def all_gather_rows(rows_t, async_op=False, counts=None):
    """rows_t: torch tensor [n, W] on this rank's device (cuda -> RCCL over xGMI, cpu -> gloo); n may differ between ranks
    (uneven env split: blocks are padded to the largest one for the collective and the padding is dropped again).
    ``counts``: the block sizes of all ranks when the caller knows them (``[hi - lo for lo, hi in (shard_range(n_envs, r, world) ...)]``,
    the steady state of a collector): no size exchange and no host synchronisation then; without it the sizes are all-gathered first.
    Returns [sum of n over ranks, W] in rank order.  ``async_op=True`` returns (work, finish) instead: wait on ``work``
    (or just call ``finish()``, which waits) -- the collective of step i then overlaps the search of step i + 1 on the
    engine's own stream."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or _single_rank(dist):
        return (None, lambda: rows_t) if async_op else rows_t
    world = dist.get_world_size()
    if counts is None:
        n = torch.tensor([rows_t.shape[0]], dtype=torch.int64, device=rows_t.device)
        cts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(cts, n)
        counts = [int(c.item()) for c in cts]
    else:
        counts = [int(c) for c in counts]
        assert len(counts) == world and counts[dist.get_rank()] == rows_t.shape[0]
    nmax = max(counts)
    send = rows_t.contiguous()
    if send.shape[0] < nmax:
        pad = torch.zeros((nmax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    out = torch.empty((world * nmax,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
    work = dist.all_gather_into_tensor(out, send, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()
        if all(c == nmax for c in counts):
            return out
        return torch.cat([out[r * nmax:r * nmax + counts[r]] for r in range(world)], 0)
    return (work, finish) if async_op else finish()


def _single_rank(dist):
    """a one-rank group needs no collective -- unless LZ_FORCE_COLLECTIVE=1 asks for it (tests on a 1-GPU box: the RCCL call sequence
    of the N > 1 path with N = 1)"""
    return dist.get_world_size() == 1 and not os.environ.get("LZ_FORCE_COLLECTIVE")


def all_gather_rows_equal(rows_t, out=None, async_op=False):
    """The steady-state form for equal blocks (weak scaling: every GPU owns the same number of envs): no size exchange, the
    output buffer can be pre-allocated and the work handle returned for overlap."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or _single_rank(dist):
        return rows_t, None
    if out is None:
        out = torch.empty((dist.get_world_size() * rows_t.shape[0],) + tuple(rows_t.shape[1:]), dtype=rows_t.dtype, device=rows_t.device)
    work = dist.all_gather_into_tensor(out, rows_t, async_op=async_op)
    return out, work


class FlatStateDict(dict):
    """name -> views of ONE flat float32 buffer in name order (what ``broadcast_state_dict(on_device=True)`` returns): ``flat`` is that
    buffer, ``layout`` the (name, offset, size) triples.  ``model.load_state_dict`` recognises it and hands the buffer to the library by
    pointer without walking the tensors (lz_model_refresh_flat)."""
    flat = None
    layout = ()


def flat_state_dict(state_dict, device="cuda"):
    """the ``FlatStateDict`` of a reference-keyed state_dict on ``device``: one concatenation + one upload (a learner in the same process
    can hand its weights to the collector's engine model this way; across ranks ``broadcast_state_dict(on_device=True)`` builds the same)"""
    import torch
    names = sorted(k for k in state_dict if not k.endswith("num_batches_tracked"))

    def t(v):
        return v.detach().reshape(-1).float() if hasattr(v, "detach") else torch.from_numpy(np.ascontiguousarray(np.asarray(v), dtype=np.float32).reshape(-1))
    flat = torch.cat([t(state_dict[k]).to(device) for k in names]) if any(getattr(state_dict[k], "is_cuda", False) for k in names) \
        else torch.cat([t(state_dict[k]) for k in names]).to(device)
    return _flat_views(flat, names, [tuple(np.shape(state_dict[k])) for k in names])


def _flat_views(flat, names, shapes):
    out, off, lay = FlatStateDict(), 0, []
    for k, sh in zip(names, shapes):
        n = int(np.prod(sh)) if len(sh) else 1
        out[k] = flat[off:off + n].view(*sh) if len(sh) else flat[off:off + n].view(())
        lay.append((k, off, n))
        off += n
    out.flat, out.layout = flat, tuple(lay)
    return out


def broadcast_state_dict(state_dict, src=0, device=None, on_device=False):
    """Weight refresh across ranks: rank ``src`` holds the new ``state_dict`` (name -> array-like, reference names), every other
    rank passes its current one (same names and shapes; only used as the layout).  One flat float32 broadcast (RCCL on cuda
    tensors, gloo on cpu).  Returns name -> numpy float32 arrays, ready for ``model.load_state_dict`` (in-place re-ingest) -- or,
    with ``on_device=True`` and an RCCL broadcast, name -> views of the flat CUDA buffer: ``load_state_dict`` then hands them to the
    library by pointer (``lz_model_set_tensor_device``), no ``.cpu()`` and no per-tensor numpy copies on the Python side."""
    import torch
    import torch.distributed as dist
    if on_device and isinstance(state_dict, FlatStateDict) and getattr(state_dict.flat, "is_cuda", False) \
            and (not dist.is_available() or not dist.is_initialized() or _single_rank(dist)):
        return state_dict        # one rank, already one flat device buffer: nothing to broadcast, nothing to walk
    names = sorted(k for k in state_dict if not k.endswith("num_batches_tracked"))

    def arr(v):
        return np.ascontiguousarray(v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v), dtype=np.float32)
    shapes = [tuple(np.shape(state_dict[k])) for k in names]
    sizes = [int(np.prod(s)) if len(s) else 1 for s in shapes]
    if not dist.is_available() or not dist.is_initialized() or _single_rank(dist):
        if on_device and torch.cuda.is_available():
            if isinstance(state_dict, FlatStateDict) and getattr(state_dict.flat, "is_cuda", False):
                return state_dict        # already one flat device buffer (nothing to broadcast to)
            return flat_state_dict(state_dict, "cuda")
        return {k: arr(state_dict[k]) for k in names}
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    def still_views(sd):
        """every entry is still the view of sd.flat it was built as (an entry re-assigned afterwards would ship a stale buffer: ADVICE r5)"""
        off = 0
        for k, n in zip(names, sizes):
            v = sd[k]
            if not hasattr(v, "data_ptr") or v.data_ptr() != sd.flat.data_ptr() + 4 * off or v.numel() != n:
                return False
            off += n
        return True
    if isinstance(state_dict, FlatStateDict) and state_dict.flat is not None and state_dict.flat.device.type == torch.device(device).type \
            and state_dict.flat.numel() == sum(sizes) and still_views(state_dict):
        # already one flat buffer in name order on the collective's device (a learner's flat_state_dict, or the result of the previous
        # broadcast on the receiving ranks): broadcast it in place -- no concatenation, no staging
        flat = state_dict.flat
        # ... but an engine model that took this buffer by pointer copies out of it asynchronously on ITS stream (lz_model_refresh_flat);
        # the collective is ordered on torch's stream only: wait for the model's "consumed" event before overwriting the buffer
        ev = getattr(state_dict, "consumed_event", None)
        if ev is not None and flat.is_cuda:
            torch.cuda.current_stream(flat.device).wait_event(ev)
    else:
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
        if dist.get_rank() == src:
            flat.copy_(torch.from_numpy(np.concatenate([arr(state_dict[k]).reshape(-1) for k in names])))
    dist.broadcast(flat, src=src)
    if on_device and flat.is_cuda:
        return _flat_views(flat, names, shapes)
    host = flat.cpu().numpy()
    out, off = {}, 0
    for k, s, n in zip(names, shapes, sizes):
        out[k] = host[off:off + n].reshape(s).copy()
        off += n
    return out
