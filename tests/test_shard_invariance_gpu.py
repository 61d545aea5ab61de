"""GPU: shard invariance -- what an 8-GPU run rests on (contiguous env blocks, no collective inside a search).
Rank q's block of a sharded run must give BIT-IDENTICAL visit counts, root values, per-simulation selections, network outputs and
env-step rows to the same envs inside the single-rank batch: a root's result may not depend on its position in the batch (LSTM 16-row
tiles, 4-root head workgroups, 16-pixel tiles of the tower, partial tiles at the block ends) nor on the batch size.  Deterministic
tie-break, host-supplied Dirichlet noise and arg-max action selection (the device-side random streams are keyed by the root's index in
its OWN batch by design: every rank explores differently).  One process, one GPU: the blocks run one after the other on the engine the
full batch ran on -- the kernels cannot tell."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, horizon=5, delta=0.01)


def _run(model, obs, noise, S, A, frame):
    from lightzero_amd import _lib as L, shard
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    lib = L.lib()
    B = obs.shape[0]
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    roots._ensure(A)
    d_obs = obs.cuda().contiguous()
    torch.cuda.synchronize()
    L.check(lib.lz_roots_enable_trace(roots._h, 1))
    L.check(lib.lz_initial_inference(roots._h, d_obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, np.ascontiguousarray(noise, np.float32).ctypes.data, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount"], CFG["horizon"], CFG["delta"]))
    dist = np.array(roots.get_distributions())
    val = np.array(roots.get_values(), np.float32)
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    vp = np.zeros((S + 1, B), np.float32); v = np.zeros_like(vp); pol = np.zeros((S + 1, B, A), np.float32)
    for s in range(S + 1):
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp[s], v[s], pol[s].reshape(-1)))
    lat = np.zeros((B, 64, 6, 6), np.float32)
    L.check(lib.lz_roots_read_latent(roots._h, S, lat.reshape(-1)))
    W = shard.row_width(A, frame)
    rows = torch.zeros(B, W, device="cuda")
    torch.cuda.synchronize()   # the engine writes the rows on its own (non-blocking) stream: torch's zero-fill must have landed first
    hdr = np.zeros((B, shard.HEADER + 2 * A), np.float32); lg = np.zeros((B, A), np.float32)
    ts = np.arange(B, dtype=np.int32) * 0 + 7
    L.check(lib.lz_roots_collect_rows(roots._h, 1.0, 1, 12345, None, frame, ts.ctypes.data, rows.data_ptr(), W, hdr, lg.ctypes.data))
    return dict(dist=dist, val=val, trace=trace, vp=vp, v=v, pol=pol, lat=lat, rows=rows.cpu().numpy(), hdr=hdr, logits=lg)


def _bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("total,world,S", [(256, 2, 50), (256, 8, 20), (129, 2, 20), (131, 4, 12)])
def test_rank_block_is_bit_identical_to_the_same_envs_in_the_single_rank_batch(total, world, S):
    from oracle import torch_models as tm
    from lightzero_amd import shard
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    A, frame = 6, 96 * 96
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A))
    model = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    obs = torch.rand(total, 4, 96, 96, generator=torch.Generator().manual_seed(21))
    noise = np.random.default_rng(3).dirichlet([0.3] * A, size=total).astype(np.float32)
    full = _run(model, obs, noise, S, A, frame)
    assert (full["dist"].sum(1) == S).all()
    for q in range(world):
        lo, hi = shard.shard_range(total, q, world)
        part = _run(model, obs[lo:hi], noise[lo:hi], S, A, frame)
        for k in ("dist", "val", "hdr", "logits", "rows", "lat"):
            assert np.array_equal(_bits(part[k]), _bits(full[k][lo:hi])), "rank %d of %d (envs %d..%d): %s differs" % (q, world, lo, hi, k)
        for k in ("trace", "vp", "v", "pol"):   # [S(+1)][B]...
            assert np.array_equal(_bits(part[k]), _bits(full[k][:, lo:hi])), "rank %d of %d (envs %d..%d): %s differs" % (q, world, lo, hi, k)
