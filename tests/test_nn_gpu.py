"""GPU parity of the network kernels (through the C ABI) against the torch fp32 restatement of the
reference model (oracle/torch_models.py), on shared seeded weights.

Tolerances (fp32 everywhere; only summation order differs; the table and the measured worst cases: tests/parity_record.py):
  * latent / LSTM state / policy logits / support-wide logits: |d| <= 1e-5 (1 + |x|)  (north_star's bound)
  * value / value-prefix scalars after h^-1: |d| <= 3e-4 (1 + |x|).  The reference's own fp32 formula
    sign(x)(((sqrt(1+4e(|x|+1+e))-1)/(2e))^2-1) (scaling_transform.py:88-91) subtracts 1 from a number
    ~1.004 and divides by 0.002, so its OUTPUT is quantised in steps of ~1.3e-4 near 0; two correct fp32
    pipelines whose inputs differ by 1e-6 can land on neighbouring steps.  The pre-transform logits are
    compared at 1e-5.
"""
import numpy as np
import pytest
import torch

import parity_record

pytestmark = pytest.mark.gpu


def _setup(B, A=6, S=12, seed=0):
    from oracle import torch_models as tm
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=seed)
    dev = EfficientZeroModel(action_space_size=A).load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(seed + 1)
    obs = torch.rand(B, 4, 96, 96, generator=g)
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=dev.engine)
    roots.set_tiebreak(0)
    roots._ensure(A)
    return ref, dev, obs, roots


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def _reldiff(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


@pytest.mark.parametrize("B", [8, 37, 256])
def test_initial_inference_matches_torch(B):
    from lightzero_amd import _lib as L
    from oracle import torch_models as tm
    ref, dev, obs, roots = _setup(B)
    d_obs = obs.cuda().contiguous()
    torch.cuda.synchronize()
    L.check(L.lib().lz_roots_enable_trace(roots._h, 1))  # the heads write their support-wide logits only while tracing
    L.check(L.lib().lz_initial_inference(roots._h, d_obs.data_ptr()))
    lat = np.zeros((B, 64, 6, 6), np.float32)
    L.check(L.lib().lz_roots_read_latent(roots._h, 0, lat.reshape(-1)))
    val = np.zeros(B, np.float32); pol = np.zeros((B, 6), np.float32)
    L.check(L.lib().lz_roots_get_root_outputs(roots._h, val, pol.reshape(-1)))
    vlog = np.zeros((B, 601), np.float32)
    L.check(L.lib().lz_roots_read_debug_logits(roots._h, 0, vlog.reshape(-1)))
    with torch.no_grad():
        o = ref.initial_inference(obs)
        rv = tm.InverseScalarTransform()(o.value).reshape(-1).numpy()
    parity_record.check("initial_inference/ez_atari96/B%d" % B,
                        dict(latent=_reldiff(lat, o.latent_state.numpy()), policy=_reldiff(pol, o.policy_logits.numpy()),
                             logits=_reldiff(vlog, o.value.numpy()), value=_reldiff(val, rv)), extra=dict(batch=B))


@pytest.mark.parametrize("B,S", [(16, 12), (67, 12), (256, 50)])
def test_recurrent_inference_matches_torch_teacher_forced(B, S):
    """Every simulation of a device search -- run as the benchmark runs it: captured HIP graph, tree step fused into the chain
    launch (k_chain_w<6,6,8,false,1>), the trace being one D2D copy per simulation inside the graph: feed the torch model the SAME
    (latent, h, c, action) the device gathered and compare everything the step produced, support-wide logits of the last
    simulation included.  B = 256, S = 50 is BASELINE configs[1] itself (16 LSTM row tiles x 32 unit tiles, 64 head workgroups);
    B = 67 leaves a partial 16-row LSTM tile and a partial 4-root head workgroup."""
    from lightzero_amd import _lib as L
    from oracle import torch_models as tm
    A = 6
    ref, dev, obs, roots = _setup(B, A, S)
    d_obs = obs.cuda().contiguous()
    torch.cuda.synchronize()
    lib = L.lib()
    L.check(lib.lz_initial_inference(roots._h, d_obs.data_ptr()))
    rng = np.random.default_rng(0)
    noises = rng.dirichlet([0.3] * A, size=B).astype(np.float32).reshape(-1)
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, noises.ctypes.data, L.i32([-1] * B)))
    L.check(lib.lz_roots_enable_trace(roots._h, 1))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    L.check(lib.lz_engine_synchronize(dev.engine))
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    lat = np.zeros((S + 1, B, 64, 6, 6), np.float32); hh = np.zeros((S + 1, B, 512), np.float32); cc = np.zeros_like(hh)
    vp = np.zeros((S + 1, B), np.float32); val = np.zeros_like(vp); pol = np.zeros((S + 1, B, A), np.float32)
    for s in range(S + 1):
        L.check(lib.lz_roots_read_latent(roots._h, s, lat[s].reshape(-1)))
        L.check(lib.lz_roots_read_hidden(roots._h, s, hh[s].reshape(-1), cc[s].reshape(-1)))
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp[s], val[s], pol[s].reshape(-1)))
    vlog = np.zeros((B, 601), np.float32); rlog = np.zeros((B, 601), np.float32)   # of the LAST head launch
    L.check(lib.lz_roots_read_debug_logits(roots._h, 0, vlog.reshape(-1)))
    L.check(lib.lz_roots_read_debug_logits(roots._h, 1, rlog.reshape(-1)))
    ist = tm.InverseScalarTransform()
    ar = np.arange(B)
    worst = dict(latent=0.0, h=0.0, c=0.0, policy=0.0, value_prefix=0.0, value=0.0, logits=0.0)
    for s in range(S):
        ix, act, slen = trace[s, :, 0], trace[s, :, 1], trace[s, :, 2]
        assert (ix <= s).all() and (act >= 0).all() and (act < A).all()
        with torch.no_grad():
            o = ref.recurrent_inference(torch.from_numpy(lat[ix, ar]),
                                        (torch.from_numpy(hh[ix, ar]).unsqueeze(0), torch.from_numpy(cc[ix, ar]).unsqueeze(0)),
                                        torch.from_numpy(act).long())
            r_vp = ist(o.value_prefix).reshape(-1).numpy(); r_val = ist(o.value).reshape(-1).numpy()
            rh = o.reward_hidden_state[0][0].numpy().copy(); rc = o.reward_hidden_state[1][0].numpy().copy()
        reset = (slen % 5 == 0)
        rh[reset] = 0; rc[reset] = 0  # mcts_ctree.py:859-863
        worst["latent"] = max(worst["latent"], _reldiff(lat[s + 1], o.latent_state.numpy()))
        worst["h"] = max(worst["h"], _reldiff(hh[s + 1], rh)); worst["c"] = max(worst["c"], _reldiff(cc[s + 1], rc))
        worst["policy"] = max(worst["policy"], _reldiff(pol[s + 1], o.policy_logits.numpy()))
        # scalars after h^-1: relative to 1 + |x| (the transform's own fp32 quantisation grows with |x|)
        worst["value_prefix"] = max(worst["value_prefix"], _reldiff(vp[s + 1], r_vp))
        worst["value"] = max(worst["value"], _reldiff(val[s + 1], r_val))
        if s == S - 1:
            worst["logits"] = max(_reldiff(vlog, o.value.numpy()), _reldiff(rlog, o.value_prefix.numpy()))
    print("B = %d, S = %d: worst |d| / (1 + |x|):" % (B, S), worst)
    parity_record.check("recurrent_teacher_forced/ez_atari96/B%d_S%d" % (B, S), worst, extra=dict(batch=B, simulations=S))
    dist = np.array(roots.get_distributions())
    assert (dist.sum(1) == S).all()
