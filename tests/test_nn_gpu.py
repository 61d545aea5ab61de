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
from parity_util import hinv_ieee

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
    # 3 = tracing + head debug buffers: EVERY simulation's support-wide logits and pre-transform expectations, whichever kernel
    # finished the heads -- 49 of the 50 production simulations finish them in the next chain launch's prologue (split heads)
    L.check(lib.lz_roots_enable_trace(roots._h, 3))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    L.check(lib.lz_engine_synchronize(dev.engine))
    trace = np.zeros((S, B, 4), np.int32)
    L.check(lib.lz_roots_read_trace(roots._h, S, trace.reshape(-1)))
    lat = np.zeros((S + 1, B, 64, 6, 6), np.float32); hh = np.zeros((S + 1, B, 512), np.float32); cc = np.zeros_like(hh)
    vp = np.zeros((S + 1, B), np.float32); val = np.zeros_like(vp); pol = np.zeros((S + 1, B, A), np.float32)
    vlog = np.zeros((S + 1, B, 601), np.float32); rlog = np.zeros_like(vlog); vexp = np.zeros((S + 1, B), np.float32); rexp = np.zeros_like(vexp)
    for s in range(S + 1):
        L.check(lib.lz_roots_read_latent(roots._h, s, lat[s].reshape(-1)))
        L.check(lib.lz_roots_read_hidden(roots._h, s, hh[s].reshape(-1), cc[s].reshape(-1)))
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp[s], val[s], pol[s].reshape(-1)))
        if s >= 1:
            L.check(lib.lz_roots_read_head_debug(roots._h, s, 0, vlog[s].ctypes.data, vexp[s].ctypes.data))
            L.check(lib.lz_roots_read_head_debug(roots._h, s, 1, rlog[s].ctypes.data, rexp[s].ctypes.data))
    vlast = np.zeros((B, 601), np.float32); rlast = np.zeros((B, 601), np.float32)   # the legacy view: the LAST head launch (k_heads_mm)
    L.check(lib.lz_roots_read_debug_logits(roots._h, 0, vlast.reshape(-1)))
    L.check(lib.lz_roots_read_debug_logits(roots._h, 1, rlast.reshape(-1)))
    assert np.array_equal(vlast, vlog[S]) and np.array_equal(rlast, rlog[S])
    ist = tm.InverseScalarTransform()
    support = np.arange(-300, 301, dtype=np.float64)
    ar = np.arange(B)
    worst = dict(latent=0.0, h=0.0, c=0.0, policy=0.0, value_prefix=0.0, value=0.0, logits=0.0, logits_split=0.0, expect=0.0, expect_split=0.0)
    # the pre-transform expectation softmax . support: what fp32 itself loses on it (torch fp32 and the device against a binary64
    # evaluation of the SAME fp32 logits of each side), so that the 1e-5 the device is held to can be read against torch's own distance
    f64 = dict(torch_fp32=0.0, device=0.0)
    hinv_on_own_expect_equal = 0   # post-transform scalars == the IEEE evaluation of h^-1 on the DEVICE's expectation, bit for bit (lz_hinv.h)
    scalars_bit_equal = 0          # post-transform scalars == the reference pipeline's, bit for bit
    n_scal = 0
    for s in range(S):
        ix, act, slen = trace[s, :, 0], trace[s, :, 1], trace[s, :, 2]
        assert (ix <= s).all() and (act >= 0).all() and (act < A).all()
        with torch.no_grad():
            o = ref.recurrent_inference(torch.from_numpy(lat[ix, ar]),
                                        (torch.from_numpy(hh[ix, ar]).unsqueeze(0), torch.from_numpy(cc[ix, ar]).unsqueeze(0)),
                                        torch.from_numpy(act).long())
            r_vp = ist(o.value_prefix).reshape(-1).numpy(); r_val = ist(o.value).reshape(-1).numpy()
            rh = o.reward_hidden_state[0][0].numpy().copy(); rc = o.reward_hidden_state[1][0].numpy().copy()
            # torch's own pre-transform expectation (scaling_transform.py:84-85)
            t_vexp = torch.softmax(o.value, 1).mul_(ist.value_support).sum(1).numpy(); t_rexp = torch.softmax(o.value_prefix, 1).mul_(ist.value_support).sum(1).numpy()
        reset = (slen % 5 == 0)
        rh[reset] = 0; rc[reset] = 0  # mcts_ctree.py:859-863
        split = s < S - 1             # every simulation but the last leaves its heads to the next chain launch
        worst["latent"] = max(worst["latent"], _reldiff(lat[s + 1], o.latent_state.numpy()))
        worst["h"] = max(worst["h"], _reldiff(hh[s + 1], rh)); worst["c"] = max(worst["c"], _reldiff(cc[s + 1], rc))
        worst["policy"] = max(worst["policy"], _reldiff(pol[s + 1], o.policy_logits.numpy()))
        # scalars after h^-1: relative to 1 + |x| (the transform's own fp32 quantisation grows with |x|)
        worst["value_prefix"] = max(worst["value_prefix"], _reldiff(vp[s + 1], r_vp))
        worst["value"] = max(worst["value"], _reldiff(val[s + 1], r_val))
        dl = max(_reldiff(vlog[s + 1], o.value.numpy()), _reldiff(rlog[s + 1], o.value_prefix.numpy()))
        de = max(_reldiff(vexp[s + 1], t_vexp), _reldiff(rexp[s + 1], t_rexp))
        worst["logits_split" if split else "logits"] = max(worst["logits_split" if split else "logits"], dl)
        worst["expect_split" if split else "expect"] = max(worst["expect_split" if split else "expect"], de)
        for lg32, dev_e, tor_e in ((o.value.numpy(), vexp[s + 1], t_vexp), (o.value_prefix.numpy(), rexp[s + 1], t_rexp)):
            z = lg32.astype(np.float64)
            p = np.exp(z - z.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
            e64 = p @ support
            f64["torch_fp32"] = max(f64["torch_fp32"], _reldiff(tor_e, e64))
        for lgd, dev_e in ((vlog[s + 1], vexp[s + 1]), (rlog[s + 1], rexp[s + 1])):
            z = lgd.astype(np.float64)
            p = np.exp(z - z.max(1, keepdims=True)); p /= p.sum(1, keepdims=True)
            f64["device"] = max(f64["device"], _reldiff(dev_e, p @ support))
        # the device's h^-1 is the reference formula evaluated in IEEE binary32, bit for bit (tests/test_hinv_gpu.py): on the production path too
        own_val, own_vp = hinv_ieee(vexp[s + 1]), hinv_ieee(rexp[s + 1])
        assert np.array_equal(own_val.view(np.uint32), val[s + 1].view(np.uint32)) and np.array_equal(own_vp.view(np.uint32), vp[s + 1].view(np.uint32)), \
            "simulation %d: the stored scalars are not the IEEE evaluation of h^-1 on the device's own expectation" % s
        hinv_on_own_expect_equal += 2 * B
        scalars_bit_equal += int((val[s + 1].view(np.uint32) == r_val.view(np.uint32)).sum() + (vp[s + 1].view(np.uint32) == r_vp.view(np.uint32)).sum())
        n_scal += 2 * B
    print("B = %d, S = %d: worst |d| / (1 + |x|):" % (B, S), worst)
    print("   pre-transform expectation against binary64 on each side's own logits:", f64,
          "; post-transform scalars bit-equal to the reference pipeline's: %d of %d" % (scalars_bit_equal, n_scal))
    parity_record.check("recurrent_teacher_forced/ez_atari96/B%d_S%d" % (B, S), worst,
                        extra=dict(batch=B, simulations=S, split_head_simulations=S - 1, expect_vs_binary64=f64,
                                   post_transform_scalars_bit_equal_fraction=scalars_bit_equal / n_scal,
                                   post_transform_scalars_equal_ieee_hinv_of_device_expectation=hinv_on_own_expect_equal / n_scal))
    dist = np.array(roots.get_distributions())
    assert (dist.sum(1) == S).all()
