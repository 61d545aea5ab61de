"""GPU: FAST MODE (lz_model_cfg.precision = 1, ``EfficientZeroModel(fast_mode=True)``; BASELINE.md section 2, last arm) -- the recurrent
chain's 3x3 convolutions and the LSTM gate product on bf16 MFMA (k_chain_b / k_lstm_b), everything else fp32.  NOT a parity mode: the
claims here are statistical and every bound is a bf16 bound, written next to the assertion.  What is asserted:
  * single recurrent steps stay within bf16 round-off of the fp32 engine on identical inputs (next latent, LSTM state, logits);
  * the fast engine equals a torch fp32 model whose multiplied operands are rounded to bf16 at the same places (the oracle with an
    emulation of the rounding: the kernel's arithmetic is what its header says, not merely "close");
  * a full 256 x 50 search agrees with the parity-mode search statistically (root values, visit distributions, chosen actions);
  * the tree arithmetic is untouched: the fast search replays bit-exactly through the CPU oracle tree on its own network outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

A = 6
CFG = dict(pb_c_base=19652, pb_c_init=1.25, discount=0.997, horizon=5, delta=0.01)


def _models():
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    ref = tm.synthetic_init(tm.EfficientZeroModel(action_space_size=A), seed=5)
    sd = ref.state_dict()
    par = EfficientZeroModel(action_space_size=A, engine=L.new_engine(0)).load_state_dict(sd)
    fast = EfficientZeroModel(action_space_size=A, engine=L.new_engine(0), fast_mode=True).load_state_dict(sd)
    return ref, par, fast


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-12))


def test_recurrent_step_is_within_bf16_roundoff_of_the_fp32_engine():
    ref, par, fast = _models()
    g = torch.Generator().manual_seed(2)
    B = 37
    obs = torch.rand(B, 4, 96, 96, generator=g)
    with torch.no_grad():
        lat = ref.initial_inference(obs).latent_state   # a realistic latent (post-ReLU, BatchNorm scale)
    h = (torch.randn(1, B, 512, generator=g) * 0.3, torch.randn(1, B, 512, generator=g) * 0.3)
    act = torch.randint(0, A, (B,), generator=g)
    op, of = par.recurrent_inference(lat, h, act), fast.recurrent_inference(lat, h, act)
    # bf16 has 8 significand bits: a K = 576 dot product of operands rounded to 2^-9 relative, accumulated in fp32, moves by ~2^-9 x
    # sqrt(K) x |term| -- a few 1e-3 of the tensor's scale per layer, six layers deep
    assert _rel(of.latent_state, op.latent_state) < 3e-2
    assert _rel(of.reward_hidden_state[0], op.reward_hidden_state[0]) < 3e-2
    assert _rel(of.reward_hidden_state[1], op.reward_hidden_state[1]) < 3e-2
    assert _rel(of.policy_logits, op.policy_logits) < 5e-2
    assert _rel(of.value, op.value) < 5e-2
    assert _rel(of.value_prefix, op.value_prefix) < 5e-2
    # and it IS a different arithmetic (the switch reached the kernels)
    assert not np.array_equal(of.latent_state.numpy(), op.latent_state.numpy())


def test_initial_inference_is_within_bf16_roundoff_of_the_fp32_engine():
    """the representation tower on k_conv_bf (eleven 3x3 convolutions deep) + the chain's residual blocks on k_chain_b"""
    from lightzero_amd import _lib as L
    ref, par, fast = _models()
    B = 21
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(6))
    op, of = par.initial_inference(obs), fast.initial_inference(obs)
    lat = []
    for m in (par, fast):
        own = m._own_roots(B, "infer", 2)
        x = np.zeros((B, 64, 6, 6), np.float32)
        L.check(L.lib().lz_roots_read_latent(own._h, 0, x.reshape(-1)))
        lat.append(x)
    with torch.no_grad():
        want = ref.initial_inference(obs).latent_state.numpy()
    assert _rel(lat[0], want) < 1e-4            # (the parity engine: fp32)
    assert 1e-5 < _rel(lat[1], want) < 3e-2     # fast: bf16 round-off, eleven layers deep -- and not the fp32 path
    assert _rel(of.policy_logits, op.policy_logits) < 5e-2
    assert float(np.abs(of.value.numpy() - op.value.numpy()).max()) < 0.02 * float(np.abs(op.value.numpy()).max())


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def test_fast_kernels_equal_the_oracle_with_bf16_rounded_operands():
    """the dynamics convolution + residual block of the fast chain against torch fp32 convolutions whose inputs and weights were rounded to
    bf16 first (products of bf16 values are exact in fp32, so only the fp32 accumulation order differs: 1e-5 territory)"""
    ref, par, fast = _models()
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(4)
    B = 19
    obs = torch.rand(B, 4, 96, 96, generator=g)
    with torch.no_grad():
        lat = ref.initial_inference(obs).latent_state
    h = (torch.zeros(1, B, 512), torch.zeros(1, B, 512))
    act = torch.randint(0, A, (B,), generator=g)
    of = fast.recurrent_inference(lat, h, act)
    dn = ref.dynamics_network
    with torch.no_grad():
        onehot = torch.zeros(B, A, 6, 6)
        onehot[torch.arange(B), act] = 1.0
        w = dn.conv.weight
        # latent channels through bf16, the one-hot action planes through the fp32 action table (exact planes of ones)
        x = F.conv2d(_bf16(lat), _bf16(w[:, :64]), padding=1) + F.conv2d(onehot, w[:, 64:], padding=1)
        x = dn.norm_common(x) + lat
        x = F.relu(x)
        for blk in dn.resblocks:
            y = F.relu(blk.conv1[1](F.conv2d(_bf16(x), _bf16(blk.conv1[0].weight), padding=1)))
            y = blk.conv2[1](F.conv2d(_bf16(y), _bf16(blk.conv2[0].weight), padding=1))
            x = F.relu(y + x)
    err = float((of.latent_state - x).abs().max()), float(x.abs().max())
    assert err[0] < 2e-4 * (1.0 + err[1]), err   # fp32 accumulation-order noise plus the rare bf16 rounding flip of an intermediate activation


def _search(model, obs, noise, S, trace=False):
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    lib = L.lib()
    B = obs.shape[0]
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0)
    roots._ensure(A)
    d_obs = obs.cuda().contiguous()
    torch.cuda.synchronize()
    if trace:
        L.check(lib.lz_roots_enable_trace(roots._h, 1))
    L.check(lib.lz_initial_inference(roots._h, d_obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, np.ascontiguousarray(noise, np.float32).ctypes.data, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, CFG["pb_c_base"], CFG["pb_c_init"], CFG["discount"], CFG["horizon"], CFG["delta"]))
    out = dict(dist=np.array(roots.get_distributions()), val=np.array(roots.get_values(), np.float32), roots=roots)
    if trace:
        tr = np.zeros((S, B, 4), np.int32)
        L.check(lib.lz_roots_read_trace(roots._h, S, tr.reshape(-1)))
        vp = np.zeros((S + 1, B), np.float32); v = np.zeros_like(vp); pol = np.zeros((S + 1, B, A), np.float32)
        for s in range(S + 1):
            L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp[s], v[s], pol[s].reshape(-1)))
        out.update(trace=tr, vp=vp, v=v, pol=pol)
    return out


def test_search_agrees_statistically_with_parity_mode():
    ref, par, fast = _models()
    B, S = 256, 50
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(9))
    noise = np.random.default_rng(1).dirichlet([0.3] * A, size=B).astype(np.float32)
    p, f = _search(par, obs, noise, S), _search(fast, obs, noise, S)
    assert (f["dist"].sum(1) == S).all()
    # root values.  The bf16-rounded weights are ONE slightly different network for every root, so the shift has a common component (the
    # random-init model's values sit near -32 for every observation: |delta| ~ 0.2 = 0.7 %, which is 0.02 support units before h^-1,
    # whose slope there is ~11); what a search consumer needs is the ordering of the roots and the size relative to the values
    tv = 0.5 * np.abs(f["dist"] / S - p["dist"] / S).sum(1)   # total-variation distance of the visit distributions per root
    stats = dict(value_rel=float(np.abs(f["val"] - p["val"]).mean() / np.abs(p["val"]).mean()),
                 value_corr=float(np.corrcoef(f["val"], p["val"])[0, 1]), tv_mean=float(tv.mean()), tv_le_02=float((tv <= 0.2).mean()),
                 identical=float((tv == 0).mean()), same_argmax=float((f["dist"].argmax(1) == p["dist"].argmax(1)).mean()))
    print("fast vs parity, 256 x 50:", stats)
    assert stats["value_rel"] < 0.02, stats
    assert stats["value_corr"] > 0.98, stats
    assert stats["tv_mean"] < 0.10 and stats["tv_le_02"] > 0.85, stats
    assert stats["same_argmax"] > 0.85, stats   # the action a greedy actor would take


def test_tree_arithmetic_is_untouched_exact_replay_of_the_fast_search():
    """fast mode changes what the network computes, not what the tree does with it: feeding the fast engine's own per-simulation network
    outputs to the CPU oracle tree (and the reference's compiled ctree where it is on the box) reproduces every selection, visit count,
    root value and min-max statistic bit for bit -- the parity-mode exact gate (tests/test_exact_replay_gpu.py) on the fast graph."""
    from test_exact_replay_gpu import _search_and_replay
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    ref, par, fast = _models()
    B, S = 256, 50
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(11)).cuda().contiguous()
    rng = np.random.default_rng(2)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=fast.engine)
    roots.set_tiebreak(0)
    _search_and_replay("ez", fast, roots, obs, legal, [-1] * B, noises, S, 0.997)
    roots.reset(legal)
    _search_and_replay("ez", fast, roots, obs, legal, [-1] * B, noises, S, 0.997, trace=True)


def test_muzero_conv_fast_mode_statistics_and_exact_replay():
    """the same switch on the convolutional MuZeroModel (BASELINE configs[2]'s model): tower and chain on bf16 MFMA (no LSTM), heads / tree fp32"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_muzero import mz_tree
    from lightzero_amd.model.muzero_model import MuZeroModel
    from test_exact_replay_gpu import _search_and_replay
    A4, B, S = 4, 192, 60
    sd = tm.synthetic_init(tm.MuZeroModel(action_space_size=A4), seed=8).state_dict()
    par = MuZeroModel(action_space_size=A4, engine=L.new_engine(0)).load_state_dict(sd)
    fast = MuZeroModel(action_space_size=A4, engine=L.new_engine(0), fast_mode=True).load_state_dict(sd)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(12)).cuda().contiguous()
    rng = np.random.default_rng(4)
    noises = [rng.dirichlet([0.3] * A4).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A4))] * B
    res = {}
    for name, m in (("par", par), ("fast", fast)):
        roots = mz_tree.Roots(B, legal, action_space_size=A4, max_simulations=S, engine=m.engine)
        roots.set_tiebreak(0)
        d, v, _ = _search_and_replay("mz", m, roots, obs, legal, [-1] * B, noises, S, 0.997)   # (includes the exact replay gate on this graph)
        res[name] = (np.array(d), np.asarray(v))
    tv = 0.5 * np.abs(res["fast"][0] / S - res["par"][0] / S).sum(1)
    stats = dict(value_rel=float(np.abs(res["fast"][1] - res["par"][1]).mean() / (np.abs(res["par"][1]).mean() + 1e-9)),
                 value_corr=float(np.corrcoef(res["fast"][1], res["par"][1])[0, 1]), tv_mean=float(tv.mean()), identical=float((tv == 0).mean()))
    print("MuZero conv, fast vs parity, %d x %d:" % (B, S), stats)
    assert not np.array_equal(res["fast"][1], res["par"][1])
    # (this synthetic model's root values are small and close together: 3 % of their mean, correlation 0.89 measured; the visit distributions
    # are what a consumer sees: 82 % of the roots identical, mean total-variation distance 0.005)
    assert stats["value_rel"] < 0.06 and stats["value_corr"] > 0.8 and stats["tv_mean"] < 0.02 and stats["identical"] > 0.7, stats


def test_fast_mode_on_the_shipped_atari_shape_64x64():
    """the reference's shipped Atari EfficientZero configuration (4 x 64 x 64 frames -> 8 x 8 x 64 latent, supports (-50, 51, 1)): k_chain_b<8, 8> (four
    pixel tiles per tap), k_lstm_b<96, 64> (16 x 64 + 512 columns), the tower's k_conv_bf on the 32^2 / 16^2 / 8^2 grids"""
    from oracle import torch_models as tm
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from test_exact_replay_gpu import _search_and_replay
    sup = (-50., 51., 1.)
    kw = dict(observation_shape=(4, 64, 64), action_space_size=A, reward_support_range=sup, value_support_range=sup)
    ref = tm.synthetic_init(tm.EfficientZeroModel(**kw), seed=14)
    par = EfficientZeroModel(engine=L.new_engine(0), **kw).load_state_dict(ref.state_dict())
    fast = EfficientZeroModel(engine=L.new_engine(0), fast_mode=True, **kw).load_state_dict(ref.state_dict())
    g = torch.Generator().manual_seed(15)
    # single steps against the fp32 engine
    Bs = 23
    obs = torch.rand(Bs, 4, 64, 64, generator=g)
    with torch.no_grad():
        lat = ref.initial_inference(obs).latent_state
    h = (torch.randn(1, Bs, 512, generator=g) * 0.3, torch.randn(1, Bs, 512, generator=g) * 0.3)
    act = torch.randint(0, A, (Bs,), generator=g)
    op, of = par.recurrent_inference(lat, h, act), fast.recurrent_inference(lat, h, act)
    assert 1e-6 < _rel(of.latent_state, op.latent_state) < 3e-2
    assert _rel(of.reward_hidden_state[0], op.reward_hidden_state[0]) < 3e-2 and _rel(of.reward_hidden_state[1], op.reward_hidden_state[1]) < 3e-2
    assert _rel(of.policy_logits, op.policy_logits) < 5e-2 and _rel(of.value, op.value) < 5e-2 and _rel(of.value_prefix, op.value_prefix) < 5e-2
    oi, fi = par.initial_inference(obs), fast.initial_inference(obs)
    assert _rel(fi.policy_logits, oi.policy_logits) < 5e-2
    # the 8x8 chain's arithmetic is the stated one: torch fp32 convolutions on bf16-rounded operands (as in the 6x6 test above)
    import torch.nn.functional as F
    dn = ref.dynamics_network
    with torch.no_grad():
        onehot = torch.zeros(Bs, A, 8, 8)
        onehot[torch.arange(Bs), act] = 1.0
        w = dn.conv.weight
        x = F.conv2d(_bf16(lat), _bf16(w[:, :64]), padding=1) + F.conv2d(onehot, w[:, 64:], padding=1)
        x = F.relu(dn.norm_common(x) + lat)
        for blk in dn.resblocks:
            y = F.relu(blk.conv1[1](F.conv2d(_bf16(x), _bf16(blk.conv1[0].weight), padding=1)))
            y = blk.conv2[1](F.conv2d(_bf16(y), _bf16(blk.conv2[0].weight), padding=1))
            x = F.relu(y + x)
    err = float((of.latent_state - x).abs().max()), float(x.abs().max())
    assert err[0] < 2e-4 * (1.0 + err[1]), err
    # a search: statistics against parity mode, exact replay of the fast graph
    B, S = 128, 50
    obs = torch.rand(B, 4, 64, 64, generator=g).cuda().contiguous()
    rng = np.random.default_rng(3)
    noises = [rng.dirichlet([0.3] * A).astype(np.float32).tolist() for _ in range(B)]
    legal = [list(range(A))] * B
    res = {}
    for name, m in (("par", par), ("fast", fast)):
        roots = ez_tree.Roots(B, legal, action_space_size=A, max_simulations=S, engine=m.engine)
        roots.set_tiebreak(0)
        d, v, _ = _search_and_replay("ez", m, roots, obs, legal, [-1] * B, noises, S, 0.997)
        res[name] = (np.array(d), np.asarray(v))
    tv = 0.5 * np.abs(res["fast"][0] / S - res["par"][0] / S).sum(1)
    stats = dict(value_rel=float(np.abs(res["fast"][1] - res["par"][1]).mean() / (np.abs(res["par"][1]).mean() + 1e-9)),
                 tv_mean=float(tv.mean()), identical=float((tv == 0).mean()))
    print("EfficientZero 64x64, fast vs parity, %d x %d:" % (B, S), stats)
    # (measured on this synthetic model: 48 % of the roots identical, mean total-variation distance 0.033 = 1.6 of 50 visits, values 5 % apart)
    assert stats["value_rel"] < 0.08 and stats["tv_mean"] < 0.06 and stats["identical"] > 0.35, stats
