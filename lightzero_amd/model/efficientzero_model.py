"""Engine-backed counterpart of lzero/model/efficientzero_model.py::EfficientZeroModel (inference graph).

Same constructor keywords as the reference class (the ones that shape the inference graph); weights
are ingested from a reference-format ``state_dict`` (same key names) and live in HBM in the kernels'
layouts.  ``initial_inference`` / the recurrent loop run as HIP kernels behind the C ABI
(lightzero_amd/csrc/lz_nn.hip, lz_search.hip); there is no torch module inside and no fallback.
"""
import ctypes

import numpy as np

from .. import _lib as L


class HbmToken(object):
    """Stands for a tensor of an engine model's network output that stayed in HBM (the root latent state / LSTM state, slot 0 of
    the pools of ``roots``).  It survives the conversions the reference applies to network outputs
    (``x.detach().cpu().numpy()``, efficientzero.py:588-593) and is what ``search(roots, model, latent_state_roots, ...)`` receives
    in their place; a fused search adopts the inference it refers to."""
    _is_lz_hbm_token = True

    def __init__(self, model, roots, what):
        self.model, self.roots, self.what = model, roots, what

    def detach(self):
        return self

    def cpu(self):
        return self

    def numpy(self):
        return self

    def __repr__(self):
        return "<HBM-resident %s of %d roots>" % (self.what, self.roots.num)


def unwrap_checkpoint(state_dict, which="model"):
    """The weight-ingest format of the path: a LightZero checkpoint is the policy's learn-mode state
    ``{'model': ..., 'target_model': ..., 'optimizer': ...}`` (lzero/policy/muzero.py:1043-1047, saved by DI-engine's learner with
    ``last_iter`` etc. beside it).  Returns the bare reference-keyed state_dict of ``which`` ('model': the online network the collector
    plays with; 'target_model': the one reanalyze uses); a bare state_dict passes through.  DistributedDataParallel's ``module.`` and
    torch.compile's ``_orig_mod.`` key prefixes are stripped.  Anything else (a dict without tensors at its leaves) is refused."""
    sd = state_dict
    if isinstance(sd, dict) and which in sd and isinstance(sd[which], dict):
        sd = sd[which]
    elif isinstance(sd, dict) and "model" in sd and isinstance(sd["model"], dict):
        raise KeyError("checkpoint has no %r entry (keys: %s)" % (which, sorted(sd)))
    if not hasattr(sd, "items"):
        raise TypeError("state_dict must be a mapping of reference parameter names to tensors / arrays")
    out = {}
    for k, v in sd.items():
        if isinstance(v, dict):
            raise TypeError("state_dict entry %r is a dict: pass the checkpoint itself, or its 'model' entry" % (k,))
        k2 = str(k)
        stripped = True
        while stripped:
            stripped = False
            for pre in ("module.", "_orig_mod."):
                if k2.startswith(pre):
                    k2, stripped = k2[len(pre):], True
        out[k2] = v
    return out


class EfficientZeroModel(object):
    _model_type = 0  # lz_model_cfg.model_type

    def __init__(self, observation_shape=(4, 96, 96), action_space_size=6, lstm_hidden_size=512, num_res_blocks=1,
                 num_channels=64, reward_head_channels=16, value_head_channels=16, policy_head_channels=16,
                 reward_head_hidden_channels=(32,), value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,),
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True,
                 categorical_distribution=True, norm_type='BN', discrete_action_encoding_type='one_hot',
                 engine=None, fast_mode=False, fp32_matrix=False, **kwargs):
        """``fast_mode=True`` (not a reference argument): lz_model_cfg.precision = 1 -- the 3x3 convolutions of the representation tower and
        of the recurrent chain and the LSTM gate product on bf16 MFMA (fp32 accumulation; heads, h^-1 and the tree unchanged); statistical
        parity only, reported separately from the parity-mode numbers (last arm).  EfficientZeroModel /
        MuZeroModel on 4x96x96 (6x6x64 latent) or 4x64x64 (8x8x64) observations.
        ``fp32_matrix=True`` (not a reference argument): lz_model_cfg.precision = 2 -- parity mode with the 3x3 convolutions on the fp32 matrix
        instructions (the kernels of rounds 1-4) instead of the split-bf16 products of the default parity mode: same parity bound, other
        roundings, slower; the per-model form of LZ_CHAIN_NO_SPLIT=1 LZ_CONV_NO_SPLIT=1."""
        if not 1 <= int(num_res_blocks) <= 3 or norm_type != 'BN' or not categorical_distribution \
                or discrete_action_encoding_type not in ('one_hot', 'not_one_hot'):
            raise NotImplementedError("engine model: num_res_blocks in 1..3, norm_type='BN', categorical_distribution=True, "
                                      "discrete_action_encoding_type 'one_hot' | 'not_one_hot'")
        if not (reward_head_hidden_channels[0] == value_head_hidden_channels[0] == policy_head_hidden_channels[0]) \
                or len(value_head_hidden_channels) != 1:
            raise NotImplementedError("the three heads must have one hidden layer of the same width")
        if value_support_range[2] != 1. or reward_support_range[2] != 1.:
            raise NotImplementedError("supports with step 1")
        if tuple(reward_support_range) != tuple(value_support_range) and self._model_type != 1:
            # The reference's EfficientZero driver transforms the value prefix with the VALUE handle (mcts_ctree.py:839-841):
            # reward_support_range only sizes the value-prefix head.  Another range of the SAME size therefore behaves exactly like
            # equal supports (in the reference and here); another size fails in the reference's own driver with a shape error.
            rs = int(round((reward_support_range[1] - reward_support_range[0]) / reward_support_range[2]))
            vs = int(round((value_support_range[1] - value_support_range[0]) / value_support_range[2]))
            if rs != vs:
                raise NotImplementedError("EfficientZero: a reward support of another SIZE than the value support fails in the reference's own "
                                          "driver (the value handle is applied to the value prefix, mcts_ctree.py:839-841); MuZeroModel takes one")
            reward_support_range = value_support_range
        if not (reward_head_channels == value_head_channels == policy_head_channels):
            raise NotImplementedError("head channel counts must be equal")
        self.observation_shape = tuple(observation_shape)
        self.action_space_size = int(action_space_size)
        self.lstm_hidden_size = int(lstm_hidden_size)
        self.num_channels = int(num_channels)
        self.num_res_blocks = int(num_res_blocks)
        self._downsample = bool(downsample)
        self.value_support_size = int(round((value_support_range[1] - value_support_range[0]) / value_support_range[2]))
        self.reward_support_size = int(round((reward_support_range[1] - reward_support_range[0]) / reward_support_range[2]))
        self.discrete_action_encoding_type = discrete_action_encoding_type
        # one model per engine: the first model of the process lives on the default engine, later ones get their own
        self._engine = engine if engine is not None else L.engine_for_new_model()
        cfg = L.ModelCfg(self._model_type, self.observation_shape[0], self.observation_shape[1], self.observation_shape[2],
                         self.action_space_size, self.num_channels, self.lstm_hidden_size, int(value_head_channels),
                         int(value_head_hidden_channels[0]), self.value_support_size, float(value_support_range[0]), 1e-5,
                         1 if downsample else 0)
        cfg.num_res_blocks = self.num_res_blocks
        cfg.action_encoding = 0 if discrete_action_encoding_type == 'one_hot' else 1
        if tuple(reward_support_range) != tuple(value_support_range):
            cfg.reward_support_size, cfg.reward_support_min = self.reward_support_size, float(reward_support_range[0])
        self.fast_mode = bool(fast_mode)
        if self.fast_mode:
            if self._model_type not in (0, 1) or not downsample or self.num_channels != 64 \
                    or self.observation_shape not in ((4, 96, 96), (4, 64, 64)) or getattr(self, "num_of_sampled_actions", 0):
                raise NotImplementedError("fast_mode: EfficientZeroModel / MuZeroModel on 4x96x96 (6x6x64 latent) or 4x64x64 (8x8x64) observations")
            cfg.precision = 1
        self.fp32_matrix = bool(fp32_matrix)
        if self.fp32_matrix:
            if self.fast_mode:
                raise ValueError("fast_mode and fp32_matrix exclude each other")
            cfg.precision = 2
        self._create(cfg)

    def _create(self, cfg):
        L.check(L.lib().lz_model_create(self._engine, ctypes.byref(cfg)))
        self._uid = L.lib().lz_engine_model_uid(self._engine)
        self._loaded = False

    def _check_owner(self):
        """another model object may have been created on this engine since (it replaces this one's weights)"""
        if L.lib().lz_engine_model_uid(self._engine) != self._uid:
            raise L.LzError("%s: its engine now holds another model (one model per engine: pass engine=L.new_engine() "
                            "or let the constructor pick one)" % type(self).__name__)

    @property
    def engine(self):
        return self._engine

    def load_state_dict(self, state_dict, strict=True):
        """state_dict: reference key -> array-like (torch tensors or numpy), e.g. a LightZero checkpoint's ``model`` -- or the
        checkpoint itself as ``torch.load`` returns it (see ``unwrap_checkpoint``).
        Calling it again on a loaded model is a weight refresh (collector after a learner update): tensors are overwritten in
        place on the device, roots and their captured search graphs stay valid."""
        self._check_owner()
        if self._loaded and getattr(state_dict, "flat", None) is not None and self._refresh_on_device(state_dict):
            return self     # a shard.FlatStateDict in this model's layout: by pointer, before anything walks its tensors
        state_dict = unwrap_checkpoint(state_dict)
        if self._loaded and self._refresh_on_device(state_dict):
            return self
        synced = set()
        for name, value in state_dict.items():
            if name.endswith("num_batches_tracked"):
                continue
            if getattr(value, "is_cuda", False) and str(value.dtype) == "torch.float32" and value.is_contiguous():
                # a device tensor (shard.broadcast_state_dict(..., on_device=True)): handed over by pointer, no host copy here
                if value.device not in synced:
                    import torch
                    torch.cuda.current_stream(value.device).synchronize()   # whoever produced the tensors ON THEIR DEVICE is done
                    synced.add(value.device)
                shape = (ctypes.c_int64 * max(value.dim(), 1))(*value.shape)
                L.check(L.lib().lz_model_set_tensor_device(self._engine, name.encode(), value.data_ptr(), shape, value.dim()))
                continue
            arr = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            L.check(L.lib().lz_model_set_tensor(self._engine, name.encode(), arr.reshape(-1), shape, arr.ndim))
        L.check(L.lib().lz_model_finalize(self._engine))
        self._loaded = True
        # the shapes this (validated) load had: a later device-side refresh compares against them -- lz_model_refresh_flat only sees a flat
        # buffer, so a tensor of the same size and another shape would otherwise be taken silently where this path rejects it
        self._loaded_shapes = {k: tuple(int(d) for d in np.shape(v)) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")}
        return self

    def _engine_waits_for_torch(self, device):
        """order the engine's stream behind what torch has enqueued on its current stream of ``device`` (an event the engine's stream
        waits for) -- the host does not wait: a collector that uploads the next frames asynchronously goes straight on to enqueue the search"""
        import torch
        ext = self.__dict__.setdefault("_ext_streams", {})
        key = (device.type, device.index)
        if key not in ext:
            ext[key] = torch.cuda.ExternalStream(L.lib().lz_engine_stream(self._engine), device=device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        ext[key].wait_event(ev)

    def _refresh_on_device(self, sd):
        """A weight REFRESH (the model is loaded already) through lz_model_refresh_flat: the state_dict as ONE flat fp32 buffer in the
        library's name order -> every kernel layout by kernels on the engine's stream (bit-identical to the host re-layout; DESIGN
        section 7).  Device tensors that already are consecutive views of one buffer (shard.broadcast_state_dict(on_device=True)) are
        passed by pointer, other device tensors through one torch.cat, host arrays through the library's pinned staging buffer.
        False: not applicable here (MLP family, fast mode, other names / shapes than the loaded ones) -> the caller's host path."""
        lib = L.lib()
        lay = self.__dict__.get("_flat_layout")
        if lay is None:
            nt, nf = ctypes.c_int64(0), ctypes.c_int64(0)
            if lib.lz_model_flat_layout(self._engine, ctypes.byref(nt), ctypes.byref(nf)) != 0:
                self._flat_layout = False
                return False
            ent, buf = [], ctypes.create_string_buffer(512)
            for i in range(nt.value):
                off, size = ctypes.c_int64(0), ctypes.c_int64(0)
                L.check(lib.lz_model_flat_entry(self._engine, i, buf, 512, ctypes.byref(off), ctypes.byref(size)))
                ent.append((buf.value.decode(), off.value, size.value))
            lay = self._flat_layout = (ent, nf.value)
        if lay is False:
            return False
        ent, total = lay
        if not self.__dict__.get("_flat_layout_key"):
            self._flat_layout_key = tuple(ent)
        flat = getattr(sd, "flat", None)
        if flat is not None and getattr(flat, "is_cuda", False) and (getattr(sd, "layout", None) is self._flat_layout_key or tuple(getattr(sd, "layout", ())) == self._flat_layout_key) and flat.numel() == total \
                and str(flat.dtype) == "torch.float32" and flat.is_contiguous():
            # a shard.FlatStateDict in this model's own layout: the buffer goes over by pointer, no walk over the tensors
            self._engine_waits_for_torch(flat.device)
            L.check(lib.lz_model_refresh_flat(self._engine, flat.data_ptr(), total, 1))
            self._flat_keep = flat
            try:   # the copy out of the buffer is asynchronous on the engine's stream: whoever overwrites the buffer next (the in-place
                import torch   # RCCL broadcast of shard.broadcast_state_dict) waits for this event on its own stream
                ev = torch.cuda.Event()
                ev.record(self._ext_streams[(flat.device.type, flat.device.index)])
                sd.consumed_event = ev
            except Exception:
                pass
            return True
        if sum(1 for k in sd if not k.endswith("num_batches_tracked")) != len(ent):
            return False
        vals = []
        shapes = self.__dict__.get("_loaded_shapes") or {}
        for name, off, size in ent:
            v = sd.get(name)
            if v is None or int(np.prod(np.shape(v))) != size:
                return False
            if name in shapes and tuple(int(d) for d in np.shape(v)) != shapes[name]:
                return False     # same size, another shape: the strict host path (lz_model_set_tensor) says what is wrong
            dt = str(getattr(v, "dtype", "float32"))
            if dt not in ("torch.float32", "float32"):
                return False     # another dtype: not cast silently here either
            vals.append(v)
        if all(getattr(v, "is_cuda", False) and str(v.dtype) == "torch.float32" and v.is_contiguous() for v in vals):
            import torch
            dev = vals[0].device
            base = vals[0].data_ptr()
            if all(v.device == dev and v.data_ptr() == base + 4 * off for v, (_, off, _) in zip(vals, ent)):
                flat, ptr = vals, base      # consecutive views of one flat buffer (an RCCL broadcast): no copy at all
            else:
                flat = torch.cat([v.reshape(-1) for v in vals])
                ptr = flat.data_ptr()
            # the engine's stream waits for whatever produced the tensors on torch's stream -- an event, not a host synchronisation
            self._engine_waits_for_torch(dev)
            L.check(lib.lz_model_refresh_flat(self._engine, ptr, total, 1))
            self._flat_keep = flat          # alive until the next refresh: the copy out of it is asynchronous
            return True
        pin = ctypes.POINTER(ctypes.c_float)()
        L.check(lib.lz_model_flat_host_buffer(self._engine, ctypes.byref(pin)))
        host = np.ctypeslib.as_array(pin, shape=(total,))
        for v, (_, off, size) in zip(vals, ent):
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            host[off:off + size] = a.reshape(-1)    # (casts to float32 where the source is another type)
        L.check(lib.lz_model_refresh_flat(self._engine, ctypes.cast(pin, ctypes.c_void_p), total, 0))
        return True

    _is_lz_engine_model = True
    training = False    # torch.nn.Module.training of a model in eval() mode (the reference's _forward_eval reads it)
    _uses_lstm = True   # the value-prefix LSTM (EfficientZero); MuZeroModel clears it

    # ---- shapes the shared inference code needs (the MLP family overrides _policy_width / _latent_shape)
    @property
    def _pw(self):
        return getattr(self, "_policy_width", self.action_space_size)

    def _latent_shape(self):
        c, h, w = self.observation_shape
        if not getattr(self, "_downsample", True):
            return (self.num_channels, h, w)
        g = h // 16 if h == 96 else h // 8   # 96 -> 6 (two poolings), 64 -> 8 (common.py:358-359)
        return (self.num_channels, g, g)

    def _tree(self):
        if self._uses_lstm:
            from ..mcts.ctree.ctree_efficientzero import ez_tree as tree
        else:
            from ..mcts.ctree.ctree_muzero import mz_tree as tree
        return tree

    _OWN_ROOTS_MAX = 4   # model-owned roots handles (initial_inference without roots / the Python recurrent_inference), LRU

    def _own_roots(self, B, slot, max_simulations, trace=False):
        """a roots handle the MODEL owns (per batch size): where initial_inference(obs) without roots leaves the root state, and
        the scratch pool of the Python recurrent_inference"""
        import collections
        cache = self.__dict__.setdefault("_own", collections.OrderedDict())
        r = cache.get((slot, B))
        if r is not None:
            cache.move_to_end((slot, B))
        if r is None:
            while len(cache) >= self._OWN_ROOTS_MAX:   # a driver whose ready-env count varies: keep the few most recent batch sizes
                _, old = cache.popitem(last=False)
                old.clear()   # destroyed, not parked: its pools go back to the device
                old._evicted_from = type(self).__name__   # an HbmToken that still points here fails with a message, not on a dead handle
            r = self._new_own_roots(B, max_simulations)
            if trace:
                L.check(L.lib().lz_roots_enable_trace(r._h, 1))   # the heads also write their support-wide logits
            cache[(slot, B)] = r
        return r

    def _new_own_roots(self, B, max_simulations):
        """a fresh roots handle of this model's tree family on its engine (the sampled families override it)"""
        A = self.action_space_size
        r = self._tree().Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=max_simulations, engine=self._engine)
        r._bind_engine(self._engine)
        r._ensure(A)
        return r

    def _run_initial(self, obs, roots):
        B = roots.num
        oshape = tuple(self.observation_shape)
        if hasattr(obs, "data_ptr"):
            if tuple(obs.shape) != (B,) + oshape or not obs.is_contiguous() or str(obs.dtype) != "torch.float32":
                raise ValueError("obs must be a contiguous float32 [B, *observation_shape] tensor")
            if getattr(obs, "is_cuda", False):
                self._engine_waits_for_torch(obs.device)   # torch produced obs on its own stream
                L.check(L.lib().lz_initial_inference(roots._h, obs.data_ptr()))
            else:
                L.check(L.lib().lz_initial_inference_host(roots._h, np.ascontiguousarray(obs.numpy(), np.float32).reshape(-1)))
        else:
            arr = np.ascontiguousarray(obs, dtype=np.float32)
            if arr.shape != (B,) + oshape:
                raise ValueError("obs must be [B, *observation_shape]")
            L.check(L.lib().lz_initial_inference_host(roots._h, arr.reshape(-1)))

    def initial_inference(self, obs, roots=None, fetch=True):
        """EfficientZeroModel.initial_inference (efficientzero_model.py:203-238).  The latent state and the zero LSTM state stay in
        HBM (slot 0 of a roots handle's pools); what comes back are the root predictions plus TOKENS for those tensors.

        ``model.initial_inference(obs)`` -- the reference's signature and call order (efficientzero.py:582-610: infer, THEN build
        and prepare the roots, then search(roots, model, latent_state_roots, reward_hidden_state_roots, to_play)): the state is
        left in a handle the model owns; the returned object has the reference's fields with torch types -- ``policy_logits``
        [B, A], ``value`` [B, 1] ALREADY passed through InverseScalarTransform on the device (the engine's
        ``value_inverse_scalar_transform_handle`` recognises it and passes it on), ``value_prefix`` / ``reward`` = [0.] * B,
        ``latent_state`` / ``reward_hidden_state`` = tokens that survive ``.detach().cpu().numpy()``; the search adopts the
        inference into the roots it is given (lz_roots_adopt_inference).

        ``model.initial_inference(obs, roots[, fetch])`` -- the engine-native form: inference straight into ``roots`` (no
        adoption copy; ``fetch=False``: no read-back and no synchronisation, returns None); numpy-typed fields, ``value`` [B].

        ``obs``: [B, *observation_shape] fp32 -- a device tensor exposing ``data_ptr()`` (used in place) or a host array."""
        if not self._loaded:
            raise L.LzError("%s: load_state_dict has not been called" % type(self).__name__)
        self._check_owner()
        import types
        if roots is None:
            import torch
            B = int(obs.shape[0])
            own = self._own_roots(B, "infer", 2)
            self._run_initial(obs, own)
            own._inferred_by = self
            values = np.zeros(B, np.float32)
            logits = np.zeros((B, self._pw), np.float32)
            L.check(L.lib().lz_roots_get_root_outputs(own._h, values, logits.reshape(-1)))
            value = torch.from_numpy(values.reshape(B, 1))
            value._lz_inverse_transformed = True
            out = types.SimpleNamespace(value=value, policy_logits=torch.from_numpy(logits), latent_state=HbmToken(self, own, "latent_state"))
            if self._uses_lstm:
                out.value_prefix = [0. for _ in range(B)]
                out.reward_hidden_state = (HbmToken(self, own, "reward_hidden_state[0]"), HbmToken(self, own, "reward_hidden_state[1]"))
            else:
                out.reward = [0. for _ in range(B)]
            return out
        B = roots.num
        roots._bind_engine(self._engine)
        roots._ensure(self.action_space_size)
        self._run_initial(obs, roots)
        roots._inferred_by = self
        if not fetch:
            # the caller reads the root predictions after the search (Roots.get_search_results): no host-device
            # synchronisation between the representation network and the search
            return None
        values = np.zeros(B, np.float32)
        logits = np.zeros((B, self._pw), np.float32)
        L.check(L.lib().lz_roots_get_root_outputs(roots._h, values, logits.reshape(-1)))
        out = types.SimpleNamespace(value=values, policy_logits=logits, latent_state=("hbm-pool", roots))
        if self._uses_lstm:
            out.value_prefix = [0. for _ in range(B)]
            out.reward_hidden_state = ("hbm-pool", roots)
        else:
            out.reward = [0. for _ in range(B)]
        return out

    def recurrent_inference(self, latent_state, *rest):
        """EfficientZeroModel.recurrent_inference(latent_state, reward_hidden_state, action) (efficientzero_model.py:240-273) /
        MuZeroModel.recurrent_inference(latent_state, action) (muzero_model.py:240-272) with ARRAYS in and out, over
        lz_recurrent_inference on a scratch pool: what a foreign driver loop (the reference's search with host-side pools,
        mcts_ctree.py:815-847) or a reanalyze caller needs.  Returns the reference's fields as torch CPU tensors: ``value`` /
        ``value_prefix`` | ``reward`` are the SUPPORT-WIDE logits [B, support] (not yet inverse-transformed, like the reference's),
        ``policy_logits`` [B, A], ``latent_state`` [B, C, H, W] | [B, L], ``reward_hidden_state`` = (h [1, B, H], c [1, B, H]).
        Every call crosses PCIe both ways -- the fused search (model.initial_inference + search) never does."""
        import types
        import torch
        if not self._loaded:
            raise L.LzError("%s: load_state_dict has not been called" % type(self).__name__)
        self._check_owner()
        if self._uses_lstm:
            if len(rest) != 2:
                raise TypeError("recurrent_inference(latent_state, reward_hidden_state, action)")
            hidden, action = rest
        else:
            if len(rest) != 1:
                raise TypeError("recurrent_inference(latent_state, action)")
            hidden, action = None, rest[0]

        def host(x, dt=np.float32):
            if hasattr(x, "detach"):
                x = x.detach().cpu().numpy()
            return np.ascontiguousarray(x, dtype=dt)
        if isinstance(latent_state, HbmToken):
            raise L.LzError("recurrent_inference takes arrays; HBM tokens belong to the fused search (search(roots, model, tokens, ...))")
        lat = host(latent_state)
        B = lat.shape[0]
        lshape = tuple(self._latent_shape())
        if tuple(lat.shape[1:]) != lshape:
            raise ValueError("latent_state must be [B, %s]" % ", ".join(str(d) for d in lshape))
        lib = L.lib()
        r = self._own_roots(B, "scratch", 2, trace=True)
        L.check(lib.lz_roots_write_latent(r._h, 0, lat.reshape(-1)))
        H = int(self.lstm_hidden_size) if self._uses_lstm else 0
        if self._uses_lstm:
            h0, c0 = host(hidden[0]).reshape(B, H), host(hidden[1]).reshape(B, H)
            L.check(lib.lz_roots_write_hidden(r._h, 0, h0.reshape(-1), c0.reshape(-1)))
        zeros = np.zeros(B, np.int32)
        cont = bool(getattr(self, "continuous_action_space", False))
        if cont:
            af = host(action).reshape(B, -1)
            L.check(lib.lz_recurrent_inference(r._h, zeros, None, af.ctypes.data, None, 0, 1))
        elif int(getattr(self, "num_of_sampled_actions", 0) or 0) > 0:
            # sampled roots with a discrete action space carry an action as the float of its index (lz_recurrent_inference)
            af = host(action, np.int64).reshape(B, 1).astype(np.float32)
            L.check(lib.lz_recurrent_inference(r._h, zeros, None, af.ctypes.data, None, 0, 1))
        else:
            a = host(action, np.int64).reshape(B).astype(np.int32)
            L.check(lib.lz_recurrent_inference(r._h, zeros, a.ctypes.data, None, None, 0, 1))
        nxt = np.zeros((B,) + lshape, np.float32)
        L.check(lib.lz_roots_read_latent(r._h, 1, nxt.reshape(-1)))
        SUP = int(self.value_support_size)
        vlog = np.zeros((B, SUP), np.float32); rlog = np.zeros((B, int(self.reward_support_size)), np.float32)
        L.check(lib.lz_roots_read_debug_logits(r._h, 0, vlog.reshape(-1)))
        L.check(lib.lz_roots_read_debug_logits(r._h, 1, rlog.reshape(-1)))
        pol = np.zeros((B, self._pw), np.float32)
        scalar = np.zeros(B, np.float32)   # (the post-h^-1 scalars; the reference's contract returns the logits)
        L.check(lib.lz_roots_read_sim_outputs(r._h, 1, scalar, scalar.copy(), pol.reshape(-1)))
        out = types.SimpleNamespace(value=torch.from_numpy(vlog), policy_logits=torch.from_numpy(pol), latent_state=torch.from_numpy(nxt))
        if self._uses_lstm:
            hh = np.zeros((B, H), np.float32); cc = np.zeros((B, H), np.float32)
            L.check(lib.lz_roots_read_hidden(r._h, 1, hh.reshape(-1), cc.reshape(-1)))
            out.value_prefix = torch.from_numpy(rlog)
            out.reward_hidden_state = (torch.from_numpy(hh).unsqueeze(0), torch.from_numpy(cc).unsqueeze(0))
        else:
            out.reward = torch.from_numpy(rlog)
        return out

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the engine model is inference-only")
        return self
