"""Engine-backed counterpart of lzero/model/efficientzero_model.py::EfficientZeroModel (inference graph).

Same constructor keywords as the reference class (the ones that shape the inference graph); weights
are ingested from a reference-format ``state_dict`` (same key names) and live in HBM in the kernels'
layouts.  ``initial_inference`` / the recurrent loop run as HIP kernels behind the C ABI
(lightzero_amd/csrc/lz_nn.hip, lz_search.hip); there is no torch module inside and no fallback.
"""
import ctypes

import numpy as np

from .. import _lib as L


class EfficientZeroModel(object):
    _model_type = 0  # lz_model_cfg.model_type

    def __init__(self, observation_shape=(4, 96, 96), action_space_size=6, lstm_hidden_size=512, num_res_blocks=1,
                 num_channels=64, reward_head_channels=16, value_head_channels=16, policy_head_channels=16,
                 reward_head_hidden_channels=(32,), value_head_hidden_channels=(32,), policy_head_hidden_channels=(32,),
                 reward_support_range=(-300., 301., 1.), value_support_range=(-300., 301., 1.), downsample=True,
                 categorical_distribution=True, norm_type='BN', discrete_action_encoding_type='one_hot',
                 engine=None, **kwargs):
        if not 1 <= int(num_res_blocks) <= 3 or norm_type != 'BN' or not categorical_distribution \
                or discrete_action_encoding_type != 'one_hot':
            raise NotImplementedError("engine model: num_res_blocks in 1..3, norm_type='BN', "
                                      "categorical_distribution=True, one_hot action encoding")
        if not (reward_head_hidden_channels[0] == value_head_hidden_channels[0] == policy_head_hidden_channels[0]) \
                or len(value_head_hidden_channels) != 1:
            raise NotImplementedError("the three heads must have one hidden layer of the same width")
        if tuple(reward_support_range) != tuple(value_support_range) or value_support_range[2] != 1.:
            raise NotImplementedError("reward and value supports must be equal with step 1")
        if not (reward_head_channels == value_head_channels == policy_head_channels):
            raise NotImplementedError("head channel counts must be equal")
        self.observation_shape = tuple(observation_shape)
        self.action_space_size = int(action_space_size)
        self.lstm_hidden_size = int(lstm_hidden_size)
        self.num_channels = int(num_channels)
        self.num_res_blocks = int(num_res_blocks)
        self.value_support_size = int(round((value_support_range[1] - value_support_range[0]) / value_support_range[2]))
        self.reward_support_size = self.value_support_size
        # one model per engine: the first model of the process lives on the default engine, later ones get their own
        self._engine = engine if engine is not None else L.engine_for_new_model()
        cfg = L.ModelCfg(self._model_type, self.observation_shape[0], self.observation_shape[1], self.observation_shape[2],
                         self.action_space_size, self.num_channels, self.lstm_hidden_size, int(value_head_channels),
                         int(value_head_hidden_channels[0]), self.value_support_size, float(value_support_range[0]), 1e-5,
                         1 if downsample else 0)
        cfg.num_res_blocks = self.num_res_blocks
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
        """state_dict: reference key -> array-like (torch tensors or numpy), e.g. a LightZero checkpoint's ``model``.
        Calling it again on a loaded model is a weight refresh (collector after a learner update): tensors are overwritten in
        place on the device, roots and their captured search graphs stay valid."""
        self._check_owner()
        for name, value in state_dict.items():
            if name.endswith("num_batches_tracked"):
                continue
            arr = value.detach().cpu().numpy() if hasattr(value, "detach") else np.asarray(value)
            arr = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (ctypes.c_int64 * max(arr.ndim, 1))(*arr.shape)
            L.check(L.lib().lz_model_set_tensor(self._engine, name.encode(), arr.reshape(-1), shape, arr.ndim))
        L.check(L.lib().lz_model_finalize(self._engine))
        self._loaded = True
        return self

    _is_lz_engine_model = True

    def initial_inference(self, obs, roots, fetch=True):
        """EfficientZeroModel.initial_inference (efficientzero_model.py:203-238) for the batch held by ``roots``
        (a lightzero_amd ez_tree.Roots): the latent state and the zero LSTM state are written into the roots'
        HBM pools (slot 0) instead of being returned.  ``obs``: [B,C,H,W] fp32 -- a device tensor exposing
        ``data_ptr()`` (used in place) or a host numpy array (staged over PCIe).
        Returns an ``EZNetworkOutput``-like namespace: ``value`` is already passed through
        InverseScalarTransform (shape [B]), ``value_prefix`` is ``[0.]*B``, ``policy_logits`` is [B,A] numpy;
        ``latent_state`` / ``reward_hidden_state`` are opaque tokens bound to ``roots``.  ``fetch=False`` skips the read-back
        (and its synchronisation) and returns None."""
        if not self._loaded:
            raise L.LzError("EfficientZeroModel: load_state_dict has not been called")
        self._check_owner()
        B = roots.num
        roots._bind_engine(self._engine)
        roots._ensure(self.action_space_size)
        if hasattr(obs, "data_ptr"):
            if tuple(obs.shape) != (B,) + self.observation_shape or not obs.is_contiguous() or str(obs.dtype) != "torch.float32":
                raise ValueError("obs must be a contiguous float32 [B,C,H,W] tensor")
            if getattr(obs, "is_cuda", False):
                import torch
                torch.cuda.current_stream().synchronize()  # torch produced obs on its own stream
                L.check(L.lib().lz_initial_inference(roots._h, obs.data_ptr()))
            else:
                L.check(L.lib().lz_initial_inference_host(roots._h, np.ascontiguousarray(obs.numpy(), np.float32).reshape(-1)))
        else:
            arr = np.ascontiguousarray(obs, dtype=np.float32)
            if arr.shape != (B,) + self.observation_shape:
                raise ValueError("obs must be [B,C,H,W]")
            L.check(L.lib().lz_initial_inference_host(roots._h, arr.reshape(-1)))
        roots._inferred_by = self
        if not fetch:
            # the caller reads the root predictions after the search (Roots.get_search_results): no host-device
            # synchronisation between the representation network and the search
            return None
        values = np.zeros(B, np.float32)
        logits = np.zeros((B, self.action_space_size), np.float32)
        L.check(L.lib().lz_roots_get_root_outputs(roots._h, values, logits.reshape(-1)))
        import types
        return types.SimpleNamespace(value=values, value_prefix=[0. for _ in range(B)], policy_logits=logits,
                                     latent_state=("hbm-pool", roots), reward_hidden_state=("hbm-pool", roots))

    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the engine model is inference-only")
        return self
