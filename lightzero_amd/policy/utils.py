"""Host-side post-search helpers (restated from lzero/policy/utils.py)."""
import numpy as np

from .. import _lib as L


def select_action(visit_counts, temperature=1, deterministic=True):
    """lzero/policy/utils.py:637-661: p ~ N^(1/T); argmax (eval) or sample (collect); entropy in bits."""
    visit_counts = np.asarray(visit_counts, dtype=np.float64)
    action_probs = visit_counts ** (1 / temperature)
    action_probs = action_probs / action_probs.sum()
    if deterministic:
        action_pos = int(np.argmax(visit_counts))
    else:
        action_pos = int(L.rs().choice(len(visit_counts), p=action_probs))
    nz = action_probs[action_probs > 0]
    entropy = float(-(nz * np.log2(nz)).sum())
    return action_pos, entropy


def ez_network_output_unpack(network_output):
    """lzero/policy/utils.py:782-794: (latent_state, value_prefix, reward_hidden_state, value, policy_logits)"""
    return (network_output.latent_state, network_output.value_prefix, network_output.reward_hidden_state, network_output.value,
            network_output.policy_logits)


def mz_network_output_unpack(network_output):
    """lzero/policy/utils.py:796-807: (latent_state, reward, value, policy_logits)"""
    return network_output.latent_state, network_output.reward, network_output.value, network_output.policy_logits


class CheckpointIngest(object):
    """The reference policy's checkpoint entry points (lzero/policy/muzero.py:1037-1061: ``_state_dict_learn`` returns
    ``{'model', 'target_model', 'optimizer'}``, ``_load_state_dict_learn`` restores them) for an inference-only engine policy: the
    ONLINE network's weights (``'model'``) go into the collect / eval model -- a weight refresh when it is already loaded, the roots and
    their captured search graphs stay valid -- and the optimizer state is ignored.  Accepts the checkpoint as ``torch.load`` returns
    it, its ``'model'`` entry, or a bare reference-keyed state_dict (lightzero_amd.model.efficientzero_model.unwrap_checkpoint)."""

    def _load_state_dict_learn(self, state_dict):
        self._collect_model.load_state_dict(state_dict)
        if self._eval_model is not self._collect_model:
            self._eval_model.load_state_dict(state_dict)

    _load_state_dict_collect = _load_state_dict_eval = load_state_dict = _load_state_dict_learn
