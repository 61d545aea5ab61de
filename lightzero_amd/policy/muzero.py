"""Collect / eval halves of lzero/policy/muzero.py::MuZeroPolicy (``_forward_collect`` :705-829, ``_forward_eval``
:842-930) on the MI355X engine; same arguments and per-env output dict as the reference."""
from ..mcts.tree_search.mcts_ctree import MuZeroMCTSCtree
from .efficientzero import EfficientZeroPolicy, _g, _mcts_seed


class MuZeroPolicy(EfficientZeroPolicy):
    def __init__(self, cfg, model):
        super().__init__(cfg, model)
        self._mcts_collect = MuZeroMCTSCtree(self._mcfg)
        self._mcts_eval = MuZeroMCTSCtree(self._mcfg)

    def _roots(self, n, legal_actions):
        roots = self._roots_cache.get(n)
        if roots is None:
            roots = MuZeroMCTSCtree.roots(n, legal_actions, action_space_size=self._collect_model.action_space_size,
                                          max_simulations=int(self._mcfg["num_simulations"]))
            roots.set_tiebreak(self._tiebreak, seed=_mcts_seed(self._cfg))
            self._roots_cache[n] = roots
        else:
            roots.reset(legal_actions)
        return roots

    def _search(self, mcts, roots, model, network_output, to_play):
        mcts.search(roots, model, network_output.latent_state, to_play)
