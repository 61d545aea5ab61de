"""Writes tests/golden/nn_<case>.npz: outputs of the REFERENCE's own model modules (/root/reference/lzero/model/*.py imported
as they lie, tests/ref_loader.py) and of its InverseScalarTransform on the seeded weights / inputs of tests/nn_cases.py.
Run in the build container (the GPU box has no /root/reference):   python tests/golden/make_golden_nn.py
The recurrent steps are teacher-forced: step s consumes the reference's own (latent, h, c) of step s - 1, all stored."""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import nn_cases  # noqa: E402
import ref_loader  # noqa: E402
from oracle import torch_models as tm  # noqa: E402


def weights_digest(state_dict):
    h = hashlib.sha256()
    for k in sorted(state_dict):
        if k.endswith("num_batches_tracked"):
            continue
        h.update(k.encode())
        v = state_dict[k]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        h.update(np.ascontiguousarray(v, np.float32).tobytes())
    return h.hexdigest()


def main():
    ref = ref_loader.load()
    assert ref is not None, "/root/reference is needed to (re)generate the goldens"
    st = ref.scaling_transform
    only = set(sys.argv[1:])   # optional: regenerate only these cases
    for name, case in sorted(nn_cases.CASES.items()):
        if only and name not in only:
            continue
        fam, kw = case["family"], case["kw"]
        ora = tm.synthetic_init(nn_cases.oracle_class(tm, fam)(**kw), seed=case["seed"])   # only as the seeded weight recipe
        rmod = nn_cases.reference_class(ref, fam)(**nn_cases.reference_kwargs(case))
        res = rmod.load_state_dict(ora.state_dict(), strict=False)
        assert not res.unexpected_keys and not res.missing_keys
        rmod.eval()
        support = kw.get("value_support_range", kw.get("support_range", (-300., 301., 1.)))
        cat = bool(kw.get("categorical_distribution", True))   # False: the heads' one output is the scaled scalar (scaling_transform.py:88-89)
        ist = st.InverseScalarTransform(st.DiscreteSupport(*support), cat)
        rist = st.InverseScalarTransform(st.DiscreteSupport(*kw.get("reward_support_range", support)), cat)   # mcts_ctree.py:726-729
        obs, actions = nn_cases.inputs(case)
        out = {"weights_sha256": np.frombuffer(weights_digest(ora.state_dict()).encode(), np.uint8)}
        lstm = nn_cases.has_lstm(fam)
        with torch.no_grad():
            r = rmod.initial_inference(torch.from_numpy(obs))
            out["init_latent"] = r.latent_state.numpy()
            out["init_value_logits"] = r.value.numpy()
            out["init_value"] = ist(r.value.clone()).reshape(-1).numpy()
            out["init_policy"] = r.policy_logits.numpy()
            lat = r.latent_state
            hc = r.reward_hidden_state if lstm else None
            for s in range(nn_cases.STEPS):
                a = torch.from_numpy(actions[s])
                out["s%d_in_latent" % s] = lat.numpy()
                if lstm:
                    out["s%d_in_h" % s], out["s%d_in_c" % s] = hc[0][0].numpy(), hc[1][0].numpy()
                    r = rmod.recurrent_inference(lat, hc, a)
                    out["s%d_h" % s], out["s%d_c" % s] = r.reward_hidden_state[0][0].numpy(), r.reward_hidden_state[1][0].numpy()
                    rew_logits = r.value_prefix
                    hc = r.reward_hidden_state
                else:
                    r = rmod.recurrent_inference(lat, a)
                    rew_logits = r.reward
                out["s%d_latent" % s] = r.latent_state.numpy()
                out["s%d_reward_logits" % s] = rew_logits.numpy()
                out["s%d_reward" % s] = rist(rew_logits.clone()).reshape(-1).numpy()
                out["s%d_value_logits" % s] = r.value.numpy()
                out["s%d_value" % s] = ist(r.value.clone()).reshape(-1).numpy()
                out["s%d_policy" % s] = r.policy_logits.numpy()
                lat = r.latent_state
        path = os.path.join(HERE, "nn_%s.npz" % name)
        np.savez_compressed(path, **{k: np.ascontiguousarray(v) for k, v in out.items()})
        print(name, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
