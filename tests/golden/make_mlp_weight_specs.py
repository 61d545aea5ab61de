"""Writes lightzero_amd/model/synthetic_mlp_specs.json: tensor names and shapes of the reference's MLP model families for
the two benchmarked configurations (BASELINE.json configs[0] and configs[4]), read off the torch restatement in
oracle/torch_models.py.  lightzero_amd.model.synthetic.mlp_state_dict() fills them with seeded synthetic weights, so that
the timing tools need neither a checkpoint nor the oracle.

    python tests/golden/make_mlp_weight_specs.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    from oracle import torch_models as tm
    specs = {
        "muzero_mlp_cartpole": tm.MuZeroModelMLP(observation_shape=4, action_space_size=2, latent_state_dim=128),
        "sampled_efficientzero_mlp_dmc": tm.SampledEfficientZeroModelMLP(observation_shape=5, action_space_size=1,
                                                                       num_of_sampled_actions=20),
    }
    out = {k: {n: list(v.shape) for n, v in m.state_dict().items() if not n.endswith("num_batches_tracked")} for k, m in specs.items()}
    with open(os.path.join(ROOT, "lightzero_amd", "model", "synthetic_mlp_specs.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print({k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
