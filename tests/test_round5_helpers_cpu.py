"""CPU: host-side helpers added in round 5 -- the prior-sharpening of the synthetic weights (bench.py's depth sweep, the sharp-prior parity
cases), the flat state_dict a device-side weight refresh takes, the CPU allowance the baselines report."""
import numpy as np
import torch


def test_sharpen_scales_only_the_last_layers_of_the_policy_and_value_heads():
    from lightzero_amd.model.synthetic import efficientzero_state_dict, sharpen_state_dict
    sd = efficientzero_state_dict(seed=0, action_space_size=6)
    out = sharpen_state_dict(sd, 10.0, 4.0)
    changed = {k for k in sd if not np.array_equal(sd[k], out[k])}
    assert changed == {"prediction_network.fc_policy.3.weight", "prediction_network.fc_policy.3.bias",
                       "prediction_network.fc_value.3.weight", "prediction_network.fc_value.3.bias"}
    assert np.array_equal(out["prediction_network.fc_policy.3.weight"], sd["prediction_network.fc_policy.3.weight"] * 10.0)
    assert np.array_equal(out["prediction_network.fc_value.3.bias"], sd["prediction_network.fc_value.3.bias"] * 4.0)
    # and the sharpened policy head really is sharp: softmax of 10 x logits drawn at the recipe's scale
    z = np.random.default_rng(0).standard_normal((64, 6)) * 0.3
    p1 = np.exp(z) / np.exp(z).sum(1, keepdims=True)
    p10 = np.exp(10 * z) / np.exp(10 * z).sum(1, keepdims=True)
    assert p10.max(1).mean() > p1.max(1).mean() + 0.3


def test_flat_state_dict_is_one_buffer_in_name_order_with_views():
    from lightzero_amd import shard
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    sd = {k: torch.from_numpy(v) for k, v in efficientzero_state_dict(seed=1, action_space_size=6).items()}
    sd["representation_network.downsample_net.norm1.num_batches_tracked"] = torch.tensor(3)
    f = shard.flat_state_dict(sd, "cpu")
    names = sorted(k for k in sd if not k.endswith("num_batches_tracked"))
    assert [n for n, _, _ in f.layout] == names and list(f) == names
    off = 0
    for n, o, size in f.layout:
        assert o == off and size == sd[n].numel()
        assert f[n].shape == sd[n].shape and torch.equal(f[n], sd[n].float())
        assert f[n].data_ptr() == f.flat.data_ptr() + 4 * o        # consecutive views of the one buffer
        off += size
    assert f.flat.numel() == off
    # a single-rank "broadcast" of it is the object itself when it already lives where the collective would put it
    assert shard.broadcast_state_dict(f, src=0, on_device=False).keys() == dict(f).keys()


def test_cpu_allowance_reports_affinity_and_quota():
    import bench
    a = bench.cpu_allowance()
    assert a["affinity"] >= 1 and (a["quota"] is None or a["quota"] > 0)
    phys, thr = bench.host_cores()
    assert 1 <= phys <= thr
