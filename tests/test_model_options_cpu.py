"""CPU: model options the engine classes refuse are refused BEFORE anything touches a device, with the reason (no silent approximation):
the convolutional Sampled EfficientZero takes discrete actions and BatchNorm only; fast mode exists for two observation shapes."""
import pytest


def test_conv_sampled_efficientzero_refuses_what_it_has_no_kernels_for():
    from lightzero_amd.model.sampled_efficientzero_model import SampledEfficientZeroModel
    base = dict(observation_shape=(4, 64, 64), action_space_size=6, num_of_sampled_actions=5, downsample=True, norm_type='BN')
    with pytest.raises(NotImplementedError, match="discrete"):
        SampledEfficientZeroModel(**dict(base, continuous_action_space=True))
    with pytest.raises(NotImplementedError, match="norm_type"):
        SampledEfficientZeroModel(**dict(base, norm_type='LN'))          # the reference class's default
    with pytest.raises(NotImplementedError, match="downsample"):
        SampledEfficientZeroModel(**dict(base, downsample=False))        # the reference class's default
    with pytest.raises(NotImplementedError, match="num_of_sampled_actions"):
        SampledEfficientZeroModel(**dict(base, num_of_sampled_actions=65))
    with pytest.raises(NotImplementedError, match="activation"):
        SampledEfficientZeroModel(**dict(base, activation="tanh"))
