import cProfile, pstats, sys, os, io
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from lightzero_amd.model.synthetic import efficientzero_state_dict
from lightzero_amd.model.efficientzero_model import EfficientZeroModel
from lightzero_amd.policy.efficientzero import EfficientZeroPolicy
CFG = dict(num_simulations=50, pb_c_base=19652, pb_c_init=1.25, discount_factor=0.997, value_delta_max=0.01, lstm_horizon_len=5, root_noise_weight=0.25, root_dirichlet_alpha=0.3)
model = EfficientZeroModel(action_space_size=6).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=6))
pol = EfficientZeroPolicy(dict(CFG, device_select_action=True), model)
B = 256
obs = torch.rand(B, 4, 96, 96).cuda()
mask = np.ones((B, 6), np.float32)
for _ in range(5): pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B)
import time
t0 = time.perf_counter()
for _ in range(30): pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B)
print("per call %.3f ms" % ((time.perf_counter() - t0) / 30 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(30): pol._forward_collect(obs, action_mask=mask, temperature=1.0, to_play=[-1] * B)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4500])
