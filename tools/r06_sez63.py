"""round 6: the conv Sampled EfficientZero network the offset-56 sweep left above its bound (fuzz_sez63: 4x64x64 -> 8x8 latent, three residual
blocks, ReLU dynamics / GELU prediction, 64-wide heads, B = 29) under the kernel switches, with the binary64 anchor: per variant the worst
|device - torch fp32| / (1 + |x|) per tensor class, the device's and torch's own distance from a binary64 evaluation of the same network on the
same teacher-forced inputs.   python tools/r06_sez63.py [seed]   (on the GPU box) -> gpurun_out/r06_sez63.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNIP = r'''
import copy, json, sys
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import nn_cases, parity_record
from oracle import torch_models as tm
from test_nn_fuzz_gpu import _sez_case, _oracle_outputs, _fp32_cost
from test_nn_golden_gpu import check_case
seed = int(sys.argv[1])
case = _sez_case(seed)
model = tm.synthetic_init(nn_cases.oracle_class(tm, "sez")(**case["kw"]), seed=case["seed"]).eval()
g32 = _oracle_outputs(case, model)
g64 = _oracle_outputs(case, copy.deepcopy(model).double(), forced=g32)
cost = _fp32_cost(g32, g64)
loose = {k: 1.0 for k in ("latent", "policy", "scalar", "logits", "hc")}
w = check_case("probe_sez%d" % seed, case, g32, model.state_dict(), record="probe/", bounds=loose, g64=g64)
print("RESULT " + json.dumps(dict(vs_torch_fp32={k: v for k, v in w.items() if k != "vs64"}, device_vs_binary64=w["vs64"], torch_fp32_vs_binary64=cost,
                                  bounds={k: max(parity_record.BOUNDS[k], 3.0 * cost[k]) for k in cost})))
'''
seed = sys.argv[1] if len(sys.argv) > 1 else "63"
out = {}
for var in ("", "LZ_CHAIN_NO_SPLIT=1", "LZ_CONV_NO_SPLIT=1", "LZ_CHAIN_NO_SPLIT=1 LZ_CONV_NO_SPLIT=1", "LZ_HEADS_VALU=1", "LZ_LSTM_NOSPLIT=1", "LZ_HEADS_LAUNCH=1", "LZ_CONV_DIRECT=1", "LZ_CHAIN_DIRECT=1"):
    env = dict(os.environ, LZ_PARITY_OUT="/tmp/probe_parity.json")
    for kv in var.split():
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", SNIP, seed], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    out[var or "default"] = json.loads(line[0][7:]) if line else dict(error=(r.stdout[-500:] + r.stderr[-1500:]))
    print(var or "default", json.dumps(out[var or "default"])[:600])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_sez%s.json" % seed), "w"), indent=1, sort_keys=True)
