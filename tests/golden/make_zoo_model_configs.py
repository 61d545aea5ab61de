"""Writes tests/golden/zoo_model_configs.json: for every configuration file the reference ships under zoo/**/config/ whose policy is one
of the four families on the hot path (muzero, efficientzero, sampled_efficientzero, gumbel_muzero), the keywords its model is built
with -- ``main_config.policy.model`` as the file sets it, completed with the defaults of the reference class the policy would
construct (lzero/policy/muzero.py:237-259 & co.: model_type 'conv' | 'mlp' -> class), so that "the class default" is the REFERENCE's
default and not the restatement's.  Data only: file path, policy type, a keyword dictionary, num_simulations, collector_env_num.

    python tests/golden/make_zoo_model_configs.py          (needs /root/reference; run from the repository root)

tests/test_zoo_configs_gpu.py builds every entry on the engine: accepted ones are held to the torch restatement of the same
keywords, refused ones must say why; the outcome table is profiles/rNN_zoo_configs.json."""
import glob
import inspect
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_loader  # noqa: E402

REF = "/root/reference"
FAMILIES = {"muzero": "mz", "gumbel_muzero": "mz", "efficientzero": "ez", "sampled_efficientzero": "sez"}


def plain(v):
    """JSON-able rendering of a keyword value; torch modules by class name (nn.GELU(approximate='tanh') -> 'GELU(tanh)')"""
    import torch.nn as nn
    if isinstance(v, nn.GELU):
        return "GELU(%s)" % v.approximate
    if isinstance(v, nn.Module):
        return type(v).__name__
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    if isinstance(v, dict):
        return {str(k): plain(x) for k, x in v.items()}
    if isinstance(v, (int, float, str, bool)) or v is None:
        return v
    return repr(v)


def main():
    import torch
    torch.cuda.set_device = lambda *a, **k: None     # two configuration files select a device at import time
    ref = ref_loader.load()
    assert ref is not None, "/root/reference is needed"
    sys.path.insert(0, REF)
    classes = {("mz", "conv"): ref.muzero_model.MuZeroModel, ("mz", "mlp"): ref.muzero_model_mlp.MuZeroModelMLP,
               ("ez", "conv"): ref.efficientzero_model.EfficientZeroModel, ("ez", "mlp"): ref.efficientzero_model_mlp.EfficientZeroModelMLP,
               ("sez", "conv"): ref.sampled_efficientzero_model.SampledEfficientZeroModel,
               ("sez", "mlp"): ref.sampled_efficientzero_model_mlp.SampledEfficientZeroModelMLP}
    out, skipped = {}, {}
    for f in sorted(glob.glob(os.path.join(REF, "zoo", "**", "config", "*.py"), recursive=True)):
        rel = os.path.relpath(f, REF)
        if f.endswith("__init__.py"):
            continue
        try:
            saved = sys.stdout
            sys.stdout = open(os.devnull, "w")
            try:
                ns = runpy.run_path(f, run_name="scan")
            finally:
                sys.stdout = saved
        except BaseException as e:   # files of other families that import packages this image lacks
            skipped[rel] = repr(e)[:120]
            continue
        mc, cc = ns.get("main_config"), ns.get("create_config")
        if mc is None or cc is None:
            continue
        ptype = cc["policy"]["type"]
        if ptype not in FAMILIES:
            continue
        model = dict(mc["policy"].get("model", {}))
        mtype = model.get("model_type", "conv")
        entry = dict(policy_type=ptype, family=FAMILIES[ptype], model_type=mtype, num_simulations=mc["policy"].get("num_simulations"),
                     collector_env_num=mc["policy"].get("collector_env_num"), env_type=mc["policy"].get("env_type", "not_board_games"),
                     discount_factor=mc["policy"].get("discount_factor", 0.997),
                     max_num_considered_actions=mc["policy"].get("max_num_considered_actions"), set_by_file=sorted(model))
        # the policy dictionary as the file sets it (without the model), for the policy surface (_forward_collect / _forward_eval)
        entry["policy"] = plain({k: v for k, v in mc["policy"].items() if k != "model"})
        cls = classes.get((FAMILIES[ptype], mtype))
        if cls is None:
            entry["model"] = plain(model)      # e.g. model_type 'conv_context': a model class outside the four families' two
            entry["reference_class"] = None
        else:
            kw = {}
            for name, p in inspect.signature(cls.__init__).parameters.items():
                if name in ("self", "args", "kwargs") or p.default is inspect.Parameter.empty:
                    continue
                kw[name] = p.default
            # what the file sets but the class has no parameter for lands in the constructor's **kwargs and changes nothing
            # (atari_muzero_config.py sets use_sim_norm=True on a MuZeroModel, which does not take it)
            entry["ignored_by_reference_class"] = sorted(k for k in model if k not in kw)
            kw.update({k: v for k, v in model.items() if k in kw})
            entry["model"] = plain(kw)
            entry["reference_class"] = cls.__name__
        out[rel] = entry
    path = os.path.join(HERE, "zoo_model_configs.json")
    json.dump(dict(configs=out, not_loadable_here=skipped), open(path, "w"), indent=1, sort_keys=True)
    print(len(out), "configurations ->", path, "(%d files of other families not loadable here)" % len(skipped))


if __name__ == "__main__":
    main()
