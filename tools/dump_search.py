#!/usr/bin/env python
"""Debugging aid: runs one seeded EfficientZero search (B x S, deterministic tie-break, host noise) and dumps everything it produced
(visit counts, root values, every slot's scalars / policy logits, LSTM states and latents of a few slots) to an .npz -- two kernel
variants (LZ_* switches, one process each) are then compared bit for bit with `python tools/dump_search.py --compare a.npz b.npz`."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if sys.argv[1] == "--compare":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint32) if a[k].dtype == np.float32 else a[k], b[k].view(np.uint32) if b[k].dtype == np.float32 else b[k])]
        for k in bad:
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            print("DIFFERS %-10s max |d| %.3g  (%d of %d elements)" % (k, d.max(), (d > 0).sum(), d.size))
        print("bit-identical: %s" % (not bad))
        sys.exit(1 if bad else 0)
    out, B, S = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256, int(sys.argv[3]) if len(sys.argv) > 3 else 50
    import torch
    from lightzero_amd import _lib as L
    from lightzero_amd.mcts.ctree.ctree_efficientzero import ez_tree
    from lightzero_amd.model.efficientzero_model import EfficientZeroModel
    from lightzero_amd.model.synthetic import efficientzero_state_dict
    A = 6
    lib = L.lib()
    model = EfficientZeroModel(action_space_size=A).load_state_dict(efficientzero_state_dict(seed=0, action_space_size=A))
    roots = ez_tree.Roots(B, [list(range(A))] * B, action_space_size=A, max_simulations=S, engine=model.engine)
    roots.set_tiebreak(0); roots._ensure(A)
    obs = torch.rand(B, 4, 96, 96, generator=torch.Generator().manual_seed(9)).cuda().contiguous()
    noise = np.random.default_rng(4).dirichlet([0.3] * A, size=B).astype(np.float32)
    torch.cuda.synchronize()
    L.check(lib.lz_initial_inference(roots._h, obs.data_ptr()))
    L.check(lib.lz_roots_prepare_from_inference(roots._h, 0.25, noise.ctypes.data, L.i32([-1] * B)))
    L.check(lib.lz_search(roots._h, S, 19652, 1.25, 0.997, 5, 0.01))
    d = dict(dist=np.array(roots.get_distributions(), np.int32), val=np.array(roots.get_values(), np.float32))
    vp = np.zeros((S + 1, B), np.float32); v = np.zeros_like(vp); pol = np.zeros((S + 1, B, A), np.float32)
    hh = np.zeros((S + 1, B, 512), np.float32); cc = np.zeros_like(hh)
    for s in range(S + 1):
        L.check(lib.lz_roots_read_sim_outputs(roots._h, s, vp[s], v[s], pol[s].reshape(-1)))
        L.check(lib.lz_roots_read_hidden(roots._h, s, hh[s].reshape(-1), cc[s].reshape(-1)))
    lat = np.zeros((B, 64, 6, 6), np.float32)
    L.check(lib.lz_roots_read_latent(roots._h, S, lat.reshape(-1)))
    d.update(vp=vp, v=v, pol=pol, h=hh, c=cc, lat=lat)
    np.savez(out, **d)
    print("wrote", out)


if __name__ == "__main__":
    main()
