"""Per-parameter gradient error of the HIP path vs tests/golden/fruit_nerf_big_small.npz (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import util
from tests.test_golden import _load, _inputs, _big_config, BIG_GOLD, GOLD
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
dev = torch.device("cuda:0")
for which in ("base", "big"):
    g, sd = _load(GOLD if which == "base" else BIG_GOLD)
    oc = util.small_config(log2=10, prop_log2=8) if which == "base" else _big_config()
    cfg = FruitNerfModelConfig()
    for k, v in vars(oc).items():
        if hasattr(cfg, k):
            setattr(cfg, k, v)
    hm = FruitModel(cfg, apple_metadata(), num_train_data=5, device=dev)
    hm.load_state_dict(sd, strict=True)
    o, d, pa, cam, jit, batch = _inputs(g, dev)
    hm.train(); hm.set_anneal(0)
    tr = hm(RayBundle(o, d, pa, cam), jitter=jit)
    ld = hm.get_loss_dict(tr, batch)
    sum(ld.values()).backward()
    torch.cuda.synchronize()
    for name, p in hm.named_parameters():
        if "hash_table" in name:
            continue
        ref = torch.from_numpy(g["grad::" + name])
        diff = (p.grad.cpu() - ref).abs()
        i = int(diff.argmax())
        print(f"{which:5s} {name:50s} max|ref| {ref.abs().max():.3e} max_err {diff.max():.3e} rel {diff.max()/max(ref.abs().max(),1e-30):.2e} "
              f"at {np.unravel_index(i, ref.shape)} n_bad {(diff > 2e-3*ref.abs().max()).sum().item()}")
