"""Shared helpers for the parity tests: build an oracle model and the HIP model with identical weights."""
import copy

import torch

from oracle import fruit_oracle as fo


def small_config(log2=14, prop_log2=12, max_res=2048):
    cfg = fo.FruitNerfModelConfig(log2_hashmap_size=log2, max_res=max_res)
    cfg.proposal_net_args_list = [
        {"hidden_dim": 16, "log2_hashmap_size": prop_log2, "num_levels": 5, "max_res": 128, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": prop_log2, "num_levels": 5, "max_res": 256, "use_linear": False},
    ]
    return cfg


def full_config():
    return fo.FruitNerfModelConfig()


def randomize_(model: fo.FruitModel, seed: int, density_boost: float = 2.0):
    """'Trained-like' parameters: O(1) hash features, non-trivial densities and logits."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        f = model.field
        f.mlp_base_grid.hash_table.copy_((torch.rand(f.mlp_base_grid.hash_table.shape, generator=g) * 2 - 1) * 0.6)
        for net in model.proposal_networks:
            net.encoding.hash_table.copy_((torch.rand(net.encoding.hash_table.shape, generator=g) * 2 - 1) * 0.8)
            net.mlp_base[1].layers[1].bias.add_(density_boost)
        f.mlp_base_mlp.layers[1].weight[0].mul_(3.0)
        f.mlp_base_mlp.layers[1].bias[0].add_(density_boost)
        f.field_head_semantics.net.weight.mul_(6.0)
        f.embedding_appearance.embedding.weight.copy_(
            torch.randn(f.embedding_appearance.embedding.weight.shape, generator=g))
    return model


def make_oracle(cfg, num_images=7, seed=0, test_mode=None, randomize=True):
    torch.manual_seed(seed)
    m = fo.FruitModel(cfg, num_train_data=num_images, test_mode=test_mode)
    if randomize:
        randomize_(m, seed + 1)
    return m


def make_hip_like(oracle_model: fo.FruitModel, device, test_mode=None):
    """HIP FruitModel with the oracle's weights, loaded through the (strict) state-dict contract."""
    from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
    oc = oracle_model.config
    cfg = FruitNerfModelConfig()
    for k, v in vars(oc).items():
        if hasattr(cfg, k):
            setattr(cfg, k, copy.deepcopy(v))
    m = FruitModel(cfg, num_train_data=oracle_model.field.num_images, device=device, test_mode=test_mode)
    missing = m.load_state_dict(oracle_model.state_dict(), strict=True)
    return m


def random_rays(R, num_images, seed=0, device="cpu"):
    """Origins on the unit sphere looking at the scene centre (SURVEY §8d micro-benchmark rays)."""
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
    tgt = (torch.rand(R, 3, generator=g) - 0.5) * 0.6
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    cam = torch.randint(0, num_images, (R, 1), generator=g)
    pa = torch.full((R, 1), 1e-6)
    return o.to(device), d.to(device), pa.to(device), cam.to(device)


def report(name, got, ref, atol=None, rtol=None):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    diff = (got - ref).abs()
    denom = ref.abs().clamp_min(1e-12)
    msg = (f"[parity] {name}: shape {tuple(ref.shape)} max_abs {diff.max().item():.3e} "
           f"max_rel {(diff / denom).max().item():.3e} mean_abs {diff.mean().item():.3e} "
           f"ref_absmax {ref.abs().max().item():.3e}")
    print(msg)
    return diff.max().item(), (diff / denom).max().item()
