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


def big_config(log2=14, prop_log2=12, max_res=4096):
    """What `fruit_nerf_big` / `fruit_nerf_huge` change in the FIELD (fruit_nerf_config.py:82-95 via fruit_nerf.py:88-103):
    geo 30, semantic MLP 30 -> 128 -> 128 -> 64, max_res 4096 — on small tables."""
    cfg = small_config(log2=log2, prop_log2=prop_log2, max_res=max_res)
    cfg.geo_feat_dim, cfg.num_layers_semantic, cfg.hidden_dim_semantics = 30, 3, 128
    cfg.proposal_weights_anneal_max_num_iters = 5000
    return cfg


def huge_config(log2=14, prop_log2=12):
    """`fruit_nerf_huge` (fruit_nerf_config.py:113-164) on small tables: the field of `fruit_nerf_big` at max_res 8192 and
    ITS proposal networks — 5 levels up to 512 and 7 levels up to 2048 (the only config with a 7-level proposal grid)."""
    cfg = big_config(log2=log2, prop_log2=prop_log2, max_res=8192)
    cfg.proposal_net_args_list = [
        {"hidden_dim": 16, "log2_hashmap_size": prop_log2, "num_levels": 5, "max_res": 512, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": prop_log2, "num_levels": 7, "max_res": 2048, "use_linear": False},
    ]
    return cfg


def fruit_nerf_big_config():
    """The model part of the `fruit_nerf_big` method at its real sizes (fruit_nerf_config.py:82-95)."""
    cfg = fo.FruitNerfModelConfig(log2_hashmap_size=21, max_res=4096)
    cfg.geo_feat_dim, cfg.num_layers_semantic, cfg.hidden_dim_semantics = 30, 3, 128
    cfg.num_nerf_samples_per_ray, cfg.num_proposal_samples_per_ray = 128, (512, 256)
    cfg.proposal_weights_anneal_max_num_iters = 5000
    return cfg


def fruit_nerf_huge_config():
    """The model part of the `fruit_nerf_huge` method at its real sizes (fruit_nerf_config.py:113-164): the big field at
    max_res 8192, 512/512/64 samples, proposal networks 5 levels -> 512 and 7 levels -> 2048 (T = 2^17)."""
    cfg = fruit_nerf_big_config()
    cfg.max_res = 8192
    cfg.num_nerf_samples_per_ray, cfg.num_proposal_samples_per_ray = 64, (512, 512)
    cfg.proposal_net_args_list = [
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 5, "max_res": 512, "use_linear": False},
        {"hidden_dim": 16, "log2_hashmap_size": 17, "num_levels": 7, "max_res": 2048, "use_linear": False},
    ]
    return cfg


def smooth_tables_(model: fo.FruitModel):
    """Give the hash tables the spectrum of a trained field: level l's entries scaled by base_res / res_l, so that every
    level contributes the same SLOPE (a white table at max_res 4096 turns 1e-6 of sample position into 4e-3 of a
    feature — the encoding of noise, not of a scene; trained fine levels are small corrections)."""
    with torch.no_grad():
        for enc in [model.field.mlp_base_grid] + [n.encoding for n in model.proposal_networks]:
            T = enc.hash_table.shape[0] // enc.num_levels
            for l in range(enc.num_levels):
                enc.hash_table[l * T:(l + 1) * T].mul_(float(enc.scalings[0]) / float(enc.scalings[l]))
    return model


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


def straddle_export_thresholds(model: fo.FruitModel, aabb, n_side: int = 24, logit_spread: float = 1.5):
    """Re-centre an export-mode oracle model so that the exporter's thresholds (density >= 70, logit >= 3,
    sigmoid(logit) > 0.9; exporter_utils.py:111-114) cut through the middle of the sample distribution instead of
    keeping ~all or ~none of the lattice: shifts the density-logit bias and rescales / shifts SemanticFieldHead."""
    was_training = model.training
    model.eval()
    model.setup_inference(True, n_side)
    corners = fo.get_corners_of_aabb(aabb)
    pts, vec = fo.sample_surface_points(corners, n=n_side)
    with torch.no_grad():
        out = model(fo.OrthographicRayGenerator(pts, vec, pts.shape[0])(1))
        dens, logit = out["density"].reshape(-1), out["semantics"].reshape(-1)
        inside = dens > 0
        f = model.field
        # + 0.0123: the median sample must not sit exactly ON the threshold (it would, to fp32 rounding)
        f.mlp_base_mlp.layers[1].bias[0].add_(float(torch.log(torch.tensor(70.0)) - dens[inside].log().median()) + 0.0123)
        head = f.field_head_semantics.net
        z = logit[inside] - head.bias
        k = logit_spread / float(z.std())
        head.weight.mul_(k)
        head.bias.fill_(2.6 - k * float(z.median()))
    model.train(was_training)
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
    from fruitnerf_amd.data.semantics import apple_metadata
    oc = oracle_model.config
    cfg = FruitNerfModelConfig()
    for k, v in vars(oc).items():
        if hasattr(cfg, k):
            setattr(cfg, k, copy.deepcopy(v))
    m = FruitModel(cfg, apple_metadata(), num_train_data=oracle_model.field.num_images, device=device, test_mode=test_mode)
    missing = m.load_state_dict(oracle_model.state_dict(), strict=True)
    return m


def named_trainable(model):
    """named_parameters() without nerfstudio's zero-length `device_indicator_param` (part of the checkpoint contract, never
    of the graph: it has no gradient on either side)."""
    return [(n, p) for n, p in model.named_parameters() if p.numel() > 0]


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
