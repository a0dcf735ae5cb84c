"""CPU, gloo world_size 2: ray-range sharding of the collective-free passes — full-image evaluation by row
blocks and the volume export by runs of ray batches.  The field kernels need a GPU, so the model is an analytic
stand-in and `export_compact` a torch restatement of the thresholds; what is under test is the host logic: block
boundaries, batch counters, variable-length all-gathers and that rank-ordered concatenation reproduces the
single-process result exactly."""
import os
import socket
import types

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fruitnerf_amd.sharding import shard_range


def test_shard_range_tiles_the_range_in_rank_order():
    for n in (0, 1, 5, 64, 799, 640000):
        for world in (1, 2, 3, 8):
            for granule in (1, 4, 800):
                blocks = [shard_range(n, r, world, granule) for r in range(world)]
                assert blocks[0][0] == 0 and blocks[-1][1] == n
                assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
                assert all(lo % granule == 0 or lo == n for lo, _ in blocks)
                sizes = [-(-(hi - lo) // granule) for lo, hi in blocks]
                assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_export_compact(lat, ray_begin, n_rays, positions, density, rgb, logit, points, colors, counts):
    """fnr_export_compact's contract (include/fruitnerf_hip.h) in torch: three order-preserving selections."""
    assert lat is None
    pos = positions.reshape(-1, 3)
    sig = torch.sigmoid(logit)
    dense = density >= 70
    masks = [(sig > 0.9) & dense, (logit >= 3) & dense, dense]
    extra = [sig, sig, torch.sigmoid(density)]
    for s, m in enumerate(masks):
        k = int(m.sum())
        counts[s] = k
        if k <= points[s].shape[0]:
            points[s][:k] = pos[m]
            colors[s][:k] = torch.cat([rgb[m], extra[s][m][:, None]], dim=1)


class _AnalyticExportModel:
    """Export-mode stand-in: N samples per orthographic ray, fields are closed-form functions of the position."""
    device = torch.device("cpu")
    num_inference_samples = None          # never equals the lattice's n_samples: the generic export path is taken

    def __init__(self, n_samples):
        self.n = n_samples

    def __call__(self, ray_bundle):
        o, d = ray_bundle.origins, ray_bundle.directions
        t = (torch.arange(self.n, dtype=torch.float32) + 0.5) / self.n * ray_bundle.fars
        pos = o[:, None, :] + d[:, None, :] * t[..., None]
        r2 = (pos ** 2).sum(-1)
        return {"point_location": pos, "density": 200.0 * torch.exp(-4.0 * r2),
                "rgb": torch.sigmoid(pos), "semantics": 6.0 - 14.0 * r2}


def _export(rank, world, batch):
    from fruitnerf_amd import _kernels as K
    from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
    from fruitnerf_amd.export.exporter_utils import sample_volume
    K.export_compact = _fake_export_compact
    dm = ExportDataManager("cpu", eval_num_rays_per_batch=batch)
    n_rays = dm.setup_inference(aabb=((-1.0, -0.6, -1.0), (1.0, 0.6, 1.0)), num_points=20)
    pipe = types.SimpleNamespace(model=_AnalyticExportModel(20), datamanager=dm)
    return sample_volume(pipe, n_rays, transform_json={"scale": 0.5}, rank=rank, world_size=world)


class _RowModel:
    """Eval stand-in for get_outputs_for_camera_ray_bundle: outputs are functions of the ray origin."""
    def __init__(self, chunk):
        self.config = types.SimpleNamespace(eval_num_rays_per_chunk=chunk)
        self.calls = []

    def forward(self, ray_bundle):
        self.calls.append(ray_bundle.origins.shape[0])
        o = ray_bundle.origins
        return {"rgb": torch.sin(o), "depth": o.sum(-1, keepdim=True), "note": "not a tensor"}


def _eval(rank, world, H, W, chunk):
    from fruitnerf_amd.fruit_nerf import FruitModel
    from fruitnerf_amd.rays import RayBundle
    g = torch.Generator().manual_seed(3)
    o = torch.rand(H, W, 3, generator=g)
    m = _RowModel(chunk)
    m._eval_output_templates = types.MethodType(FruitModel._eval_output_templates, m)
    out = FruitModel.get_outputs_for_camera_ray_bundle(m, RayBundle(o, torch.zeros_like(o)), rank=rank,
                                                       world_size=world)
    return out, m.calls


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fruitnerf_amd.sharding import all_gather_rows
    res = {}
    # ragged gather, one rank empty
    local = torch.arange(6, dtype=torch.float32).view(3, 2) + 10 * rank if rank == 1 else torch.zeros(0, 2)
    got = all_gather_rows(local, world)
    res["ragged"] = bool(torch.equal(got, torch.arange(6, dtype=torch.float32).view(3, 2) + 10))
    # export: sharded == single process, for batch sizes that give 1, 2 and 5 batches (288 rays)
    ok = True
    for batch in (4096, 200, 64):
        ref = _export(0, 1, batch)
        got = _export(rank, world, batch)
        for name in ref:
            ok &= bool(np.array_equal(ref[name]["points"], got[name]["points"]))
            ok &= bool(np.array_equal(ref[name]["colors"], got[name]["colors"]))
        ok &= ref["density"]["points"].shape[0] > ref["semantic"]["points"].shape[0] > 0
    res["export"] = ok
    # eval: sharded == single process; H=5 rows over 2 ranks -> 3 + 2 rows; H=1 leaves rank 1 without rows
    ok = True
    for H, W, chunk in ((5, 7, 10), (1, 9, 4), (6, 4, 1000)):
        ref, _ = _eval(0, 1, H, W, chunk)
        got, calls = _eval(rank, world, H, W, chunk)
        ok &= set(ref) == set(got) == {"rgb", "depth"}
        ok &= all(torch.equal(ref[k], got[k]) for k in ref)
        lo, hi = shard_range(H * W, rank, world, granule=W)
        ok &= sum(calls) == max(hi - lo, 1 if hi == lo else 0)      # only its own rows (+ the 1-ray template pass)
    res["eval"] = ok
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_export_and_eval_reproduce_the_single_process_results():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        assert results[rank] == {"ragged": True, "export": True, "eval": True}, (rank, results[rank])
