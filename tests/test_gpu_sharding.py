"""GPU, two ranks on ONE device (gloo; RCCL refuses two ranks on one device): the collective-free passes of SURVEY §8e with
the REAL kernels — full-image evaluation sharded by row blocks (FruitModel.get_outputs_for_camera_ray_bundle,
/root/reference/fruit_nerf/fruit_nerf.py:225-249) and the volume export sharded by runs of ray batches (sample_volume,
/root/reference/fruit_nerf/export/exporter_utils.py:95-172) on a 64^3 lattice, fused lattice path and generic path.  Every
rank computes the single-process result itself and compares: the sharded outputs are BIT-equal on every rank.
(tests/test_sharding_cpu.py covers the host logic with stand-in kernels; the RCCL transport is the driver's 8-GPU run.)
Collected last (tests/conftest.py): it starts processes."""
import os
import socket
import types

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _image_rays(H, W, dev):
    """A pinhole camera on the unit sphere looking at the origin: [H, W, 3] origins / directions, camera index 2."""
    from fruitnerf_amd.rays import RayBundle
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    d_cam = torch.stack([(xs - W / 2 + 0.5) / (0.9 * W), -(ys - H / 2 + 0.5) / (0.9 * W), -torch.ones_like(xs)], dim=-1)
    c2w = torch.tensor([[0.8, 0.0, 0.6], [0.0, 1.0, 0.0], [-0.6, 0.0, 0.8]])
    d = torch.nn.functional.normalize(d_cam @ c2w.T, dim=-1)
    o = (c2w @ torch.tensor([0.0, 0.0, 1.0])).expand(H, W, 3).contiguous()
    cam = torch.full((H, W, 1), 2, dtype=torch.long)
    return RayBundle(o.to(dev), d.to(dev), torch.full((H, W, 1), 1e-4, device=dev), cam.to(dev))


def _worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from fruitnerf_amd import _lib
        from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
        from fruitnerf_amd.export.exporter_utils import sample_volume
        from tests.golden.make_reference_export_golden import export_state_dict
        from tests.test_gpu_reference_pins import _hip_model, _small_sd
        _lib.load()                                       # fails loudly without the HIP library
        dev = torch.device("cuda:0")
        res = {}
        # ---- e2: image evaluation by row blocks (37 rows over 2 ranks: 19 + 18; chunks of 512 rays: ragged) -------
        hm = _hip_model(dev, _small_sd(), None)
        hm.eval()
        hm.config.eval_num_rays_per_chunk = 512
        H, W = 37, 53
        single = hm.get_outputs_for_camera_ray_bundle(_image_rays(H, W, dev))
        sharded = hm.get_outputs_for_camera_ray_bundle(_image_rays(H, W, dev), rank=rank, world_size=world)
        ok = set(single) == set(sharded) and {"rgb", "semantics", "accumulation", "depth", "semantics_colormap"} <= set(single)
        for k in single:
            ok = ok and tuple(single[k].shape[:2]) == (H, W) and bool(torch.equal(single[k], sharded[k]))
        res["eval"] = bool(ok)
        res["eval_spread"] = float(single["rgb"].std()) > 1e-3          # a real image, not a constant
        # ---- e3: 64^3 export by runs of ray batches, fused lattice path and generic (explicit positions) path ----
        em = _hip_model(dev, export_state_dict(), "export")
        em.eval()
        n = 64
        em.setup_inference(True, n, deterministic=True)
        aabb = ((-0.9, -0.8, -1.0), (0.7, 0.8, 1.0))
        for path in ("fused", "generic"):
            out = []
            for r, w in ((0, 1), (rank, world)):
                dm = ExportDataManager(dev, eval_num_rays_per_batch=700)     # 4096 rays: 6 batches, 3 per rank
                n_rays = dm.setup_inference(aabb=aabb, num_points=n)
                if path == "generic":
                    dm.export_lattice = None
                pipe = types.SimpleNamespace(model=em, datamanager=dm)
                out.append(sample_volume(pipe, n_rays, transform_json={"scale": 0.5}, rank=r, world_size=w))
            ref, got = out
            ok = True
            for name in ref:
                ok = ok and bool(np.array_equal(ref[name]["points"], got[name]["points"]))
                ok = ok and bool(np.array_equal(ref[name]["colors"], got[name]["colors"]))
            res["export_" + path] = bool(ok)
            res["export_counts_" + path] = tuple(int(ref[k]["points"].shape[0]) for k in ("semantic_colormap", "semantic", "density"))
        q.put((rank, res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:   # the parent reports it instead of timing out
        import traceback
        q.put((rank, {"error": repr(exc), "trace": traceback.format_exc()}))


def test_sharded_eval_and_export_with_real_kernels_are_bit_equal_to_the_single_process(dev):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    for rank in range(world):
        r = results[rank]
        assert "error" not in r, r.get("trace")
        assert r["eval"] and r["eval_spread"], (rank, r)
        assert r["export_fused"] and r["export_generic"], (rank, r)
        c = r["export_counts_fused"]
        assert c == r["export_counts_generic"] and c[2] > c[0] > 0 and c[1] > 0, c
    assert results[0]["export_counts_fused"] == results[1]["export_counts_fused"]
    print("[sharded, real kernels] export counts", results[0]["export_counts_fused"])
