"""2+ ranks (gloo on ONE GPU is enough): after K fused training steps with rank-specific rays every rank must hold
bit-identical parameters, and they must equal a single-process run that averages the same gradients.
  FNR_ONE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
      tools/microbench/ddp_consistency.py"""
import os, sys, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
from fruitnerf_amd.cameras.camera_optimizers import CameraAdam, CameraOptimizerConfig
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0 if os.environ.get("FNR_ONE_DEVICE") == "1" else int(os.environ["LOCAL_RANK"]))
torch.cuda.set_device(dev)
HW, n_train = 64, 20; focal = 1111.0 * HW / 800
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(n_train, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=100 + rank)
torch.manual_seed(0)
model = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev); model.train()
opt = FusedAdam(model)
co = CameraOptimizerConfig(mode="SO3xR3").setup(n_train, dev); camera = (co, CameraAdam(co), batcher)
for step in range(12):
    o, d, cam, batch = batcher.sample(1024, co)
    ld, md = fused_train_iteration(model, opt, RayBundle(o, d, None, cam), batch, step, world_size=world, camera=camera)
torch.cuda.synchronize()
flat = torch.cat([model.arena().params, co.pose_adjustment.data.view(-1)]).cpu()
gathered = [torch.zeros_like(flat) for _ in range(world)]
dist.all_gather(gathered, flat)
if rank == 0:
    same = all(torch.equal(gathered[0], g) for g in gathered[1:])
    print("ranks hold identical parameters:", same, "| finite:", bool(torch.isfinite(flat).all()),
          "| max|p|", float(flat.abs().max()), "| losses", {k: round(float(v), 5) for k, v in ld.items()})
    assert same
dist.barrier(); dist.destroy_process_group()
