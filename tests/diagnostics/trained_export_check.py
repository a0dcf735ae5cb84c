"""Export point counts, HIP vs oracle, with TRAINED weights (values near the thresholds are the risk)."""
import sys, torch, copy
sys.path.insert(0, '/root/repo')
from fruitnerf_amd.data import synthetic_apple as sa
from fruitnerf_amd.fruit_nerf import FruitModel, FruitNerfModelConfig
from fruitnerf_amd.data.semantics import apple_metadata
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
from fruitnerf_amd.data.fruit_datamanager import ExportDataManager
from fruitnerf_amd.export.exporter_utils import sample_volume
from oracle import fruit_oracle as fo
dev = torch.device('cuda:0')
HW, n_train = 96, 40; focal = 1111.0 * HW / 800
scene = sa.make_scene(seed=0, device=dev); c2w = sa.make_cameras(n_train, seed=0, device=dev)
data = sa.render_dataset(scene, c2w, H=HW, W=HW, fx=focal, fy=focal)
batcher = sa.PixelBatcher(data, torch.arange(n_train, device=dev), seed=1)
torch.manual_seed(0)
hm = FruitModel(FruitNerfModelConfig(), apple_metadata(), num_train_data=n_train, device=dev); hm.train()
opt = FusedAdam(hm)
for step in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3000):
    o, d, cam, batch = batcher.sample(4096)
    fused_train_iteration(hm, opt, RayBundle(o, d, None, cam), batch, step, want_metrics=False)
N = 96
em = FruitModel(copy.deepcopy(hm.config), apple_metadata(), num_train_data=n_train, device=dev, test_mode="export")
em.load_state_dict(hm.state_dict(), strict=True); em.eval()
class P: pass
pipe = P(); pipe.model = em; pipe.datamanager = ExportDataManager(dev, eval_num_rays_per_batch=4096)
em.setup_inference(True, N, deterministic=True)
aabb = ((-0.5, -0.5, -0.5), (0.5, 0.5, 0.5))
n_rays = pipe.datamanager.setup_inference(aabb=aabb, num_points=N)
got = sample_volume(pipe, n_rays, transform_json={"scale": 1.0})
om = fo.FruitModel(fo.FruitNerfModelConfig(), num_train_data=n_train, test_mode="export")
om.load_state_dict({k: v.detach().cpu() for k, v in hm.state_dict().items()}, strict=True)
om.field.test_mode = "export"; om.eval(); om.setup_inference(True, N)
torch.set_num_threads(32)
ref = fo.sample_volume(om, aabb, N, num_rays_per_batch=4096, dataparser_scale=1.0)
for name in ("semantic_colormap", "semantic", "density"):
    a, b = got[name]["points"], ref[name]["points"].numpy()
    same = a.shape == b.shape and (a == b).all()
    print(name, "hip", a.shape[0], "oracle", b.shape[0], "identical lists" if same else "DIFFERENT")
