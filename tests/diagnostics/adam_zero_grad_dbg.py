"""Per-step comparison of the proposal-network parameters, HIP FusedAdam(skip_groups_without_grad=False) vs torch.optim
fed zero gradients (diagnostic for tests/test_gpu_training_parity.py::test_optimizer_can_step_groups_without_gradient_like_torch_1_13)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import util
from tests.test_gpu_training_parity import _batch, _oracle_step
from fruitnerf_amd.rays import RayBundle
from fruitnerf_amd.training import FusedAdam, fused_train_iteration
dev = torch.device("cuda:0")
cfg = util.small_config(log2=12, prop_log2=10)
om = util.make_oracle(cfg, seed=31)
hm = util.make_hip_like(om, dev)
om.train(); hm.train()
groups = om.get_param_groups()
opts = {"proposal_networks": torch.optim.Adam(groups["proposal_networks"], lr=1e-2, eps=1e-15),
        "fields": torch.optim.Adam(groups["fields"], lr=1e-2, eps=1e-15)}
hopt = FusedAdam(hm, skip_groups_without_grad=False)
for m in (om, hm):
    m.proposal_sampler._step = 8
    m.proposal_sampler._steps_since_update = 1
R = 96
named_h = dict(hm.named_parameters())
for step in range(9, 13):
    o, d, pa, cam = util.random_rays(R, 7, seed=300 + step)
    jit = [torch.rand(R, 1) for _ in range(3)]
    batch = _batch(R, 70 + step)
    for op in opts.values():
        op.zero_grad(set_to_none=False)
    for p in groups["proposal_networks"]:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    before = {n: p.detach().clone() for n, p in om.named_parameters()}
    _oracle_step(om, o, d, pa, cam, jit, batch, step)
    gmax = max(float(p.grad.abs().max()) for p in groups["proposal_networks"])
    for op in opts.values():
        op.step()
    om.proposal_sampler.step_cb(step)
    hb = {n: p.detach().clone() for n, p in hm.named_parameters()}
    fused_train_iteration(hm, hopt, RayBundle(o.to(dev), d.to(dev), pa.to(dev), cam.to(dev)),
                          {k: v.to(dev) for k, v in batch.items()}, step, jitter=[j.to(dev) for j in jit])
    torch.cuda.synchronize()
    print(f"step {step}: oracle max|grad prop| {gmax:.3e}  group_steps {hopt.group_steps}  torch steps "
          f"{[int(opts['proposal_networks'].state[p]['step']) for p in groups['proposal_networks'][:2]]}")
    for n, p in om.named_parameters():
        if n.startswith("proposal_networks.0"):
            mo = float((p.detach() - before[n]).abs().max())
            mh = float((named_h[n].detach() - hb[n]).abs().max())
            err = float((named_h[n].detach().cpu() - p.detach()).abs().max())
            print(f"    {n}: moved oracle {mo:.3e} hip {mh:.3e}  |hip - oracle| {err:.3e}")
