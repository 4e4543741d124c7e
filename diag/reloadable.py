import torch, slsmi
from slsmi.model import Model
from slsmi.weights import synth_state_dict
from slsmi import training
dev = torch.device("cuda:0")
cfg = slsmi.BackboneConfig(layers=2)
m = Model(None, dev, cp_path=None, cfg=cfg, use_sae=True).train()
m.load_state_dict(synth_state_dict(cfg, "sae", 201, 1234))
m.to(dev)
training.freeze_unused(m)
for n, t in m.state_dict().items():
    if not t.is_floating_point():
        continue
    t2 = t.detach().to(dev, torch.float32).contiguous()
    if t2.data_ptr() != t.data_ptr():
        print("copied:", n, t.dtype, t.shape, t.is_contiguous(), t.stride())
wav = (0.1 * torch.randn(2, 64600)).to(dev)
y = torch.tensor([0, 1], device=dev)
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-5, fused=True)
for i in range(3):
    training.train_step(m, opt, wav, y)
    print("step", i, "reloadable", m._engine.reloadable, "dirty", m._engine_dirty)
