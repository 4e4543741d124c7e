import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slsmi
from slsmi import training
from slsmi.model import Model
from slsmi.config import XLSR_300M
from slsmi.weights import synth_state_dict
dev = torch.device('cuda:0')
model = Model(None, dev, cp_path=None, precision='f16', use_sae=True).train()
model.load_state_dict(synth_state_dict(XLSR_300M, 'sae', 201, 1234)); model.to(dev)
training.freeze_unused(model)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=1e-6, weight_decay=1e-4, fused=True)
wav = (0.1 * torch.randn(5, 64600)).clamp(-1, 1).to(dev); y = torch.arange(5, device=dev) % 2
for _ in range(5): training.train_step(model, opt, wav, y)
torch.cuda.synchronize()
# CPU enqueue time per step with an idle GPU at the start of each step
tc = []; tg = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    training.train_step(model, opt, wav, y)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    tc.append(t1 - t0); tg.append(t2 - t0)
print('cpu enqueue ms', sum(tc) / len(tc) * 1e3, 'step with sync ms', sum(tg) / len(tg) * 1e3)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): training.train_step(model, opt, wav, y)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
