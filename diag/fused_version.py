import torch
p = torch.nn.Parameter(torch.randn(1000, device="cuda"))
opt = torch.optim.Adam([p], lr=1e-3, fused=True)
for i in range(3):
    p.grad = torch.randn_like(p)
    v0 = p._version; d0 = p.data_ptr(); s0 = float(p.sum())
    opt.step()
    print("fused: version", v0, "->", p._version, "ptr same", d0 == p.data_ptr(), "sum", s0, "->", float(p.sum()))
q = torch.nn.Parameter(torch.randn(1000, device="cuda"))
opt = torch.optim.Adam([q], lr=1e-3)
q.grad = torch.randn_like(q)
v0 = q._version
opt.step()
print("foreach: version", v0, "->", q._version)
