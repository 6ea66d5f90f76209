import sys, torch
sys.path.insert(0, ".")
from materialist_amd import ops
dev = torch.device("cuda:0"); torch.manual_seed(11)
M = 128 * 200
x = torch.randn(M, 256, device=dev); x[:, 0] *= 50.0
w = (torch.rand(256, 256, device=dev) * 2 - 1) / 16; b = torch.randn(256, device=dev) * 0.1
ws = ops.mlp_split_weights(w, 256, 256)
s_ref, c_ref, s_pk = (torch.empty(M, 256, device=dev) for _ in range(3))
ops.mlp_layer_fwd_bx(x, ws, b, s_ref, c_ref, 256, 256, 6)
ops.mlp_layer_fwd_bx(x, ws, b, s_pk, None, 256, 256, 6)
sp = s_pk.double()
c_reb = torch.sqrt((1 - sp * sp).clamp_min(0)) * torch.where((s_pk.view(torch.int32) & 1).bool(), -1.0, 1.0)
c32 = torch.sqrt((1 - s_pk * s_pk).clamp_min(0))
ct = c_ref.double()
err = (c32.double() * torch.sign(c_reb) - ct).abs()
for lo, hi in ((0, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1.1)):
    sel = (ct.abs() >= lo) & (ct.abs() < hi)
    print(f"|cos| in [{lo},{hi}): n={int(sel.sum())} max abs err {float(err[sel].max()):.2e} max err*|c| {float((err*ct.abs())[sel].max()):.2e} rms rel {float((err[sel]/ct.abs()[sel].clamp_min(1e-9)).pow(2).mean().sqrt()):.2e}")
print("overall rms err / rms c", float(err.pow(2).mean().sqrt() / ct.pow(2).mean().sqrt()))
