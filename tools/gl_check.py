import sys, os
sys.path.insert(0, "/root/repo")
import torch
from materialist_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(1)
for M in (128 * 5, 128 * 300, 512 * 512):
    x = torch.randn(M, 256, device=dev); x[:, 0] *= 30
    w = torch.randn(256, 256, device=dev) / 16
    b = torch.randn(256, device=dev)
    g = torch.randn(M, 256, device=dev)
    ws = ops.mlp_split_weights(w, 256, 256)
    outs = []
    for on in (0, 1, 2):
        lib.matpbr_mlp_set_lds_dma(on)
        s = torch.zeros(M, 256, device=dev); c = torch.zeros(M, 256, device=dev)
        sp = torch.zeros(M, 256, device=dev)
        gp = torch.zeros(M, 256, device=dev); db = torch.zeros(256, device=dev)
        gq = torch.zeros(M, 256, device=dev); dq = torch.zeros(256, device=dev)
        for rep in range(3):
            ops.mlp_layer_fwd_bx(x, ws, b, s, c, 256, 256, 6)
            ops.mlp_layer_fwd_bx(x, ws, b, sp, None, 256, 256, 6)
            ops.mlp_layer_bwd_input_bx(g, ws, c, gp, 256, 256, db, 6)
            ops.mlp_layer_bwd_input_bx(g, ws, sp, gq, 256, 256, dq, 6, packed=True)
        torch.cuda.synchronize()
        outs.append((s, c, sp, gp, db, gq, dq))
    for k in (1, 2):
        for a, bb, name in zip(outs[0], outs[k], "s c sp gp db gq dq".split()):
            same = torch.equal(a, bb)
            print(M, "mode", k, name, "same bits" if same else "DIFF %g (scale %g)" % ((a - bb).abs().max().item(), a.abs().max().item()), flush=True)

# the first-layer form (W0): modes 2 (512-thread kernel) and 3 (two workgroups per CU)
M, d0 = 128 * 300, 15
g = torch.randn(M, 256, device=dev)
w1 = torch.randn(256, 256, device=dev) / 16
sp = torch.sin(torch.randn(M, 256, device=dev) * 3)
x0 = torch.zeros(M, 16, device=dev); x0[:, :d0] = torch.randn(M, d0, device=dev)
wts = ops.mlp_split_weights(w1, 256, 256, transposed=True)
res = []
for mode in (2, 3):
    lib.matpbr_mlp_set_lds_dma(mode)
    gw, gb = torch.zeros(256, 16, device=dev), torch.empty(256, device=dev)
    for _ in range(2):
        ops.mlp_first_layer_bwd_bx(g, wts, sp, x0, gw, d0, 256, 256, gb, 6, packed=True)
    torch.cuda.synchronize()
    res.append((gw, gb))
lib.matpbr_mlp_set_lds_dma(2)
for a, bb, name in zip(res[0], res[1], ("dW0", "db0")):
    print("W0", name, "max diff %g (scale %g)" % ((a - bb).abs().max().item(), a.abs().max().item()))
