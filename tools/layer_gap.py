"""Why is the stand-alone forward layer (bench.py roofline.posmlp_gemm) slower than the same launch inside the traced iteration?  Operand statistics
(the chip holds a lower clock on wide-range random data) vs cache residency (in the loop the input was written by the launch before)."""
import sys, torch
sys.path.insert(0, '.')
from materialist_amd import ops
dev = torch.device('cuda')
M = 512 * 512
def b2b(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
w = torch.randn(256, 256, device=dev) / 16
b = torch.randn(256, device=dev)
wsp = ops.mlp_split_weights(w, 256, 256)
s = torch.empty(M, 256, device=dev)
for name, x in (("randn", torch.randn(M, 256, device=dev)), ("sin(randn) (what a layer reads in the loop)", torch.sin(torch.randn(M, 256, device=dev) * 3)), ("zeros", torch.zeros(M, 256, device=dev))):
    t = b2b(lambda: ops.mlp_layer_fwd_bx(x, wsp, b, s, None, 256, 256, 6))
    print(f"forward, input {name}: {t:.1f} us")
# producer -> consumer chain as in the loop: layer k writes what layer k+1 reads
x = torch.sin(torch.randn(M, 256, device=dev)); s2 = torch.empty_like(s)
t = b2b(lambda: (ops.mlp_layer_fwd_bx(x, wsp, b, s, None, 256, 256, 6), ops.mlp_layer_fwd_bx(s, wsp, b, s2, None, 256, 256, 6)))
print(f"two chained forwards: {t / 2:.1f} us each")
