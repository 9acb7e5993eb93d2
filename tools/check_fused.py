#!/usr/bin/env python3
"""Quick check + timing of rnnt_joint_loss_forward against library GEMM + RNNTLoss on one shape.
    python tools/check_fused.py B,T,U,V,H [dtype]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rnnt_amd

shape = sys.argv[1] if len(sys.argv) > 1 else "4,50,10,4232,512"
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[2] if len(sys.argv) > 2 else "bf16"]
B, T, U, V, H = (int(x) for x in shape.split(","))
dev = torch.device("cuda:0")
torch.manual_seed(0)
hidden = torch.tanh(torch.randn(B, T, U + 1, H, device=dev)).to(dt)
W = (torch.randn(V, H, device=dev) * 0.05).to(dt)
b = (torch.randn(V, device=dev) * 0.5).to(dt)
labels = torch.randint(1, V, (B, U), device=dev, dtype=torch.int32)
al = torch.full((B,), T, device=dev, dtype=torch.int32)
ll = torch.full((B,), U, device=dev, dtype=torch.int32)
if B > 1:
    al[1] = max(1, T // 2); ll[1] = U // 2

hid = hidden.clone().requires_grad_(True)
Wp = W.clone().requires_grad_(True)
bp = b.clone().requires_grad_(True)
loss = rnnt_amd.fused_joint_loss(hid, Wp, bp, labels, al, ll, min_nodes=0)
loss.backward()
torch.cuda.synchronize()

# reference: fp32-accumulated GEMM rounded to dt, then the unfused loss
hid2 = hidden.clone().requires_grad_(True); W2 = W.clone().requires_grad_(True); b2 = b.clone().requires_grad_(True)
logits = torch.nn.functional.linear(hid2, W2, b2)
loss2 = rnnt_amd.RNNTLoss()(logits, labels, al, ll)
loss2.backward()
torch.cuda.synchronize()
print("loss fused", loss.item(), "unfused", loss2.item(), "rel", abs(loss.item() - loss2.item()) / abs(loss2.item()))
for name, a, c in (("dH", hid.grad, hid2.grad), ("dW", Wp.grad, W2.grad), ("db", bp.grad, b2.grad)):
    d = (a.float() - c.float()).abs().max().item()
    print(name, "max abs diff", d, "max abs", c.float().abs().max().item())

# logits of the fused kernel itself
from rnnt_amd.fused import _FusedJointLossFunction
ctx_logits = None
class Grab(torch.autograd.Function):
    pass
lib_logits = torch.empty(B, T, U + 1, V, dtype=dt, device=dev)
# run the C entry point directly to look at logits_out
from rnnt_amd import _lib
from rnnt_amd.loss import _DTYPES
lib = _lib.load()
ws = torch.empty(_lib.workspace_bytes(T, U + 1, B), dtype=torch.uint8, device=dev)
costs = torch.empty(B, device=dev); lo = torch.empty(1, device=dev)
opts = _lib.RnntOptions(torch.cuda.current_stream().cuda_stream, 0, T, U + 1, _DTYPES[dt], None)
b32 = b.float()
def run():
    _lib.check(lib.rnnt_joint_loss_forward(hidden.data_ptr(), W.data_ptr(), b32.data_ptr(), lib_logits.data_ptr(),
               labels.data_ptr(), al.data_ptr(), ll.data_ptr(), H, V, B, costs.data_ptr(), lo.data_ptr(), 1, ws.data_ptr(), opts), "fused")
run(); torch.cuda.synchronize()
ref = (hidden.float().view(-1, H) @ W.float().t() + b.float()).to(dt).view(B, T, U + 1, V)
diff = (lib_logits.float() - ref.float()).abs()
print("logits: exact match frac", (lib_logits == ref).float().mean().item(), "max abs diff", diff.max().item())
# loss of the unfused forward on the kernel's own logits must equal the fused costs
c2 = rnnt_amd.rnnt_loss(lib_logits, labels, al, ll, reduction="none")
print("costs fused", costs[:4].tolist(), "unfused-on-same-logits", c2[:4].tolist(),
      "max rel", ((costs - c2).abs() / c2.abs()).max().item())

def t_ms(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tf = t_ms(run)
out = torch.empty(B * T * (U + 1), V, dtype=dt, device=dev)
tg = t_ms(lambda: torch.addmm(b, hidden.view(-1, H), W.t(), out=out))
crit = rnnt_amd.RNNTLoss(check_lengths=False)
o4 = out.view(B, T, U + 1, V)
with torch.no_grad():
    tl = t_ms(lambda: crit(o4, labels, al, ll))
flop = 2.0 * B * T * (U + 1) * V * H
print(f"fused fwd {tf:.3f} ms ({flop / tf / 1e9:.0f} TFLOP/s) | library GEMM {tg:.3f} ms + loss fwd (K1+K2) {tl:.3f} ms = {tg + tl:.3f} ms")
