"""One member of a ragged fuzz call against the CPU oracle, gradient by gradient:  python tools/lab/ragged_oracle.py SEED CASE MEMBER"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ragged_repro.py")).read().split("# probes:")[0]
exec(src)
from oracle import sm_mll_oracle as orc
total = (nz[b, :n] if use_vec else torch.zeros(n, dtype=D)) + (float(ns[b]) if use_scalar else 0.0)
xx = x[b, :n] if d == 2 else x[b, :n, 0]
val, gr = orc.mll_value_grad_closed_form(xx, y[b, :n], float(mean[b, 0]), total, w[b], mu[b], v[b], order, 0.0)
K = orc.sm_kernel(xx.reshape(n, -1) if hasattr(orc, "sm_kernel") else xx, xx.reshape(n, -1), w[b], mu[b], v[b], order) if hasattr(orc, "sm_kernel") else None
s = _hip.mll_value_grad(*args()); torch.cuda.synchronize()
print(f"oracle value {float(val)!r}; HIP single {float(s['mll'])!r}; ragged {float(out['mll'][b])!r}")
for p in ("w", "mu", "v"):
    r = gr[p].reshape(-1).double(); a1 = s[f"g_{p}"].cpu().reshape(-1); a2 = out[f"g_{p}"][b].cpu().reshape(-1)
    print(p, "oracle", r.tolist(), "\n   single", a1.tolist(), "\n   ragged", a2.tolist(), f"\n   rel (max-norm): single {float((a1 - r).abs().max() / r.abs().max()):.2e} ragged {float((a2 - r).abs().max() / r.abs().max()):.2e}")
print("hyper-parameters: w", w[b].tolist(), "mu", mu[b].reshape(-1).tolist(), "v", v[b].reshape(-1).tolist(), " smallest gap in x:", float((xx[1:] - xx[:-1]).min()) if d == 1 else "2-D", " noise min", float(total.min()))
# autograd through the dense restatement as a third opinion, and the size of the terms the mu-gradient cancels from
val2, gr2 = orc.mll_value_grad_autograd(xx, y[b, :n], float(mean[b, 0]), total, w[b], mu[b], v[b], order, 0.0)
print("autograd: value", repr(float(val2)), " g_mu", gr2["mu"].reshape(-1).tolist(), " g_w", gr2["w"].reshape(-1).tolist())
import numpy as np
xs = xx.double().numpy() if d == 1 else None
if xs is not None:
    tau = xs[:, None] - xs[None, :]
    e = np.exp(-2 * np.pi ** 2 * float(v[b].reshape(-1)[0]) ** 2 * tau ** 2)
    dK = -2 * np.pi * tau * float(w[b][0]) * e * np.sin(2 * np.pi * float(mu[b].reshape(-1)[0]) * tau)
    A = float(w[b][0]) * e * np.cos(2 * np.pi * float(mu[b].reshape(-1)[0]) * tau) + np.diag(total.numpy())
    Ai = np.linalg.inv(A); al = Ai @ (y[b, :n].numpy() - float(mean[b, 0]))
    G = np.outer(al, al) - Ai
    print(f"sum |G_ij dK_ij| / 2n = {np.abs(G * dK).sum() / (2 * n):.3e}   (the gradient itself: {(G * dK).sum() / (2 * n):.12e});  cond(A) = {np.linalg.cond(A):.2e}")
