"""Host cost of one evaluation call: the Python wrapper (_hip.mll_value_grad) and the C entry point alone, launched back to back
without synchronising (the device queue absorbs them).   python tools/callcost.py [n]"""
import sys, time, ctypes, torch
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import _hip, synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
t, y, e = syn.cfg2(n_obs=n)
x, yy, nz = t.double().to(dev).reshape(n, 1), y.double().to(dev), (e.double() ** 2).to(dev)
h = syn.cfg_hypers(2, y.double())
w, mu, v = h["w"].to(dev), h["mu"].reshape(4, 1).to(dev), h["v"].reshape(4, 1).to(dev)
mean = torch.zeros(n, dtype=torch.float64, device=dev)
for _ in range(5): out = _hip.mll_value_grad(x, yy, mean, nz, None, w, mu, v)
torch.cuda.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps): out = _hip.mll_value_grad(x, yy, mean, nz, None, w, mu, v)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"n={n}: _hip.mll_value_grad {1e6 * (t1 - t0) / reps:.1f} us of host time per call")
lib = _hip.load(); ws = out["workspace"]
buf = torch.empty(1 + 4 + 4 + 4 + 2 * n, dtype=torch.float64, device=dev); info = torch.empty(1, dtype=torch.int32, device=dev)
P = lambda t_: ctypes.c_void_p(t_.data_ptr())
args = (ws.handle, 1, P(x), P(yy), P(mean), P(nz), ctypes.c_void_p(None), n, 1, P(w), P(mu), P(v), 4, 0, 0.0, 1,
        ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr() + 8), ctypes.c_void_p(buf.data_ptr() + 40), ctypes.c_void_p(buf.data_ptr() + 72),
        ctypes.c_void_p(buf.data_ptr() + 104), ctypes.c_void_p(buf.data_ptr() + 104 + 8 * n), P(info), _hip.current_stream_ptr(dev))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps): lib.pgm_mll_value_grad_batched_f64(*args)
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"n={n}: pgm_mll_value_grad_batched_f64 alone {1e6 * (t1 - t0) / reps:.1f} us of host time per call")
