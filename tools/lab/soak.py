"""tools/lab/soak.py -- 600 evaluations over ten alternating problem sizes on one device (workspace reuse, graph cache, every
schedule variant in turn): every result must equal the first one of its size bit for bit.   python tools/lab/soak.py  (GPU box)"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pgmuvi_amd import _hip, synthetic as syn
dev = torch.device("cuda:0"); D = torch.float64
sizes = [89, 4096, 256, 2048, 1000, 5120, 130, 3000, 4352, 640]
data = {}
for n in sizes:
    t, y, e = syn.cfg2(n_obs=n); h = syn.cfg_hypers(2, y.double())
    data[n] = (t.double().reshape(-1, 1).to(dev), y.double().to(dev), torch.full((n,), float(h["mean"]), dtype=D, device=dev), (e.double() ** 2).to(dev),
               h["w"].to(dev), h["mu"].reshape(4, 1).to(dev), h["v"].reshape(4, 1).to(dev))
ref = {}
t0 = time.time(); bad = 0
for rep in range(60):
    for n in sizes:
        x, y, m, nz, w, mu, v = data[n]
        o = _hip.mll_value_grad(x, y, m, nz, None, w, mu, v, 0, 0.0, True)
        key = (float(o["mll"]), float(o["g_w"].sum()), float(o["g_noise"].sum()), int(o["info"]))
        if n not in ref: ref[n] = key
        elif ref[n] != key: bad += 1; print("MISMATCH", rep, n, ref[n], key)
torch.cuda.synchronize()
print(f"{60 * len(sizes)} evaluations over {len(sizes)} alternating sizes in {time.time() - t0:.1f} s, {bad} mismatches against the first result of each size; memory {torch.cuda.memory_allocated() / 2**20:.0f} MiB")
