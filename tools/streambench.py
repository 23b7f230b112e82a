"""Two sub-batches on two streams (each with a workspace of its own) against one launch set: does the second stream fill the
CUs the diagonal-block launches and the launch tails of the first leave idle?   python tools/streambench.py [n] [B]"""
import sys, time, torch
sys.path.insert(0, '.')
from pgmuvi_amd import _hip
from pgmuvi_amd.batch import make_shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
sh = make_shard(B, 0, 1, n, "cfg3", dev)
def run(ws, sl):
    return _hip.mll_value_grad(sh["x"][sl], sh["y"][sl], sh["mean"][sl], sh["noise"][sl], None, sh["w"][sl], sh["mu"][sl], sh["v"][sl], 0, 0.0, True, workspace=ws)
def bench(f, reps=10):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
ws = _hip.Workspace(dev, n, 4, 1, B)
one = bench(lambda: run(ws, slice(0, B)))
ref = run(ws, slice(0, B))["mll"].clone()
for parts in (2, 4):
    h = B // parts
    wss = [_hip.Workspace(dev, n, 4, 1, h) for _ in range(parts)]
    sts = [torch.cuda.Stream() for _ in range(parts)]
    outs = [None] * parts
    def multi():
        cur = torch.cuda.current_stream()
        for i in range(parts):
            sts[i].wait_stream(cur)
            with torch.cuda.stream(sts[i]):
                outs[i] = run(wss[i], slice(i * h, (i + 1) * h))
        for s in sts: cur.wait_stream(s)
    t = bench(multi)
    got = torch.cat([o["mll"] for o in outs])
    print(f"n={n} B={B}: one launch set {one:.3f} ms; {parts} x {h} on {parts} streams {t:.3f} ms; max|dmll| {float((got-ref).abs().max()):.1e}")
    # sequential halves on one stream (what the overlap must beat)
    def seq():
        for i in range(parts): outs[i] = run(wss[i], slice(i * h, (i + 1) * h))
    print(f"   {parts} x {h} one after the other {bench(seq):.3f} ms")
