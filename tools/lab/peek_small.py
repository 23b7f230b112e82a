import os, sys, ctypes, struct, numpy as np, torch
sys.path.insert(0, '/root/repo')
exec(open('/root/repo/tools/lab/ragged_repro.py').read().split("# probes:")[0].replace("print(", "(lambda *a, **k: None)("))
hexd = lambda v: struct.pack(">d", float(v)).hex()
os.environ["PGM_SMALL"] = "2"
_hip.release_workspaces()
s = _hip.mll_value_grad(*args()); torch.cuda.synchronize()
lib = ctypes.CDLL(_hip.lib_path())
lib.pgm_debug_peek.restype = ctypes.c_int
lib.pgm_debug_peek.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
pk = np.zeros(8); z = np.zeros(128)
print(lib.pgm_debug_peek(s["workspace"].handle, 3, 8, pk.ctypes.data_as(ctypes.c_void_p), None), lib.pgm_debug_peek(s["workspace"].handle, 0, 128, z.ctypes.data_as(ctypes.c_void_p), None))
print("s_lo s_hi tot ld z0 val n:", [f"{v!r} {hexd(v)}" for v in pk[:7]])
sq = z * z
def wave(vv):
    v = vv.copy()
    for off in (32, 16, 8, 4, 2, 1): v = v + v[np.arange(64) ^ off]
    return v[0]
lo = sq[:64].copy(); lo[0] = lo[0] + pk[3]
print("host s_lo", repr(wave(lo)), hexd(wave(lo)), " s_hi", repr(wave(sq[64:])), hexd(wave(sq[64:])), " z0", repr(z[0]), hexd(z[0]))
