"""Sustained fp64 MFMA rate: the issue-rate probe of the library for 1 ms ... 1 s, with trivial and with busy-mantissa operands
(the datasheet peak assumes the boost clock; what the part holds under a long fp64 MFMA load is what a GEMM-bound kernel can reach)."""
import ctypes, sys
sys.path.insert(0, '/root/repo')
from pgmuvi_amd import _hip
lib = _hip.load()
f = lib.pgm_debug_probe_mfma_long
f.restype = ctypes.c_int; f.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
v = ctypes.c_double()
for iters in (4000, 40000, 400000, 2000000):
    for sign, name in ((1, "trivial operands"), (-1, "busy mantissas")):
        f(0, sign * iters, ctypes.byref(v))
        print(f"{iters:8d} iterations ({iters * 8 * 64 / 2.4e9 * 1e3:7.1f} ms at 2.4 GHz), {name}: {v.value:6.2f} TFLOP/s = {v.value / 78.6 * 100:5.1f} % of 78.6")
