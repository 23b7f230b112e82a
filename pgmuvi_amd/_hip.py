"""ctypes binding of ``libpgmuvi_hip.so`` (the C ABI in ``include/pgmuvi_hip.h``).

There is NO CPU fallback: if the shared library is missing, or a tensor is not on
an MI355X device, the calls raise.  PyTorch is used only for device memory and
streams; every pointer handed to the library is a plain device address.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p, POINTER, byref
from typing import Dict, Optional, Tuple

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpgmuvi_hip.so")
_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); must list every symbol of include/pgmuvi_hip.h
SYMBOLS = {
    "pgm_version": (c_char_p, []),
    "pgm_max_qd": (c_int, []),
    "pgm_max_n": (c_int64, []),
    "pgm_workspace_create": (c_int, [POINTER(c_void_p), c_int, c_int64, c_int, c_int, c_int]),
    "pgm_workspace_destroy": (c_int, [c_void_p]),
    "pgm_workspace_bytes": (c_size_t, [c_void_p]),
    "pgm_factorisation_status": (c_int, [c_void_p, c_void_p, c_int]),
    "pgm_last_evaluation": (c_int64, [c_void_p]),
    "pgm_factorisation_status_of": (c_int, [c_void_p, c_int64, c_void_p, c_int]),
    "pgm_sm_kernel_f64": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_double, c_int, c_void_p, c_int64, c_void_p]),
    "pgm_mll_value_grad_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int64, c_int,
                                       c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_int,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pgm_mll_value_grad_batched_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double,
                                               c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "pgm_mll_value_grad_ragged_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double,
                                              c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p]),
    "pgm_ragged_plan": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pgm_ragged_plan_ws": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pgm_predict_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "pgm_profile_enable": (c_int, [c_void_p, c_int]),
    "pgm_profile_phases": (c_int, []),
    "pgm_profile_phase_name": (c_char_p, [c_int]),
    "pgm_profile_read": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int64)]),
    "pgm_profile_early_inverse_products": (c_int64, [c_void_p]),
    "pgm_probe_mfma_f64": (c_int, [c_int, POINTER(c_double)]),
    "pgm_mll_dense_f64": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_double, c_int, c_void_p, c_void_p, c_int64,
                                  c_void_p, c_void_p, c_void_p]),
    "pgm_predict_dense_f64": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "pgm_fit_create": (c_int, [POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_int, c_int, c_double, c_double, c_double, c_double, c_double, c_int]),
    "pgm_fit_set_priors": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pgm_fit_run": (c_int, [c_void_p, c_int, c_void_p]),
    "pgm_fit_read": (c_int, [c_void_p, c_void_p, POINTER(c_int), c_void_p, c_void_p, c_void_p, POINTER(c_int)]),
    "pgm_fit_destroy": (c_int, [c_void_p]),
    "pgm_pot_create": (c_int, [POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pgm_pot_eval": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pgm_pot_destroy": (c_int, [c_void_p]),
    "pgm_lomb_scargle_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                     c_void_p]),
    "pgm_mll_kernel_value_grad_f64": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                              c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pgm_lomb_scargle_fast_scratch_doubles": (c_int64, [c_int64, c_int64, c_int]),
    "pgm_lomb_scargle_fast_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_double, c_double, c_int64, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p]),
}


class HipLibraryMissing(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


def load():
    """dlopen the library and declare every prototype (no GPU call is made)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(_LIB_PATH):
                raise HipLibraryMissing(
                    f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  pgmuvi_amd has no CPU fallback.")
            lib = ctypes.CDLL(_LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def version() -> str:
    return load().pgm_version().decode()


def max_qd() -> int:
    return int(load().pgm_max_qd())


def max_n_limit() -> int:
    """Most points one light curve may have (include/pgmuvi_hip.h: the size contract)."""
    return int(load().pgm_max_n())


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else c_void_p(t.data_ptr())


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed with status {rc} (negative = index of the bad argument, see include/pgmuvi_hip.h)")


def _dev64(t: torch.Tensor, device) -> torch.Tensor:
    # (the common case -- already float64, on the device, contiguous -- must cost nothing: this runs a dozen times per
    #  training iteration while the GPU waits for the launch)
    if t.dtype is torch.float64 and t.device == device and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(device=device, dtype=torch.float64).contiguous()


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: tensor is on '{t.device}'.  pgmuvi_amd evaluates the GP marginal likelihood only with its HIP "
            "kernels on an MI355X (move the model and data with .cuda()); there is no CPU fallback.")


class Workspace:
    """Owns a ``pgm_ws`` (device buffers sized for max_n / max_q / max_d / max_batch)."""

    def __init__(self, device: torch.device, max_n: int, max_q: int, max_d: int, max_batch: int = 1):
        lib = load()
        self.device = torch.device(device)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.key = (idx, max_n, max_q, max_d, max_batch)
        handle = c_void_p()
        if max_n > max_n_limit():
            raise RuntimeError(f"a light curve of {max_n} points: the HIP library takes at most {max_n_limit()} (128 block rows of 128; the "
                               "largest size held to an oracle fixture, include/pgmuvi_hip.h) -- pgm_workspace_create would return -3")
        _check(lib.pgm_workspace_create(byref(handle), idx, max_n, max_q, max_d, max_batch), "pgm_workspace_create")
        self.handle = handle
        self.max_n, self.max_q, self.max_d, self.max_batch = max_n, max_q, max_d, max_batch
        self.nominal_bytes = int(lib.pgm_workspace_bytes(handle))       # at creation; R and prediction scratch grow on first use

    @property
    def bytes(self) -> int:
        """Device memory held now (the scratch of the early inverse pass and of prediction grows on first use)."""
        return int(load().pgm_workspace_bytes(self.handle)) if getattr(self, "handle", None) else 0

    def close(self):
        if getattr(self, "handle", None):
            load().pgm_workspace_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_evaluation(self) -> int:
        """Number of the evaluation enqueued last on this workspace (the stamp ``factorisation_failed`` takes)."""
        return int(load().pgm_last_evaluation(self.handle))

    def factorisation_failed(self, batch: int = 1, evaluation: int = -1):
        """Did any of the ``batch`` problems of evaluation number ``evaluation`` (-1: the last one) on this workspace fail to
        factor?  Waits for the event the library records after the factorisation sweep only -- the inverse/gradient pass keeps
        running behind it -- so the caller's host work overlaps the rest of the evaluation (pgm_factorisation_status_of).
        None when there is nothing to report for THAT evaluation (it ran inside a stream capture, or another evaluation has
        been enqueued on the workspace since): the caller then reads its own ``info`` from the device."""
        rc = load().pgm_factorisation_status_of(self.handle, int(evaluation), None, int(batch))
        if rc < 0:
            return None
        return rc > 0

    # -- profiling ---------------------------------------------------------
    def profile(self, on: bool):
        _check(load().pgm_profile_enable(self.handle, 1 if on else 0), "pgm_profile_enable")

    def profile_read(self) -> Dict[str, Tuple[float, int]]:
        lib = load()
        n = lib.pgm_profile_phases()
        ms = (c_double * n)()
        cnt = (c_int64 * n)()
        _check(lib.pgm_profile_read(self.handle, ms, cnt), "pgm_profile_read")
        return {lib.pgm_profile_phase_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}

    def early_inverse_products(self) -> int:
        """128^3 products of the inverse pass the last single-curve evaluation ran inside the sweep's launches."""
        return int(load().pgm_profile_early_inverse_products(self.handle))


# Cache of workspaces, least recently used first.  A request is served by ANY cached workspace that covers it (the C side
# takes n <= max_n, q <= max_q, d <= max_d, batch <= max_batch), the smallest such; a miss allocates an exact fit and
# then drops least-recently-used entries from the cache until it is back under the byte budget.  Dropping never frees: the
# device buffers go when the last holder lets go of the Workspace object (a NativeFit, the dictionary an evaluation
# returned, a captured graph's owner ...), so raw addresses baked into graphs and handles stay valid for as long as
# anything can still use them.
_workspaces: "Dict[tuple, Workspace]" = {}
_ws_lock = threading.RLock()                       # (not ``_lock``: creating a Workspace calls load(), which takes that one)
WORKSPACE_BUDGET_BYTES = int(float(os.environ.get("PGMUVI_WORKSPACE_BUDGET_GB", "128")) * (1 << 30))


def _covers(ws: "Workspace", idx: int, np_: int, q: int, d: int, batch: int) -> bool:
    # (max_q >= q as well as the product: the per-work-item partial sums are strided by q + 2 q d + 1 slots, which a
    #  (q=8, d=1) request would overrun in a workspace made for (q=4, d=2) although q d fits)
    return (ws.handle is not None and ws.key[0] == idx and ws.max_n >= np_ and ws.max_d >= d and ws.max_q >= q
            and ws.max_q * ws.max_d >= q * d and ws.max_batch >= batch)


def get_workspace(device, n: int, q: int, d: int, batch: int = 1) -> Workspace:
    """Cached workspace large enough for (n, q, d, batch) on `device` (n rounds up to the 128-block, so that a fit loop
    with fixed shapes allocates once; alternating shapes -- SM and dense back-end, the tail chunk of a batch, a second
    model -- reuse what is there instead of reallocating)."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    np_ = (n + 127) // 128 * 128
    with _ws_lock:
        best = None
        for key, ws in _workspaces.items():
            if _covers(ws, idx, np_, q, d, batch) and (best is None or ws.nominal_bytes < _workspaces[best].nominal_bytes):
                best = key
        if best is not None:
            ws = _workspaces.pop(best)
            _workspaces[best] = ws                      # most recently used last
            return ws
        with torch.cuda.device(idx):
            ws = Workspace(torch.device("cuda", idx), np_, q, d, batch)
        _workspaces[ws.key] = ws
        total = sum(w.nominal_bytes for w in _workspaces.values())
        for key in list(_workspaces):
            if total <= WORKSPACE_BUDGET_BYTES or key == ws.key:
                continue
            total -= _workspaces.pop(key).nominal_bytes  # (freed when its last holder drops it)
        return ws


def cached_workspaces(device=None):
    """The Workspace objects in the cache (of one device): whoever replays a captured graph that has workspace addresses
    baked in holds this list for as long as the graph lives."""
    with _ws_lock:
        idx = None if device is None else (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device())
        return [w for w in _workspaces.values() if idx is None or w.key[0] == idx]


_release_hooks: list = []


def on_release(hook):
    """Registers a callable ``release_workspaces()`` runs: modules that keep Workspace objects of their own outside the cache
    (``batch._two_streams``' pair) hand them back there."""
    _release_hooks.append(hook)


def release_workspaces():
    """Empties the cache -- and the workspaces other modules registered with :func:`on_release` -- (buffers are freed as soon as
    nothing else holds the Workspace objects)."""
    with _ws_lock:
        _workspaces.clear()
    for hook in list(_release_hooks):
        hook()


def workspace_bytes_estimate(n: int, q: int, d: int, batch: int) -> int:
    """Device bytes a workspace for (n, q, d, batch) holds, to first order: the N x N matrix per light curve plus the
    per-point factors and vectors (``pgm_workspace_create``; the exact figure is ``Workspace.nominal_bytes`` once it exists)."""
    np_ = (n + 127) // 128 * 128
    nb = np_ // 128
    per = 8 * (np_ * np_ + nb * 2 * 128 * 128 + (3 * q * d + d + 40) * np_)
    return int(batch) * per


def trim_cache(reserve_bytes: int = 0):
    """Drops least-recently-used cache entries until the cache plus ``reserve_bytes`` (workspaces about to be made outside the
    cache) is within WORKSPACE_BUDGET_BYTES."""
    with _ws_lock:
        total = sum(w.nominal_bytes for w in _workspaces.values()) + int(reserve_bytes)
        for key in list(_workspaces):
            if total <= WORKSPACE_BUDGET_BYTES:
                break
            total -= _workspaces.pop(key).nominal_bytes


def current_stream_ptr(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def sm_kernel_dense(x1, x2, w, mu, v, noise=None, noise_scalar: float = 0.0, dim_order: int = 0) -> torch.Tensor:
    """Dense K(x1, x2) (+ diag noise when x1 is x2) through pgm_sm_kernel_f64."""
    require_gpu(x1, "sm_kernel_dense")
    dev = x1.device
    same = x2 is x1
    x1d = _dev64(x1.reshape(x1.shape[0], -1), dev)
    x2d = x1d if same else _dev64(x2.reshape(x2.shape[0], -1), dev)
    n1, d = x1d.shape
    n2 = x2d.shape[0]
    q = w.numel()
    wd, mud, vd = _dev64(w.reshape(q), dev), _dev64(mu.reshape(q, d), dev), _dev64(v.reshape(q, d), dev)
    nz = None if noise is None else _dev64(noise.reshape(n1), dev)
    K = torch.empty((n1, n2), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_sm_kernel_f64(_ptr(x1d), n1, _ptr(x2d), n2, d, _ptr(wd), _ptr(mud), _ptr(vd), q, _ptr(nz),
                                      float(noise_scalar), int(dim_order), _ptr(K), n2, current_stream_ptr(dev))
    _check(rc, "pgm_sm_kernel_f64")
    return K


class _Outputs(dict):
    """The result dictionary of an evaluation whose fp64 outputs share one buffer: the per-output views are made when they
    are first asked for (a training iteration reads ``mll`` and the whole buffer only; each view costs host time while the
    GPU waits for the next launch)."""

    def __init__(self, buf, offs, layout):
        super().__init__()
        self._lazy = (buf, offs, layout)

    def __missing__(self, key):
        buf, offs, layout = self._lazy
        if key not in layout:
            raise KeyError(key)
        i, shape = layout[key]
        v = buf[offs[i]:offs[i + 1]].view(shape)
        self[key] = v
        return v

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy[2]

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        for k in self._lazy[2]:
            self[k]
        return dict.items(self)

    def keys(self):
        for k in self._lazy[2]:
            self[k]
        return dict.keys(self)

    def values(self):
        self.keys()
        return dict.values(self)

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())


def mll_value_grad(x, y, mean, noise, noise_scalar, w, mu, v, dim_order=0, jitter=0.0, need_grad=True,
                   workspace: Optional[Workspace] = None):
    """One (or a batch of) MLL evaluation(s) through the C ABI.

    Unbatched shapes: x (n,d) y (n) mean (n) noise (n)|None noise_scalar float|0-dim tensor, w (q) mu (q,d) v (q,d).
    Batched: a leading batch dimension on everything (noise_scalar: (B,) tensor or None).
    Returns dict(mll, g_w, g_mu, g_v, g_noise, g_mean, info) of device tensors (fp64); no host sync.
    """
    require_gpu(x, "mll_value_grad")
    dev = x.device
    batched = y.dim() == 2
    B = y.shape[0] if batched else 1
    n = y.shape[-1]
    # (the C side takes pointers: no reshapes, only float64 / device / contiguity -- and the element counts, checked here,
    #  because a wrong size would otherwise be read out of bounds on the device)
    xd = _dev64(x, dev)
    d = xd.numel() // (B * n)
    q = w.shape[-1] if w.dim() > 0 else 1
    yd = _dev64(y, dev)
    md = _dev64(mean if mean.shape == y.shape else mean.expand(y.shape), dev)
    nz = None if noise is None else _dev64(noise if noise.shape == y.shape else noise.expand(y.shape), dev)
    wd, mud, vd = _dev64(w, dev), _dev64(mu, dev), _dev64(v, dev)
    if (xd.numel() != B * n * d or yd.numel() != B * n or wd.numel() != B * q or mud.numel() != B * q * d or vd.numel() != B * q * d):
        raise ValueError(f"mll_value_grad: inconsistent shapes x {tuple(x.shape)} y {tuple(y.shape)} w {tuple(w.shape)} "
                         f"mu {tuple(mu.shape)} v {tuple(v.shape)}")
    ws = workspace or get_workspace(dev, n, q, d, B)
    # every fp64 output in ONE allocation (this runs once per training iteration between two device launches: each
    # torch.empty / view costs microseconds of host time the GPU spends idle); `info` is written by every call
    qd = q * d
    sizes = (B, B * q, B * qd, B * qd, B * n, B * n) if need_grad else (B,)
    buf = torch.empty(sum(sizes), dtype=torch.float64, device=dev)
    info = torch.empty(B, dtype=torch.int32, device=dev)
    base = buf.data_ptr()
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + sz)
    gp = [base + 8 * offs[i] for i in range(6)] if need_grad else [base, None, None, None, None, None]
    lib = load()
    # always the batched entry point (B = 1 when unbatched): the scalar noise then
    # travels as a device pointer and no host synchronisation is needed
    ns = None
    if noise_scalar is not None:
        ns = _dev64(torch.as_tensor(noise_scalar, device=dev).expand(B).reshape(B), dev)
    rc = lib.pgm_mll_value_grad_batched_f64(
        ws.handle, B, xd.data_ptr(), yd.data_ptr(), md.data_ptr(), None if nz is None else nz.data_ptr(),
        None if ns is None else ns.data_ptr(), n, d, wd.data_ptr(), mud.data_ptr(), vd.data_ptr(), q,
        int(dim_order), float(jitter), 1 if need_grad else 0,
        gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], info.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    _check(rc, "pgm_mll_value_grad_batched_f64")
    lead = (B,) if batched else ()
    out = _Outputs(buf, offs, {"mll": (0, lead), "g_w": (1, lead + (q,)), "g_mu": (2, lead + (q, d)), "g_v": (3, lead + (q, d)),
                               "g_noise": (4, lead + (n,)), "g_mean": (5, lead + (n,))} if need_grad else {"mll": (0, lead)})
    out["info"] = info.view(lead)
    out["_keep"] = (xd, yd, md, nz, ns, wd, mud, vd)
    out["_buf"], out["_offs"] = buf, offs                    # (one contiguous buffer: a caller can scale every gradient with one multiply)
    out["workspace"] = ws
    out["evaluation"] = ws.last_evaluation()
    return out


def ragged_plan(lengths, max_batch: Optional[int] = None, workspace: Optional["Workspace"] = None):
    """The launch sets ``pgm_mll_value_grad_ragged_f64`` forms for these light-curve lengths (host only, no GPU):
    (set index of every light curve, block rows of every set).  With ``workspace``: the sets THAT workspace runs (its slot
    count, its schedule switches); otherwise those of a workspace with ``max_batch`` slots made now."""
    import numpy as np
    n = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
    set_of = np.zeros(len(n), dtype=np.int32)
    nb_of = np.zeros(len(n), dtype=np.int32)
    if workspace is not None:
        k = load().pgm_ragged_plan_ws(workspace.handle, n.ctypes.data_as(c_void_p), len(n), set_of.ctypes.data_as(c_void_p),
                                      nb_of.ctypes.data_as(c_void_p))
    else:
        k = load().pgm_ragged_plan(n.ctypes.data_as(c_void_p), len(n), int(max_batch), set_of.ctypes.data_as(c_void_p),
                                   nb_of.ctypes.data_as(c_void_p))
    if k < 0:
        raise RuntimeError(f"pgm_ragged_plan failed with status {k}")
    return set_of.tolist(), nb_of[:k].tolist()


def mll_value_grad_ragged(x, y, mean, noise, noise_scalar, lengths, w, mu, v, dim_order=0, jitter=0.0, need_grad=True,
                          workspace: Optional[Workspace] = None, max_batch: Optional[int] = None):
    """MLL (+ gradients) of B light curves of different lengths through pgm_mll_value_grad_ragged_f64.

    Padded layout: x (B,S,d) y (B,S) mean (B,S) noise (B,S)|None, ``lengths`` the B point counts (host integers, each <= S);
    noise_scalar (B,)|None; w (B,q) mu (B,q,d) v (B,q,d).  Entries beyond a light curve's length are ignored; g_noise /
    g_mean come back (B,S) with zeros there.  ``max_batch``: light curves per launch set (workspace memory)."""
    import numpy as np
    require_gpu(x, "mll_value_grad_ragged")
    dev = x.device
    B, S = y.shape
    n = np.ascontiguousarray(np.asarray(lengths, dtype=np.int64))
    if n.shape != (B,) or n.min() < 1 or n.max() > S:
        raise ValueError(f"mll_value_grad_ragged: {B} lengths in [1, {S}] expected")
    xd = _dev64(x, dev)
    d = xd.numel() // (B * S)
    q = w.shape[-1]
    yd = _dev64(y, dev)
    md = _dev64(mean if mean.shape == y.shape else mean.expand(y.shape), dev)
    nz = None if noise is None else _dev64(noise if noise.shape == y.shape else noise.expand(y.shape), dev)
    ns = None if noise_scalar is None else _dev64(torch.as_tensor(noise_scalar, device=dev).expand(B).reshape(B), dev)
    wd, mud, vd = _dev64(w, dev), _dev64(mu, dev), _dev64(v, dev)
    if xd.numel() != B * S * d or wd.numel() != B * q or mud.numel() != B * q * d or vd.numel() != B * q * d:
        raise ValueError(f"mll_value_grad_ragged: inconsistent shapes x {tuple(x.shape)} y {tuple(y.shape)} w {tuple(w.shape)}")
    ws = workspace or get_workspace(dev, int(n.max()), q, d, min(B, max_batch or B))
    qd = q * d
    sizes = (B, B * q, B * qd, B * qd, B * S, B * S) if need_grad else (B,)
    buf = torch.zeros(sum(sizes), dtype=torch.float64, device=dev)
    info = torch.zeros(B, dtype=torch.int32, device=dev)
    offs = [0]
    for sz in sizes:
        offs.append(offs[-1] + sz)
    base = buf.data_ptr()
    gp = [base + 8 * offs[i] for i in range(6)] if need_grad else [base, None, None, None, None, None]
    with torch.cuda.device(dev):
        rc = load().pgm_mll_value_grad_ragged_f64(
            ws.handle, B, xd.data_ptr(), yd.data_ptr(), md.data_ptr(), None if nz is None else nz.data_ptr(),
            None if ns is None else ns.data_ptr(), n.ctypes.data_as(c_void_p), S, d, wd.data_ptr(), mud.data_ptr(), vd.data_ptr(), q,
            int(dim_order), float(jitter), 1 if need_grad else 0, gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], info.data_ptr(),
            torch.cuda.current_stream(dev).cuda_stream)
    _check(rc, "pgm_mll_value_grad_ragged_f64")
    out = _Outputs(buf, offs, {"mll": (0, (B,)), "g_w": (1, (B, q)), "g_mu": (2, (B, q, d)), "g_v": (3, (B, q, d)),
                               "g_noise": (4, (B, S)), "g_mean": (5, (B, S))} if need_grad else {"mll": (0, (B,))})
    out["info"] = info
    out["_keep"] = (xd, yd, md, nz, ns, wd, mud, vd)
    out["workspace"] = ws
    out["launch_sets"] = ragged_plan(n, workspace=ws)[1]         # (block rows of the sets this call ran: host arithmetic only)
    return out


class _KernelProgramStruct(ctypes.Structure):
    """``pgm_kernel_program`` of include/pgmuvi_hip.h."""
    _fields_ = [("nleaf", c_int), ("nterm", c_int), ("nparam", c_int),
                ("kind", ctypes.c_ubyte * 6), ("dims", ctypes.c_ubyte * 6), ("par", ctypes.c_ubyte * 6),
                ("tmask", ctypes.c_ubyte * 4), ("tnscale", ctypes.c_ubyte * 4), ("tscale", (ctypes.c_ubyte * 3) * 4)]


def kernel_program_struct(program) -> _KernelProgramStruct:
    """C image of a ``gpytorch.kernels.KernelProgram`` (built once per program)."""
    st = getattr(program, "_struct", None)
    if st is not None:
        return st
    st = _KernelProgramStruct()
    st.nleaf, st.nterm, st.nparam = len(program.leaves), len(program.terms), len(program.params)
    for l, ((kind, mask, _), par) in enumerate(zip(program.leaves, program.leaf_par)):
        st.kind[l], st.dims[l], st.par[l] = kind, mask, par
    for t, ((lv, _), sc) in enumerate(zip(program.terms, program.term_scales)):
        st.tmask[t] = sum(1 << l for l in lv)
        st.tnscale[t] = len(sc)
        for i, p in enumerate(sc):
            st.tscale[t][i] = p
    program._struct = st
    return st


def mll_kernel_value_grad(x, y, mean, noise, noise_scalar, program, theta, jitter=0.0, need_grad=True,
                          workspace: Optional[Workspace] = None):
    """One MLL evaluation of a composed stationary kernel (``program``: gpytorch.kernels.KernelProgram, ``theta`` its (P,)
    constrained parameter values on the device) through pgm_mll_kernel_value_grad_f64.  x (n,d) y (n) mean (n)
    noise (n)|None noise_scalar 0-dim tensor|float|None.  Returns dict(mll, g_theta, g_noise, g_mean, info, workspace)."""
    require_gpu(x, "mll_kernel_value_grad")
    dev = x.device
    n = y.shape[-1]
    xd = _dev64(x.reshape(n, -1), dev)
    d = xd.shape[-1]
    yd = _dev64(y.reshape(n), dev)
    md = _dev64(mean.expand(y.shape).reshape(n), dev)
    nz = None if noise is None else _dev64(noise.expand(y.shape).reshape(n), dev)
    th = _dev64(theta.reshape(1, -1), dev)
    P = th.shape[-1]
    ns = None if noise_scalar is None else _dev64(torch.as_tensor(noise_scalar, device=dev).reshape(1), dev)
    ws = workspace or get_workspace(dev, n, max(P, 4), d, 1)
    out = dict(mll=torch.empty(1, dtype=torch.float64, device=dev), info=torch.zeros(1, dtype=torch.int32, device=dev))
    if need_grad:
        out.update(g_theta=torch.empty(P, dtype=torch.float64, device=dev), g_noise=torch.empty(n, dtype=torch.float64, device=dev),
                   g_mean=torch.empty(n, dtype=torch.float64, device=dev))
    st = kernel_program_struct(program)
    with torch.cuda.device(dev):
        rc = load().pgm_mll_kernel_value_grad_f64(ws.handle, 1, _ptr(xd), _ptr(yd), _ptr(md), _ptr(nz), _ptr(ns), n, d, byref(st), _ptr(th),
                                                  float(jitter), 1 if need_grad else 0, _ptr(out["mll"]), _ptr(out.get("g_theta")),
                                                  _ptr(out.get("g_noise")), _ptr(out.get("g_mean")), _ptr(out["info"]), current_stream_ptr(dev))
    _check(rc, "pgm_mll_kernel_value_grad_f64")
    out["_keep"] = (xd, yd, md, nz, ns, th)
    out["mll"], out["info"] = out["mll"][0], out["info"][0]
    out["workspace"] = ws
    out["evaluation"] = ws.last_evaluation()
    return out


def probe_mfma_f64(device_index: int = 0) -> float:
    val = c_double()
    _check(load().pgm_probe_mfma_f64(device_index, byref(val)), "pgm_probe_mfma_f64")
    return float(val.value)


def predict(ws: Workspace, x_test: torch.Tensor, mean_test: torch.Tensor):
    """Posterior mean and latent variance at ``x_test`` from the factor the last
    ``mll_value_grad(need_grad=True)`` call left in ``ws`` (pgm_predict_f64)."""
    require_gpu(x_test, "predict")
    dev = x_test.device
    xt = _dev64(x_test.reshape(x_test.shape[0], -1), dev)
    m = xt.shape[0]
    mt = _dev64(mean_test.expand(m).reshape(m), dev)
    pm = torch.empty(m, dtype=torch.float64, device=dev)
    pv = torch.empty(m, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_predict_f64(ws.handle, _ptr(xt), _ptr(mt), m, _ptr(pm), _ptr(pv), current_stream_ptr(dev))
    _check(rc, "pgm_predict_f64")
    return pm, pv


def lomb_scargle(t: torch.Tensor, y: torch.Tensor, dy: Optional[torch.Tensor], freq: torch.Tensor, fit_mean=True,
                 center_data=True) -> torch.Tensor:
    """Floating-mean Lomb-Scargle power (B, Nf) of B light curves (B, N) on one frequency grid (pgm_lomb_scargle_f64).
    ``center_data`` is implied by ``fit_mean`` (astropy centres in both cases unless both are switched off)."""
    require_gpu(y, "lomb_scargle")
    if not (fit_mean or center_data):
        raise NotImplementedError("fit_mean=False with center_data=False (an uncentred classical periodogram) is not implemented")
    dev = y.device
    if y.dim() == 1:
        t, y = t.reshape(1, -1), y.reshape(1, -1)
        dy = None if dy is None else dy.reshape(1, -1)
    B, n = y.shape
    td = _dev64(t.expand(B, n), dev)
    yd = _dev64(y, dev)
    dd = None if dy is None else _dev64(dy.expand(B, n), dev)
    fd = _dev64(freq.reshape(-1), dev)
    nf = fd.numel()
    scratch = torch.empty((B, 2 * n + 1), dtype=torch.float64, device=dev)
    power = torch.empty((B, nf), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_lomb_scargle_f64(_ptr(td), _ptr(yd), _ptr(dd), n, B, _ptr(fd), nf, 1 if fit_mean else 0, _ptr(scratch),
                                         _ptr(power), current_stream_ptr(dev))
    _check(rc, "pgm_lomb_scargle_f64")
    return power


def lomb_scargle_fast(t: torch.Tensor, y: torch.Tensor, dy: Optional[torch.Tensor], f0: float, df: float, nf: int, fit_mean=True,
                      center_data=True, oversampling: int = 5) -> torch.Tensor:
    """The periodogram on the regular grid f0 + df * arange(nf) by the FFT approximation of astropy's ``method='auto'``
    (pgm_lomb_scargle_fast_f64): (B, nf) for B light curves (B, N)."""
    require_gpu(y, "lomb_scargle_fast")
    if not (fit_mean or center_data):
        raise NotImplementedError("fit_mean=False with center_data=False (an uncentred classical periodogram) is not implemented")
    dev = y.device
    if y.dim() == 1:
        t, y = t.reshape(1, -1), y.reshape(1, -1)
        dy = None if dy is None else dy.reshape(1, -1)
    B, n = y.shape
    td = _dev64(t.expand(B, n), dev)
    yd = _dev64(y, dev)
    dd = None if dy is None else _dev64(dy.expand(B, n), dev)
    per = int(load().pgm_lomb_scargle_fast_scratch_doubles(n, int(nf), int(oversampling)))
    scratch = torch.empty((B, per), dtype=torch.float64, device=dev)
    power = torch.empty((B, int(nf)), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_lomb_scargle_fast_f64(_ptr(td), _ptr(yd), _ptr(dd), n, B, float(f0), float(df), int(nf), 1 if fit_mean else 0,
                                              int(oversampling), _ptr(scratch), _ptr(power), current_stream_ptr(dev))
    _check(rc, "pgm_lomb_scargle_fast_f64")
    return power


def mll_dense(A: torch.Tensor, r: torch.Tensor, jitter: float = 0.0, need_grad: bool = True, workspace: Optional[Workspace] = None):
    """Per-datum MLL of r ~ N(0, A) for a dense symmetric A (n,n) | (B,n,n) built by the caller (any kernel), with
    dmll/dA and dmll/dr (pgm_mll_dense_f64).  Returns dict(mll, g_a, g_r, info, workspace); no host sync."""
    require_gpu(A, "mll_dense")
    dev = A.device
    batched = A.dim() == 3
    B = A.shape[0] if batched else 1
    n = A.shape[-1]
    Ad = _dev64(A.reshape(B, n, n), dev)
    rd = _dev64(r.reshape(B, n), dev)
    ws = workspace or get_workspace(dev, n, 1, 1, B)
    out = dict(mll=torch.empty(B, dtype=torch.float64, device=dev), info=torch.zeros(B, dtype=torch.int32, device=dev))
    if need_grad:
        out["g_a"] = torch.empty((B, n, n), dtype=torch.float64, device=dev)
        out["g_r"] = torch.empty((B, n), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_mll_dense_f64(ws.handle, B, _ptr(Ad), n, _ptr(rd), n, float(jitter), 1 if need_grad else 0, _ptr(out["mll"]),
                                      _ptr(out.get("g_a")), n, _ptr(out.get("g_r")), _ptr(out["info"]), current_stream_ptr(dev))
    _check(rc, "pgm_mll_dense_f64")
    out["_keep"] = (Ad, rd)
    if not batched:
        for k in ("mll", "info", "g_a", "g_r"):
            if k in out:
                out[k] = out[k][0]
    out["workspace"] = ws
    return out


def predict_dense(ws: Workspace, k_star: torch.Tensor, k_ss: torch.Tensor, mean_test: torch.Tensor):
    """Posterior mean / latent variance from the factor the last ``mll_dense`` left in ``ws``: k_star = K(x_train, x_test)
    (n, m), k_ss the prior variances (m)."""
    require_gpu(k_star, "predict_dense")
    dev = k_star.device
    ks = _dev64(k_star, dev)
    n, m = ks.shape
    kss = _dev64(k_ss.expand(m).reshape(m), dev)
    mt = _dev64(mean_test.expand(m).reshape(m), dev)
    pm = torch.empty(m, dtype=torch.float64, device=dev)
    pv = torch.empty(m, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = load().pgm_predict_dense_f64(ws.handle, _ptr(ks), m, _ptr(kss), _ptr(mt), m, _ptr(pm), _ptr(pv), current_stream_ptr(dev))
    _check(rc, "pgm_predict_dense_f64")
    return pm, pv


class NativeFit:
    """Handle of a device-resident fit (``pgm_fit_*``): the optimiser loop of a constant-mean spectral-mixture exact GP as
    one hipGraph replay per iteration.  ``x`` (n,d), ``y`` (n), ``noise`` (n)|None are kept alive here."""

    OPT = {"SGD": 0, "Adam": 1, "AdamW": 2}

    def __init__(self, x, y, noise, q, dim_order, raw0, ckind, ca, cb, has_noise_param, optimizer, lr, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, max_iter=100, workspace: Optional[Workspace] = None, linear_mean=False):
        import numpy as np
        require_gpu(y, "NativeFit")
        dev = y.device
        n = y.shape[-1]
        self.x = _dev64(x.reshape(n, -1), dev)
        self.y = _dev64(y.reshape(n), dev)
        self.noise = None if noise is None else _dev64(noise.expand(n).reshape(n), dev)
        d = self.x.shape[-1]
        self.P = len(raw0)
        self.max_iter = int(max_iter)
        self.ws = workspace or get_workspace(dev, n, q, d, 1)
        self.dev = dev
        arr = lambda a, t: np.ascontiguousarray(np.asarray(a, dtype=t))
        r0, ck, a_, b_ = arr(raw0, np.float64), arr(ckind, np.int32), arr(ca, np.float64), arr(cb, np.float64)
        h = c_void_p()
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = load().pgm_fit_create(byref(h), self.ws.handle, _ptr(self.x), _ptr(self.y), _ptr(self.noise), n, d, q, int(dim_order),
                                       1 if linear_mean else 0, r0.ctypes.data_as(c_void_p), ck.ctypes.data_as(c_void_p), a_.ctypes.data_as(c_void_p),
                                       b_.ctypes.data_as(c_void_p), 1 if has_noise_param else 0, self.OPT[optimizer], float(lr),
                                       float(betas[0]), float(betas[1]), float(eps), float(weight_decay), self.max_iter)
        _check(rc, "pgm_fit_create")
        self.handle = h

    def set_priors(self, kind, loc, scale):
        """Priors on the constrained parameters (per raw-vector entry: 0 none, 1 Normal, 2 LogNormal); before the first run."""
        import numpy as np
        arr = lambda a, t: np.ascontiguousarray(np.asarray(a, dtype=t))
        k, l, s = arr(kind, np.int32), arr(loc, np.float64), arr(scale, np.float64)
        if not (k.shape == l.shape == s.shape == (self.P,)):
            raise ValueError(f"prior tables must have {self.P} entries")
        with torch.cuda.device(self.dev):
            _check(load().pgm_fit_set_priors(self.handle, k.ctypes.data_as(c_void_p), l.ctypes.data_as(c_void_p),
                                             s.ctypes.data_as(c_void_p)), "pgm_fit_set_priors")

    def run(self, iters: int):
        with torch.cuda.device(self.dev):
            _check(load().pgm_fit_run(self.handle, int(iters), current_stream_ptr(self.dev)), "pgm_fit_run")

    def read(self):
        """Synchronises; returns (iterations done, losses (it,), raw parameters after each step (it,P), current raw (P,), info)."""
        import numpy as np
        it, info = c_int(), c_int()
        loss = np.zeros(self.max_iter); hist = np.zeros((self.max_iter, self.P)); raw = np.zeros(self.P)
        with torch.cuda.device(self.dev):
            rc = load().pgm_fit_read(self.handle, current_stream_ptr(self.dev), byref(it), loss.ctypes.data_as(c_void_p),
                                     hist.ctypes.data_as(c_void_p), raw.ctypes.data_as(c_void_p), byref(info))
        _check(rc, "pgm_fit_read")
        k = int(it.value)
        return k, loss[:k], hist[:k], raw, int(info.value)

    def close(self):
        if getattr(self, "handle", None):
            load().pgm_fit_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativePotential:
    """Handle of the sampler's potential on the device (``pgm_pot_*``): for B chains, z (B,P) on the host -> (U (B,), dU/dz (B,P),
    info (B,)) on the host, one hipGraph replay per call, positions and results through host-mapped memory.  ``x`` (B,n,d),
    ``y`` (B,n), ``noise`` (B,n)|None are kept alive here; ``loc`` / ``scale`` (B,P): the Normal density of every entry of z."""

    def __init__(self, x, y, noise, q, dim_order, loc, scale, workspace: Optional[Workspace] = None):
        import numpy as np
        require_gpu(y, "NativePotential")
        dev = y.device
        B, n = y.shape
        self.x = _dev64(x.reshape(B, n, -1), dev)
        self.y = _dev64(y, dev)
        self.noise = None if noise is None else _dev64(noise.expand(B, n), dev)
        d = self.x.shape[-1]
        self.B, self.P = B, 1 + q + 2 * q * d + (1 if noise is None else 0)
        self.ws = workspace or get_workspace(dev, n, q, d, B)
        self.dev = dev
        lo = np.ascontiguousarray(np.broadcast_to(np.asarray(loc, dtype=np.float64), (B, self.P)))
        sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (B, self.P)))
        h = c_void_p()
        with torch.cuda.device(dev):
            torch.cuda.synchronize(dev)
            rc = load().pgm_pot_create(byref(h), self.ws.handle, B, _ptr(self.x), _ptr(self.y), _ptr(self.noise), n, d, q, int(dim_order),
                                       lo.ctypes.data_as(c_void_p), sc.ctypes.data_as(c_void_p))
        _check(rc, "pgm_pot_create")
        self.handle = h
        self._u = np.zeros(B); self._g = np.zeros((B, self.P)); self._info = np.zeros(B, dtype=np.int32)

    def __call__(self, z):
        import numpy as np
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (self.B, self.P):
            raise ValueError(f"NativePotential: z must be ({self.B}, {self.P}), got {z.shape}")
        with torch.cuda.device(self.dev):
            rc = load().pgm_pot_eval(self.handle, z.ctypes.data_as(c_void_p), self._u.ctypes.data_as(c_void_p),
                                     self._g.ctypes.data_as(c_void_p), self._info.ctypes.data_as(c_void_p), current_stream_ptr(self.dev))
        _check(rc, "pgm_pot_eval")
        return self._u.copy(), self._g.copy(), self._info.copy()

    def close(self):
        if getattr(self, "handle", None):
            load().pgm_pot_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
