"""CPU tests of the C-ABI boundary: the shared library loads without a GPU and exports
exactly the entry points ``include/pgmuvi_hip.h`` declares (no compute call is made)."""
import ctypes
import os
import re

import pytest

from pgmuvi_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pgmuvi_hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pgm_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_expected_entry_points():
    names = _declared()
    for must in ("pgm_workspace_create", "pgm_workspace_destroy", "pgm_sm_kernel_f64", "pgm_mll_value_grad_f64",
                 "pgm_mll_value_grad_batched_f64", "pgm_predict_f64"):
        assert must in names
    txt = open(HEADER).read()
    code = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    assert "torch" not in code.lower() and "at::" not in code and 'extern "C"' in code     # plain C ABI, no torch types
    for cite in ("pgmuvi/trainers.py:179-181", "pgmuvi/gps.py:208", "pgmuvi/lightcurve.py:9607"):
        assert cite in txt                                           # each entry point cites what it replaces


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_hip.lib_path())
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"


def test_ctypes_table_matches_header():
    assert sorted(_hip.SYMBOLS) == _declared()
    lib = _hip.load()
    assert _hip.version().startswith("pgmuvi_hip") and "gfx950" in _hip.version()
    assert _hip.max_qd() == 16
    assert lib.pgm_profile_phases() == 8
    names = [lib.pgm_profile_phase_name(i).decode() for i in range(7)]
    assert "trailing_update" in names and "diag_block" in names


def test_size_contract_is_an_argument_check():
    """include/pgmuvi_hip.h's size contract: 1 <= max_n <= pgm_max_n() = 16384 (128 block rows); anything else is refused
    with -3 BEFORE a device is touched (so it can be checked here, without a GPU), and the Python binding says why."""
    lib = _hip.load()
    assert _hip.max_n_limit() == 16384
    h = ctypes.c_void_p()
    for bad in (0, -5, 16385, 1 << 20, 1 << 40):
        assert lib.pgm_workspace_create(ctypes.byref(h), 0, bad, 4, 1, 1) == -3 and not h.value
    with pytest.raises(RuntimeError, match="at most 16384"):
        _hip.Workspace("cuda:0", 16385, 4, 1, 1)
    txt = open(HEADER).read()
    assert "SIZE CONTRACT" in txt and "pgm_max_n() = 16384" in txt and "paper/paper.md:144" in txt


def test_code_object_is_gfx950_only():
    blob = open(_hip.lib_path(), "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob


def test_workspace_cache_policy_without_a_gpu(monkeypatch):
    """``_hip.get_workspace``: a request is served by the smallest cached workspace that covers it; a miss allocates an exact
    fit; going over the byte budget only DROPS least-recently-used entries from the cache (their buffers live on for whoever
    still holds them -- a native fit, a captured graph) and never closes a handle."""
    from pgmuvi_amd import _hip

    class FakeWorkspace:
        made, closed = [], []

        def __init__(self, device, max_n, max_q, max_d, max_batch=1):
            _hip.load()                                        # (the real constructor takes the loader's lock: must not deadlock)
            self.key = (0, max_n, max_q, max_d, max_batch)
            self.max_n, self.max_q, self.max_d, self.max_batch = max_n, max_q, max_d, max_batch
            self.handle = object()
            self.nominal_bytes = 8 * max_n * max_n * max_batch
            FakeWorkspace.made.append(self)

        def close(self):
            FakeWorkspace.closed.append(self)
            self.handle = None

    import contextlib
    monkeypatch.setattr(_hip, "Workspace", FakeWorkspace)
    monkeypatch.setattr(_hip.torch.cuda, "device", lambda idx: contextlib.nullcontext())
    monkeypatch.setattr(_hip, "_workspaces", {})
    monkeypatch.setattr(_hip, "WORKSPACE_BUDGET_BYTES", 8 * 2048 * 2048 * 5)
    dev = "cuda:0"
    a = _hip.get_workspace(dev, 2000, 4, 1, 4)                 # 2048-point blocks, batch 4
    assert a.max_n == 2048 and _hip.get_workspace(dev, 2048, 4, 1, 4) is a
    assert _hip.get_workspace(dev, 100, 1, 1, 1) is a and _hip.get_workspace(dev, 1000, 2, 1, 3) is a
    b = _hip.get_workspace(dev, 1000, 2, 2, 1)                 # d = 2 is not covered by a (max_d 1): a second, small workspace
    assert b is not a and b.max_d == 2 and len(FakeWorkspace.made) == 2
    assert _hip.get_workspace(dev, 900, 1, 1, 1) is b          # the smallest that covers
    c = _hip.get_workspace(dev, 2048, 4, 1, 5)                 # batch 5 > 4: a third; 4 + 5 batches of 2048^2 exceed the budget of 5
    assert c is not a and len(FakeWorkspace.made) == 3
    assert a not in _hip._workspaces.values() and c in _hip._workspaces.values()       # the least recently used went
    assert FakeWorkspace.closed == [] and a.handle is not None                          # ... out of the cache only
    assert _hip.get_workspace(dev, 2000, 4, 1, 4) is c
    _hip.release_workspaces()
    assert _hip._workspaces == {} and FakeWorkspace.closed == []


def test_lazy_result_dictionary_behaves_like_the_plain_one():
    """``mll_value_grad`` returns its fp64 outputs as views of ONE buffer that are only made when asked for (the training
    loop reads ``mll`` and the buffer); every way the tests and the callers read the dictionary must see all of them."""
    import torch
    buf = torch.arange(12.0, dtype=torch.float64)
    out = _hip._Outputs(buf, [0, 1, 4, 12], {"mll": (0, ()), "g_w": (1, (3,)), "g_noise": (2, (2, 4))})
    out["info"] = torch.zeros((), dtype=torch.int32)
    assert "g_w" in out and "g_noise" in out and "g_mean" not in out and out.get("g_mean") is None
    assert float(out["mll"]) == 0.0 and out["g_w"].tolist() == [1.0, 2.0, 3.0] and tuple(out.get("g_noise").shape) == (2, 4)
    assert sorted(out.keys()) == ["g_noise", "g_w", "info", "mll"] and len(out) == 4 and sorted(out) == sorted(out.keys())
    assert {k: tuple(v.shape) for k, v in out.items()} == {"mll": (), "g_w": (3,), "g_noise": (2, 4), "info": ()}
    assert out["g_w"].data_ptr() == buf.data_ptr() + 8                     # views, not copies
    with pytest.raises(KeyError):
        out["g_mean"]


def _plan_check():
    f = ctypes.CDLL(_hip.lib_path()).pgm_debug_plan_check
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] * 9
    return f


def test_sweep_plans_are_valid_for_every_shape():
    """Host logic of the factorisation sweep, no GPU: `pgm_debug_plan_check` runs the planner of `run_sweep` dry (nothing is
    launched) and replays the plan -- the row masks, the `own` row, the lone diagonal tile -- through a restatement of the
    kernels' tile decode on a table of 'next source this tile expects'.  Valid = every tile receives the finished block rows in
    ascending order, none twice, none skipped (a row three behind would silently lose a source: the kernels apply at most two
    per pass); every block row is complete when its diagonal block / row solve starts; the look-ahead finds its copy of tile
    (k, k+1) and an up-to-date diagonal tile (k+1, k+1) and never shares a launch with tail tiles of that row; the early
    inverse pass's tasks (one light curve with gradient) take each tile's products in ascending order from block rows that are
    final, never the same tile twice in one launch, and leave the final pass exactly the rest.  All block-row
    counts up to 64 (N = 8192), batch sizes, value-only, look-ahead on / off / switched in mid-sweep, lazy / eager plan,
    fused / panels / windowed."""
    f = _plan_check()
    bad, n = [], 0
    for nb in range(1, 65):
        for batch in (1, 2, 3, 5, 8, 16, 48, 64):
            for need_grad in (0, 1):
                for lookahead in (0, 5, 99):
                    for lazy in (0, 1):
                        for panel, window in ((-1, -1), (0, -1), (4, -1), (-1, 0), (-1, 8)):
                            e = f(nb, batch, need_grad, lookahead, lazy, panel, window, 0, 0)
                            n += 1
                            if e != 0:
                                bad.append((nb, batch, need_grad, lookahead, lazy, panel, window, e))
    assert n == 64 * 8 * 2 * 3 * 2 * 5 and not bad, bad[:10]
    assert f(0, 1, 1, 0, 1, -1, -1, 0, 0) == -1 and f(32, 0, 1, 0, 1, -1, -1, 0, 0) == -1


def test_the_plan_check_notices_a_damaged_plan():
    """The check's own test: with the planner damaged inside the dry run (mutate = 1: the lone diagonal-tile entry is
    forgotten; 2: block rows may fall three sources behind) the replay reports violations exactly where the lazy plan is in
    use (32 block rows and more), and none for the intact planner."""
    f = _plan_check()
    for nb in (32, 36, 40):
        assert f(nb, 1, 1, 0, 1, -1, -1, 0, 0) == 0
        assert f(nb, 1, 1, 0, 1, -1, -1, 0, 1) > 0
        assert f(nb, 1, 1, 0, 1, -1, -1, 0, 2) > 0
    assert f(8, 1, 1, 0, 1, -1, -1, 0, 1) == 0                  # (every row fits every launch: nothing lazy to damage)
    # the early inverse pass's task tables (one light curve with gradient, 8 block rows and more) ride in the same check:
    # damaged so that a tile may be taken twice in one launch (read-modify-write of R within a launch), it must notice
    # (up to ~24 block rows every launch has room for all the products that are ready, a tile never has two of them
    #  waiting, and the damage cannot show)
    for nb in (8, 16, 32, 40, 48):
        assert f(nb, 1, 1, 0, 1, -1, -1, 0, 0) == 0
        assert (f(nb, 1, 1, 0, 1, -1, -1, 0, 3) > 0) == (nb >= 32)


def _decisions(n, q, d, batch, need_grad=1):
    f = ctypes.CDLL(_hip.lib_path()).pgm_debug_decisions
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    out = (ctypes.c_int * 8)()
    rc = f(n, q, d, batch, need_grad, out)
    return rc, dict(one_launch=out[0], block_rows=out[1], k_blocks_per_item=out[2], items=out[3], workgroups_per_item=out[4])


def test_which_way_a_call_goes_host_logic_of_round_6(monkeypatch):
    """Host logic, no GPU (`pgm_debug_decisions`): the one launch or the launch sequence for light curves of at most 128 points
    (small_ok's measured table: by input dimensions, rows of 16x16 sub-blocks and mixtures, for up to 20 light curves per call),
    and the split of the inverse/gradient pass -- k-blocks per work item, whole / quarter / sixteenth tiles -- by block rows and
    light curves per call; the switches that bring the other paths back."""
    for k_ in ("PGM_SMALL", "PGM_LAUUM_SUB", "PGM_LAUUM_SUB16"):
        monkeypatch.delenv(k_, raising=False)
    one = lambda *a: _decisions(*a)[1]["one_launch"]
    # the reference's published workload and the small sizes: one launch whatever the mixtures
    assert one(89, 2, 1, 1) == 1 and one(89, 4, 1, 1) == 1 and one(17, 16, 1, 1) == 1 and one(64, 8, 2, 1) == 1 and one(1, 1, 1, 1) == 1
    # 1-D: 9 / 6 / 2 mixtures at 6 / 7 / 8 rows of sub-blocks
    assert [one(96, q, 1, 1) for q in (9, 10)] == [1, 0] and [one(112, q, 1, 1) for q in (6, 7)] == [1, 0] and [one(128, q, 1, 1) for q in (2, 3)] == [1, 0]
    # 2-D: 7 / 2 / 1 / 0 at 5 .. 8 rows (the Lomb-Scargle notebook's 106 points in three bands, Q=3: the launch sequence)
    assert [one(80, q, 2, 1) for q in (7, 8)] == [1, 0] and [one(96, q, 2, 1) for q in (2, 3)] == [1, 0]
    assert [one(106, q, 2, 1) for q in (1, 2, 3)] == [1, 0, 0] and one(128, 1, 2, 1) == 0
    # the table holds for up to 20 light curves per call; beyond, every shape takes the one launch; 129 points never do
    assert one(128, 4, 1, 20) == 0 and one(128, 4, 1, 21) == 1 and one(128, 8, 2, 512) == 1 and one(129, 1, 1, 1) == 0
    monkeypatch.setenv("PGM_SMALL", "2")
    assert one(128, 16, 1, 1) == 1 and one(128, 8, 2, 1) == 1
    monkeypatch.setenv("PGM_SMALL", "0")
    assert one(17, 1, 1, 1) == 0 and one(89, 2, 1, 512) == 0
    monkeypatch.delenv("PGM_SMALL")
    # the inverse/gradient pass: one k-block per item and sixteenth tiles for 2 .. 4 block rows while the call has <= 20 items
    lau = lambda *a: tuple(_decisions(*a)[1][k] for k in ("block_rows", "k_blocks_per_item", "items", "workgroups_per_item"))
    assert lau(256, 4, 1, 1) == (2, 1, 4, 16) and lau(250, 3, 2, 1) == (2, 1, 4, 16) and lau(512, 4, 1, 1) == (4, 1, 20, 16)
    assert lau(256, 2, 1, 4) == (2, 1, 4, 16) and lau(256, 2, 1, 5) == (2, 1, 4, 16)                     # 4 items x 5 light curves = 20
    assert lau(256, 2, 1, 6) == (2, 2, 3, 16) and lau(256, 2, 1, 7) == (2, 2, 3, 4)                       # 24 > 20: round 5's split of two; 3 x 6 = 18 still sixteenths, 21: quarters
    assert lau(128, 4, 1, 1) == (1, 1, 1, 16)                                   # (the launch sequence of <= 128 points: one item)
    assert lau(640, 4, 1, 1)[3] == 4 and lau(1024, 4, 1, 1)[3] == 4              # five block rows and more: quarter tiles (<= 128 items)
    assert lau(2048, 4, 1, 1)[3] == 4 and lau(3000, 4, 1, 1)[3] == 4 and lau(4096, 4, 1, 1)[3] == 1       # 13 .. 24 block rows: quarter; beyond: whole
    assert lau(2048, 4, 1, 8)[3] == 1 and lau(2048, 4, 1, 512) == (16, 16, 136, 1)                       # batches: whole tiles, one item per tile from 64 light curves on
    assert _decisions(256, 4, 1, 1, 0)[1]["workgroups_per_item"] == 1            # value only: no such pass
    monkeypatch.setenv("PGM_LAUUM_SUB16", "0")
    assert lau(256, 4, 1, 1)[1:] == (2, 3, 4)                                    # round 5's form: k-split of two, quarter tiles
    monkeypatch.setenv("PGM_LAUUM_SUB", "0")
    assert lau(256, 4, 1, 1)[3] == 1 and lau(2048, 4, 1, 1)[3] == 1
    # bad arguments
    assert _decisions(0, 1, 1, 1)[0] == -1 and _decisions(100, 17, 1, 1)[0] == -1 and _decisions(100, 1, 3, 1)[0] == -1 and _decisions(20000, 1, 1, 1)[0] == -1
