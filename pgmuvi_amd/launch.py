"""One process per GPU without an external launcher.

``python bench.py --gpus N`` (N > 1) must work as the driver types it.  The process that
was started is then only a *parent*: before anything touches the GPU it starts N fresh
children of the same command line -- one per rank, with ``RANK`` / ``LOCAL_RANK`` /
``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT`` in their environment, exactly what
``python -m torch.distributed.run`` would have set -- relays rank 0's standard output
(the one JSON line) and returns the worst exit status of the children.  No process that
has initialised the GPU is ever replaced by another program (that takes the whole
machine down on this pool): children are started with ``subprocess.Popen``.

Nothing here imports torch.  The parent counts the GPUs without the HIP runtime (:func:`visible_gpu_count`: the kernel
driver's topology files and the ``*_VISIBLE_DEVICES`` variables), so it never opens the device it then hands to its children.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import shutil
import tempfile
import threading
import time
from typing import Dict, List, Optional, Sequence

CHILD_MARK = "PGM_LAUNCH_CHILD"


def under_a_launcher(env=None) -> bool:
    """Has somebody (torch.distributed.run, or :func:`spawn_ranks`) already made this process one rank of a job?"""
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def _kfd_gpu_nodes(root: str = "/sys/class/kfd/kfd/topology/nodes", dri: str = "/dev/dri") -> Optional[int]:
    """GPUs the kernel driver lists and this process may open: topology nodes with ``simd_count`` > 0 (CPU nodes have 0) whose
    render node ``/dev/dri/renderD<drm_render_minor>`` is readable and writable (a container that was given one GPU of eight
    still sees all eight in the topology).  None when the files are not there (no amdgpu driver, or a sandbox that hides /sys)."""
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    count = 0
    for node in nodes:
        props = {}
        try:
            with open(os.path.join(root, node, "properties")) as fh:
                for line in fh:
                    key, _, val = line.partition(" ")
                    if key in ("simd_count", "drm_render_minor"):
                        props[key] = int(val)
        except (OSError, ValueError):
            continue                                           # (a node this user may not read is not a device it may use)
        if props.get("simd_count", 0) <= 0:
            continue
        minor = props.get("drm_render_minor", -1)
        if minor > 0 and os.path.isdir(dri) and not os.access(os.path.join(dri, f"renderD{minor}"), os.R_OK | os.W_OK):
            continue
        count += 1
    return count


def visible_gpu_count(env=None, kfd_root: str = "/sys/class/kfd/kfd/topology/nodes") -> Optional[int]:
    """How many GPUs a child of this process would see -- WITHOUT touching the HIP runtime: the driver's topology count, cut
    down by ``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` (comma lists; an empty value hides
    every device) as the runtime would apply them.  None: unknown (no topology files) -- the caller then skips its check and
    lets the ranks report what they find."""
    env = os.environ if env is None else env
    total = _kfd_gpu_nodes(kfd_root)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in env:
            listed = [t for t in env[var].split(",") if t.strip() != ""]
            if any(t.strip() == "-1" for t in listed):          # (the runtime stops at the first -1)
                listed = listed[:[t.strip() for t in listed].index("-1")]
            total = len(listed) if total is None else min(total, len(listed))
    return total


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


INIT_VAR = "PGM_LAUNCH_INIT"       # rendezvous of a job started by spawn_ranks: a file:// URL in a directory the parent made
SHARE_VAR = "PGM_SHARE_GPU"        # "1": test-only job whose ranks all use device 0 over gloo (bench.py --share-gpu)


def rank_environment(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None,
                     init_method: Optional[str] = None, share_gpu: bool = False) -> Dict[str, str]:
    """Environment of rank ``rank`` of a one-node job of ``world`` ranks (one rank per GPU: LOCAL_RANK == RANK).
    ``init_method``: the job's rendezvous (``file://...``; :func:`init_method_of` hands it to ``init_process_group``) -- a
    path only this job knows, where ``MASTER_PORT`` is a port number found free a moment ago that anybody may have taken since."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # the host driver only supports dmabuf IPC (RCCL needs it)
    env[CHILD_MARK] = "1"
    if init_method:
        env[INIT_VAR] = init_method
    else:
        env.pop(INIT_VAR, None)
    if share_gpu:
        env[SHARE_VAR] = "1"
    return env


def init_method_of(env=None) -> Optional[str]:
    """What a rank passes as ``init_process_group(init_method=...)``: the launcher's file rendezvous when :func:`spawn_ranks`
    started it, None (= ``env://``: MASTER_ADDR / MASTER_PORT) under torch.distributed.run."""
    env = os.environ if env is None else env
    return env.get(INIT_VAR) or None


def spawn_ranks(argv: Sequence[str], world: int, visible_devices: Optional[int] = None, grace_s: float = 15.0,
                out=None, err=None, share_gpu: bool = False) -> int:
    """Runs ``argv`` (a full command line, e.g. ``[sys.executable, "bench.py", "--gpus", "4", ...]``) once per rank and
    waits.  Rank 0's stdout is relayed line by line to ``out`` (default: this process's stdout); the other ranks' stdout goes
    to ``err`` with everybody's stderr.  Returns 0 when every rank returned 0, otherwise the first non-zero status seen (a
    rank that fails takes the others down after ``grace_s`` seconds: they would wait in a collective for ever).
    ``visible_devices``: refuse (status 2, one line on ``err``) when fewer devices than ranks are visible; None = no check
    (CPU jobs on gloo).  ``share_gpu`` (TEST MODE, never a default): every rank uses device 0 and the ranks talk over gloo --
    the way a one-GPU box runs the HIP path in several processes of one job; refused when two or more devices are visible
    (then one rank per GPU over RCCL is the job to run) or none is."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    if world < 1:
        print(f"launch: {world} ranks requested", file=err)
        return 2
    if share_gpu:
        if visible_devices is not None and visible_devices != 1:
            print(f"launch: --share-gpu is the one-GPU test mode, this host shows {visible_devices} GPUs"
                  + (" (run one rank per GPU instead)" if visible_devices > 1 else ""), file=err)
            return 2
    elif visible_devices is not None and visible_devices < world:
        print(f"launch: --gpus {world} needs {world} visible GPUs, this host shows {visible_devices}", file=err)
        return 2
    port = free_port()                                         # (MASTER_PORT for code that insists on env://; the ranks meet through the file below)
    rdzv_dir = tempfile.mkdtemp(prefix="pgm_launch_")
    init_method = "file://" + os.path.join(rdzv_dir, "rendezvous")
    procs: List[subprocess.Popen] = []
    try:
        child_err = err.fileno()                               # (a real descriptor: the children write to it directly)
    except (AttributeError, OSError, ValueError):
        child_err = None                                       # inherit this process's stderr

    def stop_children(sig):                                    # exactly the processes started below, by handle
        for p in procs:
            if p.poll() is None:
                try:
                    p.send_signal(sig)
                except OSError:
                    pass

    # a driver that ends the parent on a time-out must not leave N processes holding GPUs: the parent passes SIGTERM / SIGINT
    # on to its children and leaves through the ``finally`` below (only in the main thread: signal handlers live there)
    previous = {}
    caught = []

    def on_signal(signum, frame):
        caught.append(signum)
        stop_children(signal.SIGTERM)
        raise KeyboardInterrupt(f"launch: signal {signum}")

    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT):
            try:
                previous[sg] = signal.signal(sg, on_signal)
            except (OSError, ValueError):
                pass
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_environment(r, world, port, init_method=init_method, share_gpu=share_gpu),
                                          stdout=subprocess.PIPE if r == 0 else (child_err if child_err is not None else 2),
                                          stderr=child_err, text=(r == 0)))
        # rank 0's lines are relayed as they come (it prints little: the JSON line at the end) by a reader thread, so that
        # this thread keeps watching every rank: one that dies early leaves the others waiting in a collective
        def relay():
            for line in procs[0].stdout:
                out.write(line)
                out.flush()
        reader = threading.Thread(target=relay, daemon=True)
        reader.start()
        status, failed_at, terminated_at = 0, None, None
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and status == 0:
                    status, failed_at = rc, time.monotonic()
                    print(f"launch: rank {r} exited with status {rc}", file=err)
            now = time.monotonic()
            if pending and failed_at is not None and terminated_at is None and now - failed_at > grace_s:
                stop_children(signal.SIGTERM)
                terminated_at = now
            elif pending and terminated_at is not None and now - terminated_at > grace_s:
                stop_children(signal.SIGKILL)                   # (a rank that ignores or blocks SIGTERM)
                terminated_at = now
            if pending:
                time.sleep(0.05)
        reader.join(timeout=10)
        return status if status >= 0 else 128 - status             # (killed by a signal: the shell's convention)
    except KeyboardInterrupt:
        # the children got SIGTERM from the handler (or the terminal's SIGINT themselves): give them ``grace_s`` to leave their
        # kernels and tear RCCL down before the ``finally`` below kills what is left; 128 + signal as a shell reports it
        interrupted = 128 + (caught[-1] if caught else signal.SIGINT)
        for sg in previous:                                    # (a second signal must not break the cleanup off)
            try:
                signal.signal(sg, signal.SIG_IGN)
            except (OSError, ValueError):
                pass
        if not caught:
            stop_children(signal.SIGTERM)
        deadline = time.monotonic() + grace_s
        while time.monotonic() < deadline and any(p.poll() is None for p in procs):
            time.sleep(0.05)
        return interrupted
    finally:
        for sg in previous:
            try:
                signal.signal(sg, signal.SIG_IGN)
            except (OSError, ValueError):
                pass
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
        shutil.rmtree(rdzv_dir, ignore_errors=True)
        for sg, h in previous.items():
            try:
                signal.signal(sg, h)
            except (OSError, ValueError):
                pass
