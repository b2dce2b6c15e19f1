"""One process per GPU, `torch.distributed` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" on CPU).

The hot paths shard into independent units (attention: (batch, head) problems; HGEMM: replicas), so the
ONLY collectives are a barrier around the timed region and a gather of per-rank timings — there is no
data-path collective.  Launch: `python -m torch.distributed.run --nproc-per-node N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT come from the environment), or `spawn()` below — the
torch.multiprocessing.spawn idiom of the reference's others/pytorch/distributed/test_dist_all.py:189-234 — which is
what a bare `python bench.py --gpus N` uses."""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass
class World:
    rank: int = 0
    local_rank: int = 0
    size: int = 1
    backend: str | None = None
    group: bool = False          # a process group exists (always for size > 1; for size == 1 only under LC_DIST_FORCE=1)

    @property
    def is_dist(self) -> bool:
        return self.size > 1 or self.group


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init(backend: str | None = None) -> World:
    """LC_DIST_FORCE=1 creates the process group even for ONE rank, so that the RCCL branch (init_process_group("nccl",
    device_id=...), barrier, all_gather_into_tensor of a device tensor) executes on a single-GPU box exactly as it will
    on the 8-GPU node (tests/test_gpu_dist.py); without it a single rank needs no group at all."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("LC_DIST_FORCE") == "1"
    if size <= 1 and not force:
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return World(0, 0, 1, None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if size <= 1:
        size, rank, local = 1, 0, 0
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
    if backend is None:
        # LC_DIST_BACKEND=gloo lets several ranks share ONE GPU (plumbing tests on a 1-GPU box)
        backend = os.environ.get("LC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        if backend == "nccl" and local >= ndev:
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible — the RCCL path is one process "
                               f"per GPU (LC_DIST_BACKEND=gloo lets ranks share a GPU for plumbing tests)")
        local = local % ndev if backend != "nccl" else local
        torch.cuda.set_device(local)
    # rendezvous: env:// (torch.distributed.run) or the file:// store spawn() prepared (no port to lose a race for)
    method = os.environ.get("LC_DIST_INIT", "env://")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", init_method=method, world_size=size, rank=rank,
                                device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend, init_method=method, world_size=size, rank=rank)
    return World(rank, local, size, backend, True)


def _dev(w: World):
    return torch.device("cuda", w.local_rank) if w.backend == "nccl" else torch.device("cpu")


def barrier(w: World):
    """barrier + device synchronize on both sides of a timed region."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if w.is_dist:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def gather_row(w: World, values) -> torch.Tensor:
    """all_gather of a small fp64 vector per rank -> [world, len(values)] (timings only)."""
    row = torch.tensor(list(values), dtype=torch.float64, device=_dev(w))
    if not w.is_dist:
        return row.unsqueeze(0).cpu()
    out = torch.empty(w.size * row.numel(), dtype=torch.float64, device=row.device)
    dist.all_gather_into_tensor(out, row)
    return out.view(w.size, -1).cpu()


def max_over_ranks(w: World, value: float) -> float:
    return float(gather_row(w, [value])[:, 0].max())


def _spawn_entry(rank: int, fn, args, world: int, store: str):
    # every rank sees every GPU and binds its own with set_device(LOCAL_RANK) (init): RCCL's xGMI peer-to-peer transport
    # wants the peers visible, so no per-rank HIP_VISIBLE_DEVICES mask is set
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "LC_DIST_INIT": store, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    fn(*args)


def spawn(fn, nprocs: int, args=()):
    """Run `fn(*args)` in `nprocs` fresh processes (one per GPU), each with the environment torch.distributed.run
    would have provided, except that the rendezvous is a file:// store in a private temp directory (a port picked here
    and handed to the children could be taken by another process in between).  `fn` must be a module-level function."""
    import tempfile

    import torch.multiprocessing as mp
    backend = os.environ.get("LC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl" and torch.cuda.is_available() and nprocs > torch.cuda.device_count():
        raise RuntimeError(f"spawn({nprocs} ranks) on {torch.cuda.device_count()} GPU(s): the RCCL path is one process per GPU; "
                           f"set LC_DIST_BACKEND=gloo to let ranks share a GPU (plumbing tests)")
    with tempfile.TemporaryDirectory(prefix="lc_dist_") as d:
        mp.spawn(_spawn_entry, args=(fn, tuple(args), nprocs, f"file://{d}/store"), nprocs=nprocs, join=True)


def shutdown(w: World):
    if w.is_dist and dist.is_initialized():
        dist.destroy_process_group()
