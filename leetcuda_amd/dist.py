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

    @property
    def is_dist(self) -> bool:
        return self.size > 1


def init(backend: str | None = None) -> World:
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if size <= 1:
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return World(0, 0, 1, None)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        # LC_DIST_BACKEND=gloo lets several ranks share ONE GPU (plumbing tests on a 1-GPU box)
        backend = os.environ.get("LC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count() if backend != "nccl" else local
        torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group(backend="nccl", init_method="env://", world_size=size, rank=rank,
                                device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend=backend, init_method="env://", world_size=size, rank=rank)
    return World(rank, local, size, backend)


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


def _spawn_entry(rank: int, fn, args, world: int, port: int):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    fn(*args)


def spawn(fn, nprocs: int, args=()):
    """Run `fn(*args)` in `nprocs` fresh processes (one per GPU), each with the environment torch.distributed.run
    would have provided (rendezvous on 127.0.0.1, a free port).  `fn` must be a module-level function."""
    import socket

    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_spawn_entry, args=(fn, tuple(args), nprocs, port), nprocs=nprocs, join=True)


def shutdown(w: World):
    if w.is_dist and dist.is_initialized():
        dist.destroy_process_group()
