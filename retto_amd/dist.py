"""Multi-GPU plumbing: one process per GPU, pages sharded by index, weights broadcast
once.  The reference has no distributed code at all
(/root/reference/retto-cli/src/main.rs:80-86 is a serial map over files); every page
is independent (/root/reference/retto-core/src/session.rs:75-106 keeps no cross-image
state), so there is no data-path collective: the only exchange is the one-time
broadcast of the packed model blobs (RCCL over xGMI when the backend is "nccl", gloo in
the CPU tests) and an optional all-reduce of counters for reporting.
"""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence, Tuple


def shard_pages(sizes: Sequence[Tuple[int, int]], world: int, rank: int, est_lines: Optional[Sequence[int]] = None) -> List[int]:
    """Page indices this rank processes.  Greedy longest-processing-time assignment on a
    cost of det pixels (+ a per-line rec term when known); identical on every rank
    (pure function of the inputs).  Uniform pages degenerate to round-robin."""
    cost = []
    for i, (h, w) in enumerate(sizes):
        c = float(h) * float(w)
        if est_lines is not None:
            c += 0.135 * 960 * 960 * est_lines[i] / 1.0  # rec/det FLOP ratio per 320-wide line (1.405 / 10.36)
        cost.append(c)
    order = sorted(range(len(sizes)), key=lambda i: (-cost[i], i))
    load = [0.0] * world
    owner = [0] * len(sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cost[i]
    return [i for i in range(len(sizes)) if owner[i] == rank]


def broadcast_blobs(blobs: Optional[Sequence[bytes]], n_blobs: int, rank: int, device: str = "cpu", src: int = 0) -> List[bytes]:
    """Broadcasts ``n_blobs`` byte strings from ``src`` to every rank through
    torch.distributed (backend chosen by the caller: "nccl" = RCCL on ROCm, "gloo" on CPU)."""
    import torch
    import torch.distributed as dist
    sizes = torch.tensor([len(b) for b in blobs] if rank == src else [0] * n_blobs, dtype=torch.int64, device=device)
    dist.broadcast(sizes, src)
    out = []
    for i, n in enumerate(sizes.tolist()):
        t = torch.empty(n, dtype=torch.uint8, device=device)
        if rank == src:
            t.copy_(torch.frombuffer(bytearray(blobs[i]), dtype=torch.uint8))
        dist.broadcast(t, src)
        out.append(t.cpu().numpy().tobytes())
    return out


def digest(blobs: Sequence[bytes]) -> str:
    h = hashlib.sha256()
    for b in blobs:
        h.update(b)
    return h.hexdigest()
