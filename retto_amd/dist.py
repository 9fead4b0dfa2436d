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


def process_shard(mine: Sequence[int], process, chunk: int = 32) -> list:
    """Runs ``process(page_ids) -> list of per-page results`` over this rank's shard in calls of at most ``chunk`` pages (input
    order inside the shard); returns [(page id, result)].  The per-rank half of ``run_global_batch``: bench.py's timed steps
    call it directly (no collective in the data path), the gather below is only taken when every rank needs all results."""
    out = []
    for c0 in range(0, len(mine), chunk):
        ids = list(mine[c0:c0 + chunk])
        res = process(ids)
        if len(res) != len(ids):
            raise RuntimeError("process() must return one result per page")
        out += list(zip(ids, res))
    return out


def run_global_batch(sizes: Sequence[Tuple[int, int]], rank: int, world: int, process, est_lines: Optional[Sequence[int]] = None,
                     chunk: int = 32) -> list:
    """The sharded form of retto-cli's loop over files (/root/reference/retto-cli/src/main.rs:80-86) for ONE list of pages:
    every rank takes its LPT shard (``shard_pages``), runs ``process(page_ids) -> list of per-page results`` over it in calls
    of at most ``chunk`` pages (input order inside the shard), and the per-page results of all ranks are gathered and returned
    in INPUT order on every rank (``torch.distributed.all_gather_object`` of small result records; no tensor collective)."""
    mine = sorted(shard_pages(sizes, world, rank, est_lines))
    out = process_shard(mine, process, chunk)
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, out)
        out = [x for part in parts for x in part]
    out.sort(key=lambda t: t[0])
    if [i for i, _ in out] != list(range(len(sizes))):
        raise RuntimeError("gather lost or duplicated pages")
    return [r for _, r in out]


def broadcast_blobs_cabi(blobs: Optional[Sequence[bytes]], n_blobs: int, rank: int, world: int, device_id: int, uid: bytes,
                         root: int = 0) -> List[bytes]:
    """The same one-time broadcast through libretto_hip's C ABI (rt_broadcast_blobs: RCCL, no torch involved) -- what a
    Rust / C++ host would call.  ``uid``: the 128 bytes of ``rccl_unique_id()`` made on ``root`` and shared out of band."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    data = (C.c_void_p * max(n_blobs, 1))()
    lens = (C.c_size_t * max(n_blobs, 1))()
    keep = []
    if rank == root:
        for i, b in enumerate(blobs):
            buf = C.create_string_buffer(bytes(b), len(b)); keep.append(buf)
            data[i] = C.addressof(buf); lens[i] = len(b)
    err = C.create_string_buffer(512)
    lib.rt_broadcast_blobs.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
    rc = lib.rt_broadcast_blobs(uid, rank, world, device_id, root, n_blobs, data, lens, err, 512)
    if rc != 0:
        raise RuntimeError("rt_broadcast_blobs failed (%d): %s" % (rc, err.value.decode("utf-8", "replace")))
    if rank == root:
        return [bytes(b) for b in blobs]
    out = []
    lib.rt_buffer_free.argtypes = [C.c_void_p]
    for i in range(n_blobs):
        out.append(C.string_at(data[i], lens[i]))
        lib.rt_buffer_free(data[i])
    return out


def rccl_unique_id() -> bytes:
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    uid = C.create_string_buffer(128)
    err = C.create_string_buffer(256)
    lib.rt_rccl_unique_id.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    rc = lib.rt_rccl_unique_id(uid, 128, err, 256)
    if rc != 0:
        raise RuntimeError("rt_rccl_unique_id failed: " + err.value.decode("utf-8", "replace"))
    return uid.raw
