"""Multi-GPU plumbing: one process (rank) per GPU, RCCL over xGMI inside the C library.

New functionality -- the reference is single-GPU (no NCCL/MPI call site anywhere).
The data path is: every rank holds a shard of the points (or one frame of a multi-frame fit),
runs the same kernels, and the library all-reduces the per-cluster sufficient statistics
((7 J + 2) float64 for flat EM, 10 x 8^(l+1) for a tree level) on the context's stream before
the redundant, identical M-step.  This module only bootstraps the communicator: the 128-byte
RCCL unique id has to travel from rank 0 to the other ranks once.

Transports for that one message
  * a plain TCP exchange (default; rank 0 listens on MASTER_ADDR:MASTER_PORT+offset) -- works the same
    under ``torch.distributed.run`` (which only has to provide RANK / WORLD_SIZE / MASTER_*), under
    ``bench.py``'s own launcher and under any other one-process-per-GPU launcher; no PyTorch involved;
  * ``torch.distributed`` (gloo, CPU) on request (``transport="torch"``), for callers that already
    hold a process group.
"""
from __future__ import annotations

import os
import socket
import struct
import time


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous partition of n points over `world` ranks (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_MAGIC = b"HGMMUID1"
PORT_OFFSETS = (37, 1037, 2037, 3037)       # tried in order when MASTER_PORT + offset is taken


def exchange_bytes_tcp(rank: int, world: int, payload: bytes | None, addr: str, port: int,
                       timeout: float = 120.0) -> bytes:
    """Rank 0 sends `payload` to every other rank over plain TCP (no third-party package).

    Rank 0 listens on the first free port of ``port + PORT_OFFSETS``; the other ranks try those ports in
    turn and accept only a peer that opens with the protocol's magic bytes, so a foreign service that
    happens to sit on one of them is skipped instead of being taken for rank 0."""
    if world == 1:
        return payload
    if rank == 0:
        srv = None
        for off in PORT_OFFSETS:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, port + off))
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise OSError("unique-id exchange: no free port among %s" % [port + o for o in PORT_OFFSETS])
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            served = 0
            while served < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(10.0)
                    try:
                        if _recv_exact(conn, len(_MAGIC)) != _MAGIC:
                            continue                      # not one of ours
                        conn.sendall(_MAGIC + struct.pack("<I", len(payload)) + payload)
                        served += 1
                    except (ConnectionError, socket.timeout, OSError):
                        continue
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    while True:
        for off in PORT_OFFSETS:
            try:
                with socket.create_connection((addr, port + off), timeout=5.0) as s:
                    s.settimeout(10.0)
                    s.sendall(_MAGIC)
                    if _recv_exact(s, len(_MAGIC)) != _MAGIC:
                        continue
                    hdr = _recv_exact(s, 4)
                    return _recv_exact(s, struct.unpack("<I", hdr)[0])
            except (ConnectionError, socket.timeout, OSError):
                continue
        if time.time() > deadline:
            raise TimeoutError("unique-id exchange: rank 0 not reachable on %s ports %s"
                               % (addr, [port + o for o in PORT_OFFSETS]))
        time.sleep(0.05)


def _recv_exact(s, n):
    buf = b""
    while len(buf) < n:
        chunk = s.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed during unique-id exchange")
        buf += chunk
    return buf


def exchange_bytes_torch(rank: int, world: int, payload: bytes | None) -> bytes:
    """Same, through torch.distributed's rendezvous (gloo backend, no GPU tensors)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    obj = [payload if rank == 0 else None]
    dist.broadcast_object_list(obj, src=0)
    return obj[0]


def broadcast_from_rank0(rank: int, world: int, payload: bytes | None, transport: str = "auto",
                         port_offset: int = 37) -> bytes:
    if world == 1:
        return payload
    if transport == "auto":
        transport = "tcp"
    if transport == "torch":
        return exchange_bytes_torch(rank, world, payload)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
    return exchange_bytes_tcp(rank, world, payload, addr, port)


def attach_communicator(ctx, rank: int | None = None, world: int | None = None, transport: str = "auto"):
    """Create the RCCL communicator for `ctx` (one call per rank, collective)."""
    r, _, w = env_rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    uid = type(ctx).comm_unique_id() if rank == 0 else None
    uid = broadcast_from_rank0(rank, world, uid, transport)
    ctx.comm_init(world, rank, uid)
    return ctx
