"""Multi-GPU plumbing: one process (rank) per GPU, RCCL over xGMI inside the C library.

New functionality -- the reference is single-GPU (no NCCL/MPI call site anywhere).
The data path is: every rank holds a shard of the points (or one frame of a multi-frame fit),
runs the same kernels, and the library all-reduces the per-cluster sufficient statistics
((7 J + 2) float64 for flat EM, 10 x 8^(l+1) for a tree level) on the context's stream before
the redundant, identical M-step.  This module only bootstraps the communicator: the 128-byte
RCCL unique id has to travel from rank 0 to the other ranks once.

Transports for that one message
  * ``torch.distributed`` (gloo, CPU) when the process was started by ``torch.distributed.run``
    -- the launcher the benchmark contract prescribes; torch is used for the rendezvous only;
  * a plain TCP exchange (rank 0 listens on MASTER_ADDR:MASTER_PORT+offset) otherwise.
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


def exchange_bytes_tcp(rank: int, world: int, payload: bytes | None, addr: str, port: int,
                       timeout: float = 120.0) -> bytes:
    """Rank 0 sends `payload` to every other rank over plain TCP."""
    if world == 1:
        return payload
    if rank == 0:
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(world)
        srv.settimeout(timeout)
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(struct.pack("<I", len(payload)) + payload)
        finally:
            srv.close()
        return payload
    deadline = time.time() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                hdr = _recv_exact(s, 4)
                return _recv_exact(s, struct.unpack("<I", hdr)[0])
        except (ConnectionRefusedError, socket.timeout, OSError):
            if time.time() > deadline:
                raise
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
        transport = "torch" if os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("GROUP_RANK") else "tcp"
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
