"""Multi-GPU plumbing: one process (rank) per GPU, RCCL over xGMI inside the C library.

New functionality -- the reference is single-GPU (no NCCL/MPI call site anywhere).
The data path is: every rank holds a shard of the points (or one frame of a multi-frame fit),
runs the same kernels, and the library all-reduces the per-cluster sufficient statistics
((7 J + 2) float64 for flat EM, 10 x 8^(l+1) for a tree level) on the context's stream before
the redundant, identical M-step.  This module only bootstraps the communicator: the 128-byte
RCCL unique id has to travel from rank 0 to the other ranks once.

Transport for that one message: a plain TCP exchange (rank 0 listens on MASTER_ADDR:MASTER_PORT+offset) --
works the same under ``torch.distributed.run`` (which only has to provide RANK / WORLD_SIZE / MASTER_*), under
``bench.py``'s own launcher and under any other one-process-per-GPU launcher.  No PyTorch anywhere in this
package (a caller that already holds a torch process group can pass its own ``exchange`` callable to
``attach_communicator``; tests/test_parallel_cpu.py shows one over gloo).
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


_MAGIC = b"HGMMUID2"
_ACK = b"HGMMACK2"
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
        served = 0
        try:
            while served < world - 1:
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError("unique-id exchange: only %d of %d ranks fetched the id from rank 0 within "
                                       "%.0f s (%s:%d)" % (served, world - 1, timeout, addr, srv.getsockname()[1]))
                with conn:
                    conn.settimeout(10.0)
                    try:
                        if _recv_exact(conn, len(_MAGIC)) != _MAGIC:
                            continue                      # not one of ours
                        conn.sendall(_MAGIC + struct.pack("<I", len(payload)) + payload)
                    except (ConnectionError, socket.timeout, OSError):
                        continue                          # it did not get the payload: it retries, the listener stays
                    # The payload is out.  The client confirms it and returns at once, without ever retrying after its
                    # ACK -- so a lost ACK or a connection closed behind a full send is a served rank too (counting only
                    # ACKs left rank 0 waiting out its timeout for a rank that already held the id); only an explicit
                    # non-ACK reply means "not taken".
                    try:
                        reply = _recv_exact(conn, len(_ACK))
                    except (ConnectionError, socket.timeout, OSError):
                        reply = _ACK
                    if reply == _ACK:
                        served += 1
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
                    data = _recv_exact(s, struct.unpack("<I", hdr)[0])
                    s.sendall(_ACK)
                    return data
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


def broadcast_from_rank0(rank: int, world: int, payload: bytes | None, transport: str = "tcp",
                         port_offset: int = 37, exchange=None) -> bytes:
    """``exchange(rank, world, payload) -> bytes``: caller-provided transport (e.g. over an existing process group)."""
    if world == 1:
        return payload
    if exchange is not None:
        return exchange(rank, world, payload)
    if transport not in ("auto", "tcp"):
        raise ValueError("unknown transport %r (this package ships the TCP exchange only; pass exchange=...)" % transport)
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
    return exchange_bytes_tcp(rank, world, payload, addr, port)


def allgather_bytes_tcp(rank: int, world: int, payload: bytes, port_offset: int = 137, timeout: float = 120.0):
    """Every rank's (short) ``payload`` on every rank, in rank order: rank 0 collects them over plain TCP and sends
    the list back.  Used to AGREE on decisions that must be collective (bench.py: which all-reduce backend to use
    after a communicator could not be built on some rank)."""
    if world == 1:
        return [payload]
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
    offsets = tuple(o - PORT_OFFSETS[0] for o in PORT_OFFSETS)       # the same fall-back ladder as the id exchange
    if rank == 0:
        srv = None
        for off in offsets:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, port + off))
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise OSError("all-gather: no free port among %s" % [port + o for o in offsets])
        srv.listen(world)
        srv.settimeout(timeout)
        got = {0: payload}
        conns = []
        try:
            while len(got) < world:
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError("all-gather: %d of %d ranks reported within %.0f s" % (len(got), world, timeout))
                conn.settimeout(10.0)
                try:
                    if _recv_exact(conn, len(_MAGIC)) != _MAGIC:
                        conn.close()
                        continue
                    r, n = struct.unpack("<II", _recv_exact(conn, 8))
                    if not (0 < r < world) or n > (1 << 20):          # not a rank of this job
                        conn.close()
                        continue
                    got[r] = _recv_exact(conn, n)
                    conns.append(conn)
                except (ConnectionError, socket.timeout, OSError):
                    conn.close()
            out = [got[r] for r in range(world)]
            blob = b"".join(struct.pack("<I", len(b)) + b for b in out)
            for conn in conns:
                conn.sendall(struct.pack("<I", len(blob)) + blob)
            return out
        finally:
            for conn in conns:
                conn.close()
            srv.close()
    if not (0 < rank < world):
        raise ValueError("all-gather: rank %d outside 0..%d" % (rank, world - 1))
    deadline = time.time() + timeout
    while True:
        for off in offsets:
            try:
                with socket.create_connection((addr, port + off), timeout=5.0) as s:
                    s.settimeout(timeout)
                    s.sendall(_MAGIC + struct.pack("<II", rank, len(payload)) + payload)
                    blob = _recv_exact(s, struct.unpack("<I", _recv_exact(s, 4))[0])
                    out, at = [], 0
                    while at < len(blob):
                        n = struct.unpack("<I", blob[at:at + 4])[0]
                        out.append(blob[at + 4:at + 4 + n])
                        at += 4 + n
                    if len(out) == world:
                        return out
            except (ConnectionError, socket.timeout, OSError, struct.error):
                continue
        if time.time() > deadline:
            raise TimeoutError("all-gather: rank 0 not reachable on %s ports %s" % (addr, [port + o for o in offsets]))
        time.sleep(0.05)


def attach_communicator(ctx, rank: int | None = None, world: int | None = None, transport: str = "tcp",
                        exchange=None):
    """Create the RCCL communicator for `ctx` (one call per rank, collective)."""
    r, _, w = env_rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    uid = type(ctx).comm_unique_id() if rank == 0 else None
    uid = broadcast_from_rank0(rank, world, uid, transport, exchange=exchange)
    ctx.comm_init(world, rank, uid)
    return ctx


class TcpGroup:
    """The ranks of one job joined by PERSISTENT plain-TCP connections to rank 0 (a star): ``allgather`` / ``barrier`` /
    ``max`` for jobs whose ranks do not share a communicator -- the replica mode, where every GPU registers its own scan
    pairs and the only thing the ranks ever exchange is "ready" and a wall-clock figure (bench.py --mode pairs).
    One round trip is ~0.1 ms on a loopback; nothing here touches the GPU."""

    def __init__(self, rank: int | None = None, world: int | None = None, port_offset: int = 237, timeout: float = 120.0):
        r, _, w = env_rank_world()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.timeout = timeout
        self.conns = {}                    # rank 0: peer rank -> socket
        self.sock = None                   # other ranks: the socket to rank 0
        if self.world == 1:
            return
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("MASTER_PORT", "29500")) + port_offset
        offsets = tuple(o - PORT_OFFSETS[0] for o in PORT_OFFSETS)
        if self.rank == 0:
            srv = None
            for off in offsets:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    s.bind((addr, port + off))
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise OSError("TcpGroup: no free port among %s" % [port + o for o in offsets])
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self.conns) < self.world - 1:
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError("TcpGroup: %d of %d ranks joined within %.0f s"
                                           % (len(self.conns) + 1, self.world, timeout))
                    conn.settimeout(10.0)
                    try:
                        if _recv_exact(conn, len(_MAGIC)) != _MAGIC:
                            conn.close()
                            continue
                        peer = struct.unpack("<I", _recv_exact(conn, 4))[0]
                        if not (0 < peer < self.world) or peer in self.conns:
                            conn.close()
                            continue
                        conn.sendall(_ACK)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        conn.settimeout(timeout)
                        self.conns[peer] = conn
                    except (ConnectionError, socket.timeout, OSError):
                        conn.close()
            finally:
                srv.close()
            return
        deadline = time.time() + timeout
        while self.sock is None:
            for off in offsets:
                try:
                    s = socket.create_connection((addr, port + off), timeout=5.0)
                    s.settimeout(10.0)
                    s.sendall(_MAGIC + struct.pack("<I", self.rank))
                    if _recv_exact(s, len(_ACK)) == _ACK:
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(timeout)
                        self.sock = s
                        break
                    s.close()
                except (ConnectionError, socket.timeout, OSError):
                    continue
            if self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError("TcpGroup: rank 0 not reachable on %s ports %s" % (addr, [port + o for o in offsets]))
                time.sleep(0.05)

    def allgather(self, payload: bytes):
        """Every rank's payload on every rank, in rank order."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            got = [payload] + [None] * (self.world - 1)
            for peer, conn in self.conns.items():
                n = struct.unpack("<I", _recv_exact(conn, 4))[0]
                got[peer] = _recv_exact(conn, n)
            blob = b"".join(struct.pack("<I", len(b)) + b for b in got)
            for conn in self.conns.values():
                conn.sendall(struct.pack("<I", len(blob)) + blob)
            return got
        self.sock.sendall(struct.pack("<I", len(payload)) + payload)
        blob = _recv_exact(self.sock, struct.unpack("<I", _recv_exact(self.sock, 4))[0])
        out, at = [], 0
        while at < len(blob):
            n = struct.unpack("<I", blob[at:at + 4])[0]
            out.append(blob[at + 4:at + 4 + n])
            at += 4 + n
        return out

    def barrier(self):
        self.allgather(b"")

    def allgather_f64(self, values):
        """[world, len(values)] float64."""
        import numpy as np
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        return np.stack([np.frombuffer(b, dtype=np.float64) for b in self.allgather(v.tobytes())])

    def close(self):
        for conn in self.conns.values():
            conn.close()
        self.conns = {}
        if self.sock is not None:
            self.sock.close()
            self.sock = None
