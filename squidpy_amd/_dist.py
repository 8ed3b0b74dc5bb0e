"""Multi-GPU plumbing: one process per GPU, the data-path collective is RCCL *inside libsqgr*.

The hot path shards by *permutation range* (nhood, autocorr, ligrec), *row tiles* (co-occurrence), *feature blocks*
(autocorr) or *clusters* (Ripley); every rank holds the full (small) inputs, so the only exchange is one all-reduce
of exact 64-bit integer accumulators (< 1 MB, latency-bound): ``sqgr_comm_allreduce_i64`` / the on-device all-reduce
of ``sqgr_nhood_run`` (``include/sqgr.h``, RCCL over xGMI).  What the ranks need from the host side is only a side
channel that (i) carries the 128-byte ``ncclUniqueId`` from rank 0 to the others and (ii) moves a few host objects
(a seed, gathered result blocks).  Two side channels, no torch in the data path:

* ``SocketGroup`` — a star of TCP connections on one node built from the launcher's environment (``RANK``,
  ``WORLD_SIZE``, ``MASTER_ADDR``, ``MASTER_PORT``: what ``python -m torch.distributed.run`` or any other launcher
  exports).  Pure Python, torch is never imported.  ``init()`` creates it.
* ``TorchGroup`` — adopts a ``torch.distributed`` process group the caller has already initialised (any backend).
  With backend ``gloo`` (CPU tests, several ranks sharing one GPU) the integer all-reduce runs on the host through
  that group, because RCCL refuses two ranks on one device.

``is_distributed()`` is true once a group with more than one rank exists: after ``init()`` or automatically when a
torch process group is already initialised (also one created programmatically without torchrun's environment)."""

from __future__ import annotations

import hmac
import io
import os
import secrets
import socket
import stat
import struct
import sys
import tempfile
import time
from typing import Any

import numpy as np

_TIMEOUT_S = float(os.environ.get("SQGR_DIST_TIMEOUT", "120"))
_HELLO_TIMEOUT_S = 3.0  # the hello of a connecting rank (rank, world, secret) follows its connect at once


# ----------------------------------------------------------------------------------------------- side channels
def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    chunks, got = [], 0
    while got < n:
        b = sock.recv(min(n - got, 1 << 20))
        if not b:
            raise ConnectionError("peer closed the rendezvous connection")
        chunks.append(b)
        got += len(b)
    return b"".join(chunks)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


# Host objects that cross the side channel (a seed, result blocks, a timing) are encoded by a small DATA-ONLY codec — no
# pickle: nothing a peer sends is ever executed.  None / bool / int / float / str / bytes / numpy arrays (the .npy format,
# allow_pickle=False) / list / tuple / dict of those.
def _enc(obj: Any, out: io.BytesIO) -> None:
    if obj is None:
        out.write(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.write(b"T" if obj else b"f")
    elif isinstance(obj, (int, np.integer)):
        b = str(int(obj)).encode()
        out.write(b"I" + struct.pack("<I", len(b)) + b)
    elif isinstance(obj, (float, np.floating)):
        out.write(b"F" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode()
        out.write(b"S" + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray)):
        out.write(b"B" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError("object arrays do not cross the side channel")
        buf = io.BytesIO()
        np.lib.format.write_array(buf, obj, allow_pickle=False)
        out.write(b"A" + struct.pack("<Q", buf.tell()) + buf.getvalue())
    elif isinstance(obj, (list, tuple)):
        out.write((b"L" if isinstance(obj, list) else b"U") + struct.pack("<Q", len(obj)))
        for item in obj:
            _enc(item, out)
    elif isinstance(obj, dict):
        out.write(b"D" + struct.pack("<Q", len(obj)))
        for k, v in obj.items():
            _enc(k, out)
            _enc(v, out)
    else:
        raise TypeError(f"cannot send a {type(obj).__name__} through the side channel")


def _dec(buf: io.BytesIO) -> Any:
    tag = buf.read(1)
    if tag == b"N":
        return None
    if tag in (b"T", b"f"):
        return tag == b"T"
    if tag == b"I":
        (n,) = struct.unpack("<I", buf.read(4))
        return int(buf.read(n).decode())
    if tag == b"F":
        return struct.unpack("<d", buf.read(8))[0]
    if tag in (b"S", b"B", b"A"):
        (n,) = struct.unpack("<Q", buf.read(8))
        raw = buf.read(n)
        if len(raw) != n:
            raise ValueError("truncated side-channel message")
        if tag == b"S":
            return raw.decode()
        if tag == b"B":
            return raw
        return np.lib.format.read_array(io.BytesIO(raw), allow_pickle=False)
    if tag in (b"L", b"U"):
        (n,) = struct.unpack("<Q", buf.read(8))
        items = [_dec(buf) for _ in range(n)]
        return items if tag == b"L" else tuple(items)
    if tag == b"D":
        (n,) = struct.unpack("<Q", buf.read(8))
        return {_dec(buf): _dec(buf) for _ in range(n)}
    raise ValueError(f"unknown tag {tag!r} in a side-channel message")


def encode_object(obj: Any) -> bytes:
    out = io.BytesIO()
    _enc(obj, out)
    return out.getvalue()


def decode_object(payload: bytes) -> Any:
    return _dec(io.BytesIO(payload))


def _rendezvous_dir() -> str:
    """A directory only this user can enter (0700, owned by us, not a symlink): the rendezvous file and the group's secret
    live there, so another local user can neither pre-create the file (redirecting ranks to a server of theirs) nor read it."""
    path = os.path.join(tempfile.gettempdir(), f"sqgr-{os.getuid()}")
    try:
        os.mkdir(path, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(path)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"{path} must be a directory owned by uid {os.getuid()} with mode 0700 (found mode {oct(st.st_mode & 0o7777)}, uid {st.st_uid})")
    return path


class SocketGroup:
    """All ranks of ONE node connected to rank 0 over loopback TCP.

    Rank 0 listens on an ephemeral loopback port and publishes ``port`` and a fresh 32-byte secret in
    ``$TMPDIR/sqgr-<uid>/rdzv_<MASTER_PORT>_<launcher pid>`` (a 0700 directory, a 0600 file created exclusively;
    ``MASTER_PORT`` itself belongs to the launcher's own store); the others poll that file, connect and present the secret —
    a peer without it is dropped.  Every collective is an all-gather of byte strings through the hub, framed by lengths;
    nothing received is unpickled or otherwise executed.  The payloads on this path are a unique id, a seed, a few result blocks."""

    kind = "socket"

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", key: str | None = None):
        # `addr` is accepted for the launcher's MASTER_ADDR and ignored: the group is one node, the hub binds and the peers dial
        # 127.0.0.1 only (a routable listener was never offered; a switch that pretended to offer one is gone)
        self.rank, self.world = int(rank), int(world)
        if addr not in ("127.0.0.1", "localhost", "::1", "") and addr != socket.gethostname():
            import warnings

            warnings.warn(f"squidpy_amd: MASTER_ADDR={addr!r} is ignored — the socket rendezvous connects the ranks of ONE node over 127.0.0.1 "
                          "(ranks on other hosts cannot join; use a torch.distributed group for that)", RuntimeWarning, stacklevel=2)
        key = key or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        self._path = os.path.join(_rendezvous_dir(), f"rdzv_{key}")
        self._peers: list[socket.socket] = []
        self._hub: socket.socket | None = None
        deadline = time.monotonic() + _TIMEOUT_S
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            srv.settimeout(_TIMEOUT_S)
            token = secrets.token_bytes(32)
            try:
                os.remove(self._path)  # a stale file of an earlier launch with the same key (ours: the directory is private)
            except OSError:
                pass
            fd = os.open(self._path, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as fh:
                fh.write(f"{srv.getsockname()[1]} {token.hex()}")
            by_rank: dict[int, socket.socket] = {}
            try:
                while len(by_rank) < self.world - 1:
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: {self.world - 1 - len(by_rank)} of {self.world - 1} peers missing after {_TIMEOUT_S:.0f} s")
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    # the 40-byte hello of a genuine rank is already on the wire when accept() returns: a local client that
                    # connects and stays silent costs the accept loop this long and no more (it used to hold it for _TIMEOUT_S)
                    conn.settimeout(_HELLO_TIMEOUT_S)
                    try:
                        r, w = struct.unpack("<ii", _recv_exact(conn, 8))
                        ok = hmac.compare_digest(_recv_exact(conn, 32), token)
                        conn.settimeout(_TIMEOUT_S)
                    except (ConnectionError, OSError, struct.error):
                        conn.close()
                        continue
                    if not ok:  # not one of ours: drop it and keep waiting for the real ranks
                        conn.close()
                        continue
                    if w != self.world or not 0 < r < self.world:
                        conn.close()
                        raise RuntimeError(f"rendezvous: unexpected peer (rank {r} of {w}) for a {self.world}-rank group")
                    try:
                        conn.sendall(b"\x01")  # the rank keeps dialling until it has read this: a hello that came too late is re-sent
                        # ... and confirms that it has: a rank that gave up waiting for this byte (the hub was busy with silent
                        # connections in front of it), closed and dialled again leaves a buffered hello on a dead socket — the write
                        # above "succeeds" on it, the read below does not, and the dead connection is never registered
                        conn.settimeout(_HELLO_TIMEOUT_S)
                        if _recv_exact(conn, 1) != b"\x02":
                            raise ConnectionError("no confirmation")
                        conn.settimeout(_TIMEOUT_S)
                    except (ConnectionError, OSError):
                        conn.close()
                        continue
                    # the same rank twice all the same (a redial that crossed a confirmation): the newest dial wins
                    if r in by_rank:
                        try:
                            by_rank[r].close()
                        except OSError:
                            pass
                    by_rank[r] = conn
            finally:
                srv.close()
                try:
                    os.remove(self._path)
                except OSError:
                    pass
            self._peers = [by_rank[r] for r in range(1, self.world)]
        else:
            last: Exception | None = None
            while time.monotonic() < deadline:
                try:
                    with open(self._path) as fh:
                        port, token_hex = fh.read().split()
                    s = socket.create_connection(("127.0.0.1", int(port)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(_TIMEOUT_S)
                    s.sendall(struct.pack("<ii", self.rank, self.world) + bytes.fromhex(token_hex))
                    # the hub acknowledges a hello it has accepted; one it gave up on (it waits _HELLO_TIMEOUT_S per connection) closes
                    # the socket instead, and this rank dials again — it used to keep the dead socket and fail later in barrier()
                    s.settimeout(2.0 * _HELLO_TIMEOUT_S + 5.0)
                    if _recv_exact(s, 1) != b"\x01":
                        raise ConnectionError("rendezvous: the hub did not acknowledge the hello")
                    s.sendall(b"\x02")  # confirmation: this socket is the one the rank keeps
                    s.settimeout(_TIMEOUT_S)
                    self._hub = s
                    break
                except (OSError, ValueError) as exc:  # file not there yet / stale port of an earlier run / hello not acknowledged
                    last = exc
                    try:
                        s.close()
                    except (NameError, OSError):
                        pass
                    time.sleep(0.05)
            if self._hub is None:
                raise TimeoutError(f"rank {self.rank}: no rendezvous with rank 0 within {_TIMEOUT_S:.0f} s ({last!r})")
        self.barrier()

    def allgather_bytes(self, payload: bytes) -> list[bytes]:
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(p) for p in self._peers]
            blob = struct.pack("<Q", len(parts)) + b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for p in self._peers:
                _send_msg(p, blob)
            return parts
        assert self._hub is not None
        _send_msg(self._hub, payload)
        blob = _recv_msg(self._hub)
        (count,) = struct.unpack_from("<Q", blob, 0)
        if count != self.world:
            raise ValueError("malformed all-gather reply from the hub")
        parts, off = [], 8
        for _ in range(count):
            (n,) = struct.unpack_from("<Q", blob, off)
            off += 8
            parts.append(blob[off : off + n])
            off += n
        return parts

    def barrier(self) -> None:
        self.allgather_bytes(b"")

    def close(self) -> None:
        for s in self._peers + ([self._hub] if self._hub is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._hub = [], None


class TorchGroup:
    """An already initialised ``torch.distributed`` default group, used as the side channel."""

    kind = "torch"

    def __init__(self) -> None:
        import torch.distributed as dist

        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()

    def allgather_bytes(self, payload: bytes) -> list[bytes]:
        objs: list[Any] = [None] * self.world
        self._dist.all_gather_object(objs, payload)
        return objs

    def barrier(self) -> None:
        self._dist.barrier()

    def allreduce_i64_host(self, flat: np.ndarray) -> np.ndarray:
        import torch

        t = torch.from_numpy(flat.copy())
        if self.backend == "nccl":
            t = t.cuda(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def close(self) -> None:
        pass


# ----------------------------------------------------------------------------------------------- module state
_group: SocketGroup | TorchGroup | None = None
_comm: Any = None          # squidpy_amd._lib.Comm once created
_comm_tried = False
_collective = "none"      # what allreduce_sum_ last used: "rccl-in-library" | "torch.distributed" | "socket-hub"


def _torch_group_ready() -> bool:
    if "torch" not in sys.modules and int(os.environ.get("WORLD_SIZE", "1")) <= 1 and os.environ.get("SQGR_DIST_FORCE") != "1":
        return False  # do not import torch just to find out that nothing was initialised
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return False
    return dist.is_available() and dist.is_initialized()


def init(method: str = "auto") -> None:
    """Join the process group of this launch.  ``"auto"``: adopt an initialised ``torch.distributed`` group if there
    is one, else build the socket rendezvous from ``RANK`` / ``WORLD_SIZE`` / ``MASTER_ADDR`` / ``MASTER_PORT``
    (``"socket"`` forces the latter).  A no-op for a single process."""
    global _group
    if _group is not None:
        return
    if method == "auto" and _torch_group_ready():
        _group = TorchGroup()
        return
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size <= 1:
        return
    _group = SocketGroup(int(os.environ["RANK"]), world_size, os.environ.get("MASTER_ADDR", "127.0.0.1"))


def shutdown() -> None:
    global _group, _comm, _comm_tried
    if _comm is not None:
        _comm.close()
    if _group is not None:
        _group.close()
    _group, _comm, _comm_tried = None, None, False


def group() -> SocketGroup | TorchGroup | None:
    global _group
    if _group is not None and isinstance(_group, TorchGroup) and not _torch_group_ready():
        _group = None  # the caller destroyed its process group
    if _group is None and _torch_group_ready():
        _group = TorchGroup()
    return _group


def is_distributed() -> bool:
    g = group()
    if g is None:
        return False
    return g.world > 1 or os.environ.get("SQGR_DIST_FORCE") == "1"  # FORCE: exercise the collective path with one rank


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) without a process group."""
    if not is_distributed():
        return 0, 1
    g = group()
    assert g is not None
    return g.rank, g.world


def shard_range(n: int, rank: int, world_size: int, begin: int = 0) -> tuple[int, int]:
    """Contiguous split of [begin, begin+n) (same chunking as ``np.array_split`` / the reference's
    contiguous per-job chunks, /root/reference/src/squidpy/_utils.py:223-231)."""
    base, rem = divmod(n, world_size)
    lo = begin + rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def device_comm() -> Any:
    """The RCCL communicator libsqgr owns for this group (created on first use: rank 0 draws the unique id, the side
    channel carries it), or ``None`` when the device collective is unavailable — a gloo group (several ranks may
    share a GPU), ``SQGR_DIST_COLLECTIVE=host``, or a creation failure on any rank (all ranks then agree on the host
    path)."""
    global _comm, _comm_tried
    if _comm is not None or _comm_tried:
        return _comm
    g = group()
    if g is None:
        return None
    _comm_tried = True
    if os.environ.get("SQGR_DIST_COLLECTIVE", "") == "host" or (isinstance(g, TorchGroup) and g.backend != "nccl"):
        return None
    from . import _lib

    uid, err = b"", ""
    if g.rank == 0:
        try:
            uid = _lib.comm_unique_id()
        except Exception as exc:  # RCCL missing
            err = repr(exc)
    uid = g.allgather_bytes(uid)[0]
    comm = None
    if len(uid) == _lib.UNIQUE_ID_BYTES:
        try:
            comm = _lib.Comm(_lib.default_context(), uid, g.rank, g.world)
        except Exception as exc:
            err = repr(exc)
    oks = g.allgather_bytes(b"1" if comm is not None else b"0")
    if all(o == b"1" for o in oks):
        _comm = comm
    else:
        if comm is not None:
            comm.close()
        if g.rank == 0:
            print(f"squidpy_amd: RCCL communicator unavailable ({err or 'a peer failed'}); integer all-reduce falls back to the host side channel", file=sys.stderr)
    return _comm


def collective_kind() -> str:
    """Which implementation the last ``allreduce_sum_`` used (reported by bench.py)."""
    return _collective


def allreduce_sum_(arrays: list[np.ndarray]) -> list[np.ndarray]:
    """Exact all-reduce(sum) of 64-bit integer arrays (uint64 is summed modulo 2**64 via its int64 view).
    No-op without a process group."""
    global _collective
    if not is_distributed():
        return arrays
    g = group()
    assert g is not None
    flat = np.concatenate([np.ascontiguousarray(a).view(np.int64).reshape(-1) for a in arrays])
    comm = device_comm()
    if comm is not None:
        flat = comm.allreduce_i64(np.ascontiguousarray(flat))
        _collective = "rccl-in-library"
    elif isinstance(g, TorchGroup):
        flat = g.allreduce_i64_host(flat)
        _collective = "torch.distributed"
    else:
        parts = g.allgather_bytes(flat.tobytes())
        flat = np.sum([np.frombuffer(p, dtype=np.int64) for p in parts], axis=0, dtype=np.int64)
        _collective = "socket-hub"
    out, off = [], 0
    for a in arrays:
        out.append(flat[off : off + a.size].view(a.dtype).reshape(a.shape).copy())
        off += a.size
    return out


def allgather_object(a: Any) -> list[Any]:
    """One copy of ``a`` from every rank, in rank order (autocorr: feature blocks are gathered, not reduced)."""
    if not is_distributed():
        return [a]
    g = group()
    assert g is not None
    return [decode_object(b) for b in g.allgather_bytes(encode_object(a))]


def broadcast_object(a: Any, src: int = 0) -> Any:
    """``a`` of rank ``src`` on every rank."""
    return allgather_object(a)[src] if is_distributed() else a


def barrier() -> None:
    g = group()
    if g is not None:
        g.barrier()
