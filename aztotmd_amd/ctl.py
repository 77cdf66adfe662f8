"""Control plane of a multi-process run (one process per GPU) WITHOUT torch: a handful of small collectives over plain TCP.

Why not torch.distributed: the rank processes bring up RCCL inside libaztot (system ROCm runtime); importing torch into the same
process puts the HIP/HSA copies bundled with its wheel in front of it, and RCCL then reports "no ROCm-capable device".  The
launcher (bench.py's own, or python -m torch.distributed.run) is a process of its own; the ranks only read RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT from the environment it sets.  Nothing here is on the data path: the halo travels over RCCL
(exchange.cpp); this carries the RCCL unique id, barriers and a few scalars.

Rank 0 listens on MASTER_PORT + AZTOT_CTL_PORT_OFFSET (default 29); every collective is a gather to rank 0 followed by a reply.

Wire format (nothing received from the network is ever unpickled or evaluated): a frame is
    magic "AZC1" | u32 header bytes | u32 blobs | JSON header | per blob: u64 bytes + raw data
The JSON header is the value with every bytes object / numpy array replaced by a reference to a blob ({"__b": i} /
{"__a": i, "dtype": "<f8", "shape": [...]}; tuples travel as lists).  Only plain data types are representable, and array dtypes are
restricted to fixed-size numeric kinds.
Handshake: a peer's first frame is {"rank": r, "token": t}.  Rank 0 accepts it only if 1 <= r < world, r is not connected yet and the
token equals its own (constant-time comparison); anything else is dropped and rank 0 keeps listening.  The token is AZTOT_CTL_TOKEN
when the launcher set one (bench.py's launcher draws 128 random bits per run), else a digest of what every rank of one launch shares
(master address and port, world size, the launcher's run id).  Rank 0 binds to the loopback interface unless MASTER_ADDR names another
one (single-node runs, which is all this package launches, never leave 127.0.0.1).
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import time

import numpy as np

MAGIC = b"AZC1"
MAX_HEADER = 1 << 24          # 16 MiB of JSON
MAX_BLOBS = 1 << 16
MAX_BLOB = 1 << 32            # 4 GiB per blob (host-staged halos of the test transport are a few MB)
_DTYPE_KINDS = "biuf"         # bool, signed / unsigned integers, floats: nothing with object references


class ProtocolError(ConnectionError):
    pass


def _pack(obj, blobs):
    if obj is None or isinstance(obj, (bool, str)):
        return obj
    if isinstance(obj, (int, np.integer)):
        return int(obj)
    if isinstance(obj, (float, np.floating)):
        return float(obj)
    if isinstance(obj, (bytes, bytearray, memoryview)):
        blobs.append(bytes(obj))
        return {"__b": len(blobs) - 1}
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind not in _DTYPE_KINDS:
            raise TypeError("control plane: arrays of dtype %s do not travel" % obj.dtype)
        a = np.ascontiguousarray(obj)
        blobs.append(a.tobytes())
        return {"__a": len(blobs) - 1, "dtype": a.dtype.str, "shape": list(a.shape)}
    if isinstance(obj, (list, tuple)):
        return [_pack(v, blobs) for v in obj]
    if isinstance(obj, dict):
        if any(not isinstance(k, str) or k.startswith("__") for k in obj):
            raise TypeError("control plane: dictionary keys must be plain strings")
        return {k: _pack(v, blobs) for k, v in obj.items()}
    raise TypeError("control plane: %s does not travel" % type(obj).__name__)


def _unpack(obj, blobs):
    if isinstance(obj, list):
        return [_unpack(v, blobs) for v in obj]
    if isinstance(obj, dict):
        if "__b" in obj:
            return _blob(obj["__b"], blobs)
        if "__a" in obj:
            dt = np.dtype(str(obj["dtype"]))
            if dt.kind not in _DTYPE_KINDS:
                raise ProtocolError("control plane: refused array dtype %r" % (obj["dtype"],))
            shape = tuple(int(v) for v in obj["shape"])
            raw = _blob(obj["__a"], blobs)
            if int(np.prod(shape, dtype=np.int64)) * dt.itemsize != len(raw) or any(v < 0 for v in shape):
                raise ProtocolError("control plane: array shape does not match its payload")
            return np.frombuffer(raw, dtype=dt).reshape(shape).copy()
        return {str(k): _unpack(v, blobs) for k, v in obj.items()}
    return obj


def _blob(i, blobs):
    if not isinstance(i, int) or not 0 <= i < len(blobs):
        raise ProtocolError("control plane: reference to a blob that is not there")
    return blobs[i]


def encode(obj):
    """bytes of one frame"""
    blobs = []
    hdr = json.dumps(_pack(obj, blobs), allow_nan=True, separators=(",", ":")).encode()
    parts = [MAGIC, struct.pack("<II", len(hdr), len(blobs)), hdr]
    for b in blobs:
        parts.append(struct.pack("<Q", len(b)))
        parts.append(b)
    return b"".join(parts)


def _read(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send(sock, obj):
    sock.sendall(encode(obj))


def _recv(sock):
    head = _read(sock, 12)
    if head[:4] != MAGIC:
        raise ProtocolError("control plane: not a frame of this protocol")
    nh, nb = struct.unpack("<II", head[4:])
    if nh > MAX_HEADER or nb > MAX_BLOBS:
        raise ProtocolError("control plane: oversized frame refused")
    try:
        hdr = json.loads(_read(sock, nh).decode())
    except (UnicodeDecodeError, ValueError) as ex:
        raise ProtocolError("control plane: malformed frame header") from ex
    blobs = []
    for _ in range(nb):
        n = struct.unpack("<Q", _read(sock, 8))[0]
        if n > MAX_BLOB:
            raise ProtocolError("control plane: oversized blob refused")
        blobs.append(_read(sock, n))
    return _unpack(hdr, blobs)


def launch_token(addr, port, world):
    """what every rank of one launch shares and a stranger does not (AZTOT_CTL_TOKEN when the launcher set one)"""
    tok = os.environ.get("AZTOT_CTL_TOKEN")
    if tok:
        return tok
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "|" + os.environ.get("GROUP_WORLD_SIZE", "")
    return hashlib.sha256(("aztot-ctl|%s|%d|%d|%s" % (addr, port, world, run)).encode()).hexdigest()


def _is_loopback(addr):
    return addr in ("localhost", "", None) or str(addr).startswith("127.") or addr == "::1"


class Control:
    """rank / world from the arguments or the launcher's environment; world == 1 needs no sockets at all."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0, token=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.peers = []
        self.sock = None
        self.rejected = 0            # connections rank 0 turned away (wrong token, rank out of range, duplicate, not this protocol)
        if self.world <= 1:
            return
        if not 0 <= self.rank < self.world:
            raise ValueError("control plane: rank %d of %d" % (self.rank, self.world))
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("AZTOT_CTL_PORT_OFFSET", "29"))
        tok = token if token is not None else launch_token(addr, port, self.world)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1" if _is_loopback(addr) else addr, port))
            srv.listen(self.world + 8)
            deadline = time.time() + timeout
            got = {}
            while len(got) < self.world - 1:
                left = deadline - time.time()
                if left <= 0:
                    srv.close()
                    raise TimeoutError("control plane: %d of %d ranks connected within %.0f s" % (len(got) + 1, self.world, timeout))
                srv.settimeout(left)
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                try:
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(min(10.0, timeout))          # a stranger that says nothing must not hold the ranks up
                    hello = _recv(c)
                    r = hello.get("rank") if isinstance(hello, dict) else None
                    t = hello.get("token") if isinstance(hello, dict) else None
                    if (not isinstance(r, int) or isinstance(r, bool) or not 1 <= r < self.world or r in got or not isinstance(t, str)
                            or not hmac.compare_digest(t.encode(), tok.encode())):
                        raise ProtocolError("refused")
                    _send(c, {"ok": True})
                    c.settimeout(timeout)
                    got[r] = c
                except (OSError, ValueError, TypeError, AttributeError):
                    self.rejected += 1
                    try:
                        c.close()
                    except OSError:
                        pass
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            _send(s, {"rank": self.rank, "token": tok})
            try:
                ack = _recv(s)
            except ConnectionError as ex:
                raise ConnectionError("control plane: rank 0 refused rank %d (token or rank id not accepted)" % self.rank) from ex
            if not (isinstance(ack, dict) and ack.get("ok") is True):
                raise ProtocolError("control plane: unexpected handshake reply")
            self.sock = s

    # every collective: gather to rank 0, combine, reply
    def _collective(self, value, combine):
        if self.world <= 1:
            return combine([value])
        if self.rank == 0:
            vals = [value] + [_recv(p) for p in self.peers]
            out = combine(vals)
            for p in self.peers:
                _send(p, out)
            return out
        _send(self.sock, value)
        return _recv(self.sock)

    def barrier(self):
        self._collective(0, lambda v: 0)

    def broadcast(self, obj):
        """rank 0's object on every rank"""
        return self._collective(obj, lambda v: v[0])

    def all_gather(self, obj):
        return self._collective(obj, lambda v: list(v))

    def all_max(self, x):
        return self._collective(x, max)

    def all_min(self, x):
        return self._collective(x, min)

    def all_sum(self, arr):
        """element-wise sum of equal-length sequences of floats"""
        return self._collective([float(v) for v in arr], lambda v: [sum(col) for col in zip(*v)])

    def sendrecv(self, send_peer, payload, recv_peer):
        """exchange of opaque payloads between ring neighbours, relayed by rank 0 (tests of the host-staged transport only)"""
        got = self._collective([int(send_peer), bytes(payload)], lambda v: list(v))
        for src, (dst, data) in enumerate(got):
            if dst == self.rank and src == recv_peer:
                return data
        raise RuntimeError("control plane: nothing addressed to rank %d from %d" % (self.rank, recv_peer))

    def close(self):
        for p in self.peers:
            try:
                p.close()
            except OSError:
                pass
        if self.sock is not None:
            try:
                self.sock.close()
            except OSError:
                pass
        self.peers, self.sock = [], None
