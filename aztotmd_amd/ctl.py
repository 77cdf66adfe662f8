"""Control plane of a multi-process run (one process per GPU) WITHOUT torch: a handful of small collectives over plain TCP.

Why not torch.distributed: the rank processes bring up RCCL inside libaztot (system ROCm runtime); importing torch into the same
process puts the HIP/HSA copies bundled with its wheel in front of it, and RCCL then reports "no ROCm-capable device".  The
launcher (python -m torch.distributed.run) is a process of its own and stays as it is; the ranks only read RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT from the environment it sets.  Nothing here is on the data path: the halo travels over RCCL
(exchange.cpp); this carries the RCCL unique id, barriers and a few scalars.

Rank 0 listens on MASTER_PORT + AZTOT_CTL_PORT_OFFSET (default 29); every collective is a gather to rank 0 followed by a reply.
"""
import os
import pickle
import socket
import struct
import time


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv(sock):
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        hdr += chunk
    n = struct.unpack("<Q", hdr)[0]
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("control plane: peer closed the connection")
        buf += chunk
    return pickle.loads(bytes(buf))


class Control:
    """rank / world from the arguments or the launcher's environment; world == 1 needs no sockets at all."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.peers = []
        self.sock = None
        if self.world <= 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("MASTER_PORT", "29500")) + int(os.environ.get("AZTOT_CTL_PORT_OFFSET", "29"))
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = _recv(c)
                got[int(r)] = c
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
            _send(s, self.rank)
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
        return self._collective(list(arr), lambda v: [sum(col) for col in zip(*v)])

    def sendrecv(self, send_peer, payload, recv_peer):
        """exchange of opaque payloads between ring neighbours, relayed by rank 0 (tests of the host-staged transport only)"""
        got = self._collective((send_peer, payload), lambda v: list(v))
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
