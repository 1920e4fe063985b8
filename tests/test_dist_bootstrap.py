"""CPU: the one-shot TCP exchange that carries the ncclUniqueId from rank 0 to the other ranks (distributed.Distributed.from_environment)
-- three real processes on 127.0.0.1, no GPU."""
import multiprocessing as mp
import os
import socket


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oldoceananigans_jl_amd import distributed as dist
    payload = bytes(range(128)) if rank == 0 else None
    q.put((rank, dist._broadcast_bytes(payload, rank, world, timeout=30.0)))


def test_unique_id_reaches_every_rank_and_skips_a_busy_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    # a stranger already listens on MASTER_PORT + 1: the exchange moves to the next port, the clients tell the listeners apart
    stranger = socket.socket()
    stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        stranger.bind(("127.0.0.1", port + 1))
        stranger.listen(4)
    except OSError:
        stranger = None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    if stranger is not None:
        stranger.close()
    assert all(got[r] == bytes(range(128)) for r in range(3))
