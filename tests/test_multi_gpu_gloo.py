"""N > 1 path on CPU: world_size-2 gloo processes, one independent IQ stream per rank, gather of waterfall lines to rank 0
(the only exchange step of the path).  Line contents come from the oracle here — the point is the sharding / gather logic
bench.py uses with RCCL on the GPU node."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import support as S
    from sdrplusplus_amd import multi, workloads

    streams = multi.stream_for_rank(rank, world, world)
    assert streams == [rank]
    N, W = 4096, 256
    x = workloads.synth(2, 3 * N, seed=multi.stream_seed(100, streams[0]))
    w = S.oracle_fft_window(2, N)
    lines = S.OracleSpectrum(N, N, 0, w).push(x)
    zoomed = np.stack([S.oracle_do_zoom(0, N, W, l) for l in lines])
    got = multi.gather_lines(torch.from_numpy(zoomed), dst=0)
    if rank == 0:
        q.put(got.numpy())
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_streams_gather_lines_gloo():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import support as S
    from sdrplusplus_amd import multi, workloads

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == (2, 3, 256)
    N, W = 4096, 256
    for r in range(2):  # rank r's slot holds stream r's lines (distinct seeds -> distinct content)
        x = workloads.synth(2, 3 * N, seed=multi.stream_seed(100, r))
        lines = S.OracleSpectrum(N, N, 0, S.oracle_fft_window(2, N)).push(x)
        exp = np.stack([S.oracle_do_zoom(0, N, W, l) for l in lines])
        assert np.array_equal(got[r], exp)
    assert not np.array_equal(got[0], got[1])


def test_stream_dealing():
    from sdrplusplus_amd import multi

    assert multi.stream_for_rank(0, 8, 8) == [0] and multi.stream_for_rank(7, 8, 8) == [7]
    assert multi.stream_for_rank(1, 2, 8) == [1, 3, 5, 7]
    assert [multi.stream_seed(0, s) for s in range(8)] == list(range(8))
