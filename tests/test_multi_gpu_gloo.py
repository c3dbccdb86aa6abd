"""N > 1 path on CPU: world_size-2 gloo processes, one independent IQ stream per rank, gather of waterfall lines to rank 0
(the only exchange step of the path).  Line contents come from the oracle here — the point is the sharding / gather logic
bench.py uses with RCCL on the GPU node: test_stream_runner_protocol_gloo drives multi.StreamRunner — the very object whose
step() / timed() bench.py calls — with a stub context."""
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


class _StubCtx:
    """What StreamRunner needs of a context: push_device + fft_copy_device.  The 'lines' of push i are rank- and step-specific numbers."""

    def __init__(self, rank, lines):
        self.rank, self.lines, self.pushes = rank, lines, []

    def push_device(self, ptr, count):
        self.pushes.append((ptr, count))

    def fft_copy_device(self, first, n, zoomed_ptr=None):
        assert first == 0 and n == self.lines.shape[0] and zoomed_ptr == self.lines.data_ptr()
        self.lines.fill_(1000.0 * self.rank + len(self.pushes))
        return n


def _runner_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdrplusplus_amd import multi

    lines = torch.zeros((3, 16), dtype=torch.float32)
    bufs = [torch.zeros(8), torch.ones(8)]
    ctx = _StubCtx(rank, lines)
    r = multi.StreamRunner(ctx, bufs, push=4, lines=lines)
    assert (r.world, r.rank) == (world, rank)
    for i in range(2):
        r.step(i)
    elapsed = r.timed(5, first=2)
    assert elapsed > 0.0 and len(ctx.pushes) == 7
    assert [p for p, _ in ctx.pushes] == [bufs[i % 2].data_ptr() for i in range(7)] and all(c == 4 for _, c in ctx.pushes)
    if rank == 0:
        q.put((elapsed, r.gathered.numpy().copy()))
    else:
        assert r.gathered is None
        q.put((elapsed, None))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_runner_protocol_gloo():
    """bench.py's per-rank protocol (multi.StreamRunner: step = push + copy out this rank's lines + gather on rank 0; timed = exactly K steps
    between barriers, MAX over ranks) with world size 2 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    times = [t for t, _ in res]
    assert abs(times[0] - times[1]) < 1e-9  # the all-reduced maximum: identical on both ranks
    got = [g for _, g in res if g is not None]
    assert len(got) == 1 and got[0].shape == (2, 3, 16)
    assert np.all(got[0][0] == 7.0) and np.all(got[0][1] == 1007.0)  # slot r = rank r's lines of the last (7th) push


class _StubPipeCtx:
    """Pipelined protocol: push_device + ticket / result_wait / result_release.  Block t of rank r "completes" (t % 3) + 1 lines, every
    value of which is 1000 r + t."""

    def __init__(self, rank, width):
        self.rank, self.width, self.pushes, self.held, self.released = rank, width, 0, set(), []

    def push_device(self, ptr, count):
        self.pushes += 1

    def ticket(self):
        return self.pushes

    def result_wait(self, ticket, copy=True):
        assert 1 <= ticket <= self.pushes and ticket not in self.held and not copy
        self.held.add(ticket)
        n = (ticket % 3) + 1
        return {"ticket": ticket, "n_lines": n, "zoomed": np.full((n, self.width), 1000.0 * self.rank + ticket, np.float32)}

    def result_release(self, ticket):
        self.held.remove(ticket)
        self.released.append(ticket)


def _pipe_runner_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sdrplusplus_amd import multi

    W, G, ML = 16, 4, 3
    lines = torch.zeros((G, ML + 1, W), dtype=torch.float32)
    bufs = [torch.zeros(8), torch.ones(8)]
    ctx = _StubPipeCtx(rank, W)
    r = multi.StreamRunner(ctx, bufs, push=4, lines=lines, pipelined=True, lag=3, gather_every=G)
    elapsed = r.timed(10)  # 10 blocks: two full batches + a partial one of 2
    assert elapsed > 0.0 and ctx.pushes == 10 and ctx.released == list(range(1, 11)) and not ctx.held
    assert r.collected == 10 and r.batches == 3 and not r.tickets
    q.put((rank, elapsed, None if r.gathered is None else r.gathered.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_stream_runner_pipelined_protocol_gloo():
    """The protocol bench.py runs in pipelined mode (one launch per block, lines taken from the result slot of the block `lag` pushes back,
    batches of `gather_every` blocks gathered on rank 0, finish() inside the timed region) with world size 2 over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_runner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda x: x[0])
    assert abs(res[0][1] - res[1][1]) < 1e-9
    assert res[1][2] is None
    got = res[0][2]
    assert got.shape == (2, 4, 4, 16)  # [rank, block of the batch, max_lines + 1, width]
    for rank in range(2):  # last batch = blocks 9 and 10, then two empty slots
        for k, t in enumerate((9, 10)):
            n = (t % 3) + 1
            assert got[rank, k, 3, 0] == n and np.all(got[rank, k, :n] == 1000.0 * rank + t)
        assert got[rank, 2, 3, 0] == 0 and got[rank, 3, 3, 0] == 0


def test_stream_dealing():
    from sdrplusplus_amd import multi

    assert multi.stream_for_rank(0, 8, 8) == [0] and multi.stream_for_rank(7, 8, 8) == [7]
    assert multi.stream_for_rank(1, 2, 8) == [1, 3, 5, 7]
    assert [multi.stream_seed(0, s) for s in range(8)] == list(range(8))


def test_bench_spawns_its_ranks_when_started_like_the_one_gpu_run():
    """`python bench.py --gpus 2` with NO launcher around it (how the driver starts `--gpus 1`) must start its own ranks: the launcher branch
    re-executes bench.py through torch.distributed.run with one rank per GPU.  --dry-launch takes the ranks through everything of the N > 1
    path that needs no device — rendezvous (gloo here, RCCL on the node), device naming, StreamRunner's pipelined protocol on a stub context,
    per-rank rates, ONE JSON line from rank 0 as the last line of stdout."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--steps", "9", "--warmup", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    out = json.loads(last)
    assert out["dry_launch"] and out["ok"] and out["n_gpus"] == 2 and out["steps"] == 9 and len(out["devices"]) == 2
    assert [p["rank"] for p in out["per_rank"]] == [0, 1] and out["gathered_shape"][0] == 2
    # ONE metric across N (BASELINE.json: "... 1/2/4/8 GPU"): the line of --gpus 1 and of --gpus 2 name the same metric and the same per-stream
    # workload — the headline cfg 3 on every rank's own stream; cfg 5 only on request
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dry-launch", "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r1.returncode == 0, (r1.stdout[-2000:], r1.stderr[-2000:])
    one = json.loads([l for l in r1.stdout.splitlines() if l.strip()][-1])
    assert one["n_gpus"] == 1 and len(one["per_rank"]) == 1
    assert one["metric"] == out["metric"] and "32 VFO WFM" in out["metric"] and "cfg5" not in out["metric"]
    assert one["config"]["workload"] == out["config"]["workload"] and out["config"]["workload"].startswith("cfg3:")
    assert out["config"]["streams"] == 2 and one["config"]["streams"] == 1
    r5 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch", "--cfg", "5", "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    five = json.loads([l for l in r5.stdout.splitlines() if l.strip()][-1])
    assert "cfg5" in five["metric"] and five["config"]["workload"].startswith("cfg5:")
    # a launcher that already set WORLD_SIZE to something else is an error message, not an AssertionError
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True, timeout=120, env=env2)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)
