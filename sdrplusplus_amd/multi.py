"""Multi-GPU layout of the hot path: independent wideband IQ streams, one per GPU / process.

There is no data-path collective: stream i lives entirely on rank i.  The only exchange is the gather of finished
(zoomed) waterfall lines to the display rank — `gather_lines` — over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU node; "gloo" in the CPU test).  `StreamRunner` is the per-rank step / timing protocol bench.py runs: the same
object drives a real context on a GPU and a stub context in tests/test_multi_gpu_gloo.py, so the N > 1 control flow that
the driver launches on the 8-GPU node is the one the CPU test exercises."""
import time

import torch
import torch.distributed as dist


def stream_for_rank(rank, world, n_streams):
    """Streams are dealt round-robin to ranks; with n_streams == world, stream i -> GPU i (BASELINE cfg 5)."""
    return [s for s in range(n_streams) if s % world == rank]


def stream_seed(base_seed, stream_index):
    """cfg 5: 8 copies of cfg 4 with seeds 0..7 -> per-stream seed, independent of which rank hosts the stream."""
    return int(base_seed) + int(stream_index)


def gather_lines(lines, dst=0, group=None):
    """Gather each rank's [n_lines, data_width] tensor of finished waterfall lines on `dst`.
    Returns a [world, n_lines, data_width] tensor on dst, None elsewhere.  All ranks must pass the same shape (the per-step
    line count is fixed by the framing: samples_per_step / (nz + skip))."""
    if not dist.is_available() or not dist.is_initialized():
        return lines.contiguous().unsqueeze(0)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lines = lines.contiguous()
    if world == 1:
        return lines.unsqueeze(0)
    out = [torch.empty_like(lines) for _ in range(world)] if rank == dst else None
    dist.gather(lines, out, dst=dst, group=group)
    return torch.stack(out) if rank == dst else None


class StreamRunner:
    """One rank = one IQ stream.  step(i): one push of the hot path over input batch i (already resident on the device), then —
    with more than one rank — this rank's finished zoomed lines are copied out of the context and gathered on rank 0.

    ctx needs push_device(ptr, count) and fft_copy_device(first, n, zoomed_ptr=...); `bufs` are the resident input batches
    (anything with data_ptr()); `lines` is this rank's [lines_per_push, data_width] staging tensor; sync() blocks until the
    device is idle (torch.cuda.synchronize on a GPU, a no-op for the CPU stub)."""

    def __init__(self, ctx, bufs, push, lines, sync=None):
        self.ctx, self.bufs, self.push, self.lines = ctx, bufs, int(push), lines
        self.sync = sync or (lambda: None)
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.gathered = None

    def step(self, i):
        self.ctx.push_device(self.bufs[i % len(self.bufs)].data_ptr(), self.push)
        if self.world > 1:
            self.ctx.fft_copy_device(0, self.lines.shape[0], zoomed_ptr=self.lines.data_ptr())
            self.gathered = gather_lines(self.lines, dst=0)

    def barrier(self):
        if self.world > 1:
            dist.barrier()

    def timed(self, steps, first=0):
        """EXACTLY `steps` steps bracketed by barrier + device sync on both sides; returns the MAX over ranks of the wall time."""
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(first + i)
        self.sync()
        self.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=self.lines.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed
