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
    out = [torch.empty_like(lines) for _ in range(world)] if rank == dst else None
    dist.gather(lines, out, dst=dst, group=group)
    return torch.stack(out) if rank == dst else None


class StreamRunner:
    """One rank = one IQ stream.  Two protocols, the same object for a real context on a GPU and for the stub context of
    tests/test_multi_gpu_gloo.py:

    ordinary (pipelined=False): step(i) = one push of the hot path over input batch i (already resident on the device), then — with more
    than one rank — this rank's finished zoomed lines are copied out of the context (`fft_copy_device`) and gathered on rank 0.
    `lines` is this rank's [lines_per_push, data_width] staging tensor.

    pipelined (sdrpp_set_pipelined, one launch per block, results `depth` launches late): step(i) = one push; the zoomed lines of the
    block pushed `lag` steps earlier are taken from its page-locked result slot (`result_wait` never has to flush the pipeline for a block
    that old) into a host staging buffer, and every `gather_every` blocks the batch goes to the device tensor `lines`
    ([gather_every, max_lines + 1, data_width]; row max_lines, column 0 of a block's slot = its line count — blocks at the stream cap
    complete a varying number of frames) and, with more than one rank, is gathered on rank 0.  Copy and gather run on a side stream:
    nothing the host does for the lines waits for the launches queued on the compute stream.  finish() collects what is outstanding.

    ctx needs push_device(ptr, count) and fft_copy_device(first, n, zoomed_ptr=...) resp. ticket() / result_wait(ticket, copy=False) /
    result_release(ticket); `bufs` are the resident input batches (anything with data_ptr()); sync() blocks until the device is idle
    (torch.cuda.synchronize on a GPU, a no-op for the CPU stub)."""

    def __init__(self, ctx, bufs, push, lines, sync=None, pipelined=False, lag=8, gather_every=4):
        self.ctx, self.bufs, self.push, self.lines = ctx, bufs, int(push), lines
        self.sync = sync or (lambda: None)
        # (a process group of ONE rank still runs the collectives: how a single-GPU box exercises the RCCL leg)
        self.collective = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if self.collective else 1
        self.rank = dist.get_rank() if self.collective else 0
        self.gathered = None
        self.pipelined = bool(pipelined)
        self.lag, self.gather_every = int(lag), int(gather_every)
        self.tickets = []
        self.collected = 0   # blocks whose lines have been taken from their result slots
        self.batches = 0     # batches handed to the device / gathered
        self.nb = 0
        self.local_elapsed = 0.0
        self._next_ticket = None
        if self.pipelined:
            assert lines.dim() == 3 and lines.shape[0] == self.gather_every, "pipelined: lines = [gather_every, max_lines + 1, data_width]"
            self.max_lines = lines.shape[1] - 1
            on_gpu = lines.is_cuda
            self.host = [torch.zeros(tuple(lines.shape), dtype=lines.dtype).pin_memory() if on_gpu else torch.zeros(tuple(lines.shape), dtype=lines.dtype) for _ in range(2)]
            self.side = torch.cuda.Stream(device=lines.device) if on_gpu else None
            self.host_free = [torch.cuda.Event() if on_gpu else None for _ in range(2)]
            self.cur = 0
            # a context with result_lines_into (capi.Context) copies a block's lines straight into the staging buffer
            self.lean = hasattr(ctx, "result_lines_into")
            self.host_np = [h.numpy() for h in self.host]
            self.host_addr = [h.data_ptr() for h in self.host]
            self.slot_bytes = lines.shape[1] * lines.shape[2] * lines.element_size()
        self.buf_ptrs = [b.data_ptr() for b in bufs]

    # ---- pipelined protocol ----
    def _collect(self, ticket):
        if self.lean:  # one call: wait, one memmove of the lines into the staging buffer, release
            n = self.ctx.result_lines_into(ticket, self.host_addr[self.cur] + self.nb * self.slot_bytes, self.max_lines)
            self.host_np[self.cur][self.nb, self.max_lines, 0] = n
        else:
            r = self.ctx.result_wait(ticket, copy=False)
            n = int(r["n_lines"])
            if n > self.max_lines:
                raise RuntimeError("block %d completed %d lines, staging holds %d" % (ticket, n, self.max_lines))
            h = self.host[self.cur]
            if n:
                h[self.nb, :n].copy_(torch.from_numpy(r["zoomed"]))
            h[self.nb, self.max_lines, 0] = float(n)
            self.ctx.result_release(ticket)
        self.collected += 1
        self.nb += 1
        if self.nb == self.gather_every:
            self._flush_batch()

    def _flush_batch(self):
        if self.nb == 0:
            return
        h = self.host[self.cur]
        for k in range(self.nb, self.gather_every):  # a partial last batch: the unused slots say "no lines"
            self.host_np[self.cur][k, self.max_lines, 0] = 0.0
        if self.side is not None:
            with torch.cuda.stream(self.side):
                self.lines.copy_(h, non_blocking=True)
                self.host_free[self.cur].record(self.side)
                if self.collective:
                    self.gathered = gather_lines(self.lines, dst=0)
            self.cur ^= 1
            self.host_free[self.cur].synchronize()  # the staging buffer about to be refilled has left the host (two batches ago)
        else:
            self.lines.copy_(h)
            if self.collective:
                self.gathered = gather_lines(self.lines, dst=0)
            self.cur ^= 1
        if not self.collective:
            self.gathered = self.lines.unsqueeze(0)
        self.batches += 1
        self.nb = 0

    def finish(self):
        """Collect every outstanding block (the first result_wait runs the queued stages), hand over the last partial batch."""
        if not self.pipelined:
            return
        # end of the stream: everything still queued is launched in one go (sdrpp_pipeline_flush does not wait) — collecting the last blocks
        # one by one would launch one tick, wait for it, launch the next ...
        if self.tickets and hasattr(self.ctx, "pipeline_flush"):
            self.ctx.pipeline_flush()
        # the tickets are counted locally between two finish() calls (one ctypes call less per step); here — once per run, outside the per-step
        # path — the count is checked against the context: another caller pushing on the same context, or a context whose pipeline was reset,
        # would otherwise make the runner collect the wrong slots without noticing
        if self.tickets and self._next_ticket is not None and hasattr(self.ctx, "ticket"):
            now = self.ctx.ticket()
            if now != self._next_ticket:
                raise RuntimeError("StreamRunner: the context stands at ticket %d, the runner counted %d — somebody else pushed on this context or its pipeline was reset" % (now, self._next_ticket))
        while self.tickets:
            self._collect(self.tickets.pop(0))
        self._next_ticket = None  # re-read from the context at the next step (the context may be re-configured between runs)
        self._flush_batch()
        if self.side is not None:
            self.side.synchronize()

    def step(self, i):
        self.ctx.push_device(self.buf_ptrs[i % len(self.buf_ptrs)], self.push)
        if self.pipelined:
            if self._next_ticket is None:  # (tickets count the pushes of the context: one call to learn where it stands, then counted here)
                self._next_ticket = self.ctx.ticket()
            else:
                self._next_ticket += 1
            self.tickets.append(self._next_ticket)
            if len(self.tickets) > self.lag:
                self._collect(self.tickets.pop(0))
        elif self.collective:
            self.ctx.fft_copy_device(0, self.lines.shape[0], zoomed_ptr=self.lines.data_ptr())
            self.gathered = gather_lines(self.lines, dst=0)

    def barrier(self):
        if self.collective:
            dist.barrier()

    def timed(self, steps, first=0):
        """EXACTLY `steps` steps (pipelined: and the delivery of all their lines) bracketed by barrier + device sync on both sides;
        returns the MAX over ranks of the wall time."""
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(first + i)
        self.finish()
        self.sync()
        self.local_elapsed = time.perf_counter() - t0  # this rank's own steps + deliveries (before it waits for the slowest rank)
        self.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.collective:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=self.lines.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed
