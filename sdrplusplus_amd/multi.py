"""Multi-GPU layout of the hot path: independent wideband IQ streams, one per GPU / process.

There is no data-path collective: stream i lives entirely on rank i.  The only exchange is the gather of finished
(zoomed) waterfall lines to the display rank — `gather_lines` — over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU node; "gloo" in the CPU test)."""
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
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lines = lines.contiguous()
    if world == 1:
        return lines.unsqueeze(0)
    out = [torch.empty_like(lines) for _ in range(world)] if rank == dst else None
    dist.gather(lines, out, dst=dst, group=group)
    return torch.stack(out) if rank == dst else None
