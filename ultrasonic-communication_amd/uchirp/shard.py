"""Frame sharding across the GPUs of one node.

Frames are independent, so the frame index space is block-partitioned: rank r
of W owns frames [lo_r, hi_r), contiguous, sizes differing by at most one.
There is NO data-path collective; the only exchange is the gather of the
decoded symbol stream (1 byte per frame) at the end of a batch -- RCCL
(`nccl` backend) on GPUs, `gloo` in the CPU tests.
"""
import numpy as np


def partition(n_frames, world, rank):
    """[lo, hi) of `rank`: contiguous blocks, the first n_frames % world ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(int(n_frames), world)
    lo = rank * q + min(rank, r)
    hi = lo + q + (1 if rank < r else 0)
    return lo, hi


def frame_span(lo, hi, n, stride, halo=0):
    """Element range [e0, e1) of the sample stream a rank must hold to process its
    frames: overlapping FIFO reads (stride < n) and the FIR halo are satisfied by
    overlapping the INPUT partition (read-only duplication), never by an exchange."""
    if hi <= lo:
        return 0, 0
    return lo * stride - halo, (hi - 1) * stride + n


def stream_span(n_samples, world, rank, halo, hop, decim):
    """UC_STREAM: the overlap-save BLOCKS are independent, so a stream shards like frames do.
    Returns (s0, s1, q0, q1): rank processes samples[s0:s1] (its first `halo` samples are the
    history it shares, read-only, with the previous rank) and produces outputs [q0, q1) of the
    whole stream; block boundaries coincide with the single-GPU run, so the results are identical."""
    n_out = (n_samples - halo) // decim if n_samples > halo else 0
    n_blocks = -(-n_out // hop)
    b0, b1 = partition(n_blocks, world, rank)
    q0, q1 = min(b0 * hop, n_out), min(b1 * hop, n_out)
    if q1 <= q0:
        return 0, 0, q0, q0
    return q0 * decim, halo + q1 * decim, q0, q1


def gather_symbols(local_symbols, n_frames, dist=None, group=None):
    """All-gather the per-rank symbol bytes into the full stream (every rank gets it).

    local_symbols: torch uint8 tensor with this rank's partition, in frame order.
    Ragged partitions are padded to the largest shard for the collective and
    trimmed afterwards.
    """
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_symbols
    world = dist.get_world_size(group)
    sizes = [partition(n_frames, world, r) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.full((mx,), 0xFF, dtype=torch.uint8, device=local_symbols.device)
    pad[: local_symbols.numel()] = local_symbols
    out = torch.empty(world * mx, dtype=torch.uint8, device=local_symbols.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * mx: r * mx + (h - l)] for r, (l, h) in enumerate(sizes)]
    return torch.cat(parts)


def symbols_to_bytes(symbols):
    """MSB-first bit packing of the receiver: msg = (msg << 1) + bit, a char every 8 bits
    (receiver/Src/main.c:523-537).  symbols: array of 0/1."""
    s = np.asarray(symbols, dtype=np.uint8)
    nb = s.size // 8
    return np.packbits(s[: nb * 8].reshape(nb, 8), axis=1, bitorder="big").reshape(-1).tobytes()
