"""Entity-range sharding over the GPUs of one node: one process per GPU, rows split into
contiguous 256-aligned ranges, every GPU culls its own rows against ALL views, and ONE all-gather (RCCL over
xGMI; gloo in the CPU tests) of the packed ViewVisibility bitmask gives every rank the full per-view masks.
Payload is tiny (10M rows x 4 views = 5 MB in total) so the exchange is latency-bound: a single
all_gather_into_tensor on an in-place layout, no bucketing.

Layout of the gathered buffer (uint64 words): [world][n_views][words_per_shard]; rank r's kernels write
straight into block r (mi_bind_visibility_output), so the collective is in place and needs no packing pass.
"""
import numpy as np

ROW_ALIGN = 256  # one workgroup; keeps every shard's first row on a 64-bit mask word boundary


def shard_rows(n_rows, world, rank):
    """Contiguous row range [lo, hi) of `rank`: ceil-divided, aligned to ROW_ALIGN (the last shards may be short
    or empty)."""
    per = -(-n_rows // world)
    per = -(-per // ROW_ALIGN) * ROW_ALIGN
    lo = min(n_rows, rank * per)
    hi = min(n_rows, lo + per)
    return lo, hi


def words_per_shard(n_rows, world):
    lo, hi = shard_rows(n_rows, world, 0)
    per = -(-max(hi - lo, 1) // ROW_ALIGN) * ROW_ALIGN
    return per // 64


def gathered_words(n_rows, world, n_views):
    return world * n_views * words_per_shard(n_rows, world)


def block_offset_words(n_rows, world, n_views, rank):
    """(words_per_view, word_offset) to pass to mi_bind_visibility_output for this rank."""
    w = words_per_shard(n_rows, world)
    return w, rank * n_views * w


def all_gather_visibility(full, n_rows, world, n_views, rank, group=None):
    """In-place all-gather of the [world][n_views][W] uint64 buffer `full` (a torch tensor viewed as int64)."""
    import torch.distributed as dist
    w = words_per_shard(n_rows, world)
    blk = n_views * w
    if world == 1:
        return full
    dist.all_gather_into_tensor(full, full[rank * blk:(rank + 1) * blk], group=group)
    return full


def unpack_view(full_words, n_rows, world, n_views, view):
    """numpy helper: the gathered buffer -> uint8[n_rows] visibility of one view (rows in global order)."""
    w = words_per_shard(n_rows, world)
    full_words = np.asarray(full_words).view(np.uint64).reshape(world, n_views, w)
    out = np.zeros(n_rows, np.uint8)
    for r in range(world):
        lo, hi = shard_rows(n_rows, world, r)
        if hi <= lo:
            continue
        bits = np.unpackbits(full_words[r, view].view(np.uint8), bitorder="little")
        out[lo:hi] = bits[:hi - lo]
    return out
