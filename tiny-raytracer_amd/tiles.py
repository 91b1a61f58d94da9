"""Image sharding across the GPUs of one node.

Every pixel-sample is independent (renderer/sampler/cpu.rs:39-65 reads only the immutable BVH), so
the image is the only shared output.  The scene is replicated; the image is cut into bands of
`band_rows` rows dealt round-robin (band b -> rank b % world_size) so that expensive regions spread
over all ranks.  Each rank renders its bands into one contiguous [rows_local, W, 3] f32 buffer; the
RNG is keyed by the *image* pixel index, so the assembled frame is bit-identical for any world size.
The only data-path collective is one gather of those buffers to rank 0 (RCCL over xGMI when the
tensors are on GPUs, gloo on CPU tensors in the tests), followed by a row un-interleave.
"""
import torch
import torch.distributed as dist

DEFAULT_BAND_ROWS = 16


def band_layout(height, world_size, rank, band_rows=DEFAULT_BAND_ROWS):
    """Rows owned by `rank`.  Returns dict(band_rows, band_stride, band_offset, rows_local, rows) where rows is the
    list of image rows in local order (local row r -> rows[r]), matching trt_render_params' mapping."""
    n_bands = (height + band_rows - 1) // band_rows
    rows = []
    for b in range(rank, n_bands, world_size):
        rows.extend(range(b * band_rows, min((b + 1) * band_rows, height)))
    return dict(band_rows=band_rows, band_stride=world_size, band_offset=rank, rows_local=len(rows), rows=rows)


def max_rows_local(height, world_size, band_rows=DEFAULT_BAND_ROWS):
    return max(band_layout(height, world_size, r, band_rows)["rows_local"] for r in range(world_size))


_PLACE_CACHE = {}


def row_placement(height, world_size, band_rows=DEFAULT_BAND_ROWS, pad_rows=None, device="cpu"):
    """Index tensor `src` with full[y] = stacked[src[y]], where `stacked` is the [world_size * pad_rows, W, 3] concatenation
    of the ranks' padded buffers in rank order: the whole un-interleave is ONE index_select."""
    pad_rows = pad_rows or max_rows_local(height, world_size, band_rows)
    key = (height, world_size, band_rows, pad_rows, str(device))
    if key not in _PLACE_CACHE:
        src = torch.empty(height, dtype=torch.long)
        for r in range(world_size):
            rows = band_layout(height, world_size, r, band_rows)["rows"]
            src[torch.tensor(rows, dtype=torch.long)] = r * pad_rows + torch.arange(len(rows), dtype=torch.long)
        _PLACE_CACHE[key] = src.to(device)
    return _PLACE_CACHE[key]


def gather_image(local, height, width, world_size, rank, band_rows=DEFAULT_BAND_ROWS, group=None):
    """One gather of the per-rank accumulators to rank 0, then one index_select that puts every row in its place.
    `local` is this rank's [rows_local, width, 3] f32 tensor.  Returns the full [height, width, 3] image on rank 0, None
    elsewhere.  Ranks may own different row counts (ragged last band), so buffers are padded to the common maximum."""
    if world_size == 1:
        return local
    pad_rows = max_rows_local(height, world_size, band_rows)
    send = local
    if local.shape[0] != pad_rows:
        send = torch.zeros((pad_rows, width, 3), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    if rank == 0:
        stacked = torch.empty((world_size * pad_rows, width, 3), dtype=local.dtype, device=local.device)
        recv = list(stacked.view(world_size, pad_rows, width, 3).unbind(0))          # views: the gather fills `stacked` in place
        dist.gather(send, gather_list=recv, dst=0, group=group)
        return stacked.index_select(0, row_placement(height, world_size, band_rows, pad_rows, local.device))
    dist.gather(send, gather_list=None, dst=0, group=group)
    return None
