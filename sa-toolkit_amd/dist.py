"""Data-parallel sharding of an utterance list over the GPUs of one node.

The reference's `anonymize` splits `wav.scp` into contiguous shards, one process per GPU, and each
process writes its own results (satools/satools/bin/anonymize:80-93, script_utils.split_dict); there
is no collective.  Here the same contiguous sharding and the same fixed batch order are kept (batch
composition is part of the result: the F0 normalisation is batch-coupled, cmvn.py:147-151), one
process per GPU under torch.distributed, and the anonymized waveforms are exchanged with ONE
all-gather (RCCL over xGMI on the GPUs; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """contiguous shard [lo, hi) of rank; the first n_items % world ranks hold one item more
    (numpy.array_split-style, the split the reference's split_dict produces)"""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def batches(lo, hi, batch_size):
    """fixed-size batches in index order inside a shard"""
    return [(s, min(s + batch_size, hi)) for s in range(lo, hi, batch_size)]


def pcm16_rows(local):
    """f32 waveforms -> the int16 PCM the reference writes (bin/pipeline.py:160, torchaudio.save(..., bits_per_sample=16):
    round(x * 32768) clipped): half the bytes on the xGMI links (82 MB per rank instead of 164 at 512 utterances)"""
    return (local * 32768.0).round_().clamp_(-32768.0, 32767.0).to(torch.int16)


def all_gather_rows(local, n_items, group=None):
    """gather per-rank row blocks [n_local, ...] (contiguous shards of shard_bounds) into
    [n_items, ...] on every rank.  Equal shards use one all_gather_into_tensor; ragged shards are
    padded to the largest shard."""
    if local is not None and local.dtype == torch.int16:
        # ProcessGroupNCCL has no int16: the PCM rows travel as bytes
        out = all_gather_rows(local.contiguous().view(torch.uint8), n_items, group)
        return out.view(torch.int16)
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]
    mx = max(sizes)
    mine = sizes[dist.get_rank(group)]
    if local is None or local.shape[0] != mine:
        raise ValueError(f"all_gather_rows: this rank's block must hold {mine} rows, got "
                         f"{None if local is None else local.shape[0]}")
    if local.shape[0] != mx:
        pad = torch.zeros((mx - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(parts, local.contiguous(), group=group)
        out = torch.cat(parts, 0)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx:r * mx + sizes[r]] for r in range(world)], 0)


def convert_sharded(convert_fn, n_items, batch_size, gather=True, group=None, local_out=None, before_gather=None, transform=None):
    """run `convert_fn(lo, hi) -> [hi-lo, ...]` over this rank's shard in fixed batches and
    (optionally) all-gather the results in global index order.

    `local_out` [shard rows, ...]: a preallocated buffer `convert_fn` fills itself (row `i - lo` for item `i`, e.g.
    from several HIP streams); its return values are then ignored and nothing is concatenated.  `before_gather()`
    runs after the last batch has been issued and before the collective (e.g. make the current stream wait for the
    job streams); `transform(local)` then maps the shard to what is gathered (e.g. `pcm16_rows`).  Every rank must hold at least one item: with n_items < world some shard is empty, its rank has no
    rows to give the trailing shape of, and the collective would hang on the others — refused on ALL ranks before
    anything is launched (the reference's `split_dict` never produces more shards than entries either)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n_items < world:
        raise ValueError(f"convert_sharded: {n_items} items cannot be sharded over {world} ranks (every rank needs one)")
    lo, hi = shard_bounds(n_items, rank, world)
    outs = [convert_fn(s, e) for s, e in batches(lo, hi, batch_size)]
    if local_out is not None:
        if local_out.shape[0] != hi - lo:
            raise ValueError(f"convert_sharded: local_out holds {local_out.shape[0]} rows, the shard {hi - lo}")
        local = local_out
    else:
        local = torch.cat(outs, 0)
    if before_gather is not None:
        before_gather()
    if transform is not None:
        local = transform(local)
    if not gather:
        return local
    return all_gather_rows(local, n_items, group)
