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


def _shard_sizes(n_items, world):
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


class ChunkedGather:
    """The same exchange as `all_gather_rows`, issued in pieces while the shard is still being computed: chunk c = local rows
    [c * rows, (c + 1) * rows) of every rank (the last chunk of a ragged shard is shorter on some ranks: padded), one asynchronous
    all-gather per chunk straight into the rows of `out` where they belong.  Same bytes on the links as the single collective,
    hidden behind the batches still running, and a slow rank shows up at the chunk it delays (per-chunk issue / wait times in
    `.stats`).  Every rank must call `issue(c, ...)` for the same chunks in the same order."""

    def __init__(self, n_items, rows_per_chunk, group=None):
        self.group, self.n_items, self.rows = group, int(n_items), int(rows_per_chunk)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.sizes = _shard_sizes(n_items, self.world)
        self.starts = [shard_bounds(n_items, r, self.world)[0] for r in range(self.world)]
        self.n_chunks = -(-max(self.sizes) // self.rows)
        self.out = None
        self.pending = []
        self.stats = []
        self._int16 = False

    def chunk_rows(self, c, rank=None):
        """local row range of chunk c on `rank` (empty past the end of a shorter shard)"""
        n = self.sizes[self.rank if rank is None else rank]
        return min(c * self.rows, n), min((c + 1) * self.rows, n)

    def issue(self, c, local_chunk):
        """local_chunk: this rank's rows of chunk c (already transformed), on the stream the collective may start behind"""
        import time
        if local_chunk.dtype == torch.int16:      # ProcessGroupNCCL has no int16: the PCM rows travel as bytes
            local_chunk = local_chunk.contiguous().view(torch.uint8)
            self._int16 = True
        if self.out is None:
            self.out = torch.empty((self.n_items,) + tuple(local_chunk.shape[1:]), dtype=local_chunk.dtype, device=local_chunk.device)
        spans = [self.chunk_rows(c, r) for r in range(self.world)]
        lens = [b - a for a, b in spans]
        if local_chunk.shape[0] != lens[self.rank]:
            raise ValueError(f"ChunkedGather.issue: chunk {c} of this rank holds {lens[self.rank]} rows, got {local_chunk.shape[0]}")
        t0 = time.perf_counter()
        if all(n == lens[0] for n in lens):
            views = [self.out[self.starts[r] + spans[r][0]: self.starts[r] + spans[r][1]] for r in range(self.world)]
            work = dist.all_gather(views, local_chunk.contiguous(), group=self.group, async_op=True)
            self.pending.append((work, None, ()))
        else:
            mx = max(lens)
            mine = local_chunk
            if mine.shape[0] != mx:
                mine = torch.cat([mine, torch.zeros((mx - mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)], 0)
            parts = [torch.empty_like(mine) for _ in range(self.world)]
            work = dist.all_gather(parts, mine.contiguous(), group=self.group, async_op=True)

            def place(parts=parts, spans=spans, lens=lens):
                for r in range(self.world):
                    if lens[r]:
                        self.out[self.starts[r] + spans[r][0]: self.starts[r] + spans[r][1]].copy_(parts[r][:lens[r]])
            self.pending.append((work, place, tuple(parts)))
        self.stats.append({"chunk": c, "rows": lens[self.rank], "issue_ms": (time.perf_counter() - t0) * 1e3})

    def finish(self):
        """wait for every chunk (the current stream waits for the collectives), place padded chunks; -> [n_items, ...]"""
        import time
        # `out` and the padded chunks' `parts` were allocated under the caller's `before_chunk` context, i.e. (on the GPUs) on the
        # communication stream; from here on they are read and written on the CURRENT stream.  The caching allocator knows a block
        # only by the stream it was allocated on: without `record_stream` it could hand the block to the next allocation on the
        # communication stream (a later `transform(piece)`) while the current stream still works on it.
        cur = torch.cuda.current_stream() if self.out is not None and self.out.is_cuda else None
        for i, (work, place, parts) in enumerate(self.pending):
            t0 = time.perf_counter()
            work.wait()
            if cur is not None:
                for p in parts:
                    p.record_stream(cur)
            if place is not None:
                place()
            self.stats[i]["wait_ms"] = (time.perf_counter() - t0) * 1e3
        self.pending = []
        if cur is not None:
            self.out.record_stream(cur)
        return self.out.view(torch.int16) if self._int16 else self.out


def convert_sharded(convert_fn, n_items, batch_size, gather=True, group=None, local_out=None, before_gather=None, transform=None,
                    gather_chunk_batches=0, before_chunk=None, stats=None, gather_chunk_lag=1):
    """run `convert_fn(lo, hi) -> [hi-lo, ...]` over this rank's shard in fixed batches and
    (optionally) all-gather the results in global index order.

    `local_out` [shard rows, ...]: a preallocated buffer `convert_fn` fills itself (row `i - lo` for item `i`, e.g.
    from several HIP streams); its return values are then ignored and nothing is concatenated.  `before_gather()`
    runs after the last batch has been issued and before the collective (e.g. make the current stream wait for the
    job streams); `transform(local)` then maps the shard to what is gathered (e.g. `pcm16_rows`).  Every rank must hold at least one item: with n_items < world some shard is empty, its rank has no
    rows to give the trailing shape of, and the collective would hang on the others — refused on ALL ranks before
    anything is launched (the reference's `split_dict` never produces more shards than entries either).

    `gather_chunk_batches` = K > 0 (needs `local_out`): the exchange is issued in chunks of K batches as they are enqueued
    (`ChunkedGather`) instead of once at the end — `before_chunk(c, a, b)` runs before chunk c (local rows [a, b)) is handed to the
    collective and may return a context manager the issue runs inside (e.g. a communication stream that waits for the job streams of
    those batches).  The result equals the one-collective path's; `stats` (a dict) receives the per-chunk host timings.
    `gather_chunk_lag` = L (default 1): chunk c is handed to the collective once the batches of chunk c + L have been enqueued too — a
    caller whose batches carry deferred work (`convert(..., defer_status=True)`: rows may be rewritten until the batch's status has been
    checked, which it does before the same job's next batch) then never has to finish that work for batches it has only just launched."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if n_items < world:
        raise ValueError(f"convert_sharded: {n_items} items cannot be sharded over {world} ranks (every rank needs one)")
    lo, hi = shard_bounds(n_items, rank, world)
    chunked = None
    if gather and gather_chunk_batches > 0:
        if local_out is None:
            raise ValueError("convert_sharded: gather_chunk_batches needs a preallocated local_out")
        chunked = ChunkedGather(n_items, gather_chunk_batches * batch_size, group)
    if local_out is not None and local_out.shape[0] != hi - lo:
        raise ValueError(f"convert_sharded: local_out holds {local_out.shape[0]} rows, the shard {hi - lo}")
    import contextlib

    def issue_chunk(c):
        a, b = chunked.chunk_rows(c)
        ctx = before_chunk(c, a, b) if before_chunk is not None else None
        with (ctx if ctx is not None else contextlib.nullcontext()):
            piece = local_out[a:b]
            chunked.issue(c, transform(piece) if transform is not None else piece)

    outs, done = [], 0
    for i, (s, e) in enumerate(batches(lo, hi, batch_size)):
        outs.append(convert_fn(s, e))
        if chunked is not None:
            ready = (i + 1) // gather_chunk_batches - max(0, int(gather_chunk_lag))     # chunks enqueued at least `lag` chunks ago
            while done < min(ready, chunked.n_chunks):
                issue_chunk(done)
                done += 1
    if local_out is not None:
        local = local_out
    else:
        local = torch.cat(outs, 0)
    if chunked is not None:
        while done < chunked.n_chunks:            # the ragged tail of this shard, and chunks only longer shards have (empty here)
            issue_chunk(done)
            done += 1
    if before_gather is not None:
        before_gather()
    if chunked is not None:
        out = chunked.finish()
        if stats is not None:
            stats["chunks"] = chunked.stats
        return out
    if transform is not None:
        local = transform(local)
    if not gather:
        return local
    return all_gather_rows(local, n_items, group)
