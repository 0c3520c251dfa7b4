"""Multi-GPU host logic: one process per GPU, partial radiance buffers combined on a root rank.

Two routes to the same exchange step:
  * NativeExchange — the library's own RCCL step (ir_comm_init / ir_combine, csrc/insploray_comm.hip): a grouped reduce for
    the sample-range partition, a gather of owned tiles for the tile partition, on a stream of its own so that the next
    render overlaps it.  torch.distributed (any backend) only carries the 128-byte rendezvous id.  This is what a Rust
    host would call, and what bench.py uses;
  * combine_partials / DistributedFilm — reduce of torch tensors bound as the accumulator (backend "nccl" = RCCL on the
    GPU, "gloo" on CPU tensors in the tests): for hosts that already live in torch.

The reference has no distributed code (its only parallelism is the 4-thread tile pool, renderer.rs:105-159); the
path shards naturally because every pixel-sample is independent and the accumulator is additive
(tile_buffer.rs:24-28, full_buffer.rs:107-125):
  * tile partition  — 64x64 tiles dealt round-robin over ranks; supports are disjoint, so sum == gather exactly;
  * sample partition — every rank renders all pixels for its own sample indices; sum equals the 1-GPU result up to
    f32 summation order.
Both end in the same exchange step: reduce(sum) of the float4 radiance sums and of the u32 sample counts.
"""
import torch
import torch.distributed as dist

from . import _capi

TILE = 64  # renderer.rs:105


def tile_grid(width, height):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tile_owner(tile_index, world):
    """Round-robin tile deal of IR_PARTITION_TILES."""
    return tile_index % world


def owned_pixel_mask(width, height, rank, world):
    """bool (h, w) mask of the pixels rank renders under the tile partition."""
    tx, ty = tile_grid(width, height)
    ys = torch.arange(height).unsqueeze(1) // TILE
    xs = torch.arange(width).unsqueeze(0) // TILE
    return ((ys * tx + xs) % world) == rank


def sample_base(rank, world, spp_per_rank, done_per_rank):
    """First sample index rank renders in a call under IR_PARTITION_SAMPLES (see ir_set_partition)."""
    return done_per_rank * world + rank * spp_per_rank


def combine_partials(sums, counts, dst=0, group=None, all_ranks=False):
    """The exchange step: sum the partial radiance buffers (float32 (h,w,4)) and the sample counts (int32 (h,w))
    onto rank `dst` (or onto every rank).  In place; returns (sums, counts)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return sums, counts
    if all_ranks:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    else:
        dist.reduce(sums, dst=dst, op=dist.ReduceOp.SUM, group=group)
        dist.reduce(counts, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return sums, counts


class DistributedFilm:
    """Torch-owned accumulator bound to a RayTracer so that RCCL reduces it in place (ir_bind_accumulator)."""

    def __init__(self, tracer, width, height, mode, rank=None, world=None):
        self.tracer = tracer
        self.rank = dist.get_rank() if rank is None and dist.is_initialized() else (rank or 0)
        self.world = dist.get_world_size() if world is None and dist.is_initialized() else (world or 1)
        self.sums = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
        self.counts = torch.zeros((height, width), dtype=torch.int32, device="cuda")
        _capi.check(tracer._ctx, _capi.lib.ir_bind_accumulator(tracer._ctx, self.sums.data_ptr(),
                                                              self.counts.data_ptr(), width, height))
        # launch on torch's current stream so that the collective (which torch.distributed orders against that stream)
        # and the next render's accumulator reset cannot overlap
        _capi.check(tracer._ctx, _capi.lib.ir_set_stream(tracer._ctx, torch.cuda.current_stream().cuda_stream))
        tracer.set_partition(mode if self.world > 1 else _capi.IR_PARTITION_NONE, self.rank, self.world)

    def combine(self, dst=0, all_ranks=False):
        return combine_partials(self.sums, self.counts, dst=dst, all_ranks=all_ranks)


class NativeExchange:
    """ir_comm_init + ir_combine for a RayTracer.  `broadcast_id(bytes_or_None) -> bytes` hands rank 0's rendezvous id to
    every rank; the default uses torch.distributed.broadcast_object_list on the default group."""

    def __init__(self, tracer, rank, world, broadcast_id=None):
        import ctypes as C
        self.tracer, self.rank, self.world = tracer, int(rank), int(world)
        buf = C.create_string_buffer(_capi.IR_COMM_ID_BYTES)
        ident = None
        if self.rank == 0:
            # a failure here (no RCCL on this host) must still reach the broadcast below, or the other ranks wait for
            # ever: they receive None and every rank raises together
            if _capi.lib.ir_comm_unique_id(buf) == _capi.IR_OK:
                ident = buf.raw
        if self.world > 1:
            if broadcast_id is None:
                box = [ident]
                dist.broadcast_object_list(box, src=0)
                ident = box[0]
            else:
                ident = broadcast_id(ident)
        if not isinstance(ident, (bytes, bytearray)) or len(ident) != _capi.IR_COMM_ID_BYTES:
            raise ValueError("no rendezvous id of %d bytes (rank 0: %s)" % (
                _capi.IR_COMM_ID_BYTES, (_capi.lib.ir_last_global_error() or b"").decode("utf-8", "replace") or "not received"))
        _capi.check(tracer._ctx, _capi.lib.ir_comm_init(tracer._ctx, bytes(ident), self.rank, self.world))

    def combine(self, root=0):
        """Blocking: when it returns on the root, get_radiance / get_output show the combined film."""
        _capi.check(self.tracer._ctx, _capi.lib.ir_combine(self.tracer._ctx, int(root)))

    def combine_async(self, root=0):
        _capi.check(self.tracer._ctx, _capi.lib.ir_combine_async(self.tracer._ctx, int(root)))

    def wait(self):
        _capi.check(self.tracer._ctx, _capi.lib.ir_combine_wait(self.tracer._ctx))

    def last_ms(self):
        return float(_capi.lib.ir_last_combine_ms(self.tracer._ctx))

    def close(self):
        if getattr(self.tracer, "_ctx", None):
            _capi.lib.ir_comm_destroy(self.tracer._ctx)
