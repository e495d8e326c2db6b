"""Multi-GPU plumbing for the batched many-camera mode.

The path shards naturally: a stereo pair of one camera stream depends on nothing from other streams, so stream
s goes to rank s mod G, whole pairs stay on one GPU (stereo matching needs both pyramids) and NO collective is
needed on the data path.  The only optional exchange is the cross-camera descriptor all-gather of config C5:
every rank contributes fixed-capacity blocks {int32 n; uint8 desc[cap][32]} per image so that every GPU ends
up with all cameras' descriptors.  torch.distributed is used as plumbing only (backend "nccl" = RCCL over xGMI
on the GPUs, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def assign_streams(n_streams, world, rank):
    """Camera streams handled by `rank`: stream s -> rank s mod world (round robin)."""
    return [s for s in range(n_streams) if s % world == rank]


def pack_descriptor_blocks(counts, desc, cap):
    """counts [I] int32, desc [I, cap, 32] uint8 -> one uint8 tensor [I, 4 + cap*32] (count header + rows)."""
    I = counts.shape[0]
    out = torch.zeros((I, 4 + cap * 32), dtype=torch.uint8, device=desc.device)
    out[:, :4] = counts.to(torch.int32).contiguous().view(torch.uint8).reshape(I, 4)
    out[:, 4:] = desc.reshape(I, cap * 32)
    return out


def unpack_descriptor_blocks(blocks, cap):
    """inverse of pack_descriptor_blocks -> (counts [I] int32, desc [I, cap, 32] uint8)"""
    I = blocks.shape[0]
    counts = blocks[:, :4].contiguous().view(torch.int32).reshape(I)
    return counts, blocks[:, 4:].reshape(I, cap, 32)


def allgather_descriptor_blocks(counts, desc, cap, group=None):
    """All-gather every rank's descriptor blocks.  Returns (counts [G*I], desc [G*I, cap, 32]) ordered by rank,
    then by local image index (rank r, image i is camera stream i*G + r under assign_streams)."""
    world = dist.get_world_size(group)
    local = pack_descriptor_blocks(counts, desc, cap)
    out = torch.empty(world * local.numel(), dtype=torch.uint8, device=local.device)  # flat: backend agnostic
    dist.all_gather_into_tensor(out, local.reshape(-1), group=group)
    return unpack_descriptor_blocks(out.reshape(world * local.shape[0], local.shape[1]), cap)


def allgather_members(counts, desc, group=None):
    """The two collectives orbx_allgather_descriptors issues (csrc/orbx_api_comm.hip: ncclGroupStart, ncclAllGather of the
    descriptor rows [I * cap * 32 bytes], ncclAllGather of the counts [I int32], ncclGroupEnd) as torch.distributed calls on the
    same buffers: the block {n, desc[cap][32]} travels as its two members, each rank's slice lands at rank * slice.  Used by the
    world-size-2 gloo test to show that this layout and the packed-block twin above agree, so that on an 8-GPU node the only
    step never exercised is RCCL itself."""
    world = dist.get_world_size(group)
    I, cap = desc.shape[0], desc.shape[1]
    out_d = torch.empty(world * desc.numel(), dtype=torch.uint8, device=desc.device)
    out_c = torch.empty(world * I, dtype=torch.int32, device=counts.device)
    dist.all_gather_into_tensor(out_d, desc.contiguous().reshape(-1), group=group)
    dist.all_gather_into_tensor(out_c, counts.to(torch.int32).contiguous(), group=group)
    return out_c, out_d.reshape(world * I, cap, 32)


class DescriptorExchange:
    """Config C5's exchange step through the C ABI: orbx_allgather_descriptors (one grouped RCCL call straight from the
    handle's result arrays, on the handle's stream) -- the product path on the GPUs.  torch.distributed only ferries the
    128-byte RCCL id from rank 0 to the other ranks at construction; the destination arrays are torch allocations
    (device memory plumbing) reused every step.  allgather_descriptor_blocks above is the torch-only twin the gloo CPU
    tests use to check the block order."""

    def __init__(self, n_images, cap, device, group=None, comm=None):
        """comm: an orbx.Comm to share -- ONE communicator per rank serves all handles (include/orbx.h, ordering rule: every
        rank issues the same gathers in the same order; the C ABI chains them on the device).  None creates it (collective)."""
        import orb_slam3_fast_amd as orbx
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if comm is None:
            uid = [orbx.comm_unique_id() if self.rank == 0 else None]
            if self.world > 1:
                dist.broadcast_object_list(uid, src=0, group=group)
            comm = orbx.Comm(uid[0], self.world, self.rank, device)
        self.comm = comm
        self.n_images, self.cap = n_images, cap
        dev = torch.device("cuda", device)
        self.desc = torch.empty((self.world * n_images, cap, 32), dtype=torch.uint8, device=dev)
        self.counts = torch.empty((self.world * n_images,), dtype=torch.int32, device=dev)

    def gather(self, ex):
        """Enqueue the all-gather of ex's last batch behind its extraction; returns (counts [G*I], desc [G*I, cap, 32]),
        complete once ex's stream reaches this point (ex.sync())."""
        ex.allgather_descriptors(self.comm, self.n_images, self.desc.data_ptr(), self.counts.data_ptr())
        return self.counts, self.desc


def max_over_ranks(seconds, device="cpu", group=None):
    """bench.py contract: the step time is the MAX over ranks."""
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def whole_job_rate(units_per_rank, steps, seconds, group=None):
    """units all ranks processed / max-over-ranks time"""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    return world * units_per_rank * steps / seconds
