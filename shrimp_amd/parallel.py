"""Multi-GPU layer: reads are independent units, so the node-level scheme is
  * one process per GPU (torch.distributed; backend "nccl" == RCCL on ROCm),
  * ONE broadcast of the resident index arrays from the building rank at start-up,
  * contiguous blocks of whole read chunks per rank, no per-step collective,
  * output re-serialised in input order (rank order == chunk order), as the reference's
    chunk-ordered output heap does (ref: gmapper/gmapper.c:588-607).
Nothing here computes alignments.
"""
from __future__ import annotations


def shard_bounds(n_reads: int, world: int, chunk: int = 1000, paired: bool = False) -> list[tuple[int, int]]:
    """Contiguous [lo, hi) block of reads per rank, in whole chunks (ref chunk_size = 1000,
    gmapper-defaults.h:13; made even in paired mode, gmapper.c:2319-2322) so that mates never split."""
    if paired and chunk % 2:
        chunk += 1
    n_chunks = (n_reads + chunk - 1) // chunk
    out = []
    for r in range(world):
        c0 = (n_chunks * r) // world
        c1 = (n_chunks * (r + 1)) // world
        out.append((min(n_reads, c0 * chunk), min(n_reads, c1 * chunk)))
    return out


class _DevArray:
    """__cuda_array_interface__ shim: lets torch view a raw device pointer without copying."""
    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_view(ptr: int, nbytes: int, device):
    """uint8 tensor over `nbytes` of device memory at `ptr` (no copy): what RCCL broadcasts from / into"""
    import torch
    return torch.as_tensor(_DevArray(ptr, nbytes), device=device)


def host_view(ptr: int, nbytes: int, device=None):
    """the same over host memory (the gloo tests drive broadcast_index through this view)"""
    import ctypes, numpy as np, torch
    return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))


def broadcast_index(index, rank: int, device, src: int = 0, chunk_bytes: int = 1 << 30, alloc_like=None, view=device_view):
    """Replicate the index built on `src` to every rank: metadata by object broadcast, then each
    resident array (packed genome, per-seed directory, per-seed positions) by dist.broadcast straight
    from/into HBM.  xGMI is point-to-point, so one ring broadcast of the 39 GB hg-sized index is per-link
    bound (~0.3 s); it happens once.  Returns the local Index.
    `index` needs meta() and device_arrays(); `alloc_like(meta, device)` makes the receiving side (default: gmapper.Index.alloc_like);
    `view(ptr, nbytes, device)` turns one array into a tensor (device_view for HBM; host_view in the CPU tests)."""
    import torch.distributed as dist
    meta = [index.meta() if rank == src else None]
    dist.broadcast_object_list(meta, src=src)
    if rank != src:
        if alloc_like is None:
            from . import gmapper as gm
            alloc_like = gm.Index.alloc_like
        index = alloc_like(meta[0], device=device.index if hasattr(device, "index") else device)
    for ptr, nb in index.device_arrays():
        if not nb or not ptr:
            continue
        t = view(ptr, nb, device)
        for o in range(0, nb, chunk_bytes):
            dist.broadcast(t[o:o + chunk_bytes], src=src)
    return index


def gather_ordered(local: bytes, rank: int, world: int, dst: int = 0):
    """Concatenate per-rank SAM text in rank (= input) order on `dst`; other ranks get None."""
    import torch.distributed as dist
    parts = [None] * world if rank == dst else None
    dist.gather_object(local, parts, dst=dst)
    return b"".join(parts) if rank == dst else None


def contig_groups(contig_len, world: int) -> list[list[int]]:
    """Genome-sharded mode (the reference's SPLIT-DB workflow, SPLITTING_AND_MERGING:25-60, for a genome whose index does not fit one device): contigs
    dealt to `world` groups, longest first onto the lightest group, each group in genome order.  Deterministic, so every rank computes the same deal."""
    order = sorted(range(len(contig_len)), key=lambda i: (-int(contig_len[i]), i))
    load = [0] * world; groups = [[] for _ in range(world)]
    for i in order:
        g = min(range(world), key=lambda k: (load[k], k))
        groups[g].append(i); load[g] += int(contig_len[i])
    return [sorted(g) for g in groups]


def merge_genome_shards(local_sam: bytes, reads_text: bytes, rank: int, world: int, dst: int = 0, **merge_options):
    """Every rank mapped ALL reads against its own contig group (local_sam: header + records, Z fields present); rank `dst` gathers the texts (one
    gather_object, the only exchange of the scheme) and merges them with the mapping qualities recomputed across groups (gm_merge_sam, ref:
    mergesam/mergesam.c).  Other ranks get None.  Give all_contigs=1 when no further merge follows (SPLITTING_AND_MERGING:100-148)."""
    import torch.distributed as dist
    from . import gmapper
    parts = [None] * world if rank == dst else None
    dist.gather_object(local_sam, parts, dst=dst)
    if rank != dst:
        return None
    return gmapper.merge_sam(reads_text, parts, **merge_options)
