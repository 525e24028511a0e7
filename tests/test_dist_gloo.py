"""N > 1 path on CPU: world_size-2 gloo.  Every rank maps its shard (here with the CPU oracle standing
in for the device pipeline -- the multi-GPU layer never touches alignments), rank 0 re-serialises the
shards in input order and must reproduce the unsharded output byte for byte; the start-up broadcast of
index metadata + arrays is exercised with host tensors."""
import os, socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from tests import oracle_api as oa
from shrimp_amd import parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs, reads, _ = oa.load_golden("stress_60bp")
    # start-up broadcast: rank 0 owns the "index arrays", everyone else receives them
    meta = [{"n": int(sum(len(c) for c in contigs))} if rank == 0 else None]
    dist.broadcast_object_list(meta, src=0)
    arr = torch.from_numpy(np.concatenate(contigs).copy()) if rank == 0 else torch.zeros(meta[0]["n"], dtype=torch.uint8)
    for o in range(0, arr.numel(), 1 << 16):
        dist.broadcast(arr[o:o + (1 << 16)], src=0)
    assert int(arr.sum()) == int(np.concatenate(contigs).sum())
    lo, hi = parallel.shard_bounds(len(reads), world)[rank]
    s = oa.Session(contigs)
    local = s.map_sam(reads[lo:hi], nthreads=2)
    # names are positional inside a call; re-base them to global read indices like the host does
    local = b"".join(b"r%d\t" % (lo + int(l.split(b"\t", 1)[0][1:])) + l.split(b"\t", 1)[1] + b"\n" for l in local.split(b"\n") if l)
    merged = parallel.gather_ordered(local, rank, world)
    if rank == 0:
        whole = s.map_sam(reads, nthreads=2)
        q.put(merged == whole)
    s.close()
    dist.barrier(); dist.destroy_process_group()


def test_world2_sharded_equals_unsharded():
    oa.load()   # build the oracle once, before forking
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get() is True
