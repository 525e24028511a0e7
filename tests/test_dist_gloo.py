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


class _HostIndex:
    """stand-in for gmapper.Index with its arrays in host memory: the same two methods parallel.broadcast_index uses"""
    def __init__(self, arrays, meta_blob):
        self.arrays = arrays; self.blob = meta_blob
    def meta(self): return self.blob
    def device_arrays(self): return [(a.ctypes.data, a.nbytes) for a in self.arrays] + [(0, 0)]       # a kind that is not resident: skipped
    @classmethod
    def alloc_like(cls, meta, device=None):
        import pickle
        shapes = pickle.loads(meta)
        return cls([np.zeros(n, dtype=np.uint8) for n in shapes], meta)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    contigs, reads, _ = oa.load_golden("stress_60bp")
    # start-up broadcast through parallel.broadcast_index itself (host view instead of the HBM view): rank 0 owns the arrays, the others receive them
    import pickle
    if rank == 0:
        arrays = [np.concatenate(contigs).copy(), np.arange(100003, dtype=np.uint32).view(np.uint8).copy(), np.zeros(0, dtype=np.uint8)]
        ix = _HostIndex(arrays, pickle.dumps([a.nbytes for a in arrays]))
    else:
        ix = None
    ix = parallel.broadcast_index(ix, rank, None, src=0, chunk_bytes=1 << 16, alloc_like=_HostIndex.alloc_like, view=parallel.host_view)
    assert int(ix.arrays[0].astype(np.int64).sum()) == int(np.concatenate(contigs).astype(np.int64).sum())
    assert (ix.arrays[1].view(np.uint32) == np.arange(100003, dtype=np.uint32)).all()
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


def _bench_rank_worker(rank, world, port, q):
    """bench.py's rank logic (seeds per rank, barrier-bracketed timing, all_reduce MAX, rank-0-only JSON) without a GPU"""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "LOCAL_WORLD_SIZE": str(world)})
    import io, contextlib, json, sys
    sys.path.insert(0, oa.ROOT)
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bench.selftest_ranks(steps=3)
    q.put((rank, buf.getvalue()))


def test_world2_bench_rank_logic():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    outs = dict(q.get() for _ in range(2))
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    import json
    assert outs[1] == ""                                    # only rank 0 prints
    lines = [l for l in outs[0].split("\n") if l]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["seeds"][0] != d["seeds"][1] and len(set(sum(d["seeds"], []))) == 4          # every (rank, pool slot) draws its own reads
    # the job's time is the slowest rank's (rank 1 sleeps twice as long per step): value = all ranks' units over that time
    assert abs(d["ms_per_step"] - 1e3 * d["rank_step_s"][1]) < 0.5 * 1e3 * d["rank_step_s"][1]
    assert d["ms_per_step"] >= 1e3 * d["rank_step_s"][0]
    assert abs(d["value"] - 2 * d["units_per_rank_step"] / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"]
    assert d["host_threads_per_rank"] >= 1


def _genome_shard_worker(rank, world, port, q):
    import gzip, json, torch.distributed as dist
    from shrimp_amd import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    M = os.path.join(os.path.dirname(__file__), "golden", "merge")
    rd = lambda n: gzip.open(os.path.join(M, n), "rb").read()
    c = json.load(open(os.path.join(M, "cases.json")))["ls_db2"]
    # rank r holds what the device path produces for contig group r (byte-identical to these files: tests/test_gpu_parity.py checks that on the GPU)
    local = rd("ls_db2.in%d.sam.gz" % rank)
    d = c["sets"]["single_best_all"]
    out = parallel.merge_genome_shards(local, rd("ls_db2.reads.gz"), rank, world, command_line=d["command_line"], single_best=1, all_contigs=1, threads=2)
    q.put((rank, None if out is None else out == rd("ls_db2@single_best_all.out.gz")))
    dist.barrier(); dist.destroy_process_group()


def test_world2_genome_sharded_merge():
    """genome-sharded scheme: one contig group per rank, one gather, the merge on rank 0 -- equal to the reference's mergesam on the same shard files"""
    from shrimp_amd import parallel
    assert parallel.contig_groups([160212, 120000, 70, 74200], 2) == [[0, 2], [1, 3]]
    assert parallel.contig_groups([5, 5, 5], 4) == [[0], [1], [2], []]
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_genome_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    res = dict(q.get(timeout=180) for _ in ps)
    for p in ps: p.join(60)
    assert res == {0: True, 1: None}


def test_bench_gpus_n_starts_n_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must itself start two ranks: the launcher function bench.py uses for that (a child
    torch.distributed.run on 127.0.0.1; the parent never touches the GPU) driven here at world 2 with the rank logic on gloo (--selftest-ranks)."""
    import json, subprocess, sys
    sys.path.insert(0, oa.ROOT)
    import bench
    env_backup = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT") if k in os.environ}
    try:
        rc, out, err = bench.launch_ranks(2, ["--gpus", "2", "--selftest-ranks"], capture=True, timeout=300)
        assert rc == 0, err[-2000:]
        lines = [l for l in out.split("\n") if l.startswith("{")]
        assert len(lines) == 1, out                          # rank 0 only
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and len(d["host_threads"]) == 2 and len(d["seeds"]) == 2
        # the plain command line does the same (the parent sees no WORLD_SIZE, --gpus 2 > 1: it launches)
        p = subprocess.run([sys.executable, os.path.join(oa.ROOT, "bench.py"), "--gpus", "2", "--selftest-ranks"], capture_output=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        assert json.loads([l for l in p.stdout.decode().split("\n") if l.startswith("{")][0])["n_gpus"] == 2
        # a launcher whose rank count disagrees with --gpus is an error, not a silent one-GPU run
        p = subprocess.run([sys.executable, os.path.join(oa.ROOT, "bench.py"), "--gpus", "2", "--selftest-ranks"], capture_output=True, timeout=120,
                           env=dict(os.environ, WORLD_SIZE="1", RANK="0"))
        assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr
    finally:
        os.environ.update(env_backup)
