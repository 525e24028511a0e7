#!/usr/bin/env python3
"""bench.py -- units/sec mapped by the MI355X gmapper hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5] [--reads-per-step R]

A step = one pass of the hot path (seed lookup -> windows -> vector SW -> full SW -> SAM records) over one batch of R synthetic reads / pairs.
Workloads:
  cfg3 (default) 100 bp letter-space reads vs the 24-contig 3.0 Gbp genome (BASELINE configs[2]: the configuration the metric is quoted on; its
                 38.7 GB index fits one GPU, so it is the workload at every N); reads already resident in HBM (gm_map_reads_device)
  cfg2           100 bp reads vs 4 x 25 Mbp (BASELINE configs[1]); reads resident in HBM
  cfg4           50-colour SOLiD reads (one indel, 4 % colour errors) vs the 3.0 Gbp genome (BASELINE configs[3]): sw_full_cs + post_sw path
  cfg5           2 x 150 bp opp-in pairs, -I 100,600, vs the 3.0 Gbp genome (BASELINE configs[4]): paired-mode path; unit = pairs/s
  (cfg4 / cfg5 go through the host-buffer entry points: packed reads are uploaded every step, 26 / 152 bytes per unit.)
N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL); rank 0 builds the index, its arrays are broadcast once over RCCL (no per-step
collective); every rank maps its own read shard (weak scaling: R units per rank per step).  Prints ONE JSON line on rank 0.  The ranks come either from
the caller (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: WORLD_SIZE must then equal --gpus) or from this script: a plain
`python bench.py --gpus N` starts them itself as a CHILD torch.distributed.run (launch_ranks; the parent never touches the GPU and exits with the child's code).
The default N = 1 line also carries `other_workloads`: cfg4 and cfg5, three steps each, with their own roofline and parity-sample fields.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np

METRIC = "reads/sec mapped (whole node), 100bp LS reads vs 3Gbp ref, 1/2/4/8 GPUs"
WORKLOADS = {
    # name: (kind, genome cfg, genome seed, read length, read seed, default units per step, metric, unit, description)
    "cfg3": ("ls", "cfg3", 3, 100, 3, 1_000_000, METRIC, "reads/s", "100bp LS reads vs 24-contig 3.0Gbp uniform genome (BASELINE configs[2]), default 3 seeds w12"),
    "cfg2": ("ls", "cfg2", 2, 100, 2, 1_000_000, METRIC.replace("3Gbp", "100Mbp"), "reads/s", "1M x 100bp LS reads vs 4x25Mbp uniform genome (BASELINE configs[1]), default 3 seeds w12"),
    "cfg1": ("ls", "cfg1", 12345, 36, 12345, 100_000, METRIC.replace("100bp", "36bp").replace("3Gbp", "1Mbp"), "reads/s", "36bp LS reads vs 1Mbp (BASELINE configs[0])"),
    "cfg4": ("cs", "cfg3", 3, 50, 4, 1_000_000, "reads/sec mapped (whole node), 50-colour CS reads vs 3Gbp ref", "reads/s",
             "50-colour SOLiD reads (1 indel, 4% colour errors) vs 24-contig 3.0Gbp genome (BASELINE configs[3]), sw_full_cs + post_sw, default CS seeds"),
    "cfg5": ("pairs", "cfg3", 3, 150, 5, 131_072, "pairs/sec mapped (whole node), 2x150bp LS pairs vs 3Gbp ref", "pairs/s",
             "2x150bp opp-in pairs, -I 100,600, vs 24-contig 3.0Gbp genome (BASELINE configs[4]), half-paired rescue on"),
}


def shard_seed(base: int, rank: int, n_pool: int, b: int) -> int:
    """distinct synthetic batch per (rank, pool slot): what makes the N-GPU run weak scaling over different reads"""
    return base + 7919 * (rank * n_pool + b)


def reduce_max_time(dt: float, world: int, dist=None, device=None) -> float:
    """the job's step time = the slowest rank's (all_reduce MAX); a no-op at world 1"""
    if world <= 1:
        return dt
    import torch
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def host_threads_for(local_world: int) -> int:
    """host threads per rank for the finalisation (selection, MAPQ, SAM text): this rank's share of the cores the job may run on"""
    try: n_cores = len(os.sched_getaffinity(0))
    except Exception: n_cores = os.cpu_count() or 1
    return max(1, min(32, n_cores // max(1, local_world)))      # (not cut to the CPU quota: the finalisation comes in bursts, see gm_usable_cores in gm_host.hip)


def usable_cores() -> int:
    """the cores this job may really use: the affinity mask, cut to the container's CPU quota where there is one (cgroup v2 cpu.max, v1 cfs quota) -- a one-GPU box of
    the pool shows 256 processors and grants 16"""
    try: n = len(os.sched_getaffinity(0))
    except Exception: n = os.cpu_count() or 1
    quota = period = -1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": quota, period = int(q), int(p)
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except Exception: pass
    if quota > 0 and period > 0: n = min(n, max(1, -(-quota // period)))
    return n


def launch_ranks(n: int, argv, capture: bool = False, timeout=None):
    """Start `n` ranks of this script on this node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P bench.py argv`
    as a child process (never an exec: the caller may not have initialised HIP, and must not be replaced once it has).  Returns the child's exit code
    (with capture: (code, stdout, stderr)).  The reference's analogue is the OpenMP chunk loop of launch_scan_threads (ref: gmapper/gmapper.c:322-345,588-607)."""
    import socket, subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
    if capture:
        p = subprocess.run(cmd, env=env, capture_output=True, timeout=timeout)
        return p.returncode, p.stdout.decode(errors="replace"), p.stderr.decode(errors="replace")
    return subprocess.run(cmd, env=env, timeout=timeout).returncode


def selftest_ranks(steps: int = 3, units: int = 1000):
    """The N > 1 logic of main() without a GPU (gloo): per-rank seeds, barrier-bracketed timing, MAX over ranks, rank-0-only JSON.
    A step is a sleep that grows with the rank, so that the slowest rank decides.  Used by tests/test_dist_gloo.py."""
    import torch, torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1: dist.init_process_group("gloo", rank=rank, world_size=world)
    n_pool = 2
    seeds = [shard_seed(3, rank, n_pool, b) for b in range(n_pool)]
    per_step = 0.02 * (rank + 1)
    if world > 1: dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps): time.sleep(per_step)
    if world > 1: dist.barrier()
    dt = reduce_max_time(time.perf_counter() - t0, world, dist, torch.device("cpu"))
    allseeds = [None] * world; allstep = [None] * world; allthr = [None] * world
    if world > 1:
        dist.all_gather_object(allseeds, seeds); dist.all_gather_object(allstep, per_step); dist.all_gather_object(allthr, host_threads_for(local_world))
    else:
        allseeds, allstep, allthr = [seeds], [per_step], [host_threads_for(local_world)]
    if rank == 0:
        print(json.dumps({"selftest": True, "value": units * steps * world / dt, "n_gpus": world, "steps": steps, "ms_per_step": 1e3 * dt / steps, "scaling": "weak",
                          "seeds": allseeds, "rank_step_s": allstep, "units_per_rank_step": units, "host_threads_per_rank": host_threads_for(local_world),
                          "host_threads": allthr}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def make_pools(synth, contigs, kind, R, L, rseed, rank, n_pool, dev):
    """n_pool distinct synthetic batches for this rank (letter space: packed and resident in HBM; colour space / pairs: packed host buffers)"""
    import torch
    pools = []
    for b in range(n_pool):
        sd = shard_seed(rseed, rank, n_pool, b)
        if kind == "ls":
            reads, _ = synth.make_reads(contigs, R, L, sd)
            pools.append(torch.from_numpy(synth.pack_reads(reads).view(np.int32)).to(dev))          # resident in HBM
        elif kind == "cs":
            reads, _ = synth.make_cs_reads(contigs, R, L, sd)
            pools.append((np.ascontiguousarray(synth.pack_reads(np.ascontiguousarray(reads[:, 1:]))), np.ascontiguousarray(reads[:, 0])))
        else:
            reads, _ = synth.make_pairs(contigs, R, L, sd)
            pools.append((np.ascontiguousarray(synth.pack_reads(np.ascontiguousarray(reads[0::2]))), np.ascontiguousarray(synth.pack_reads(np.ascontiguousarray(reads[1::2])))))
    return pools


def make_sample(synth, contigs, kind, n, L, rseed):
    if kind == "ls": return synth.make_reads(contigs, n, L, rseed + 104729)[0]
    if kind == "cs": return synth.make_cs_reads(contigs, n, L, rseed + 104729)[0]
    return synth.make_pairs(contigs, n, L, rseed + 104729)[0]


def timed_steps(gm, sess, kind, pools, R, L, steps, warmup, world, dist, dev, emit_sam=True):
    """W untimed steps, then EXACTLY `steps` timed ones between barrier + synchronize on both sides; the job's time is the slowest rank's.
    Returns (seconds, K1 event milliseconds, K1 algorithmic bytes, K1 launches, summed stats)."""
    import torch
    popts = gm.PairOpts.default("opp-in", 100, 600)
    n_pool = len(pools)

    def step(i):
        p = pools[i % n_pool]
        if kind == "ls": sess.map_device(p.data_ptr(), R, L, emit_sam=emit_sam, return_bytes=False)
        elif kind == "cs": sess.map_cs_packed(p[0], p[1], R, L, return_bytes=False)
        else: sess.map_pairs_packed(p[0], p[1], R, L, L, popts, return_bytes=False)
        return sess.stats

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    lk_ms = 0.0; lk_bytes = 0; lk_launch = 0
    agg = {}
    t0 = time.perf_counter()
    for i in range(steps):
        st = step(warmup + i)
        ms, nb, nl = sess.lookup_timing()                    # K1's own HIP events on its stream (paired mode: both mate sets of every sub-batch)
        lk_ms += ms; lk_bytes += (nb or st["list_bytes"]); lk_launch += nl
        for k, v in st.items():
            agg[k] = agg.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = reduce_max_time(time.perf_counter() - t0, world, dist, dev)
    return dt, lk_ms, lk_bytes, lk_launch, agg


def roofline_of(gm, lk_ms, lk_bytes, lk_launch, units):
    """dominant kernel = seed lookup (K1): algorithmic bytes (12 B per lookup + 4 B per list entry) / its own HIP-event time"""
    ach = (lk_bytes / 1e9) / (lk_ms / 1e3) if lk_ms > 0 else 0.0
    kname = gm.lib().gm_last_lookup_kernel().decode()       # k_lookup_bkt (one slab, short lists), k_lookup_v4 (folded count), k_lookup_v3 (slab sweep), k_lookup_v5
    return {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
            "traffic": None,       # HBM bytes are not measurable inside this run; see traffic_from_profile
            "kernel": kname, "launches": lk_launch, "avg_launch_ms": lk_ms / max(1, lk_launch), "alg_bytes_per_launch": lk_bytes / max(1, lk_launch),
            "alg_bytes_per_unit": lk_bytes / max(1, units)}


def oracle_sample(oa, o, kind, sample, ncores):
    """the CPU restatement on a bounded sample: (seconds, SAM text)"""
    t0 = time.perf_counter()
    if kind == "pairs":
        o.set_pairing("opp-in", 100, 600); sam = o.map_pairs_sam(sample[0::2], sample[1::2], nthreads=ncores)
    else:
        sam = o.map_sam(sample, nthreads=ncores)
    return time.perf_counter() - t0, sam


def product_sample(sess, kind, sample):
    if kind == "ls": return sess.map_reads(sample)
    if kind == "cs": return sess.map_reads_cs(sample)
    return sess.map_pairs(sample[0::2], sample[1::2], mode="opp-in", min_insert=100, max_insert=600)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("GM_BENCH_WORKLOAD", "cfg3"), choices=sorted(WORKLOADS))
    ap.add_argument("--reads-per-step", type=int, default=0, help="units (reads, or pairs for cfg5) per rank per step; 0 = the workload's default")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genome (debugging only; makes the result invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units in the CPU-baseline sample; 0 = the workload's default (10-30 s of CPU work)")
    ap.add_argument("--no-sam", action="store_true", help="skip SAM text emission on the host (alignment records only; letter-space unpaired only)")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the cfg4 / cfg5 entries of the default line")
    ap.add_argument("--selftest-ranks", action="store_true", help="rank logic only (gloo, sleeps instead of GPU steps): what tests/test_dist_gloo.py drives through launch_ranks")
    args = ap.parse_args()

    have_ranks = "WORLD_SIZE" in os.environ
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not have_ranks and args.gpus > 1:
        # plain `python bench.py --gpus N`: start the N ranks as a child job.  This process makes no GPU call of any kind (not even a device count): a rank without
        # a GPU fails by itself ("local rank k has no GPU") and the child's exit code is ours.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: the launcher's rank count and --gpus must agree (n_gpus in the JSON is the number of ranks that ran)" % (args.gpus, world))
    if args.selftest_ranks:
        return selftest_ranks()

    # The library measured: the release build (make release: no tuning knobs compiled in) when it is there -- unless the caller asks for a build (GM_LIB_PATH) or sets
    # one of the kernel-variant knobs, which only the tuning build reads.
    rel = os.path.join(ROOT, "shrimp_amd", "libgmapper_hip_release.so")
    knobs = [k for k in os.environ if k.startswith("GM_") and k not in ("GM_HOST_THREADS", "GM_SUBBATCH", "GM_BENCH_WORKLOAD", "GM_CPU_THREADS", "GM_LIB_PATH")]
    if "GM_LIB_PATH" not in os.environ and not knobs and os.path.exists(rel):
        os.environ["GM_LIB_PATH"] = rel
    import torch
    import torch.distributed as dist
    from shrimp_amd import gmapper as gm, synth

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d has no GPU (%d visible)" % (rank, local, torch.cuda.device_count()))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        if dist.get_world_size() != args.gpus:
            raise SystemExit("only %d of %d ranks came up" % (dist.get_world_size(), args.gpus))
    if gm.lib().gm_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host threads per rank for the finalisation (selection, MAPQ, SAM text): this rank's share of the host cores
    host_threads = host_threads_for(local_world)
    os.environ.setdefault("GM_HOST_THREADS", str(host_threads))
    host_threads = int(os.environ["GM_HOST_THREADS"])

    kind, gname, gseed, L, rseed, R_def, metric, unit, descr = WORKLOADS[args.workload]
    R = args.reads_per_step or R_def
    # ---- synthetic inputs (same generators as the parity fixtures) ----
    t0 = time.time()
    contigs = synth.make_genome(synth.contig_lengths(gname, args.scale), gseed)
    t_gen = time.time() - t0
    n_pool = min(10, args.steps + args.warmup)           # distinct batches, cycled: 10 x 1 M reads = the 10 M reads of BASELINE configs[2] (round 3 cycled two)
    pools = make_pools(synth, contigs, kind, R, L, rseed, rank, n_pool, dev)
    n_sample = args.cpu_sample or {"ls": 200_000, "cs": 50_000, "pairs": 20_000}[kind]
    sample = make_sample(synth, contigs, kind, n_sample, L, rseed)

    # ---- index: built on rank 0's GPU, broadcast once ----
    params = gm.default_params_cs() if kind == "cs" else gm.default_params()
    t0 = time.time()
    ix = gm.Index(contigs, device=local, params=params) if rank == 0 else None
    t_index = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        from shrimp_amd import parallel
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        ix = parallel.broadcast_index(ix, rank, dev, src=0)      # the single collective of the whole job
        torch.cuda.synchronize(); dist.barrier()
        t_bcast = time.time() - t0
    sub_batch = lambda k: int(os.environ.get("GM_SUBBATCH", "131072" if k != "pairs" else "65536"))
    sess = gm.Session(ix, params=params, max_batch_reads=sub_batch(kind))

    dt, lk_ms, lk_bytes, lk_launch, agg = timed_steps(gm, sess, kind, pools, R, L, args.steps, args.warmup, world, dist, dev, emit_sam=not args.no_sam)
    total_units = R * args.steps * world
    value = total_units / dt
    threads_all = [host_threads]
    if world > 1:
        threads_all = [None] * world
        dist.all_gather_object(threads_all, host_threads)

    out = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {"workload": descr, ("pairs_per_step_per_gpu" if kind == "pairs" else "reads_per_step_per_gpu"): R, "read_len": L,
                   "genome_bp": int(sum(len(c) for c in contigs)),
                   "parallelism": "read-sharded x%d, index replicated (1 RCCL broadcast at start-up)" % world,
                   "inputs": "reads resident in HBM" if kind == "ls" else "packed reads in host buffers, uploaded every step", "distinct_batches": n_pool,
                   "library": os.path.basename(gm.LIB_PATH) + ("" if not knobs else " (tuning knobs set: %s)" % ",".join(sorted(knobs))),
                   "sam_emitted": not args.no_sam, "scale": args.scale, "host_threads_per_rank": host_threads, "host_threads": threads_all,
                   "sub_batch_pipeline": "stage order" if os.environ.get("GM_OVERLAP") == "0" else "two streams (lookup of sub-batch i+1 beside SW of sub-batch i)",
                   "vector_sw_filter": ("every window swept to its end" if os.environ.get("GM_P1_EARLY") == "0" else
                                        ("the mates' own pass (pair sums) sweeps every window; the unpaired pass behind it stops a window once no alignment can reach the vector threshold" if kind == "pairs" else
                                         "a window stops once no alignment can reach the vector threshold (exact bound, same SAM; DESIGN.md section 4, K3)"))},
    }
    if rank == 0:
        U = R * args.steps
        out["roofline"] = roofline_of(gm, lk_ms, lk_bytes, lk_launch, U)
        kname = out["roofline"]["kernel"]
        try:      # a stored figure from separate rocprofv3 --pmc passes (profiles/traffic.json), NOT something this run measured
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            t = tj.get("%s/%s/131072" % (args.workload, kname))
            if t and args.scale == 1.0:
                out["roofline"]["traffic_from_profile"] = {"bytes_per_131072_read_launch": t["bytes_per_launch"], "source": t["source"]}
        except Exception:
            pass
        out["stages_ms_per_step"] = {k: agg[k] / args.steps for k in agg if k.startswith("ms_")}
        out["per_unit"] = {"lookups": agg["lookups"] / U, "list_entries": agg["list_entries"] / U, "alg_bytes": agg["list_bytes"] / U, "survivors": agg["survivors"] / U,
                           "survivors_pruned": agg["survivors_pruned"] / U, "vec_sw_calls": agg["vec_calls"] / U, "full_sw_calls": agg["full_calls"] / U,
                           "mapped_frac": agg["reads_matched"] / U, "exact_order_frac": agg["exact_order_reads"] / (2 * U),
                           "post_sw_host_redo": agg.get("post_sw_host_redo", 0) / U, "mp_unfiltered": agg.get("mp_unfiltered", 0) / U}
        out["setup_s"] = {"genome_gen": t_gen, "index_build": t_index, "index_bcast": t_bcast, "index_bytes": ix.nbytes}
        o_ls = None
        if not args.no_cpu_baseline and world == 1:
            # the CPU restatement (oracle, "port") on this box's host cores over a bounded sample of the same workload -- on as many threads as the GPU run's
            # host side used (this process's CPU affinity share), so that the two numbers sit on the same cores
            from tests import oracle_api as oa
            ncores = int(os.environ.get("GM_CPU_THREADS", min(host_threads, usable_cores())))      # the CPU leg runs flat out: its threads = the cores the container grants
            oa.load().gmo_set_threads(ncores)
            t0 = time.time(); o = oa.Session(contigs, opts="colour=1" if kind == "cs" else None); t_oidx = time.time() - t0
            cdt, sam = oracle_sample(oa, o, kind, sample, ncores)
            if kind == "cs": o.close()
            else: o_ls = o                                   # the letter-space oracle index also serves the cfg5 entry below
            out["cpu_baseline"] = {"value": n_sample / cdt, "unit": unit, "cores": ncores, "gpu_run_host_threads": host_threads, "cpu_quota_cores": usable_cores(), "kind": "port",
                                   "sample": "%d %s of the same workload (same genome, same error model), oracle/gm_oracle.hpp with OpenMP over reads; "
                                             "index build %.1fs not included" % (n_sample, "pairs" if kind == "pairs" else "reads", t_oidx)}
            # the reference binary itself does not travel; its speed relative to the port was measured in the build container
            try:
                rb = json.load(open(os.path.join(ROOT, "profiles", "r03_ref_baseline.json")))
                key = ("cfg3_group" if gname == "cfg3" else "cfg2") if (kind == "ls" and L == 100) else ("cfg1" if kind == "ls" else "")
                c = rb["cases"].get(key) or (rb["cases"].get("cfg2") if key == "cfg3_group" else None)
                if c:
                    out["cpu_baseline"]["ref_ratio"] = c["ref_over_oracle"]
                    out["cpu_baseline"]["ref_ratio_measured_at"] = "%d bp genome, %d x %d bp reads, %d threads" % (c["genome_bp"], c["reads"], c["read_len"], rb["threads"])
                    out["cpu_baseline"]["ref_ratio_source"] = c.get("what", "reference gmapper-ls vs the port on the same reads") + "; " + rb["host"] + " (profiles/r03_ref_baseline.json)"
                    out["cpu_baseline"]["reference_equivalent"] = out["cpu_baseline"]["value"] * c["ref_over_oracle"]
                else:
                    out["cpu_baseline"]["ref_ratio"] = None
            except Exception:
                out["cpu_baseline"]["ref_ratio"] = None
            # parity spot check on the sample while we are here
            out["cpu_baseline"]["sample_sam_identical"] = bool(product_sample(sess, kind, sample) == sam)
        # ---- the other single-GPU configurations, three steps each, so that their numbers are measured by whoever runs the default line ----
        if world == 1 and args.workload == "cfg3" and args.scale == 1.0 and not args.no_other_workloads and not args.no_sam:
            others = {}
            for wname in ("cfg5", "cfg4"):
                k2, _, _, L2, rs2, R2, m2, u2, d2 = WORKLOADS[wname]
                try:
                    p2 = gm.default_params_cs() if k2 == "cs" else params
                    t0 = time.time()
                    ix2 = gm.Index(contigs, device=local, params=p2) if k2 == "cs" else ix
                    t_ix2 = time.time() - t0
                    s2 = gm.Session(ix2, params=p2, max_batch_reads=sub_batch(k2))
                    pools2 = make_pools(synth, contigs, k2, R2, L2, rs2, 0, 2, dev)
                    dt2, ms2, by2, nl2, agg2 = timed_steps(gm, s2, k2, pools2, R2, L2, 3, 1, 1, dist, dev)
                    e = {"metric": m2, "unit": u2, "workload": d2, "value": R2 * 3 / dt2, "steps": 3, "warmup": 1, "ms_per_step": 1e3 * dt2 / 3,
                         "roofline": {k: v for k, v in roofline_of(gm, ms2, by2, nl2, R2 * 3).items() if k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "alg_bytes_per_unit")},
                         "stages_ms_per_step": {k: agg2[k] / 3 for k in agg2 if k.startswith("ms_")}}
                    if k2 == "cs": e["index_build_s"] = t_ix2
                    if not args.no_cpu_baseline:
                        from tests import oracle_api as oa
                        ns2 = {"cs": 20_000, "pairs": 5_000}[k2]
                        smp2 = make_sample(synth, contigs, k2, ns2, L2, rs2)
                        o2 = oa.Session(contigs, opts="colour=1") if k2 == "cs" else (o_ls or oa.Session(contigs))
                        cdt2, sam2 = oracle_sample(oa, o2, k2, smp2, ncores)
                        if o2 is not o_ls: o2.close()
                        e["cpu_port_value"] = ns2 / cdt2; e["cpu_cores"] = ncores; e["sample_units"] = ns2
                        e["sample_sam_identical"] = bool(product_sample(s2, k2, smp2) == sam2)
                    s2.close()
                    if ix2 is not ix: ix2.close()
                    del pools2
                    others[wname] = e
                except Exception as ex:                      # the headline must not be lost to a problem in a side entry: say what happened instead
                    others[wname] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            # configs[1] (100 Mbp) has a genome and an index of its own -- small ones: k_lookup_bkt with real Poisson(6) lists
            try:
                k2, g2, gs2, L2, rs2, R2, m2, u2, d2 = WORKLOADS["cfg2"]
                t0 = time.time(); contigs2 = synth.make_genome(synth.contig_lengths(g2, 1.0), gs2); ix2 = gm.Index(contigs2, device=local, params=params); t_ix2 = time.time() - t0
                s2 = gm.Session(ix2, params=params, max_batch_reads=sub_batch(k2))
                pools2 = make_pools(synth, contigs2, k2, R2, L2, rs2, 0, 2, dev)
                dt2, ms2, by2, nl2, agg2 = timed_steps(gm, s2, k2, pools2, R2, L2, 3, 1, 1, dist, dev)
                e = {"metric": m2, "unit": u2, "workload": d2, "value": R2 * 3 / dt2, "steps": 3, "warmup": 1, "ms_per_step": 1e3 * dt2 / 3,
                     "roofline": {k: v for k, v in roofline_of(gm, ms2, by2, nl2, R2 * 3).items() if k in ("kernel", "achieved", "frac", "avg_launch_ms", "launches", "alg_bytes_per_unit")},
                     "stages_ms_per_step": {k: agg2[k] / 3 for k in agg2 if k.startswith("ms_")}, "genome_and_index_s": t_ix2}
                if not args.no_cpu_baseline:
                    from tests import oracle_api as oa
                    smp2 = make_sample(synth, contigs2, k2, 50_000, L2, rs2)
                    o2 = oa.Session(contigs2); cdt2, sam2 = oracle_sample(oa, o2, k2, smp2, ncores); o2.close()
                    e["cpu_port_value"] = 50_000 / cdt2; e["cpu_cores"] = ncores; e["sample_units"] = 50_000
                    e["sample_sam_identical"] = bool(product_sample(s2, k2, smp2) == sam2)
                s2.close(); ix2.close(); del pools2, contigs2
                others["cfg2"] = e
            except Exception as ex:
                others["cfg2"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            # cfg3 end to end from TEXT: the same 1 M reads as FASTA-style lines through gm_map_reads_text -- parsing, packing to 4-bit codes and the upload are inside the
            # timed region, as they are inside the reference's "Read Mapping Time" (ref: gmapper.c:322-398,800-804)
            try:
                reads_t, _ = synth.make_reads(contigs, R_def, L, shard_seed(rseed, 0, 10, 0))
                lut = np.frombuffer(b"ACGT", dtype=np.uint8)
                lines = np.empty((R_def, L + 1), dtype=np.uint8); lines[:, :L] = lut[reads_t]; lines[:, L] = 10
                text = lines.tobytes()[:-1]; del lines, reads_t
                sess.map_text_buffer(text, R_def, L, return_bytes=False)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3): sess.map_text_buffer(text, R_def, L, return_bytes=False)
                torch.cuda.synchronize(); dt3 = time.perf_counter() - t0
                others["cfg3_text"] = {"metric": metric, "unit": unit, "workload": descr + "; input = text lines (gm_map_reads_text: parse + pack + upload inside the timed region)",
                                       "value": R_def * 3 / dt3, "steps": 3, "warmup": 1, "ms_per_step": 1e3 * dt3 / 3, "text_bytes_per_step": len(text)}
                # ... and from a FASTA FILE through the streaming entry (gm_map_reads_file_cb: the reference's reader, chunks of 2^20 reads read and preprocessed by a second
                # thread while the chunk before is mapped, records handed to a write function): 2 M reads, so that two chunks overlap
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    fpath = os.path.join(td, "reads.fa")
                    nm = np.char.add(np.char.add(">r", np.arange(R_def).astype("U8")), "\n").astype("S")
                    body = text.split(b"\n")
                    with open(fpath, "wb") as f:
                        for rep in range(2): f.write(b"".join(a + b + b"\n" for a, b in zip(nm.tolist(), body)))
                    del nm, body
                    sess.map_reads_file_chunks(fpath, collect=False)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    parts = sess.map_reads_file_chunks(fpath, collect=False)
                    torch.cuda.synchronize(); dtf = time.perf_counter() - t0
                    others["cfg3_file"] = {"metric": metric, "unit": unit, "workload": descr + "; input = a FASTA file of 2 M reads (gm_map_reads_file_cb: read + parse + preprocess + pack + upload inside the timed region)",
                                           "value": 2 * R_def / dtf, "steps": 1, "warmup": 1, "ms_per_step": 1e3 * dtf, "chunks": len(parts), "file_bytes": os.path.getsize(fpath), "sam_bytes": int(sum(parts))}
                del text
            except Exception as ex:
                others["cfg3_text"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            out["other_workloads"] = others
            # how the step depends on the host threads a rank gets (an 8-rank job on this node leaves each rank cores / 8): the headline workload and the host-bound
            # 100 Mbp one again at 16 and 8 threads, three steps each (the library reads GM_HOST_THREADS at every call)
            try:
                hs = {}
                keep = os.environ.get("GM_HOST_THREADS")
                contigs2 = synth.make_genome(synth.contig_lengths("cfg2", 1.0), 2); ix2 = gm.Index(contigs2, device=local, params=params)
                s2 = gm.Session(ix2, params=params, max_batch_reads=sub_batch("ls"))
                pools2 = make_pools(synth, contigs2, "ls", 1_000_000, 100, 2, 0, 2, dev)
                for nt in (32, 16, 8):
                    if nt > host_threads and nt != 32: continue
                    os.environ["GM_HOST_THREADS"] = str(min(nt, host_threads))
                    d3 = timed_steps(gm, sess, kind, pools[:2], R, L, 3, 1, 1, dist, dev)
                    d2 = timed_steps(gm, s2, "ls", pools2, 1_000_000, 100, 3, 1, 1, dist, dev)
                    hs[str(min(nt, host_threads))] = {"cfg3_reads_per_s": R * 3 / d3[0], "cfg3_ms_host": d3[4]["ms_host"] / 3, "cfg2_reads_per_s": 3e6 / d2[0], "cfg2_ms_host": d2[4]["ms_host"] / 3}
                if keep is None: os.environ.pop("GM_HOST_THREADS", None)
                else: os.environ["GM_HOST_THREADS"] = keep
                s2.close(); ix2.close(); del pools2, contigs2
                out["host_sensitivity"] = hs
            except Exception as ex:
                out["host_sensitivity"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if o_ls is not None: o_ls.close()
        print(json.dumps(out))
    sess.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
