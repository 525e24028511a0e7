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
N > 1: launched by torch.distributed.run, one rank per GPU; rank 0 builds the index, its arrays are broadcast once over RCCL (no per-step
collective); every rank maps its own read shard (weak scaling: R units per rank per step).  Prints ONE JSON line on rank 0.
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
    "cfg4": ("cs", "cfg3", 3, 50, 4, 500_000, "reads/sec mapped (whole node), 50-colour CS reads vs 3Gbp ref", "reads/s",
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
    return max(1, min(32, n_cores // max(1, local_world)))


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
    allseeds = [None] * world; allstep = [None] * world
    if world > 1:
        dist.all_gather_object(allseeds, seeds); dist.all_gather_object(allstep, per_step)
    else:
        allseeds, allstep = [seeds], [per_step]
    if rank == 0:
        print(json.dumps({"selftest": True, "value": units * steps * world / dt, "n_gpus": world, "steps": steps, "ms_per_step": 1e3 * dt / steps, "scaling": "weak",
                          "seeds": allseeds, "rank_step_s": allstep, "units_per_rank_step": units, "host_threads_per_rank": host_threads_for(local_world)}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GM_BENCH_WORKLOAD", "cfg3"), choices=sorted(WORKLOADS))
    ap.add_argument("--reads-per-step", type=int, default=0, help="units (reads, or pairs for cfg5) per rank per step; 0 = the workload's default")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genome (debugging only; makes the result invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="units in the CPU-baseline sample; 0 = the workload's default (10-30 s of CPU work)")
    ap.add_argument("--no-sam", action="store_true", help="skip SAM text emission on the host (alignment records only; letter-space unpaired only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from shrimp_amd import gmapper as gm, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if gm.lib().gm_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # host threads per rank for the finalisation (selection, MAPQ, SAM text): this rank's share of the host cores
    host_threads = host_threads_for(local_world)
    os.environ.setdefault("GM_HOST_THREADS", str(host_threads))

    kind, gname, gseed, L, rseed, R_def, metric, unit, descr = WORKLOADS[args.workload]
    R = args.reads_per_step or R_def
    # ---- synthetic inputs (same generators as the parity fixtures) ----
    t0 = time.time()
    contigs = synth.make_genome(synth.contig_lengths(gname, args.scale), gseed)
    t_gen = time.time() - t0
    n_pool = min(2, args.steps + args.warmup)            # distinct batches, cycled
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
    n_sample = args.cpu_sample or {"ls": 200_000, "cs": 50_000, "pairs": 20_000}[kind]
    if kind == "ls": sample, _ = synth.make_reads(contigs, n_sample, L, rseed + 104729)
    elif kind == "cs": sample, _ = synth.make_cs_reads(contigs, n_sample, L, rseed + 104729)
    else: sample, _ = synth.make_pairs(contigs, n_sample, L, rseed + 104729)

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
    sess = gm.Session(ix, params=params, max_batch_reads=int(os.environ.get("GM_SUBBATCH", "131072" if kind != "pairs" else "65536")))
    popts = gm.PairOpts.default("opp-in", 100, 600)

    def step(i):
        p = pools[i % n_pool]
        if kind == "ls": sess.map_device(p.data_ptr(), R, L, emit_sam=not args.no_sam, return_bytes=False)
        elif kind == "cs": sess.map_cs_packed(p[0], p[1], R, L, return_bytes=False)
        else: sess.map_pairs_packed(p[0], p[1], R, L, L, popts, return_bytes=False)
        return sess.stats

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    lk_ms = 0.0; lk_bytes = 0; lk_launch = 0
    agg = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        st = step(args.warmup + i)
        ms, nb, nl = sess.lookup_timing()                    # K1's own HIP events on its stream (paired mode: both mate sets of every sub-batch)
        lk_ms += ms; lk_bytes += (nb or st["list_bytes"]); lk_launch += nl
        for k, v in st.items():
            agg[k] = agg.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = reduce_max_time(time.perf_counter() - t0, world, dist, dev)
    total_units = R * args.steps * world
    value = total_units / dt

    out = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {"workload": descr, ("pairs_per_step_per_gpu" if kind == "pairs" else "reads_per_step_per_gpu"): R, "read_len": L,
                   "genome_bp": int(sum(len(c) for c in contigs)),
                   "parallelism": "read-sharded x%d, index replicated (1 RCCL broadcast at start-up)" % world,
                   "inputs": "reads resident in HBM" if kind == "ls" else "packed reads in host buffers, uploaded every step",
                   "sam_emitted": not args.no_sam, "scale": args.scale, "host_threads_per_rank": int(os.environ["GM_HOST_THREADS"]),
                   "sub_batch_pipeline": "stage order" if os.environ.get("GM_OVERLAP") == "0" else "two streams (lookup of sub-batch i+1 beside SW of sub-batch i)",
                   "vector_sw_filter": ("every window swept to its end" if (kind == "pairs" or os.environ.get("GM_P1_EARLY") == "0") else
                                        "a window stops once no alignment can reach the vector threshold (exact bound, same SAM; DESIGN.md section 4, K3)")},
    }
    if rank == 0:
        U = R * args.steps
        # dominant kernel = seed lookup (K1): algorithmic bytes (12 B per lookup + 4 B per list entry) / its own HIP-event time
        ach = (lk_bytes / 1e9) / (lk_ms / 1e3) if lk_ms > 0 else 0.0
        kname = gm.lib().gm_last_lookup_kernel().decode()       # k_lookup_bkt (one slab, short lists), k_lookup_v4 (folded count), k_lookup_v3 (slab sweep), k_lookup_v5
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": None,       # HBM bytes are not measurable inside this run; see traffic_from_profile
                           "kernel": kname, "launches": lk_launch, "avg_launch_ms": lk_ms / max(1, lk_launch), "alg_bytes_per_launch": lk_bytes / max(1, lk_launch),
                           "alg_bytes_per_unit": lk_bytes / max(1, U)}
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
                           "mapped_frac": agg["reads_matched"] / U, "exact_order_frac": agg["exact_order_reads"] / (2 * U)}
        out["setup_s"] = {"genome_gen": t_gen, "index_build": t_index, "index_bcast": t_bcast, "index_bytes": ix.nbytes}
        if not args.no_cpu_baseline and world == 1:
            # the CPU restatement (oracle, "port") on this box's host cores over a bounded sample of the same workload
            from tests import oracle_api as oa
            ncores = int(os.environ.get("GM_CPU_THREADS", min(os.cpu_count() or 1, 16)))     # the GPU box gives one GPU job a 16-core share of the host
            oa.load().gmo_set_threads(ncores)
            t0 = time.time(); o = oa.Session(contigs, opts="colour=1" if kind == "cs" else None); t_oidx = time.time() - t0
            t0 = time.perf_counter()
            if kind == "pairs":
                o.set_pairing("opp-in", 100, 600); sam = o.map_pairs_sam(sample[0::2], sample[1::2], nthreads=ncores)
            else:
                sam = o.map_sam(sample, nthreads=ncores)
            cdt = time.perf_counter() - t0
            o.close()
            out["cpu_baseline"] = {"value": n_sample / cdt, "unit": unit, "cores": ncores, "kind": "port",
                                   "sample": "%d %s of the same workload (same genome, same error model), oracle/gm_oracle.hpp with OpenMP over reads; "
                                             "index build %.1fs not included" % (n_sample, "pairs" if kind == "pairs" else "reads", t_oidx)}
            # the reference binary itself cannot travel; its speed relative to the port was measured in the build container
            try:
                rb = json.load(open(os.path.join(ROOT, "profiles", "r02_ref_baseline.json")))
                c = rb["cases"].get("cfg2" if kind == "ls" and L == 100 else ("cfg1" if kind == "ls" else ""))
                if c:
                    out["cpu_baseline"]["ref_ratio"] = c["ref_over_oracle"]
                    out["cpu_baseline"]["ref_ratio_source"] = "reference gmapper-ls -N %d vs the port, %d x %d bp reads vs %d bp, %s (profiles/r02_ref_baseline.json)" % (
                        rb["threads"], c["reads"], c["read_len"], c["genome_bp"], rb["host"])
                    out["cpu_baseline"]["reference_equivalent"] = out["cpu_baseline"]["value"] * c["ref_over_oracle"]
                else:
                    out["cpu_baseline"]["ref_ratio"] = None
            except Exception:
                out["cpu_baseline"]["ref_ratio"] = None
            # parity spot check on the sample while we are here
            if kind == "ls": got = sess.map_reads(sample)
            elif kind == "cs": got = sess.map_reads_cs(sample)
            else: got = sess.map_pairs(sample[0::2], sample[1::2], mode="opp-in", min_insert=100, max_insert=600)
            out["cpu_baseline"]["sample_sam_identical"] = bool(got == sam)
        print(json.dumps(out))
    sess.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
