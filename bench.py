#!/usr/bin/env python3
"""bench.py -- reads/sec mapped by the MI355X gmapper hot path (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3] [--reads-per-step R]

A step = one pass of the hot path (seed lookup -> windows -> vector SW -> full SW -> SAM records)
over one batch of R synthetic 100 bp letter-space reads that is already resident in HBM.
Workloads:  cfg3 = 24 contigs / 3.0 Gbp (BASELINE configs[2]: the configuration the metric is quoted on;
                                    its 38.7 GB index fits one GPU, so it is the default at every N),
                             cfg2 = 4 x 25 Mbp uniform genome (BASELINE configs[1]).
N > 1: launched by torch.distributed.run, one rank per GPU; rank 0 builds the index, its arrays
are broadcast once over RCCL (no per-step collective); every rank maps its own read shard (weak
scaling: R reads per rank per step).  Prints ONE JSON line on rank 0.
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("GM_BENCH_WORKLOAD", "cfg3"))
    ap.add_argument("--reads-per-step", type=int, default=1_000_000)
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the genome (debugging only; makes the result invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=200000)
    ap.add_argument("--no-sam", action="store_true", help="skip SAM text emission on the host (alignment records only)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from shrimp_amd import gmapper as gm, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    if gm.lib().gm_device_count() < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- synthetic inputs (same generator as the parity fixtures) ----
    gname, gseed, _, L, rseed = synth.CONFIGS[args.workload]
    t0 = time.time()
    contigs = synth.make_genome(synth.contig_lengths(gname, args.scale), gseed)
    t_gen = time.time() - t0
    R = args.reads_per_step
    n_pool = min(2, args.steps + args.warmup)            # distinct resident batches, cycled
    pools = []
    for b in range(n_pool):
        reads, _ = synth.make_reads(contigs, R, L, rseed + 7919 * (rank * n_pool + b))
        pools.append(torch.from_numpy(synth.pack_reads(reads).view(np.int32)).to(dev))
    sample_reads, _ = synth.make_reads(contigs, args.cpu_sample, L, rseed + 104729)

    # ---- index: built on rank 0's GPU, broadcast once ----
    t0 = time.time()
    ix = gm.Index(contigs, device=local) if rank == 0 else None
    t_index = time.time() - t0
    t_bcast = 0.0
    if world > 1:
        from shrimp_amd import parallel
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.time()
        ix = parallel.broadcast_index(ix, rank, dev, src=0)      # the single collective of the whole job
        torch.cuda.synchronize(); dist.barrier()
        t_bcast = time.time() - t0
    sess = gm.Session(ix, max_batch_reads=int(os.environ.get("GM_SUBBATCH", "131072")))

    def step(i):
        p = pools[i % n_pool]
        sess.map_device(p.data_ptr(), R, L, emit_sam=not args.no_sam, return_bytes=False)
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
        ms, nb, nl = sess.lookup_timing()
        lk_ms += ms; lk_bytes += nb; lk_launch += nl
        for k, v in st.items():
            agg[k] = agg.get(k, 0) + v
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_reads = R * args.steps * world
    value = total_reads / dt

    out = {
        "metric": "reads/sec mapped (whole node), 100bp LS reads vs 3Gbp ref, 1/2/4/8 GPUs",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int16", "data": "synthetic",
        "config": {"workload": {"cfg2": "1M x 100bp LS reads vs 4x25Mbp uniform genome (BASELINE configs[1]), default 3 seeds w12",
                                "cfg3": "100bp LS reads vs 24-contig 3.0Gbp uniform genome (BASELINE configs[2]), default 3 seeds w12",
                                "cfg1": "36bp LS reads vs 1Mbp (BASELINE configs[0])"}[args.workload],
                   "reads_per_step_per_gpu": R, "read_len": L, "genome_bp": int(sum(len(c) for c in contigs)),
                   "parallelism": "read-sharded x%d, index replicated (1 RCCL broadcast at start-up)" % world,
                   "sam_emitted": not args.no_sam, "scale": args.scale,
                   "sub_batch_pipeline": "stage order" if os.environ.get("GM_OVERLAP") == "0" else "two streams (lookup of sub-batch i+1 beside SW of sub-batch i)"},
    }
    if rank == 0:
        # dominant kernel = seed lookup (k_lookup): algorithmic bytes / its own HIP-event time
        ach = (lk_bytes / 1e9) / (lk_ms / 1e3) if lk_ms > 0 else 0.0
        kname = gm.lib().gm_last_lookup_kernel().decode()       # k_lookup_bkt (one slab, short lists), k_lookup_v4 (folded count), k_lookup_v3 (slab sweep)
        # HBM bytes per launch from separate rocprofv3 --pmc passes (FETCH_SIZE + WRITE_SIZE, profiles/traffic.json),
        # valid only for the workload / sub-batch they were measured on
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            t = tj.get("%s/%s/%d" % (args.workload, kname, min(R, int(os.environ.get("GM_SUBBATCH", "131072")))))
            if t and args.scale == 1.0:
                # measured on launches of 131 072 reads; the launches of this run differ in size (the sub-batch sizes ramp up and
                # down around the overlapped pipeline), so the figure is scaled to this run's mean reads per launch
                traffic = t["bytes_per_launch"] * ((R * args.steps / max(1, lk_launch)) / 131072.0)
        except Exception:
            pass
        out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0,
                           "traffic": traffic, "kernel": kname, "launches": lk_launch,
                           "avg_launch_ms": lk_ms / max(1, lk_launch), "alg_bytes_per_launch": lk_bytes / max(1, lk_launch)}
        out["stages_ms_per_step"] = {k: agg[k] / args.steps for k in agg if k.startswith("ms_")}
        out["per_read"] = {"lookups": agg["lookups"] / (R * args.steps), "list_entries": agg["list_entries"] / (R * args.steps),
                           "alg_bytes": agg["list_bytes"] / (R * args.steps), "survivors": agg["survivors"] / (R * args.steps),
                           "survivors_pruned": agg["survivors_pruned"] / (R * args.steps),
                           "vec_sw_calls": agg["vec_calls"] / (R * args.steps), "full_sw_calls": agg["full_calls"] / (R * args.steps),
                           "mapped_frac": agg["reads_matched"] / (R * args.steps), "exact_order_frac": agg["exact_order_reads"] / (2 * R * args.steps)}
        out["setup_s"] = {"genome_gen": t_gen, "index_build": t_index, "index_bcast": t_bcast, "index_bytes": ix.nbytes}
        if not args.no_cpu_baseline and world == 1:
            # the CPU restatement (oracle, "port") on this box's host cores over a bounded sample of the same workload
            from tests import oracle_api as oa
            # the GPU box gives one GPU job a 16-core share of the host; use what we can actually run on
            ncores = int(os.environ.get("GM_CPU_THREADS", min(os.cpu_count() or 1, 16)))
            oa.load().gmo_set_threads(ncores)
            t0 = time.time(); o = oa.Session(contigs); t_oidx = time.time() - t0
            t0 = time.perf_counter(); sam = o.map_sam(sample_reads, nthreads=ncores); cdt = time.perf_counter() - t0
            o.close()
            out["cpu_baseline"] = {"value": len(sample_reads) / cdt, "unit": "reads/s", "cores": ncores, "kind": "port",
                                   "sample": "%d reads of the same workload (same genome, same error model), oracle/gm_oracle.hpp with OpenMP over reads; "
                                             "index build %.1fs not included" % (len(sample_reads), t_oidx)}
            # parity spot check on the sample while we are here
            got = sess.map_reads(sample_reads)
            out["cpu_baseline"]["sample_sam_identical"] = bool(got == sam)
        print(json.dumps(out))
    sess.close()
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
