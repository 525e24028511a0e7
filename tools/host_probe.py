"""Host-side probe for the GPU box: first-touch memory rate, core count, oracle index build time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
t0 = time.time(); a = np.zeros(1 << 30, dtype=np.uint32); a[::1024] = 1; print("first touch 4 GiB: %.2f s" % (time.time() - t0)); del a
from tests import oracle_api as oa
from shrimp_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
t0 = time.time(); contigs = synth.make_genome([n // 2, n // 2], 3); print("genome %d bp: %.2f s" % (n, time.time() - t0))
t0 = time.time(); s = oa.Session(contigs); print("oracle session: %.2f s" % (time.time() - t0)); s.close()
