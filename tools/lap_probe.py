"""the lap-parallel pre-pass on the bench's own descriptors: time of one run of a resident chained batch, links that broke"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import bench
pkg = g.load_package()
fs, nsamp, nch = 25e6, 2500000, 16
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with pkg.Synth(0) as s:
    for where in ([int(sys.argv[3])] if len(sys.argv) > 3 else [3, 1]):
        s.set_option(pkg.OPT_SEED_WHERE, where)
        for nb in [int(x) for x in sys.argv[1].split(",")]:
            ch = bench.stream_descriptors(pkg, 128000, nch, first=first, count=nb)
            r0, w0 = s.info(pkg.INFO_CHAIN_REPAIRS), s.info(pkg.INFO_CHAIN_FALLBACKS)
            b = s.batch(ch, 1.0 / fs, nsamp, flags=pkg.CHAIN_CARRIER)
            ts = []
            for k in range(4):
                t0 = time.perf_counter()
                b.run(); s.sync()
                ts.append(time.perf_counter() - t0)
            st = b.timing_stats(reset=True)
            print("where %d blocks %d: run times %s ms  prepass %d  repairs %d rewalked %d  timing %s" % (
                where, nb, " ".join("%.2f" % (1e3 * t) for t in ts), s.info(pkg.INFO_PREPASS),
                s.info(pkg.INFO_CHAIN_REPAIRS) - r0, s.info(pkg.INFO_CHAIN_FALLBACKS) - w0, st), flush=True)
            b.close()
