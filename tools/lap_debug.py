"""one small chained batch through the lap-parallel pre-pass of a debug build (GPSBB_PY_LIB=dbg: -DGPSBB_LAP_DEBUG prints every link that did not hold)"""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
fs, nsamp, nch, nb, seed = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 3)
    ch = pkg.synth_descriptors(nb, nch=nch, seed=seed)
    b = s.batch(ch, 1.0 / fs, nsamp, flags=pkg.CHAIN_CARRIER)
    b.run(); s.sync()
    print("prepass", s.info(pkg.INFO_PREPASS), "repairs", s.info(pkg.INFO_CHAIN_REPAIRS), "rewalked", s.info(pkg.INFO_CHAIN_FALLBACKS))
    b.close()
