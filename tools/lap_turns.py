"""wave-turns of the lap walks per launch (debug build, GPSBB_PY_LIB=dbg: hazards[3] counts one per turn of a wavefront's lockstep loop;
k_lap_scan prints the running total): python tools/lap_turns.py <fs> <nsamp> <nch> <nblocks> [chain]"""
import sys, os
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
fs, nsamp, nch, nb = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
chain = len(sys.argv) > 5
with pkg.Synth(0) as s:
    s.set_option(pkg.OPT_SEED_WHERE, 3)
    ch = pkg.synth_descriptors(nb, nch=nch, seed=11)
    b = s.batch(ch, 1.0 / fs, nsamp, flags=pkg.CHAIN_CARRIER if chain else 0)
    for _ in range(3):
        b.run(); s.sync()
    print("prepass", s.info(pkg.INFO_PREPASS), "laps: carrier about", float(np.abs(ch["f_carr"]).sum() / fs * nsamp), "code about", float(ch["f_code"].sum() / fs * nsamp / 1023.0))
    b.close()
