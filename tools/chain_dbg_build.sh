# usage: bash tools/chain_dbg_build.sh <ch.npy> fs nsamp bps channel  -> gpurun_out/chain_dbg.log (debug build, then restored)
make -C pluto-gps-sim_amd/csrc EXTRA="-DGPSBB_CHAIN_DEBUG" -B >/dev/null 2>&1
python tools/chain_repro.py "$@" > gpurun_out/chain_dbg.log 2>&1
grep -n "^batch\|^stream\|   got" gpurun_out/chain_dbg.log
make -C pluto-gps-sim_amd/csrc -B >/dev/null 2>&1
