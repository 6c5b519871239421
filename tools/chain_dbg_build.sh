make -C pluto-gps-sim_amd/csrc EXTRA="-DGPSBB_CHAIN_DEBUG" -B >/dev/null 2>&1
python tools/chain_repro.py tools/_fail_case28.npy 30e6 24607 33 6 > gpurun_out/chain_dbg.log 2>&1
grep -n "stream depth 2" gpurun_out/chain_dbg.log
make -C pluto-gps-sim_amd/csrc -B >/dev/null 2>&1
