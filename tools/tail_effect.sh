# synthesis kernel alone for different numbers of blocks: how much of a launch is the tail of the last round of workgroups
for nb in ${BLOCKS:-128 256 384 400 512 768}; do
  python tools/kbench.py --no-cpu --steps 10 --synth-only --blocks $nb 2>/dev/null | tail -1 | NB=$nb python -c "import sys,json,os; d=json.loads(sys.stdin.read()); nb=int(os.environ['NB']); ms=d['roofline']['ms_per_launch']; print('%4d blocks: %.3f ms per launch = %.2f us per block' % (nb, ms, ms*1e3/nb))"
done
