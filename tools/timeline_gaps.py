#!/usr/bin/env python3
"""start-to-start intervals of one kernel and what else ran: python tools/timeline_gaps.py <db> [kernel]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
kern = sys.argv[2] if len(sys.argv) > 2 else "k_synth_ev"
rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
t0 = rows[0][1]
ks = [(s, e) for n, s, e, st in rows if kern in n]
mid = ks[len(ks) // 3: 2 * len(ks) // 3]
gaps = [(b[0] - a[0]) / 1e6 for a, b in zip(mid, mid[1:])]
print("%s: %d launches; middle third: start-to-start min %.2f median %.2f max %.2f ms; busy %.2f ms avg" %
      (kern, len(ks), min(gaps), sorted(gaps)[len(gaps) // 2], max(gaps), sum((e - s) for s, e in mid) / len(mid) / 1e6))
# union busy time of all kernels in the middle window
w0, w1 = mid[0][0], mid[-1][1]
for name in ("k_lap_plan", "k_lap_pass1", "k_lap_scan", "k_lap_pass2", "k_lap_repair", "k_walk<1>", "k_walk<2>", "k_chain_fix", "k_tiles", "k_synth_ev"):
    iv = sorted((max(s, w0), min(e, w1)) for n, s, e, st in rows if name in n and e > w0 and s < w1)
    tot = sum(e - s for s, e in iv)
    # concurrency: average number in flight
    print("  %-12s in flight on average %.2f  (avg dur %.2f ms, %d launches)" % (name, tot / (w1 - w0), tot / max(len(iv), 1) / 1e6, len(iv)))
