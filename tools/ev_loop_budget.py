#!/usr/bin/env python3
"""Why k_synth_ev's channel loop was not rewritten in assembly: the compiler's instruction count per channel PAIR, for every number
of carrier breakpoints a run can hold (KC), against a count by hand of the same algorithm's vector operations — what a hand-written
loop could save at most.  Reads the ISA dump `make -C pluto-gps-sim_amd/csrc asm` leaves in /tmp.

The count by hand, per channel and run (gpsbb_events.hip.h):
  ev_first   carrier model fma, fract, time-to-next-change fma (3); clamp of every change KC, the changes after the first KC - 1;
             code model fma, fract, fma, clamp (4); the danger test: KC + 3 low words -> ceil((KC + 2) / 2) min3 / min, 1 compare;
             2 table addresses
  ev_second  2 signs (sign-extended chip bytes xor data bit), chip-change-or-not 2, sample 0's contribution 2;
             per index change 7: before-the-chip-change?, which sign, amplitude difference, its sign (xor, sub), the row's
             address, amplitude in force before the chip change; the chip change 4 (shift, xor, sub, address)
"""
import re

s = open('/tmp/gpsbb-hip-amdgcn-amd-amdhsa-gfx950.s').read()
m = re.search(r'^(_ZN10gpsbb_impl\d+k_synth_evENS[^\n]*):(.*?)\.Lfunc_end\d+:', s, re.S | re.M)
body = m.group(2)
parts = re.split(r'\n(\.LBB\d+_\d+):([^\n]*)', body)
blocks = []  # (label, comment, text)
for k in range(1, len(parts), 3):
    blocks.append((parts[k], parts[k + 1], parts[k + 2]))


def count(t):
    lines = [l.strip() for l in t.split('\n') if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    return (sum(l.startswith('v_') for l in lines), sum(l.startswith('ds_') for l in lines), sum(l.startswith('s_') and not l.startswith('s_waitcnt') for l in lines),
            sum(l.startswith('s_waitcnt') for l in lines))


def by_hand(kc):
    first = 3 + kc + (kc - 1) + 4 + (kc + 3) // 2 + 1 + 2
    second = 6 + 7 * kc + 4
    return first + second


print("KC   compiler: VALU per pair (loop head + two ev_second tails)   DS   SALU   waits   by hand, per pair   to gain")
for kc in (1, 2, 3, 4):
    # the pair loop of ev_channels<KC, false, false>: the block that holds the two ev_first (a loop header at depth 2) is followed by
    # the two blocks named after ev_second<KC, false, false> exits
    idx = [i for i, b in enumerate(blocks) if re.search(r'ev_secondILi%dELb0ELb0E' % kc, b[1])]
    if len(idx) < 2:
        continue
    i0 = idx[0] - 1
    v = d = sa = w = 0
    for b in blocks[i0:i0 + 3]:
        c = count(b[2])
        v += c[0]; d += c[1]; sa += c[2]; w += c[3]
    h = 2 * by_hand(kc)
    print("%d    %3d  (%s)%s %3d   %3d   %3d     %3d                 %.1f %%" %
          (kc, v, " + ".join(str(count(b[2])[0]) for b in blocks[i0:i0 + 3]), " " * 22, d, sa, w, h, 100.0 * (v - h) / v))
