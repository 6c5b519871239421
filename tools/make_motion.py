#!/usr/bin/env python3
"""Write tests/golden/circle_motion.csv: a 10 Hz user-motion file in the format readUserMotion parses
(plutogpssim.c:1794-1818: `t,x,y,z` ECEF metres per line).  The trajectory is our own: a 200 m radius
horizontal circle flown at 20 m/s, 100 m above the BASELINE site (30.286502 N, 120.032669 E), 3000 points
(300 s), i.e. the same shape of input as the reference's sample circle.csv but not its data."""
import math
import os

A, E2 = 6378137.0, 0.0818191908426 ** 2
LAT, LON, H = math.radians(30.286502), math.radians(120.032669), 100.0


def llh2xyz(lat, lon, h):
    n = A / math.sqrt(1 - E2 * math.sin(lat) ** 2)
    return ((n + h) * math.cos(lat) * math.cos(lon), (n + h) * math.cos(lat) * math.sin(lon),
            (n * (1 - E2) + h) * math.sin(lat))


def main():
    x0, y0, z0 = llh2xyz(LAT, LON, H)
    east = (-math.sin(LON), math.cos(LON), 0.0)
    north = (-math.sin(LAT) * math.cos(LON), -math.sin(LAT) * math.sin(LON), math.cos(LAT))
    r, v = 200.0, 20.0
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "..", "tests", "golden", "circle_motion.csv")
    with open(path, "w") as f:
        for k in range(3000):
            t = 0.1 * k
            a = v * t / r
            de, dn = r * math.sin(a), r * (1 - math.cos(a))
            p = [c + de * e + dn * n for c, e, n in zip((x0, y0, z0), east, north)]
            f.write("%5.1f,%12.3f,%12.3f,%12.3f\n" % (t, p[0], p[1], p[2]))
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
