#!/usr/bin/env python3
"""Generate the synthetic RINEX-2 GPS navigation files used as test/bench inputs.

BASELINE.json's configs name `brdc3540.14n`, which is not shipped with the reference and cannot be
fetched (no network).  This writes stand-in broadcast-ephemeris files for the same day
(2014-12-20, GPS week 1823, tow 518400) in exactly the fixed-column layout the reference's reader
parses (readRinex2, plutogpssim.c:874-1233: header labels at column 60, record fields at columns
0/3/22/41/60, width 19, 'D' exponents):

  tests/golden/synth3540.14n   32 SVs in 6 planes (a plausible full constellation), 3 two-hourly sets
  tests/golden/dense3540.14n   32 SVs of which >= 16 are above the horizon at the BASELINE site
                               (30.286502 N, 120.032669 E, 100 m) for the first hours of the day

Everything is deterministic (fixed seed); the committed files are the fixtures, this script documents
how they were made:  python tools/make_synth_rinex.py
"""
import math
import os
import random

GM = 3.986005e14
OMEGA_E = 7.2921151467e-5
PI = 3.1415926535898
WEEK = 1823
TOW0 = 518400.0  # 2014-12-20 00:00:00 GPST
SITE_LLH = (30.286502, 120.032669, 100.0)


def llh2xyz(lat_deg, lon_deg, h):
    a, e = 6378137.0, 0.0818191908426
    lat, lon = math.radians(lat_deg), math.radians(lon_deg)
    n = a / math.sqrt(1.0 - (e * math.sin(lat)) ** 2)
    return ((n + h) * math.cos(lat) * math.cos(lon), (n + h) * math.cos(lat) * math.sin(lon),
            (n * (1 - e * e) + h) * math.sin(lat))


def sat_ecef(el, tow):
    """Plain Kepler orbit -> ECEF (no harmonic corrections): only used to pick visible geometries."""
    a = el["sqrta"] ** 2
    n = math.sqrt(GM / a ** 3) + el["deltan"]
    tk = tow - el["toe"]
    m = el["m0"] + n * tk
    e = el["ecc"]
    ek = m
    for _ in range(20):
        ek = m + e * math.sin(ek)
    nu = math.atan2(math.sqrt(1 - e * e) * math.sin(ek), math.cos(ek) - e)
    u = nu + el["aop"]
    r = a * (1 - e * math.cos(ek))
    inc = el["inc0"] + el["idot"] * tk
    om = el["omg0"] + (el["omgdot"] - OMEGA_E) * tk - OMEGA_E * el["toe"]
    xp, yp = r * math.cos(u), r * math.sin(u)
    return (xp * math.cos(om) - yp * math.cos(inc) * math.sin(om),
            xp * math.sin(om) + yp * math.cos(inc) * math.cos(om), yp * math.sin(inc))


def elevation_deg(sat, site_xyz, lat_deg, lon_deg):
    lat, lon = math.radians(lat_deg), math.radians(lon_deg)
    d = [sat[i] - site_xyz[i] for i in range(3)]
    up = (math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat))
    rng = math.sqrt(sum(x * x for x in d))
    return math.degrees(math.asin(sum(d[i] * up[i] for i in range(3)) / rng))


def base_elements(rng, m0, omg0):
    return {
        "sqrta": 5153.6 + rng.uniform(-0.3, 0.3),
        "ecc": rng.uniform(0.002, 0.018),
        "inc0": math.radians(55.0) + rng.uniform(-0.02, 0.02),
        "aop": rng.uniform(-PI, PI),
        "m0": m0,
        "omg0": omg0,
        "omgdot": -8.0e-9 + rng.uniform(-4e-10, 4e-10),
        "idot": rng.uniform(-4e-10, 4e-10),
        "deltan": 4.5e-9 + rng.uniform(-8e-10, 8e-10),
        "cuc": rng.uniform(-3e-6, 3e-6), "cus": rng.uniform(2e-6, 9e-6),
        "crc": rng.uniform(180.0, 330.0), "crs": rng.uniform(-90.0, 90.0),
        "cic": rng.uniform(-2e-7, 2e-7), "cis": rng.uniform(-2e-7, 2e-7),
        "af0": rng.uniform(-4e-4, 4e-4), "af1": rng.uniform(-6e-12, 6e-12), "af2": 0.0,
        "tgd": rng.uniform(-1.8e-8, 4e-9),
        "toe": TOW0,
    }


def advance(el, dt):
    """Same orbit referred to a later toe (so consecutive sets describe one smooth trajectory)."""
    out = dict(el)
    a = el["sqrta"] ** 2
    n = math.sqrt(GM / a ** 3) + el["deltan"]
    out["m0"] = math.remainder(el["m0"] + n * dt, 2 * PI)
    out["omg0"] = math.remainder(el["omg0"] + el["omgdot"] * dt, 2 * PI)
    out["inc0"] = el["inc0"] + el["idot"] * dt
    out["af0"] = el["af0"] + el["af1"] * dt
    out["toe"] = el["toe"] + dt
    return out


def fmt(v):
    s = "%19.12E" % v
    return s.replace("E", "D")


def header():
    def line(body, label):
        return "%-60s%-20s\n" % (body, label)
    out = []
    out.append(line("     2.10           N: GPS NAV DATA", "RINEX VERSION / TYPE"))
    out.append(line("gpsbb-synth         gpsbb               20141220 000000 UTC", "PGM / RUN BY / DATE"))
    out.append(line("synthetic broadcast ephemeris, not real data", "COMMENT"))
    out.append(line("  %12s%12s%12s%12s" % ("1.1176D-08", "7.4506D-09", "-5.9605D-08", "-5.9605D-08"),
                    "ION ALPHA"))
    out.append(line("  %12s%12s%12s%12s" % ("9.0112D+04", "0.0000D+00", "-1.9661D+05", "-6.5536D+04"),
                    "ION BETA"))
    out.append(line("   %19s%19s%9d%9d" % (fmt(-1.862645149231e-09), fmt(-1.687538997430e-14), 503808, WEEK),
                    "DELTA-UTC: A0,A1,T,W"))
    out.append(line("%6d" % 16, "LEAP SECONDS"))
    out.append(line("", "END OF HEADER"))
    return "".join(out)


def record(prn, hour, el, iode):
    l = []
    l.append("%2d %02d %2d %2d %2d %2d%5.1f%s%s%s\n" % (prn, 14, 12, 20, hour, 0, 0.0, fmt(el["af0"]),
                                                      fmt(el["af1"]), fmt(el["af2"])))
    rows = [
        (float(iode), el["crs"], el["deltan"], el["m0"]),
        (el["cuc"], el["ecc"], el["cus"], el["sqrta"]),
        (el["toe"], el["cic"], el["omg0"], el["cis"]),
        (el["inc0"], el["crc"], el["aop"], el["omgdot"]),
        (el["idot"], 1.0, float(WEEK), 0.0),
        (2.0, 0.0, el["tgd"], float(iode)),
        (el["toe"] - 7200.0 + 6.0 * 0 + 0.0, 4.0, 0.0, 0.0),
    ]
    for r in rows:
        l.append("   " + "".join(fmt(v) for v in r) + "\n")
    return "".join(l)


def header3():
    def line(body, label):
        return "%-60s%-20s\n" % (body, label)
    out = []
    out.append(line("     3.02           N: GNSS NAV DATA    G: GPS", "RINEX VERSION / TYPE"))
    out.append(line("gpsbb-synth         gpsbb               20141220 000000 UTC", "PGM / RUN BY / DATE"))
    out.append(line("synthetic broadcast ephemeris, not real data", "COMMENT"))
    out.append(line("GPSA %12s%12s%12s%12s" % ("1.1176D-08", "7.4506D-09", "-5.9605D-08", "-5.9605D-08"),
                    "IONOSPHERIC CORR"))
    out.append(line("GPSB %12s%12s%12s%12s" % ("9.0112D+04", "0.0000D+00", "-1.9661D+05", "-6.5536D+04"),
                    "IONOSPHERIC CORR"))
    out.append(line("GPUT %17s%16s%7d%5d" % ("%17.10E" % -1.862645149231e-09, "%16.9E" % -1.687538997e-14, 503808, WEEK),
                    "TIME SYSTEM CORR"))
    out.append(line("%6d" % 16, "LEAP SECONDS"))
    out.append(line("", "END OF HEADER"))
    return "".join(out)


def record3(prn, hour, el, iode):
    l = []
    l.append("G%02d %4d %02d %02d %02d %02d %02d%s%s%s\n" % (prn, 2014, 12, 20, hour, 0, 0, fmt(el["af0"]),
                                                          fmt(el["af1"]), fmt(el["af2"])))
    rows = [
        (float(iode), el["crs"], el["deltan"], el["m0"]),
        (el["cuc"], el["ecc"], el["cus"], el["sqrta"]),
        (el["toe"], el["cic"], el["omg0"], el["cis"]),
        (el["inc0"], el["crc"], el["aop"], el["omgdot"]),
        (el["idot"], 1.0, float(WEEK), 0.0),
        (2.0, 0.0, el["tgd"], float(iode)),
        (el["toe"] - 7200.0, 4.0, 0.0, 0.0),
    ]
    for r in rows:
        l.append("    " + "".join(fmt(v) for v in r) + "\n")
    return "".join(l)


def write_file3(path, sats, hours=(0, 2, 4)):
    """The same ephemerides in RINEX 3.02 layout (readRinex3, plutogpssim.c:1241-1610), plus two records of
    another constellation that the reader must skip (c:1380-1382)."""
    with open(path, "w") as f:
        f.write(header3())
        for k, hour in enumerate(hours):
            for prn in sorted(sats):
                el = advance(sats[prn], 3600.0 * hour)
                f.write(record3(prn, hour, el, 10 + 3 * k + (prn % 3)))
                if prn == 5:
                    f.write(record3(prn, hour, el, 1).replace("G05", "R05", 1))


def write_file(path, sats, hours=(0, 2, 4)):
    with open(path, "w") as f:
        f.write(header())
        for k, hour in enumerate(hours):
            for prn in sorted(sats):
                el = advance(sats[prn], 3600.0 * hour)
                f.write(record(prn, hour, el, 10 + 3 * k + (prn % 3)))


def full_constellation(seed):
    rng = random.Random(seed)
    sats = {}
    prn = 1
    for plane in range(6):
        omg0 = math.radians(-170.0 + 60.0 * plane) + rng.uniform(-0.03, 0.03)
        n_in_plane = 6 if plane < 2 else 5
        for k in range(n_in_plane):
            m0 = math.remainder(2 * PI * k / n_in_plane + 0.37 * plane + rng.uniform(-0.15, 0.15), 2 * PI)
            sats[prn] = base_elements(rng, m0, omg0)
            prn += 1
    assert prn == 33
    return sats


def dense_constellation(seed, want_visible=20):
    """32 SVs: want_visible of them stay above ~12 deg elevation at the site for the first half hour."""
    rng = random.Random(seed)
    site = llh2xyz(*SITE_LLH)
    sats = {}
    prn = 1
    tries = 0
    while prn <= 32:
        tries += 1
        el = base_elements(rng, rng.uniform(-PI, PI), rng.uniform(-PI, PI))
        elevs = [elevation_deg(sat_ecef(el, TOW0 + t), site, SITE_LLH[0], SITE_LLH[1]) for t in (0, 900, 1800)]
        vis = min(elevs) > 12.0
        if prn <= want_visible and not vis:
            continue
        if prn > want_visible and max(elevs) > -5.0:
            continue  # the rest stays clearly below the horizon
        sats[prn] = el
        prn += 1
    return sats


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    gold = os.path.join(here, "..", "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    write_file(os.path.join(gold, "synth3540.14n"), full_constellation(3582))
    write_file(os.path.join(gold, "dense3540.14n"), dense_constellation(35401))
    write_file3(os.path.join(gold, "synth3540_v3.rnx"), full_constellation(3582))
    print("wrote", os.path.normpath(os.path.join(gold, "synth3540.14n")), "and dense3540.14n")
