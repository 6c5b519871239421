#!/bin/bash
# Build the REAL reference code for the hot path and its front end, from the sources where they lie
# under /root/reference, into oracle/_ref/ (git-ignored; nothing of the reference is copied into the
# repo).  See oracle/ref/README.md for what is reference text and what is harness glue.
#
# The whole program cannot be built in this image: plutogpssim.c #includes <iio.h>, <ad9361.h> and
# <curl/curl.h>, none of which are installed, and writing stand-ins for them is not allowed.  What CAN
# be compiled from the reference's own text with only libc/libm/zlib is everything that matters for
# this path: lines 93-1989 (tables, codegen, geodesy, satpos, nav-message encoder, RINEX readers,
# computeRange, computeCodePhase, allocateChannel) and the statement ranges of main() that make up the
# scenario loop, including the sample loop 2690-2756 itself.  Those ranges are cut out by line number
# (the file is pinned by SHA-256 below) into a temporary directory, #included by the harness
# translation unit oracle/ref/ref_harness.c, compiled, and the temporary directory is deleted.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${GPSBB_REFERENCE_DIR:-/root/reference}"
OUT="$HERE/../_ref"
SRC="$REF/plutogpssim.c"
HDR="$REF/plutogpssim.h"

if [ ! -f "$SRC" ] || [ ! -f "$HDR" ]; then
    echo "build_ref: $REF not present - nothing built (prebuilt oracle/_ref is used if it exists)" >&2
    exit 0
fi
want_c=0a8413caebcd484787022d886a84b6dfcf958bf7315ef1c5a475469cac7a9b08
want_h=8516be6beaaf7c1a82d838098f4a717c1cdf8a7882f83692ac0521686bc0fdf2
got_c=$(sha256sum "$SRC" | cut -d' ' -f1)
got_h=$(sha256sum "$HDR" | cut -d' ' -f1)
if [ "$got_c" != "$want_c" ] || [ "$got_h" != "$want_h" ]; then
    echo "build_ref: reference sources differ from the pinned revision; line ranges are not valid" >&2
    exit 1
fi

TMP="$(mktemp -d)"
trap 'rm -rf "$TMP"' EXIT
cut_lines() { sed -n "$1,$2p" "$SRC" > "$TMP/$3"; }
cut_lines 93 1989 slice_front.inc       # tables ... allocateChannel
cut_lines 2497 2569 slice_timewin.inc   # gmin/gmax, -T overwrite, start-time check
cut_lines 2576 2597 slice_ephsel.inc    # current ephemeris set
cut_lines 2620 2632 slice_chaninit.inc  # clear channels, first allocateChannel
cut_lines 2645 2646 slice_antpat.inc    # antenna pattern table
cut_lines 2653 2653 slice_grx0.inc      # grx += 0.1
cut_lines 2656 2687 slice_seed.inc      # per-block computeRange/computeCodePhase/gain
cut_lines 2690 2756 slice_loop.inc      # THE SAMPLE LOOP
cut_lines 2764 2805 slice_maint.inc     # 30 s maintenance, time/motion index update

# sanity anchors: first/last statements of the hot-loop slice
head -1 "$TMP/slice_loop.inc" | grep -q 'for (isamp = 0; isamp < NUM_SAMPLES; isamp++) {'
tail -2 "$TMP/slice_loop.inc" | head -1 | grep -q 'iq_buff\[isamp \* 2 + 1\] = (short) q_acc;'

mkdir -p "$OUT"
CC="${CC:-gcc}"
# the reference's own flags (Makefile:1-2) minus -g; -I order: temp slices, reference header
REFFLAGS="-std=c11 -O0 -W -Wall -D_GNU_SOURCE"
# optimised variant: bit-identical provided libm calls are not folded/merged by the compiler
FASTFLAGS="-std=c11 -O2 -fno-builtin -ffp-contract=off -W -Wall -D_GNU_SOURCE"
INC="-I$TMP -I$REF -I$HERE/../../include"

for mc in 12 16; do
    $CC $REFFLAGS $INC -DREF_MAX_CHAN=$mc -DREF_BUILD_MAIN "$HERE/ref_harness.c" -o "$OUT/ref_sim$mc" -lm -lz
done
$CC $REFFLAGS $INC -DREF_MAX_CHAN=16 -shared -fPIC "$HERE/ref_harness.c" -o "$OUT/libplutoref.so" -lm -lz
$CC $FASTFLAGS $INC -DREF_MAX_CHAN=16 -shared -fPIC "$HERE/ref_harness.c" -o "$OUT/libplutoref_O2.so" -lm -lz
$CC $FASTFLAGS $INC -DREF_MAX_CHAN=16 -DREF_BUILD_MAIN "$HERE/ref_harness.c" -o "$OUT/ref_sim16_O2" -lm -lz
# The reference's other carrier NCO (the `#ifndef FLOAT_CARR_PHASE` code: 32-bit phase accumulator).  The
# header defines FLOAT_CARR_PHASE unconditionally (h:12), so this variant is compiled against a temporary
# copy of the header with exactly that one line removed; everything else is as above.
HFIX="$TMP/hfixed"
mkdir -p "$HFIX"
sed -n '12p' "$HDR" | grep -q '^#define FLOAT_CARR_PHASE'
sed '12d' "$HDR" > "$HFIX/plutogpssim.h"
INCF="-I$TMP -I$HFIX -I$HERE/../../include"
$CC $REFFLAGS $INCF -DREF_MAX_CHAN=16 -shared -fPIC "$HERE/ref_harness.c" -o "$OUT/libplutoref_fixed.so" -lm -lz
$CC $REFFLAGS $INCF -DREF_MAX_CHAN=12 -DREF_BUILD_MAIN "$HERE/ref_harness.c" -o "$OUT/ref_sim12_fixed" -lm -lz
echo "build_ref: built $(ls "$OUT" | tr '\n' ' ')"
