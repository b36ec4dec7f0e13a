#!/bin/bash
# Builds libgdhip.so in-tree for gfx950.  Usage: build.sh [outdir]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${1:-$HERE}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $GDHIP_EXTRA_FLAGS"
OBJS=""
for f in core stats binning density1d kopt2d density2d fft thin contours limits1d convolve batch2d comm; do
  if [ ! -f "$OUT/$f.o" ] || [ "$HERE/$f.hip" -nt "$OUT/$f.o" ] || [ "$HERE/ctx.hpp" -nt "$OUT/$f.o" ] || [ "$HERE/ldsfft.hpp" -nt "$OUT/$f.o" ] || [ "$HERE/fft288.hpp" -nt "$OUT/$f.o" ] || [ "$HERE/solvers.hpp" -nt "$OUT/$f.o" ] || [ "$f" = batch2d -a \( "$HERE/batch2d.hpp" -nt "$OUT/$f.o" -o "$HERE/batch1d.hpp" -nt "$OUT/$f.o" \) ] || [ "$HERE/../../include/gdhip.h" -nt "$OUT/$f.o" ]; then
    rm -f "$OUT/$f.o"  # a failed compile must not leave a stale object for the link
    $HIPCC $FLAGS -c "$HERE/$f.hip" -o "$OUT/$f.o" &
  fi
  OBJS="$OBJS $OUT/$f.o"
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT/libgdhip.so" -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
echo "built $OUT/libgdhip.so"
