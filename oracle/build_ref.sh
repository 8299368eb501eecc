#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Builds the *real* reference as oracle binaries.
#
# Compiles the reference modules where they lie under /root/reference (CRLF stripped in a temp
# dir) with amdflang -O2, SERIAL (no -fopenmp: the reference's documented compile line has none
# and its OpenMP pair loop races), and links our own dump drivers:
#   SUMMER_SPH.f90            lines 1-931  (module SPH_routines_module) + oracle/ref_driver.f90
#   SUMMER_SPH - Variable.f90 lines 1-1165 (same module name, variable-h) + oracle/ref_driver_v.f90
# Only the resulting binaries land in oracle/_ref/ (git-ignored); no reference source is copied
# into the repo.
#
# This only works in the build container (where /root/reference exists); on the GPU box the
# binaries are not needed: tests use the committed fixtures in tests/golden.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${SUMMERSPH_REFERENCE:-/root/reference}"
if [ ! -f "$ref/SUMMER_SPH.f90" ]; then
  echo "build_ref: $ref/SUMMER_SPH.f90 not present -- skipping (fixtures in tests/golden are used instead)"
  exit 0
fi
FC="${FC:-amdflang}"
command -v "$FC" >/dev/null || { echo "build_ref: no $FC"; exit 0; }
tmp="$(mktemp -d /tmp/summersph_ref.XXXXXX)"
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$here/_ref" "$tmp/f" "$tmp/v"

sed -n '1,931p' "$ref/SUMMER_SPH.f90" | tr -d '\r' > "$tmp/f/ref_module.f90"
( cd "$tmp/f" && "$FC" -O2 -w -c ref_module.f90 -o ref_module.o \
  && "$FC" -O2 -w -I"$tmp/f" "$here/ref_driver.f90" ref_module.o -o "$here/_ref/ref_driver" )
echo "build_ref: built $here/_ref/ref_driver"

if [ -f "$ref/SUMMER_SPH - Variable.f90" ]; then
  sed -n '1,1165p' "$ref/SUMMER_SPH - Variable.f90" | tr -d '\r' > "$tmp/v/ref_module_v.f90"
  ( cd "$tmp/v" && "$FC" -O2 -w -c ref_module_v.f90 -o ref_module_v.o \
    && "$FC" -O2 -w -I"$tmp/v" "$here/ref_driver_v.f90" ref_module_v.o -o "$here/_ref/ref_driver_v" )
  echo "build_ref: built $here/_ref/ref_driver_v"
fi
