#!/usr/bin/env bash
# TEST INFRASTRUCTURE.  Builds the *real* reference as an oracle binary.
#
# Compiles the reference module where it lies (/root/reference/SUMMER_SPH.f90,
# lines 1-931 = module SPH_routines_module; CRLF stripped in a temp dir) with
# amdflang -O2, SERIAL (no -fopenmp: the reference's documented compile line has
# none and its OpenMP pair loop races), and links our own dump
# driver oracle/ref_driver.f90.  Only the resulting binary lands in
# oracle/_ref/ (git-ignored); no reference source is copied into the repo.
#
# This only works in the build container (where /root/reference exists); on
# the GPU box the prebuilt binary is not needed: tests use committed fixtures.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${SUMMERSPH_REFERENCE:-/root/reference}"
src="$ref/SUMMER_SPH.f90"
if [ ! -f "$src" ]; then
  echo "build_ref: $src not present -- skipping (fixtures in tests/golden are used instead)"
  exit 0
fi
FC="${FC:-amdflang}"
command -v "$FC" >/dev/null || { echo "build_ref: no $FC"; exit 0; }
tmp="$(mktemp -d /tmp/summersph_ref.XXXXXX)"
trap 'rm -rf "$tmp"' EXIT
sed -n '1,931p' "$src" | tr -d '\r' > "$tmp/ref_module.f90"
mkdir -p "$here/_ref"
( cd "$tmp" && "$FC" -O2 -w -c ref_module.f90 -o ref_module.o \
  && "$FC" -O2 -w -I"$tmp" "$here/ref_driver.f90" ref_module.o -o "$here/_ref/ref_driver" )
echo "build_ref: built $here/_ref/ref_driver"
