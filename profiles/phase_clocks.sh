#!/bin/bash
# Builds the profiling variant of the library (tiled.hip with -DSPH_PHASE_CLOCKS) beside the product one:
#   bash profiles/phase_clocks.sh          (here, no GPU needed)
# then on the GPU box:  SUMMERSPH_LIB=summersph_amd/libsummersph_hip_prof.so python tests/tools/phase_clocks.py
set -e -o pipefail
cd "$(dirname "$0")/../summersph_amd/csrc"
make -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-result -DSPH_PHASE_CLOCKS -c tiled.hip -o /tmp/tiled_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fno-gpu-rdc -shared -o ../libsummersph_hip_prof.so api.o grid.o pairs.o integrate.o varh.o /tmp/tiled_prof.o gravity.o accrete.o domain.o
echo built summersph_amd/libsummersph_hip_prof.so
