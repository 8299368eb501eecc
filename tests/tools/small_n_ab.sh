for n in 12000 100000 300000; do
  python tests/tools/step_time.py $n 40 default
  SPH_TILE_MIN_GROUPS_D=0 SPH_TILE_MIN_GROUPS_F=0 python tests/tools/step_time.py $n 40 tile_always
  SPH_TILE_MIN_GROUPS_D=100000 SPH_TILE_MIN_GROUPS_F=100000 SPH_GATHER_BLOCK=256 python tests/tools/step_time.py $n 40 gather256
  SPH_TILE_MIN_GROUPS_D=100000 SPH_TILE_MIN_GROUPS_F=100000 SPH_GATHER_BLOCK=64 python tests/tools/step_time.py $n 40 gather64
  SPH_TILE_MIN_GROUPS_D=100000 SPH_TILE_MIN_GROUPS_F=0 SPH_GATHER_BLOCK=256 python tests/tools/step_time.py $n 40 dens_gather_forces_tile
done
