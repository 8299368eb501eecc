"""Where a group's time goes inside forces_q / density_wt (profiling build, see profiles/phase_clocks.sh):
   SUMMERSPH_LIB=summersph_amd/libsummersph_hip_prof.so python tests/tools/phase_clocks.py [N] [steps]
Ticks of the constant 100-MHz counter, summed by the kernels per phase of a group; printed per group in microseconds."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = C.CDLL(capi.LIB_PATH)
if not hasattr(lib, "sph_debug_phase_clocks"):
    sys.exit("not the profiling build: " + capi.LIB_PATH)
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=214, nngb=85.0))
ctx = capi.Context(device=0)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(5, 1e-2, 0.0); ctx.synchronize()
buf = (C.c_ulonglong * 64)()
lib.sph_debug_phase_clocks(buf)          # clear
dt, t = ctx.run(steps, dt, t); ctx.synchronize()
assert lib.sph_debug_phase_clocks(buf) == 0
v = list(buf)
tick_us = 0.01
out = {}
for name, o in (("forces_q", 0), ("density_wt", 8)):
    g = max(v[o], 1)
    out[name] = {"groups": v[o], "launches": 2 * steps,
                 "us_per_group": {"sync+stage+targets": round(v[o + 1] / g * tick_us, 2), "pairs_wave0": round(v[o + 2] / g * tick_us, 2),
                                  "pairs_mean_wave": round(v[o + 3] / max(v[o + 4], 1) * tick_us, 2),
                                  "reduce+epilogue": round(v[o + 5] / g * tick_us, 2)}}
    tot = (v[o + 1] + v[o + 2] + v[o + 5]) / g * tick_us
    out[name]["us_per_group"]["total_wave0"] = round(tot, 2)
names = ["loop back..top", "own rows", "barrier1", "tile loads+writes", "barrier2", "own record", "pair loop", "lane reduction", "epilogue"]
g = max(v[0], 1)
for label, o in (("wave0 (epilogue wave)", 32), ("last wave (stager)", 48)):
    tot = sum(v[o:o + len(names)]) or 1
    out["forces_q " + label] = {nm: round(100.0 * v[o + k] / tot, 1) for k, nm in enumerate(names)}
    out["forces_q " + label]["ticks_per_group"] = round(tot / g, 1)
print(json.dumps(out, indent=1))
