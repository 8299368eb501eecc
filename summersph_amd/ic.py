"""Seeded synthetic initial conditions in the reference's units (AU, Msun, yr; G = 4 pi^2).

The reference ships no IC files; every IC used for parity and for
the benchmark is generated here.  Rows follow the reference's ingest convention
(/root/reference/SUMMER_SPH.f90:647,658-696): columns x y z vx vy vz u m, a row
with u == 0 is a sink particle, everything else is gas.

All generators return an (n_rows, 8) float64 array, gas rows first, sink rows last.
"""
from __future__ import annotations

import numpy as np

G_DP = 4.0 * np.pi ** 2  # the IC uses the full-precision constant; the solver uses the reference's REAL(4)-rounded one
H_REF = 2.5              # SUMMER_SPH.f90:11 `smoothing`

# midplane number density that gives N_ngb neighbours inside 2h for a uniform medium
def _n0_for(nngb: float, h: float) -> float:
    return nngb / (4.0 / 3.0 * np.pi * (2.0 * h) ** 3)


def sod_column(seed: int = 101, nx_left: int = 50, nx_right: int = 50, nngb: float = 50.0,
               h: float = H_REF, mass: float = 1.0e-6) -> np.ndarray:
    """3-D 'Sod' column along x (BASELINE config 1).  The reference has no 1-D mode and no
    boundaries, so the tube is a lattice column with free lateral surfaces:
    left half 4x4 particles per layer at spacing d, right half 2x2 at spacing 2d (density
    ratio 8 with equal masses), u_L = 2.5, u_R = 2.0 (P_L/P_R = 10, gamma = 1.4), v = 0.
    No sink row, so the reference's dummy-sink path (SUMMER_SPH.f90:663-665,698-707) runs."""
    rng = np.random.default_rng(seed)
    d = _n0_for(nngb, h) ** (-1.0 / 3.0)
    rows = []
    # left: x in (-nx_left*d, 0)
    for ix in range(nx_left):
        for iy in range(4):
            for iz in range(4):
                rows.append((-(ix + 0.5) * d, (iy - 1.5) * d, (iz - 1.5) * d, 2.5))
    d2 = 2.0 * d
    for ix in range(nx_right):
        for iy in range(2):
            for iz in range(2):
                rows.append(((ix + 0.5) * d2, (iy - 0.5) * d2, (iz - 0.5) * d2, 2.0))
    a = np.asarray(rows)
    n = a.shape[0]
    out = np.zeros((n, 8))
    out[:, 0:3] = a[:, 0:3] + rng.normal(0.0, 1.0e-3 * d, size=(n, 3))
    out[:, 6] = a[:, 3]
    out[:, 7] = mass
    return out


def keplerian_disc(n: int, seed: int = 202, r_in: float = 10.0, nngb: float = 60.0, h: float = H_REF,
                   scale_height: float = 2.5, m_disc: float = 0.01, m_star: float = 1.0,
                   u0: float = 0.25, with_sink: bool = True) -> np.ndarray:
    """Uniform-surface-density Keplerian disc (BASELINE configs 2/3).  The outer radius is set
    from n so that the MIDPLANE neighbour count inside 2h is ~nngb at the reference's h = 2.5
    whatever n is (SURVEY.md section 7 'neighbour count realism'); z ~ N(0, H); circular
    velocities about a central mass m_star; one sink row (u = 0) of mass m_star last."""
    rng = np.random.default_rng(seed)
    sigma_n = _n0_for(nngb, h) * np.sqrt(2.0 * np.pi) * scale_height  # particles per AU^2
    r_out = np.sqrt(n / (np.pi * sigma_n) + r_in ** 2)
    r = np.sqrt(rng.uniform(r_in ** 2, r_out ** 2, size=n))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=n)
    z = rng.normal(0.0, scale_height, size=n)
    vk = np.sqrt(G_DP * m_star / r)
    out = np.zeros((n + (1 if with_sink else 0), 8))
    out[:n, 0] = r * np.cos(phi)
    out[:n, 1] = r * np.sin(phi)
    out[:n, 2] = z
    out[:n, 3] = -vk * np.sin(phi)
    out[:n, 4] = vk * np.cos(phi)
    out[:n, 6] = u0
    out[:n, 7] = m_disc / n
    if with_sink:
        out[n, 7] = m_star  # u = 0 marks the sink
    return out


def thin_ring(n: int, seed: int = 404, nngb: float = 60.0, h: float = H_REF, rel_width: float = 0.05,
              scale_height: float = 2.5, m_ring: float = 0.01, m_star: float = 1.0,
              u0: float = 0.25, with_sink: bool = True) -> np.ndarray:
    """Thin Keplerian ring r ~ N(r0, rel_width r0) (BASELINE config 4); r0 follows from n so that the
    peak neighbour count is ~nngb at h = 2.5."""
    rng = np.random.default_rng(seed)
    n0 = _n0_for(nngb, h)
    # n = n0 * (2 pi r0) * (sqrt(2 pi) w r0) * (sqrt(2 pi) H)  ->  r0
    r0 = np.sqrt(n / (n0 * (2.0 * np.pi) ** 2 * rel_width * scale_height))
    r = np.abs(rng.normal(r0, rel_width * r0, size=n))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=n)
    z = rng.normal(0.0, scale_height, size=n)
    vk = np.sqrt(G_DP * m_star / r)
    out = np.zeros((n + (1 if with_sink else 0), 8))
    out[:n, 0] = r * np.cos(phi)
    out[:n, 1] = r * np.sin(phi)
    out[:n, 2] = z
    out[:n, 3] = -vk * np.sin(phi)
    out[:n, 4] = vk * np.cos(phi)
    out[:n, 6] = u0
    out[:n, 7] = m_ring / n
    if with_sink:
        out[n, 7] = m_star
    return out


def keplerian_disc_var(n: int, seed: int = 303, r_in: float = 10.0, eta: float = 1.2, h_mid: float = H_REF,
                       scale_height: float = 2.5, m_disc: float = 0.01, m_star: float = 1.0, u0: float = 0.25,
                       alpha0: float = 0.1, h_cap: float = 8.0, with_sink: bool = True) -> np.ndarray:
    """Variable-h disc in the 10-column format of the reference's variable-h reader
    ("SUMMER_SPH - Variable.f90":782): x y z vx vy vz u m alpha h.  h_i = eta (m / rho_est)^(1/3)
    with the analytic density estimate of the vertical Gaussian; the surface density is chosen so
    that h = h_mid in the midplane whatever n is.  (BASELINE config 3.)"""
    rng = np.random.default_rng(seed)
    n0 = (eta / h_mid) ** 3                                   # midplane number density
    sigma_n = n0 * np.sqrt(2.0 * np.pi) * scale_height
    r_out = np.sqrt(n / (np.pi * sigma_n) + r_in ** 2)
    r = np.sqrt(rng.uniform(r_in ** 2, r_out ** 2, size=n))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=n)
    z = rng.normal(0.0, scale_height, size=n)
    vk = np.sqrt(G_DP * m_star / r)
    out = np.zeros((n + (1 if with_sink else 0), 10))
    out[:n, 0] = r * np.cos(phi)
    out[:n, 1] = r * np.sin(phi)
    out[:n, 2] = z
    out[:n, 3] = -vk * np.sin(phi)
    out[:n, 4] = vk * np.cos(phi)
    out[:n, 6] = u0
    out[:n, 7] = m_disc / n
    out[:n, 8] = alpha0
    nz = n0 * np.exp(-0.5 * (z / scale_height) ** 2)
    out[:n, 9] = np.minimum(eta * nz ** (-1.0 / 3.0), h_cap)
    if with_sink:
        out[n, 7] = m_star
    return out


def split_rows(rows: np.ndarray):
    if rows.shape[1] >= 10:
        return split_rows_var(rows)
    return _split_rows_fixed(rows)


def split_rows_var(rows: np.ndarray):
    """10-column ingest of the variable-h reference (Variable.f90:793-843): alpha and h come from
    columns 9 and 10, sinks get radius 5."""
    gas, sinks = _split_rows_fixed(rows[:, :8], sink_radius=5.0)
    is_sink = rows[:, 6] == 0.0
    gas["alpha"] = rows[~is_sink, 8].copy()
    gas["h"] = rows[~is_sink, 9].copy()
    return gas, sinks


def _split_rows_fixed(rows: np.ndarray, sink_radius: float = 3.5):
    """Gas/sink split exactly as the reference's reader does it (SUMMER_SPH.f90:658-707):
    u != 0 -> gas (alpha starts at 0), u == 0 -> sink (radius 3.5); with no sink row a single
    massless dummy sink at the origin is created.  Returns (gas dict, sink dict)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64)
    is_sink = rows[:, 6] == 0.0
    g = rows[~is_sink]
    s = rows[is_sink]
    gas = {
        "x": g[:, 0].copy(), "y": g[:, 1].copy(), "z": g[:, 2].copy(),
        "vx": g[:, 3].copy(), "vy": g[:, 4].copy(), "vz": g[:, 5].copy(),
        "u": g[:, 6].copy(), "m": g[:, 7].copy(), "alpha": np.zeros(g.shape[0]),
    }
    if s.shape[0] > 0:
        sinks = {
            "x": s[:, 0].copy(), "y": s[:, 1].copy(), "z": s[:, 2].copy(),
            "vx": s[:, 3].copy(), "vy": s[:, 4].copy(), "vz": s[:, 5].copy(),
            "m": s[:, 7].copy(), "radius": np.full(s.shape[0], sink_radius),
        }
    else:
        z1 = np.zeros(1)
        sinks = {"x": z1.copy(), "y": z1.copy(), "z": z1.copy(), "vx": z1.copy(), "vy": z1.copy(),
                 "vz": z1.copy(), "m": z1.copy(), "radius": z1.copy()}
    return gas, sinks


def uniform_box(n: int, seed: int = 213, nngb: float = 50.0, h: float = H_REF, u0: float = 0.25, m_total: float = 0.01,
                v_sigma: float = 0.05) -> np.ndarray:
    """Uniform random cube with ~nngb neighbours inside 2h, no sink row: a THICK domain (the candidate intervals of a
    run of consecutive particles span whole columns, nothing fits an LDS tile) for the bench's side record and the tests."""
    rng = np.random.default_rng(seed)
    dens = nngb / (4.0 / 3.0 * np.pi * (2.0 * h) ** 3)
    edge = (n / dens) ** (1.0 / 3.0)
    out = np.zeros((n, 8))
    out[:, 0:3] = rng.uniform(0.0, edge, (n, 3))
    out[:, 3:6] = rng.normal(0.0, v_sigma, (n, 3))
    out[:, 6] = u0
    out[:, 7] = m_total / n
    return out
