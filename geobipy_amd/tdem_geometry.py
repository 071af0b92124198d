"""Transmitter / receiver geometry of the time-domain path: attitude (roll, pitch, yaw of both loops) and the X / Y / Z outputs.

The reference hands GA-AEM the tuple ``Geometry(tx_height, tx_roll, -tx_pitch, -tx_yaw, dx, dy, dz, rx_roll, -rx_pitch,
-rx_yaw)`` on every forward (system/Loop_pair.py:63-77) and Tempest solves for the receiver pitch
(data/datapoint/Tempest_datapoint.py:192-213).  gatdaem1d is absent, so the conventions are restated from GA-AEM's published
geometry description -- x = flight direction, y = left, z = up; roll "left side up", pitch "nose down", yaw "turn left"
positive, i.e. right-handed rotations about x, y, z in degrees; body -> earth matrix R = Rz(yaw) Ry(pitch) Rx(roll) -- and held
against closed forms (tests/test_tdem_attitude.py), not against a reference vector: **parity unpinned** for non-zero angles.

How the kernels see it.  Above the ground the secondary field of a magnetic dipole m at height h is -grad of
``Phi = (1/4pi) (m_x d/dx + m_y d/dy - m_z d/dZ) G``,  ``G = Int rTE(lam) e^{-lam Z} J0(lam rho) dlam``,  Z = 2 h + dz.  In the
frame whose x' axis points from the transmitter to the receiver (horizontal distance rho) every field component is a combination
of FIVE Hankel integrals -- the "basis integrals" of one spline node:

    B0L = Int K s(lam)  J0(lam rho) dlam        vertical moment (the system's horizontal loop of radius a:
    B1L = Int K s(lam)  J1(lam rho) dlam          s = lam J1(lam a) / (2 pi a); a = 0: lam^2 / 4 pi)
    B0  = (1/4pi)     Int K lam^2 J0 dlam       horizontal moment (a dipole)
    B1  = (1/4pi)     Int K lam^2 J1 dlam
    BA  = (1/4pi rho) Int K lam   J1 dlam                                        K = rTE e^{-lam Z}

    H'_x = (B0 - BA) m'_x + B1L m'_z      H'_y = BA m'_y      H'_z = -B1 m'_x + B0L m'_z

Each (basis integral, spline node) is one "frequency" of the raw Hankel handle the frequency-domain kernels run on -- the tables
depend on (rho, dz) only -- and the geometry of a sounding is a small REAL matrix that mixes the nodal spectra of the basis
integrals into the nodal spectra of the output components, per row, inside the window kernel (``gbp_td_apply_mix``):

    out_k = sign_k scale_k sum_j V[k, j] H'_j,     V = R_rx^T Rz(phi),   m' = Rz(-phi) R_tx z^,   phi = atan2(dy, dx)

so soundings of any attitude share one launch and one table set per (rho, dz), level flight costs what it did (B0L for Z, B1L
for X), and a receiver-only rotation costs nothing extra.  ``sign`` = (-1, -1, +1) for (x, y, z): the reference negates
GA-AEM's z (TdemDataPoint.py:1013-1015) and the time operator carries the -dB/dt convention (pinned on the CSV known answers).
"""
import numpy as np

BASIS = ("B0L", "B1L", "B0", "B1", "BA")
COMPONENTS = "xyz"
OUTPUT_SIGN = np.array([-1.0, -1.0, 1.0])
ON_AXIS_RHO = 1.0e-2     # metres: horizontal-moment integrals of an on-axis receiver are taken at this distance (J0 ~ 1 - 1e-7)


def rotation(roll, pitch, yaw):
    """Body -> earth rotation matrices [..., 3, 3] for angles in degrees (GA-AEM sign semantics, see the module docstring)."""
    r, p, y = (np.deg2rad(np.asarray(v, dtype=np.float64)) for v in (roll, pitch, yaw))
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    R = np.empty(np.broadcast(r, p, y).shape + (3, 3))
    R[..., 0, 0] = cy * cp
    R[..., 0, 1] = cy * sp * sr - sy * cr
    R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp
    R[..., 1, 1] = sy * sp * sr + cy * cr
    R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp
    R[..., 2, 1] = cp * sr
    R[..., 2, 2] = cp * cr
    return R


def gaaem_geometry(height, offset, attitude=None):
    """[B, 10] GA-AEM tuples from heights [B], offsets (3,) or [B, 3] and attitudes (6,) / [B, 6] = (tx roll, pitch, yaw, rx roll,
    pitch, yaw) in GA-AEM's sign convention (None: level flight)."""
    h = np.atleast_1d(np.asarray(height, dtype=np.float64))
    g = np.zeros((h.size, 10))
    g[:, 0] = h
    g[:, 4:7] = np.asarray(offset, dtype=np.float64)
    if attitude is not None:
        a = np.broadcast_to(np.asarray(attitude, dtype=np.float64), (h.size, 6))
        g[:, 1:4], g[:, 7:10] = a[:, :3], a[:, 3:]
    return g


def from_loops(tx, rx):
    """(offset (dx, dy, dz), attitude (6,) in GA-AEM's convention) of a loop pair, with the sign changes of Loop_pair.Geometry
    (system/Loop_pair.py:70-77: roll as is, pitch and yaw negated)."""
    f = lambda v: float(np.atleast_1d(v)[0])
    off = (f(rx.x) - f(tx.x), f(rx.y) - f(tx.y), f(rx.z) - f(tx.z))
    att = (f(tx.roll), -f(tx.pitch), -f(tx.yaw), f(rx.roll), -f(rx.pitch), -f(rx.yaw))
    return off, att


# The scalars of a loop pair the reference can sample, in the order Loop_pair.perturb visits them (system/Loop_pair.py:161-164:
# the pair's own Point = the receiver offset, then the transmitter -- Point.perturb x, y, z, then EmLoop.perturb pitch, roll, yaw,
# system/EmLoop.py:222-240 --, then the receiver, whose x / y / z priors go to the offset instead, Loop_pair.set_priors :172-178).
# name -> (option-key stem, cells of the posterior: EmLoop.set_pitch_posterior uses Uniform.bins(199), everything else 99)
LOOP_PAIR_SCALARS = (("dx", "receiver_x", 99), ("dy", "receiver_y", 99), ("dz", "receiver_z", 99),
                     ("tx_x", "transmitter_x", 99), ("tx_y", "transmitter_y", 99), ("tx_z", "transmitter_z", 99),
                     ("tx_pitch", "transmitter_pitch", 199), ("tx_roll", "transmitter_roll", 99), ("tx_yaw", "transmitter_yaw", 99),
                     ("rx_pitch", "receiver_pitch", 199), ("rx_roll", "receiver_roll", 99), ("rx_yaw", "receiver_yaw", 99))


def loop_pair_values(tx, rx):
    """{name: value} of the scalars above for a loop pair, angles as the loops store them (the reference's own convention)."""
    f = lambda v: float(np.atleast_1d(v)[0])
    return dict(dx=f(rx.x) - f(tx.x), dy=f(rx.y) - f(tx.y), dz=f(rx.z) - f(tx.z), tx_x=f(tx.x), tx_y=f(tx.y), tx_z=f(tx.z),
                tx_pitch=f(tx.pitch), tx_roll=f(tx.roll), tx_yaw=f(tx.yaw), rx_pitch=f(rx.pitch), rx_roll=f(rx.roll), rx_yaw=f(rx.yaw))


def gaaem_tuple(values):
    """Loop_pair.Geometry (system/Loop_pair.py:63-77) from such a dict: (tx height, tx roll, -tx pitch, -tx yaw, dx, dy, dz, rx roll,
    -rx pitch, -rx yaw) -- the transmitter's x / y do not enter the forward."""
    v = values
    return np.array([v["tx_z"], v["tx_roll"], -v["tx_pitch"], -v["tx_yaw"], v["dx"], v["dy"], v["dz"], v["rx_roll"], -v["rx_pitch"], -v["rx_yaw"]])


def loop_pair_moves(values, options):
    """The ScalarMove list of a loop pair from the reference's option keys -- ``solve_<stem>``, ``maximum_<stem>_change``,
    ``<stem>_proposal_variance`` (all False in the options files the reference ships) -- centred on the pair's current values."""
    from .rjmcmc import ScalarMove
    return [ScalarMove(name, values[name], options["maximum_" + stem + "_change"], options[stem + "_proposal_variance"], n_bins=nb)
            for name, stem, nb in LOOP_PAIR_SCALARS if options.get("solve_" + stem, False)]


# The moves the DEVICE sampler takes (gbp_td_moves): name -> (entry of the GA-AEM tuple, sign of Loop_pair.Geometry).  Angles change
# the mixing weights only; the receiver offset (dx, dy, dz) and the transmitter height keep the chain's table set and are evaluated with
# a per-chain distance scale and effective height (gbp_td_moves.rho_scale, gbp_fdem_*_rows_scaled).  The transmitter's x / y are not in
# the tuple -- they never change a prediction (Loop_pair keeps the offset as a Point of its own) -- and stay with the host sampler.
DEVICE_ANGLE_MOVES = {"tx_pitch": (2, -1.0), "tx_roll": (1, 1.0), "tx_yaw": (3, -1.0), "rx_pitch": (8, -1.0), "rx_roll": (7, 1.0), "rx_yaw": (9, -1.0)}
DEVICE_POSITION_MOVES = {"dx": (4, 1.0), "dy": (5, 1.0), "dz": (6, 1.0), "tx_z": (0, 1.0)}


def device_angle_moves(options):
    """[(name, tuple entry, sign, maximum change, proposal scale, posterior cells)] in the reference's order for the options'
    solve_transmitter_* / solve_receiver_* keys: the attitude angles, the receiver offset (solve_receiver_x / _y / _z) and the
    transmitter height (solve_transmitter_z); solve_transmitter_x / _y raise (no effect on the forward: host sampler)."""
    out = []
    for name, stem, nb in LOOP_PAIR_SCALARS:
        if not options.get("solve_" + stem, False):
            continue
        if name in DEVICE_ANGLE_MOVES:
            e, sg = DEVICE_ANGLE_MOVES[name]
        elif name in DEVICE_POSITION_MOVES:
            e, sg = DEVICE_POSITION_MOVES[name]
        else:
            raise NotImplementedError("solve_" + stem + ": a position move of the transmitter's x / y never changes a prediction (the GA-AEM tuple "
                                      "holds the receiver OFFSET, Loop_pair.py:63-77): the device sampler does not sample it (the host sampler, "
                                      "inference.Inference1D, takes it)")
        out.append((name, e, sg, float(options["maximum_" + stem + "_change"]), float(options[stem + "_proposal_variance"]), nb))
    return out


def basis_weights(geometry, loop_radius):
    """w[B, 3, 5]: field along the receiver's axis k (x, y, z) = sum_i w[b, k, i] * BASIS_i, before output sign and scaling.
    For a dipole transmitter (loop_radius = 0) B0L / B1L ARE B0 / B1 and take their weights (B0, B1 get 0); for a receiver on
    the transmitter's axis (rho = 0) B1L = B1 = 0 and BA -> B0 / 2, evaluated at ``ON_AXIS_RHO``."""
    g = np.asarray(geometry, dtype=np.float64).reshape(-1, 10)
    rho = np.hypot(g[:, 4], g[:, 5])
    on = rho == 0.0
    c, s = np.where(on, 1.0, g[:, 4] / np.where(on, 1.0, rho)), np.where(on, 0.0, g[:, 5] / np.where(on, 1.0, rho))
    Rz = np.zeros((g.shape[0], 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = c, -s, s, c, 1.0
    u = np.einsum("bji,bj->bi", Rz, rotation(g[:, 1], g[:, 2], g[:, 3])[:, :, 2])          # Rz(-phi) R_tx z^
    V = np.einsum("bji,bjk->bik", rotation(g[:, 7], g[:, 8], g[:, 9]), Rz)                  # R_rx^T Rz(phi)
    w = np.zeros((g.shape[0], 3, 5))
    w[:, :, 0] = V[:, :, 2] * u[:, None, 2]                      # B0L
    w[:, :, 1] = V[:, :, 0] * u[:, None, 2]                      # B1L
    w[:, :, 2] = V[:, :, 0] * u[:, None, 0]                      # B0
    w[:, :, 3] = -V[:, :, 2] * u[:, None, 0]                     # B1
    w[:, :, 4] = -V[:, :, 0] * u[:, None, 0] + V[:, :, 1] * u[:, None, 1]    # BA
    if not loop_radius > 0.0:
        w[:, :, 0] += w[:, :, 2]
        w[:, :, 1] += w[:, :, 3]
        w[:, :, 2:4] = 0.0
    if on.any():
        w[on, :, 2] += 0.5 * w[on, :, 4]
        w[on, :, 1] = w[on, :, 3] = w[on, :, 4] = 0.0
    return w


class GeometryMix:
    """Everything a batch needs to turn GA-AEM geometry tuples into launches, for ``systems`` (list of TdemSystem):

      ``basis``     the basis integrals some row needs (indices into BASIS), in table order
      ``set_keys``  distinct (rho, dz) pairs = table sets; ``set_of_row`` [B] int32
      ``weights``   [B, n_w] per-row mixing weights, one block per (system, component, active basis integral)
      ``src``, ``col``  [n_nodal_out, T] int32: nodal_out[m] = sum_t weights[b, col[m, t]] * nodal_in[src[m, t]]  (-1: unused)
      ``n_in`` / ``n_out``  lengths of the nodal vectors [Re(all frequencies) | Im(all frequencies)] before / after mixing;
                   in: per system (basis integral, node); out: per system (component, node)
    All rows must be on the same side of the axis (rho = 0 uses other filters): ``on_axis`` says which."""

    def __init__(self, systems, geometry, force_basis=()):
        """``force_basis``: basis integrals to keep in the layout although no row's weights need them now (sampled attitudes)."""
        g = np.ascontiguousarray(np.asarray(geometry, dtype=np.float64).reshape(-1, 10))
        assert np.all(np.isfinite(g)), ValueError("geometry must be finite")
        self.geometry, self.systems = g, list(systems)
        rho = np.hypot(g[:, 4], g[:, 5])
        assert np.all(rho == 0.0) or np.all(rho > 0.0), ValueError("soundings on and off the transmitter's axis need separate batches")
        self.on_axis = bool(g.shape[0] > 0 and rho[0] == 0.0)
        keys = np.stack([rho, g[:, 6]], axis=1)
        self.set_keys, inverse = np.unique(keys, axis=0, return_inverse=True)
        self.set_of_row = inverse.ravel().astype(np.int32)
        radii = {float(s.loopRadius()) > 0.0 for s in self.systems}
        assert len(radii) == 1, ValueError("systems of one acquisition share their transmitter loop (all dipoles or all loops)")
        self.loop = radii.pop()
        if self.on_axis:
            assert self.loop, ValueError("a receiver on the transmitter's axis needs a finite ModellingLoopRadius")
        w = basis_weights(g, 1.0 if self.loop else 0.0)                     # [B, 3, 5]
        blocks, used = [], np.zeros(5, dtype=bool)
        for s in self.systems:
            for comp in s.components:
                k = COMPONENTS.index(comp)
                blk = w[:, k, :] * (OUTPUT_SIGN[k] * s.scaling[comp])
                used |= np.any(blk != 0.0, axis=0)
                blocks.append(blk)
        used[0] = True                                                      # (an empty batch still has a layout)
        used[list(force_basis)] = True
        self.basis = [i for i in range(5) if used[i]]
        nb = len(self.basis)
        self.weights = np.ascontiguousarray(np.concatenate([blk[:, self.basis] for blk in blocks], axis=1)) if blocks else np.zeros((g.shape[0], 0))
        # per (system, component) block: component index, output sign x scaling, factor of the free-space field, windows (gbp_td_moves)
        self.block_comp = np.array([COMPONENTS.index(c_) for s in self.systems for c_ in s.components], dtype=np.int32)
        self.block_scale = np.array([OUTPUT_SIGN[COMPONENTS.index(c_)] * s.scaling[c_] for s in self.systems for c_ in s.components])
        self.block_primary = np.array([(1.0 if COMPONENTS.index(c_) < 2 else -1.0) * s.scaling[c_] * 4.0e-7 * np.pi * s.moment
                                       for s in self.systems for c_ in s.components])
        self.block_windows = np.array([s.nwindows for s in self.systems for _ in s.components], dtype=np.int32)
        # index maps
        n_nodes = [s.node_frequencies().size for s in self.systems]
        nF_in = sum(nb * n for n in n_nodes)
        nF_out = sum(s.n_components * n for s, n in zip(self.systems, n_nodes))
        self.n_in, self.n_out = 2 * nF_in, 2 * nF_out
        src = np.full((self.n_out, nb), -1, dtype=np.int32)
        col = np.zeros((self.n_out, nb), dtype=np.int32)
        fi0 = fo0 = c0 = 0
        for s, n in zip(self.systems, n_nodes):
            for j in range(s.n_components):
                for t in range(nb):
                    rows = fo0 + j * n + np.arange(n)
                    src[rows, t] = fi0 + t * n + np.arange(n)
                    src[nF_out + rows, t] = nF_in + fi0 + t * n + np.arange(n)
                    col[rows, t] = col[nF_out + rows, t] = c0 + j * nb + t
            fi0, fo0, c0 = fi0 + nb * n, fo0 + s.n_components * n, c0 + s.n_components * nb
        self.src, self.col = np.ascontiguousarray(src), np.ascontiguousarray(col)

    def tables(self, system, key):
        """Raw Hankel tables of ``system`` for the table set ``key`` = (rho, dz) with this batch's basis integrals."""
        return system.hankel_tables(float(key[0]), float(key[1]), self.basis)

    def primary_field(self):
        """[B, sum n_components] free-space field of the rotated transmitter dipole along the receiver's axes, in the output
        units and the reference's sign convention (x, y as they are, z negated: TdemDataPoint.py:1004-1015)."""
        g = self.geometry
        m = rotation(g[:, 1], g[:, 2], g[:, 3])[:, :, 2]
        R = g[:, 4:7]
        Rn = np.linalg.norm(R, axis=1, keepdims=True)
        H = (3.0 * np.sum(m * R, axis=1, keepdims=True) * R / Rn ** 2 - m) / (4.0 * np.pi * Rn ** 3)
        c = np.einsum("bji,bj->bi", rotation(g[:, 7], g[:, 8], g[:, 9]), H)
        out = []
        for s in self.systems:
            for comp in s.components:
                k = COMPONENTS.index(comp)
                out.append((1.0 if k < 2 else -1.0) * s.scaling[comp] * 4.0e-7 * np.pi * s.moment * c[:, k])
        return np.stack(out, axis=1)
