"""Layered-earth model containers: the inputs of the forward solve.

Mirrors the parts of ``geobipy/src/classes/mesh/RectilinearMesh1D.py`` (edges, widths :435-437,
nCells, relative_to :412-425) and ``geobipy/src/classes/model/Model.py`` (mesh, values :127-138)
that ``fdem1dfwd`` reads (FD/fdem1d.py:29, 50-52), plus the methods ``Inference1D.accept_reject`` calls on a model
(inversion/Inference1D.py:559, 589, 600): ``perturb`` (model/Model.py:325-345 -> stochastic_newton_perturbation :368-419),
``probability`` (:533-575) and ``proposal_probabilities`` (:577-659) -- thin methods over the host restatement in
``rjmcmc.py`` (pinned to the reference's seeded chains), so that the reference's own accept / reject body runs against these
objects and walks the reference's chain (tests/test_object_api.py).
"""
from copy import deepcopy

import numpy as np

from . import rjmcmc

_ACTION_NAMES = {rjmcmc.NONE: "none", rjmcmc.INSERT: "insert", rjmcmc.DELETE: "delete", rjmcmc.PERTURB: "perturb"}


class RectilinearMesh1D:
    """1-D mesh defined by its cell edges; the last edge of an earth model is +/-inf."""

    def __init__(self, centres=None, edges=None, widths=None, relative_to=None, **kwargs):
        if edges is not None:
            self._edges = np.asarray(edges, dtype=np.float64).copy()
        elif widths is not None:
            w = np.asarray(widths, dtype=np.float64)
            self._edges = np.r_[0.0, np.cumsum(w)]
        elif centres is not None:
            c = np.asarray(centres, dtype=np.float64)
            d = np.diff(c)
            self._edges = np.r_[c[0] - 0.5 * d[0], c[:-1] + 0.5 * d, c[-1] + 0.5 * d[-1]] if c.size > 1 \
                else np.r_[c[0] - 0.5, c[0] + 0.5]
        else:
            raise ValueError("one of centres, edges, widths is required")
        self._relative_to = None if relative_to is None else np.float64(relative_to)

    @property
    def edges(self):
        return self._edges

    @edges.setter
    def edges(self, values):
        self._edges = np.asarray(values, dtype=np.float64).copy()

    @property
    def nCells(self):
        return np.int32(self._edges.size - 1)

    @property
    def widths(self):
        return np.abs(np.diff(self._edges))      # RectilinearMesh1D.py:435-437 (last = inf)

    @property
    def relative_to(self):
        return np.float64(0.0) if self._relative_to is None else self._relative_to

    @property
    def shape(self):
        return (int(self.nCells),)

    @property
    def action(self):
        """('insert' | 'delete' | 'perturb' | 'none', index, value) of the move that produced this mesh
        (RectilinearMesh1D.perturb, mesh/RectilinearMesh1D.py:993-1120)."""
        return getattr(self, "_action", ("none", 0, 0.0))


class Model:
    """Cell values (conductivity, S/m) on a mesh (reference: model/Model.py:18-138)."""

    def __init__(self, mesh=None, values=None, **kwargs):
        assert isinstance(mesh, RectilinearMesh1D), TypeError("mesh must be a RectilinearMesh1D")
        self._mesh = mesh
        self.values = values

    @property
    def mesh(self):
        return self._mesh

    @property
    def nCells(self):
        return self.mesh.nCells

    @property
    def shape(self):
        return self.mesh.shape

    @property
    def values(self):
        return self._values

    @values.setter
    def values(self, values):
        if values is None:
            self._values = np.zeros(self.shape)
            return
        v = np.asarray(values, dtype=np.float64).copy()
        assert v.shape == self.shape, ValueError("values must have shape {}".format(self.shape))
        self._values = v

    # -- rjMCMC interface (what Inference1D.accept_reject calls; host logic in rjmcmc.py) -----------------------------------
    def set_priors(self, structure_prior=None, value_prior=None, prng=None, **kwargs):
        """Attach the priors the moves and probabilities need.  Either ``rjmcmc.StructurePrior`` / ``rjmcmc.ValuePrior``
        objects, or the reference's options-file keys (Inference1D.initialize_model, inversion/Inference1D.py:485-535):
        maximum_number_of_layers, minimum_depth, maximum_depth, minimum_thickness, probability_of_birth / death / perturb /
        no_change, value_mean (the best half-space), factor, gradient_standard_deviation, solve_gradient, solve_parameter,
        parameter_limits."""
        if structure_prior is None:
            k = kwargs
            structure_prior = rjmcmc.StructurePrior(k["maximum_number_of_layers"], k["minimum_depth"], k["maximum_depth"],
                                                    k.get("minimum_thickness", 1.0),
                                                    [k["probability_of_birth"], k["probability_of_death"],
                                                     k["probability_of_perturb"], k["probability_of_no_change"]])
        if value_prior is None:
            k = kwargs
            value_prior = rjmcmc.ValuePrior(k["value_mean"], k.get("factor", 10.0), k.get("gradient_standard_deviation", 1.5),
                                            k.get("solve_gradient", True), bool(k.get("solve_parameter", False)),
                                            k.get("parameter_limits"))
        self._structure_prior, self._value_prior, self._prng = structure_prior, value_prior, prng
        return self

    def _interior_edges(self):
        return np.asarray(self.mesh.edges[1:-1], dtype=np.float64)

    def _like(self, edges, values, action=None):
        out = Model(mesh=RectilinearMesh1D(edges=np.r_[self.mesh.edges[0], edges, np.inf]), values=values)
        out._structure_prior, out._value_prior, out._prng = self._structure_prior, self._value_prior, self._prng
        if action is not None:
            out.mesh._action = action
        return out

    def perturb(self, observation=None, low_variance=-np.inf, high_variance=np.inf, alpha=1.0):
        """Model.perturb (model/Model.py:325-345): structural move + stochastic-Newton value proposal.  Returns
        (remapped_model, perturbed_model); ``observation`` (a data point with fm_dlogc / sensitivity_matrix / predictedData /
        std / data) supplies the Jacobian -- recomputed at the remapped model when the structure changed (:383-384).
        The variance limiters are inert in the reference (perturb ignores them) and here."""
        sp, vp, prng = self._structure_prior, self._value_prior, self._prng
        action, index, value, edges, rem = rjmcmc.perturb_structure(prng, sp, self._interior_edges(), self.values)
        remapped = self._like(edges, rem, (_ACTION_NAMES[action], index, value))
        if observation is None:                                 # prior-only proposals (ignore_likelihood; Model.py:380, 269, 352): no data term
            none = np.zeros(0)
            mean, H = rjmcmc.stochastic_newton(vp, edges, rem, np.zeros((0, rem.size)), none, none, none, alpha)
        else:
            if action != rjmcmc.NONE:
                observation.fm_dlogc(remapped)
            J = np.asarray(observation.sensitivity_matrix)[:, : rem.size]
            mean, H = rjmcmc.stochastic_newton(vp, edges, rem, J, observation.predictedData, observation.data, observation.std, alpha)
        prop = rjmcmc.propose_values(prng, mean, H)
        perturbed = self._like(edges, prop, (_ACTION_NAMES[action], index, value))
        perturbed._proposal_covariance = H                      # values.proposal.variance in the reference
        return remapped, perturbed

    def probability(self, solve_value, solve_gradient):
        """Model.probability (model/Model.py:533-575): log prior of the layer count, the values (``solve_value``) and their
        vertical gradient (``solve_gradient``)."""
        vp = deepcopy(self._value_prior)
        vp.solve_value, vp.solve_gradient = bool(solve_value), bool(solve_gradient)
        return np.float64(rjmcmc.model_log_prior(self._structure_prior, vp, self._interior_edges(), self.values))

    def proposal_probabilities(self, remapped_model, observation=None, structure_only=False, alpha=1.0):
        """Model.proposal_probabilities (model/Model.py:577-659): (forward, reverse) log proposal densities of a
        dimension-changing move -- the Jacobian is re-evaluated at this (the proposed) model (:612); 1.0, 1.0 otherwise."""
        action = self.mesh.action[0]
        if action not in ("insert", "delete") or structure_only:
            return 1.0, 1.0
        vp = self._value_prior
        H = self._proposal_covariance
        grad = rjmcmc.model_prior_derivative(vp, self._interior_edges(), self.values, 1)
        if observation is not None:         # (the reference dereferences a None observation here, Model.py:619: its prior-only run ends at the
            observation.sensitivity(self)   #  first birth or death; local_gradient(observation=None) is what its other branches do)
            J = np.asarray(observation.sensitivity_matrix)[:, : self.values.size]
            data, pred, std = observation.data, observation.predictedData, observation.std
            a = data > 0.0
            grad = grad + J[a].T @ ((pred[a] - data[a]) / std[a] ** 2.0)
        mean_r = np.exp(np.longdouble(1.0) * (np.log(self.values) + alpha * (H @ grad)))
        if np.any(np.isinf(mean_r)) or np.any(mean_r == 0.0):
            return -np.inf, -np.inf
        rem = remapped_model.values
        return (rjmcmc.mvn_logpdf(np.log(rem), np.log(mean_r).astype(np.float64), H),
                rjmcmc.mvn_logpdf(np.log(self.values), np.log(rem), H))
