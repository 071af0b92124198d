"""Layered-earth model containers: the inputs of the forward solve.

Mirrors the parts of ``geobipy/src/classes/mesh/RectilinearMesh1D.py`` (edges, widths :435-437,
nCells, relative_to :412-425) and ``geobipy/src/classes/model/Model.py`` (mesh, values :127-138)
that ``fdem1dfwd`` reads (FD/fdem1d.py:29, 50-52).  The rjMCMC moves (birth / death / perturb) and
priors of the reference classes are SURVEY row f-2 and not part of this path.
"""
import numpy as np


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
