"""SparseTensor / PointTensor containers with the torchsparse 1.4.0 surface the reference uses
(network/utils.py:27-31,58-59,84-99; network/spvcnn.py:114,131; train.py:134).

coords are int32 [N,4] = (x, y, z, batch) -- batch LAST -- kept as multiples of the tensor
stride; `cmaps` maps stride -> coords and `kmaps` maps (stride, kernel, conv stride, dilation) ->
lidal_amd.nn.functional.conv.KernelMap; both dicts are shared by every tensor derived from one
input so a kernel map is built once per resolution level.
"""
from .utils import make_ntuple

__all__ = ['SparseTensor', 'PointTensor', 'MapCache']


class MapCache(dict):
    """The per-input dict behind `cmaps` / `kmaps`: a plain dict that can be weakly referenced, so
    caches derived from a level's coordinates (the hash table, nn/functional/query.py) can be
    scoped to the input they were built for -- a NEW SparseTensor over the same coordinate tensor
    rebuilds everything, as the reference does every iteration."""
    __slots__ = ('__weakref__', 'trace')     # trace: nn.Conv3d's record of the maps a forward pass asks for (look-back prefetch)


class SparseTensor:
    """`feats` may be DEFERRED (lidal_amd.nn.Deferred): the output of a train-mode spnn.BatchNorm is computed when it
    is first read, so that an in-place spnn.ReLU and / or the sum of a residual block (`__add__`) that follow it in the
    user's nn.Sequential ride in the normalising kernel -- the fusions the package's own networks are written with,
    carried by the SURFACE for networks that are not (the reference's files, scripts/surface_unet.py).  Reading `feats`
    / `F` resolves it; nothing else changes."""

    def __init__(self, feats, coords, stride=1):
        self._feats = feats
        self._deferred = None
        self.coords = coords
        self.stride = make_ntuple(stride, ndim=3)
        self.cmaps = MapCache()
        self.kmaps = MapCache()

    @property
    def feats(self):
        d = self._deferred
        if d is not None:
            self._feats = d.resolve()
            self._deferred = None
        return self._feats

    @feats.setter
    def feats(self, value):
        self._feats = value
        self._deferred = None

    F = property(lambda self: self.feats, lambda self, v: setattr(self, 'feats', v))
    C = property(lambda self: self.coords, lambda self, v: setattr(self, 'coords', v))
    s = property(lambda self: self.stride, lambda self, v: setattr(self, 'stride', v))

    def _map(self, fn):
        self.coords = fn(self.coords)
        self.feats = fn(self.feats)
        return self

    def cpu(self):
        return self._map(lambda t: t.cpu())

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking=True):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def __add__(self, other):
        # a deferred BatchNorm output + anything: the sum rides in the normalising pass (see Deferred.plus)
        for a, b in ((self, other), (other, self)):
            d = a._deferred
            if d is not None and d.can_take_sum():
                out = SparseTensor(None, self.coords, self.stride)
                out._deferred = d.plus(b.feats)
                out.cmaps = self.cmaps
                out.kmaps = self.kmaps
                return out
        out = SparseTensor(self.feats + other.feats, self.coords, self.stride)
        out.cmaps = self.cmaps
        out.kmaps = self.kmaps
        return out


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = idx_query if idx_query is not None else {}
        self.weights = weights if weights is not None else {}
        self.additional_features = {'idx_query': {}, 'counts': {}}

    def _map(self, fn):
        self.F = fn(self.F)
        self.C = fn(self.C)
        return self

    def cpu(self):
        return self._map(lambda t: t.cpu())

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking=True):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def __add__(self, other):
        out = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        out.additional_features = self.additional_features
        return out
