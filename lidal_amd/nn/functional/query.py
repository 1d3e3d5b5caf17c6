"""F.sphashquery (torchsparse/nn/functional/query.py; network/utils.py:19,48,76).

HashTable is the reusable half (build once, probe many); sphashquery is the one-shot API."""
import torch

from ... import backend as B

__all__ = ['sphashquery', 'HashTable', 'coords_table']


class HashTable:
    """Open-addressing table of i64 keys in HBM; value = index of the first occurrence."""

    def __init__(self, references):
        B.require_gpu(references)
        references = references.contiguous().view(-1)
        assert references.dtype == torch.int64
        self.n = references.numel()
        self.nbytes = B.lib().lidal_hash_table_bytes(self.n)
        self.buf = B.empty(self.nbytes, torch.uint8, references.device)
        B.check(B.lib().lidal_hash_table_build(B.ptr(references), self.n, B.ptr(self.buf),
                                               self.nbytes, B.stream()), 'hash_table_build')

    @classmethod
    def from_coords(cls, coords, stride):
        """The table of sphash(coords) built from the int32 [N, 4] coordinates themselves (tensor stride `stride`, a power
        of two): the same slots, plus the spatial occupancy bitmap the symmetric kernel-map probes read (csrc/common.h)."""
        B.require_gpu(coords)
        self = cls.__new__(cls)
        coords = coords.contiguous()
        self.n = coords.shape[0]
        self.nbytes = B.lib().lidal_hash_table_bytes(self.n)
        self.buf = B.empty(self.nbytes, torch.uint8, coords.device)
        B.check(B.lib().lidal_hash_table_build_coords(B.ptr(coords), self.n, int(stride), B.ptr(self.buf), self.nbytes,
                                                      B.stream()), 'hash_table_build')
        return self

    def query(self, queries):
        B.require_gpu(queries)
        sizes = queries.size()
        q = queries.contiguous().view(-1)
        assert q.dtype == torch.int64
        out = B.empty(q.shape, q.dtype, q.device)
        B.check(B.lib().lidal_hash_table_query(B.ptr(self.buf), self.nbytes, B.ptr(q), q.numel(),
                                               B.ptr(out), B.stream()), 'hash_table_query')
        return out.view(*sizes)


def sphashquery(queries, references):
    return HashTable(references).query(queries)


def coords_table(coords, scope=None, stride=None):
    """HashTable over sphash(coords) for an int32 [N, 4] coordinate tensor, built once per level of
    one input: the kernel-map builds and the point<->voxel look-ups of a level all probe the same
    table.  The table is cached ON the coordinate tensor, keyed by its version counter / storage AND
    by `scope` -- the `cmaps` dict (tensor.MapCache) of the SparseTensor family it was built for --
    so a caller that keeps one coordinate tensor resident over many forward passes (bench.py) gets a
    fresh table per pass, exactly the work the reference does per iteration.  scope=None: no
    cross-call caching.  `stride` (the level's tensor stride, if the caller knows it): the table is then built from the
    coordinates in one launch and carries the spatial bitmap (HashTable.from_coords)."""
    import weakref
    from .hash import sphash
    key = (coords._version, coords.data_ptr(), coords.shape[0])
    cached = getattr(coords, '_lidal_table', None)
    if (cached is not None and cached[0] == key and scope is not None and cached[2] is not None
            and cached[2]() is scope):
        return cached[1]
    s0 = int(stride[0] if isinstance(stride, (tuple, list)) else stride) if stride is not None else 0
    if s0 >= 1 and s0 & (s0 - 1) == 0 and coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4:
        table = HashTable.from_coords(coords, s0)
    else:
        table = HashTable(sphash(coords))
    ref = None
    if scope is not None:
        try:
            ref = weakref.ref(scope)
        except TypeError:           # a plain dict assigned by foreign code: cannot be scoped
            ref = None
    coords._lidal_table = (key, table, ref)
    return table
