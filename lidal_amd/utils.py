"""Small host helpers shared by the operator API (mirrors torchsparse/utils)."""


def make_ntuple(x, ndim):
    if isinstance(x, int):
        x = tuple([x] * ndim)
    elif isinstance(x, list):
        x = tuple(x)
    assert isinstance(x, tuple) and len(x) == ndim, x
    return x
