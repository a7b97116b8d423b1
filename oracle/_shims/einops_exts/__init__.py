from einops import rearrange


def rearrange_many(tensors, pattern, **kw):
    return tuple(rearrange(t, pattern, **kw) for t in tensors)


def check_shape(tensor, pattern, **kw):
    return rearrange(tensor, f"{pattern} -> {pattern}", **kw)
