class _Empty:
    """attribute access yields inert callables (the reference evaluates
    ``T.ToTensor()`` as a default argument at import time; nothing is ever used)."""

    def __getattr__(self, name):
        return lambda *a, **k: None


transforms = _Empty()
utils = _Empty()
