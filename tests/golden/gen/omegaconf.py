"""Minimal stand-in for the `omegaconf` package (absent from this image).

Only used by tests/golden/make_golden.py so that the reference's fake-quant
package (`qdiff`, under /root/reference) can be imported in the build
container to produce golden vectors.  It is NOT part of the product and never
travels to the GPU box as a dependency of anything.

Provides exactly what qdiff touches: `ListConfig` (isinstance checks) and
`OmegaConf.create` returning an attribute-style dict.
"""


class ListConfig:
    """Sequence that is deliberately NOT a `list` subclass (as in the real package):
    qdiff tells the two apart with isinstance(..., list) (base_quantizer.py:23)."""

    def __init__(self, items=()):
        self._items = list(items)

    def __getitem__(self, i):
        return self._items[i]

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(o):
    if isinstance(o, dict):
        return DictConfig({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return ListConfig([_wrap(v) for v in o])
    return o


class OmegaConf:
    @staticmethod
    def create(o):
        return _wrap(o)
