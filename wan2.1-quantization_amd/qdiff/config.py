"""Quant-config loading without omegaconf (absent from the image): attribute-style dicts + a list type that is not a
`list`, i.e. exactly what the reference's code relies on (`cfg.weight.n_bits`, `cfg.get('viditq')`,
`isinstance(n_bits, ListConfig)`).  Schema: ViDiT-Q/examples/Wan2.1/quant_configs/config.yaml."""
import yaml


class ListConfig:
    def __init__(self, items=()):
        self._items = list(items)

    def __getitem__(self, i):
        return self._items[i]

    def __len__(self):
        return len(self._items)

    def __iter__(self):
        return iter(self._items)

    def __repr__(self):
        return f"ListConfig({self._items})"


class DictConfig(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def create(obj):
    if isinstance(obj, dict):
        return DictConfig({k: create(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return ListConfig([create(v) for v in obj])
    return obj


def load(path):
    with open(path) as f:
        return create(yaml.safe_load(f))


class OmegaConf:  # the two calls the reference's scripts make
    create = staticmethod(create)
    load = staticmethod(load)
