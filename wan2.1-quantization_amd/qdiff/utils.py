"""Module-tree visitor, seeding and logging helpers (interface of ViDiT-Q/quant_utils/qdiff/utils.py:15-94)."""
import logging
import os
import random

import numpy as np
import torch


def apply_func_to_submodules(module, class_type, function, parent_name="", return_d=None, **kwargs):
    """Depth-first over named_children(); calls `function(submodule, **kwargs)` on every instance of class_type.
    kwargs named `name`, `full_name`, `parent_module` are filled in per visited module (reference utils.py:31-38).
    The children list is snapshotted first, so `function` may replace the child on its parent; recursion continues
    into the original child (a replaced nn.Linear has no children, and its replacement's `fp_module` must not be visited)."""
    for name, sub in list(module.named_children()):
        full = f"{parent_name}.{name}" if parent_name else name
        if "name" in kwargs:
            kwargs["name"] = name
        if "full_name" in kwargs:
            kwargs["full_name"] = full
        if "parent_module" in kwargs:
            kwargs["parent_module"] = module
        if isinstance(sub, class_type):
            res = function(sub, **kwargs)
            if return_d is not None:
                return_d[full] = res
        apply_func_to_submodules(sub, class_type, function, full, return_d, **kwargs)  # the ORIGINAL child, as the reference
    return return_d


def seed_everything(seed=42):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def setup_logging(log_file=None, level=logging.INFO):
    handlers = [logging.StreamHandler()]
    if log_file:
        handlers.append(logging.FileHandler(log_file, mode="a"))
    logging.basicConfig(level=level, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s", handlers=handlers, force=True)
