"""Stand-in for the parts of `diffusers` that the reference's scheduler files import (wan/utils/fm_solvers*.py): configuration
plumbing only -- `register_to_config` keeps the constructor arguments in `self.config`, the mixins are empty, `deprecate` is
silent.  None of the schedulers' arithmetic lives in diffusers: it is all in the reference's own files.  Used by
tests/golden/make_golden_schedulers.py in the build container only."""
