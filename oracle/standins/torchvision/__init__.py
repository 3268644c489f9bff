"""Minimal stand-in for the absent third-party ``torchvision`` package.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  It exists so that the
unmodified ``/root/reference/model.py`` can be imported in the build container
to generate golden vectors (reference ``model.py:6-14`` imports torchvision and
gates on its version).  Written from the published ResNet-50 v1.5 architecture,
not from torchvision sources.  No pretrained weights (no network): every weight
is whatever the caller loads.
"""
__version__ = "0.15.0"
from . import models  # noqa: E402,F401
