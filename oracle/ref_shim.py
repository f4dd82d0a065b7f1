"""Import the UNMODIFIED reference ``src/modules.py`` from /root/reference  --  TEST
INFRASTRUCTURE, build-container only (``/root/reference`` does not exist on the GPU
box; nothing in ``-m gpu`` tests, ``smoke()`` or ``bench.py`` may call this).

``modules.py:3`` does ``from utils import *`` and the reference's ``utils.py:11-19``
imports wget / torch._six / torchmetrics / torchvision, none of which exist here.
We pre-register a stub ``utils`` exporting exactly the names ``modules.py`` uses
(``nn, F, torch, np``) so the reference file itself is executed byte-for-byte.
"""
import os
import sys
import types

REFERENCE_SRC = "/root/reference/src"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "modules.py"))


def load_reference_modules():
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_SRC)
    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    if "utils" not in sys.modules or not getattr(sys.modules["utils"], "_stego_stub", False):
        stub = types.ModuleType("utils")
        stub.nn, stub.F, stub.torch, stub.np = nn, F, torch, np
        stub.__all__ = ["nn", "F", "torch", "np"]
        stub._stego_stub = True
        sys.modules["utils"] = stub
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import modules as ref_modules  # noqa: the reference's src/modules.py
    return ref_modules
