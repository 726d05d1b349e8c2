"""Import harness for the reference modules (this container only; never shipped to the GPU box's run-time path).

Used only by tests/golden/make_golden.py to GENERATE fixtures.  /root/reference is read-only and is imported in
place; nothing is copied.  Three inert import stand-ins replace packages the image lacks and that
the reference only touches at import time: torchaudio (never called on the paths exercised here), loguru (logger),
transformers.utils.model_parallel_utils (only used by the never-called ``parallelize``).
"""
import logging
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install():
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before the torchaudio stand-in exists)
    from transformers import GPT2Config, GPT2Model  # noqa: F401

    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.transforms = types.ModuleType("torchaudio.transforms")
        ta.functional = types.ModuleType("torchaudio.functional")
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.transforms"] = ta.transforms
        sys.modules["torchaudio.functional"] = ta.functional
    if "loguru" not in sys.modules:
        lg = types.ModuleType("loguru")
        lg.logger = logging.getLogger("loguru_standin")
        sys.modules["loguru"] = lg
    name = "transformers.utils.model_parallel_utils"
    if name not in sys.modules:
        mp = types.ModuleType(name)
        mp.assert_device_map = lambda *a, **k: None
        mp.get_device_map = lambda *a, **k: None
        sys.modules[name] = mp
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
