"""MI355X-native VL-SAT eval forward (PointNet object encoder -> MMG attention-GNN ->
node/edge heads) behind the reference's ``Mmgnet.forward`` tensor-in/tensor-out contract
(reference ``src/model/SGFN_MMG/model.py:288-335``).  Compute lives in ``csrc/*.hip``
(gfx950) behind the C ABI declared in ``include/vlsat.h``; Python is glue only."""
from .config import VLSATConfig, param_shapes  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["VLSATConfig", "param_shapes", "synth"]
