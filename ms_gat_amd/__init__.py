"""ms_gat_amd -- MI355X-native implementation of the MS-GAT graph-attention hot path.

The directory is `ms_gat_amd` (an importable spelling of "ms-gat_amd").  It holds only
what the hot path needs: the HIP kernels and C ABI (`csrc/`, include/msgat_hip.h) and the
host-side mirror of the reference's operator interface (`GraphAttention`, `GACN`), plus the
callers either side of it that SURVEY.md section 8 marks "keep API" / "next": the MS-GAT blocks
(`model.py`), the train/eval loop (`engine.py`), data slicing (`data.py`) and batch-sharded data
parallelism (`parallel.py`).
"""
from .attention import GACN, GraphAttention, StackedGACN  # noqa: F401
from .graph import SparseGraph, graph_of, random_edges, sym_norm_adjacency, synthetic_adjacency  # noqa: F401
from .ops import gacn, graph_attention  # noqa: F401
from .model import MEAM, MSGAT, TPC, msgat48, msgat72, msgat96  # noqa: F401
from .engine import Evaluator, HuberLoss, Metrics, Trainer  # noqa: F401

__all__ = ["GACN", "GraphAttention", "StackedGACN", "SparseGraph", "graph_of", "random_edges",
           "sym_norm_adjacency", "synthetic_adjacency", "gacn", "graph_attention", "MEAM", "TPC", "MSGAT",
           "msgat48", "msgat72", "msgat96", "Trainer", "Evaluator", "HuberLoss", "Metrics"]
