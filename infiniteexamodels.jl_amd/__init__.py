"""MI355X-native NLP evaluation backend for InfiniteOpt → ExaModels transcriptions.

Host-side mirror (Python) of the reference's plug point
``ExaTranscriptionBackend(solver; backend = …)`` and of the ExaModels builder /
NLPModels evaluation surface it drives; all arithmetic runs in HIP for gfx950 behind the
C-ABI of ``include/iem.h`` (``libiem_hip.so``).  There is no CPU evaluation path.
"""
from .nodes import (Const, DataField, DataSource, Node, Null, ParameterNode, Var, FUNCS, nary)
from .items import Items
from .core import ExaCore, Variable, Parameter
from .infinite import InfiniteModel, DomainRestriction
from .transcribe import ExaMappingData, exa_core
from .model import ExaModel, MI355XBackend
from .backend import ExaTranscriptionBackend

__all__ = ["Const", "DataField", "DataSource", "Node", "Null", "ParameterNode", "Var", "FUNCS",
           "nary", "Items", "ExaCore", "Variable", "Parameter", "InfiniteModel", "DomainRestriction",
           "ExaMappingData", "exa_core", "ExaModel", "MI355XBackend", "ExaTranscriptionBackend"]
