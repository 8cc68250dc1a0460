"""Minimal stand-ins for the Flower 1.0 message types the reference's client/strategy protocol uses
(/root/reference/code/flower_common.py:8-17): same names, same fields.  If the real ``flwr`` package
is importable its types are used instead, so an unmodified Flower server can drive these clients.
"""
from __future__ import annotations

import io
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np

try:  # pragma: no cover - flwr is not installed in the build image
    from flwr.common import (Code, EvaluateIns, EvaluateRes, FitIns, FitRes, GetParametersIns, GetParametersRes,
                             GetPropertiesIns, GetPropertiesRes, Parameters, Status, ndarrays_to_parameters,
                             parameters_to_ndarrays)
    HAVE_FLWR = True
except Exception:  # noqa: BLE001
    HAVE_FLWR = False

    class Code:
        OK = 0

    @dataclass
    class Status:
        code: object = "OK"
        message: str = "Success"

    @dataclass
    class Parameters:
        tensors: List[bytes]
        tensor_type: str = "numpy.ndarray"

    @dataclass
    class FitIns:
        parameters: Parameters
        config: Dict = field(default_factory=dict)

    @dataclass
    class FitRes:
        status: Status
        parameters: Parameters
        num_examples: int
        metrics: Dict = field(default_factory=dict)

    @dataclass
    class EvaluateIns:
        parameters: Parameters
        config: Dict = field(default_factory=dict)

    @dataclass
    class EvaluateRes:
        status: Status
        loss: float
        num_examples: int
        metrics: Dict = field(default_factory=dict)

    @dataclass
    class GetParametersIns:
        config: Dict = field(default_factory=dict)

    @dataclass
    class GetParametersRes:
        status: Status
        parameters: Parameters

    @dataclass
    class GetPropertiesIns:
        config: Dict = field(default_factory=dict)

    @dataclass
    class GetPropertiesRes:
        status: Status
        properties: Dict = field(default_factory=dict)

    def ndarray_to_bytes(a: np.ndarray) -> bytes:
        buf = io.BytesIO()
        np.save(buf, a, allow_pickle=False)          # Flower's wire format: np.save per array
        return buf.getvalue()

    def bytes_to_ndarray(b: bytes) -> np.ndarray:
        return np.load(io.BytesIO(b), allow_pickle=False)

    def ndarrays_to_parameters(arrs) -> Parameters:
        return Parameters(tensors=[ndarray_to_bytes(np.asarray(a)) for a in arrs])

    def parameters_to_ndarrays(p: Parameters):
        return [bytes_to_ndarray(t) for t in p.tensors]

if HAVE_FLWR:  # pragma: no cover
    from flwr.common import bytes_to_ndarray, ndarray_to_bytes  # noqa: F401
