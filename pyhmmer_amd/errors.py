"""Error types of the search path, same names and meaning as ``pyhmmer.errors``
(reference ``src/pyhmmer/errors.pyx:43-275``) plus :class:`DeviceUnavailable`, raised -- instead
of silently computing on the CPU -- when a device entry point is used without an MI355X."""
from __future__ import annotations

statuscode = {
    0: "eslOK", 1: "eslFAIL", 2: "eslEOL", 3: "eslEOF", 4: "eslEOD", 5: "eslEMEM", 6: "eslENOTFOUND",
    7: "eslEFORMAT", 8: "eslEAMBIGUOUS", 9: "eslEDIVZERO", 10: "eslEINCOMPAT", 11: "eslEINVAL",
    12: "eslESYS", 13: "eslECORRUPT", 14: "eslEINCONCEIVABLE", 15: "eslESYNTAX", 16: "eslERANGE",
    17: "eslEDUP", 18: "eslENOHALT", 19: "eslENORESULT", 100: "p7xENODEVICE", 101: "p7xEDEVICE",
}


class UnexpectedError(RuntimeError):
    def __init__(self, code: int, function: str):
        super().__init__(code, function)
        self.code = code
        self.function = function

    def __str__(self):
        return "Unexpected error occurred in {!r}: {} (status code {})".format(
            self.function, statuscode.get(self.code, "<unknown>"), self.code)


class AllocationError(MemoryError):
    def __init__(self, ctype: str, itemsize: int, count: int = 1):
        super().__init__(ctype, itemsize, count)
        self.ctype, self.itemsize, self.count = ctype, itemsize, count


class AlphabetMismatch(ValueError):
    def __init__(self, expected, actual):
        super().__init__(expected, actual)
        self.expected, self.actual = expected, actual

    def __str__(self):
        return "Expected {}, found {}".format(self.expected, self.actual)


class MissingCutoffs(ValueError):
    """The model is missing the bit-score cutoffs a pipeline was asked to use
    (reference ``errors.pyx:180-210``; raised at ``plan7.pyx:6424-6425``)."""

    def __init__(self, model_name=None, bit_cutoffs=None):
        super().__init__(model_name, bit_cutoffs)
        self.model_name, self.bit_cutoffs = model_name, bit_cutoffs

    def __str__(self):
        if self.model_name is not None and self.bit_cutoffs is not None:
            return f"HMM {self.model_name!r} is missing {self.bit_cutoffs} cutoffs"
        return "Model is missing required bit-score cutoffs"


class InvalidParameter(ValueError):
    def __init__(self, name, value, choices=None, hint=None):
        super().__init__(name, value, choices, hint)
        self.name, self.value, self.choices, self.hint = name, value, choices, hint

    def __str__(self):
        opt = f" (expected one of {self.choices})" if self.choices else (f" ({self.hint})" if self.hint else "")
        return f"Invalid {self.name!r} parameter value: {self.value!r}{opt}"


class DeviceUnavailable(RuntimeError):
    """No usable HIP device / libp7x device entry point failed.  The product never falls back to a CPU path."""


def status_to_exception(status: int, function: str, detail: str = "") -> Exception:
    """Error convention of the boundary: eslEINVAL -> MissingCutoffs is decided by
    the caller; eslERANGE -> OverflowError; p7x device codes -> DeviceUnavailable; else UnexpectedError."""
    if status == 16:
        return OverflowError(detail or f"numeric overflow in {function}")
    if status == 5:
        return AllocationError(function, 0)
    if status in (100, 101):
        return DeviceUnavailable(f"{function}: {detail or statuscode[status]} (no CPU fallback exists)")
    if status == 11 and detail:
        return ValueError(f"{function}: {detail}")
    return UnexpectedError(status, function)
