"""Error types raised at the drop-in boundary.

Mirrors the error convention of the reference's public API for the
``preprocess_data`` path (reference: marEx/exceptions.py:11-81 base class,
84-119 ``DataValidationError``, 180-214 ``ConfigurationError``, 338-360
``create_data_validation_error``).  The reference's tests match on the
class and on a regex of the first line of the message
(tests/test_error_handling.py), so only those two properties plus the
``details / suggestions / context / error_code`` attributes are kept.
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional


class MarExError(Exception):
    """Root of the error hierarchy (reference: marEx/exceptions.py:11)."""

    default_code: Optional[str] = None

    def __init__(
        self,
        message: str,
        details: Optional[str] = None,
        suggestions: Optional[List[str]] = None,
        error_code: Optional[str] = None,
        context: Optional[Dict[str, Any]] = None,
    ):
        self.message = message
        self.details = details
        self.suggestions = list(suggestions) if suggestions else []
        self.error_code = error_code if error_code is not None else self.default_code
        self.context = dict(context) if context else {}
        super().__init__(self._render())

    def _render(self) -> str:
        lines = [self.message]
        if self.details:
            lines.append(f"Details: {self.details}")
        if self.context:
            lines.append("Context: " + ", ".join(f"{k}={v}" for k, v in self.context.items()))
        if self.suggestions:
            lines.append("Suggestions:\n" + "\n".join(f"  - {s}" for s in self.suggestions))
        if self.error_code:
            lines.append(f"Error Code: {self.error_code}")
        return "\n".join(lines)

    def add_suggestion(self, suggestion: str) -> None:
        self.suggestions.append(suggestion)

    def add_context(self, key: str, value: Any) -> None:
        self.context[key] = value


class DataValidationError(MarExError):
    """Problems with the input data (reference: marEx/exceptions.py:84)."""

    default_code = "DATA_VALIDATION"


class ConfigurationError(MarExError):
    """Invalid option / option combination (reference: marEx/exceptions.py:180)."""

    default_code = "CONFIGURATION"


class ProcessingError(MarExError):
    """Failure inside the compute path (reference: marEx/exceptions.py:151)."""

    default_code = "PROCESSING"


class DependencyError(MarExError):
    """A required native component is missing (reference: marEx/exceptions.py:217).

    Raised when the HIP extension ``libmarex_hip.so`` cannot be loaded: the
    product path never falls back to a CPU implementation.
    """

    default_code = "DEPENDENCY"


def create_data_validation_error(
    message: str, data_info: Optional[Dict[str, Any]] = None, **kwargs: Any
) -> DataValidationError:
    """Convenience constructor (reference: marEx/exceptions.py:338-360)."""
    context = dict(kwargs.pop("context", None) or {})
    if data_info:
        context.update(data_info)
    return DataValidationError(message, context=context, **kwargs)
