"""Model builders with the reference's signatures (deephar/models/__init__.py:1-8)."""
from . import blocks  # noqa: F401
from . import reception  # noqa: F401
from . import action  # noqa: F401
from . import spnet  # noqa: F401
from .spnet import split_model  # noqa: F401
