"""Model builders with the reference's signatures (deephar/models/__init__.py)."""
from . import blocks  # noqa: F401
from . import reception  # noqa: F401
