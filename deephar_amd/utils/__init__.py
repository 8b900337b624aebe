"""Small helpers the builders and callers share with the reference (deephar/utils/__init__.py)."""
from .pose import *  # noqa: F401,F403

TEST_MODE, TRAIN_MODE, VALID_MODE = 0, 1, 2     # deephar/utils/parser.py:12-14


def appstr(s, a):
    """Safe string append: None stays None (deephar/utils/parser.py:254-259)."""
    try:
        return s + a
    except Exception:
        return None
