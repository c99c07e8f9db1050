"""Stand-in for aenum: the stdlib enum plus an ``extend_enum`` that refuses."""
from enum import *  # noqa: F401,F403
from enum import Enum, EnumMeta, IntEnum  # noqa: F401


def extend_enum(cls, name, value):
    raise NotImplementedError("extend_enum is not available in the stand-in")
