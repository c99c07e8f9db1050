from . import registration  # noqa: F401

registry: dict = {}
