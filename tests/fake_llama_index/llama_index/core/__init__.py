from .schema import QueryBundle  # noqa: F401
