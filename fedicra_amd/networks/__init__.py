from .net_factory import net_factory  # noqa: F401
