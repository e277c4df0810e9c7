from .config import DEFAULTS, load_config, load_default_config  # noqa: F401
