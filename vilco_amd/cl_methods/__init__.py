from .prompt import Prompt  # noqa: F401
