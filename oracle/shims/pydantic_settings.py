"""Stand-in for ``pydantic_settings`` (absent in this image; test infrastructure only).

The reference's config classes (``alignn/models/alignn.py:19-45``,
``alignn/utils.py:13-21``) only need ``BaseSettings`` to behave like a pydantic
model; the environment-variable loading of the real package is irrelevant to
the arithmetic.
"""

from pydantic import BaseModel


class BaseSettings(BaseModel):
    model_config = {"extra": "forbid", "protected_namespaces": ()}
