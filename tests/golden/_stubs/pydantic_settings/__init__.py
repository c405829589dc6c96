"""Test-tooling stub (not reference code): stands in for the missing
pydantic-settings wheel when importing the reference in the build container.
Same substitution the reference's own tests/conftest.py:12-17 makes."""
from pydantic import BaseModel as BaseSettings  # noqa: F401

SettingsConfigDict = dict
