"""Test-tooling stub."""


def load_dotenv(*a, **k):
    return False
